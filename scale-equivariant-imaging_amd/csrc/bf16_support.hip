// Support kernels of the bf16 throughput mode: casts, weight shadows, bf16-output LayerNorm, bf16 column sums.
// All HBM-bound streaming kernels.
#include "sei_common.h"

namespace {

__device__ __forceinline__ unsigned short f2bf(float v) {
    const __bf16 b = (__bf16)v;
    return __builtin_bit_cast(unsigned short, b);
}
__device__ __forceinline__ float bf2f(unsigned short v) { return __uint_as_float((unsigned)v << 16); }

__global__ __launch_bounds__(256) void cast_bf16_kernel(const float *__restrict__ x, unsigned short *__restrict__ y,
                                                        size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t n4 = n / 4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const float4 v = reinterpret_cast<const float4 *>(x)[i];
        ushort4 o;
        o.x = f2bf(v.x); o.y = f2bf(v.y); o.z = f2bf(v.z); o.w = f2bf(v.w);
        reinterpret_cast<ushort4 *>(y)[i] = o;
    }
    for (size_t i = n4 * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) y[i] = f2bf(x[i]);
}

__device__ __forceinline__ unsigned short to_bf(float v) { return f2bf(v); }
__device__ __forceinline__ unsigned short to_bf(unsigned short v) { return v; }

// x (R, C) f32 or bf16 -> x16 (R, C) bf16 (optional) and xt16 (C, ldt) bf16 (optional; columns R..ldt-1
// are zero-filled so that the transposed matrix can be a K-padded GEMM operand). 64x64 tiles through LDS.
__device__ __forceinline__ float to_f(float v) { return v; }
__device__ __forceinline__ float to_f(unsigned short v) { return bf2f(v); }

// colsum (optional): colsum[c] += sum_r x[r][c] in float32 from the UN-rounded input (bias gradients),
// one atomic per column per 64-row tile.
template <typename T>
__global__ __launch_bounds__(256) void cast_transpose_kernel(const T *__restrict__ x,
                                                             unsigned short *__restrict__ x16,
                                                             unsigned short *__restrict__ xt16, int R, int C,
                                                             int ldt, float *__restrict__ colsum) {
    __shared__ unsigned short tile[64][66];
    __shared__ float csum[4][64];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    float part = 0.f;                      // this thread's column c = tid & 63, rows (tid >> 6) + 4k
    for (int e = threadIdx.x; e < 64 * 64; e += 256) {
        const int r = e >> 6, c = e & 63;
        unsigned short v = 0;
        if (r0 + r < R && c0 + c < C) {
            const T raw = x[(size_t)(r0 + r) * C + c0 + c];
            part += to_f(raw);
            v = to_bf(raw);
            if (x16) x16[(size_t)(r0 + r) * C + c0 + c] = v;
        }
        tile[r][c] = v;
    }
    if (colsum) csum[threadIdx.x >> 6][threadIdx.x & 63] = part;
    __syncthreads();
    if (colsum && threadIdx.x < 64 && c0 + threadIdx.x < C && r0 < R)
        atomicAdd(colsum + c0 + threadIdx.x,
                  csum[0][threadIdx.x] + csum[1][threadIdx.x] + csum[2][threadIdx.x] + csum[3][threadIdx.x]);
    if (xt16)
        for (int e = threadIdx.x; e < 64 * 64; e += 256) {
            const int c = e >> 6, r = e & 63;
            if (r0 + r < ldt && c0 + c < C) xt16[(size_t)(c0 + c) * ldt + r0 + r] = tile[r][c];   // zeros past R
        }
}

// f32 (R, C) -> bf16 copy (+ optional column sums of the un-rounded input), no transpose: a lane owns 4
// consecutive columns (float4 in, 8 bytes out) and walks rows, 4 in flight; the row sub-groups of a workgroup
// are folded through LDS and the column sums leave as coalesced atomics (one per column per workgroup).
// row_weight (optional): colsum[c] += sum_r row_weight[r] x[r][c] (the bias gradient of the convolution evaluated behind the
// ideal downsampler, whose bias enters as bias[c] * s[row]: models/_ops.DownsampleFn16).
__global__ __launch_bounds__(256) void cast_colsum_kernel(const float *__restrict__ x,
                                                          unsigned short *__restrict__ x16,
                                                          float *__restrict__ colsum, size_t R, int C, int tpr,
                                                          size_t rows_per_block, const float *__restrict__ row_weight,
                                                          float *__restrict__ part) {
    __shared__ float red[256 * 4];
    const int rsubs = 256 / tpr;
    const int cl = threadIdx.x % tpr, rsub = threadIdx.x / tpr;
    const int cq = blockIdx.y * tpr + cl;                       // column quad: columns 4*cq .. 4*cq+3
    const size_t r0 = (size_t)blockIdx.x * rows_per_block, r1 = min(R, r0 + rows_per_block);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    auto emit = [&](size_t r, const float4 v) {
        if (row_weight) {
            const float w = row_weight[r];
            s.x = fmaf(w, v.x, s.x); s.y = fmaf(w, v.y, s.y); s.z = fmaf(w, v.z, s.z); s.w = fmaf(w, v.w, s.w);
        } else {
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        uint2 o;
        o.x = (unsigned)to_bf(v.x) | ((unsigned)to_bf(v.y) << 16);
        o.y = (unsigned)to_bf(v.z) | ((unsigned)to_bf(v.w) << 16);
        *reinterpret_cast<uint2 *>(x16 + r * C + 4 * (size_t)cq) = o;
    };
    if (4 * cq < C) {
        const float *col = x + 4 * (size_t)cq;
        size_t r = r0 + rsub;
        for (; r + 3 * (size_t)rsubs < r1; r += 4 * (size_t)rsubs) {
            const float4 v0 = *reinterpret_cast<const float4 *>(col + r * C);
            const float4 v1 = *reinterpret_cast<const float4 *>(col + (r + rsubs) * C);
            const float4 v2 = *reinterpret_cast<const float4 *>(col + (r + 2 * (size_t)rsubs) * C);
            const float4 v3 = *reinterpret_cast<const float4 *>(col + (r + 3 * (size_t)rsubs) * C);
            emit(r, v0); emit(r + rsubs, v1); emit(r + 2 * (size_t)rsubs, v2); emit(r + 3 * (size_t)rsubs, v3);
        }
        for (; r < r1; r += rsubs) emit(r, *reinterpret_cast<const float4 *>(col + r * C));
    }
    if (!colsum && !part) return;                               // uniform over the workgroup
    const int width = tpr * 4;
    red[rsub * width + cl * 4 + 0] = s.x;
    red[rsub * width + cl * 4 + 1] = s.y;
    red[rsub * width + cl * 4 + 2] = s.z;
    red[rsub * width + cl * 4 + 3] = s.w;
    __syncthreads();
    const int c0 = blockIdx.y * width;
    for (int e = threadIdx.x; e < width; e += 256) {
        if (c0 + e >= C) break;
        float t = 0.f;
        for (int q = 0; q < rsubs; ++q) t += red[q * width + e];
        // part: this row block's sums as one row of [row blocks][C] partial sums (sei_cast_bf16_colsum_parts: folded later by
        // sei_fold_many, no atomics here -- they are what bounds the atomics form's grid, tools/exp_cast.py)
        if (part) part[(size_t)blockIdx.x * C + c0 + e] = t;
        else atomicAdd(colsum + c0 + e, t);
    }
}

template <int G>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int off = G / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// LayerNorm over C with a bf16 output (and f32 mean / rstd for the backward). One workgroup row-group
// layout as in unet_kernels.hip: G lanes per row for C <= 512, one workgroup per row above.
template <int G>
__global__ __launch_bounds__(256) void ln_fwd_bf16_group_kernel(const float *__restrict__ x,
                                                                const float *__restrict__ gamma,
                                                                const float *__restrict__ beta,
                                                                unsigned short *__restrict__ y,
                                                                float *__restrict__ mean, float *__restrict__ rstd,
                                                                size_t rows, int C, float eps) {
    const int lg = threadIdx.x % G, rsub = threadIdx.x / G;
    constexpr int RPB = 256 / G;
    const float invC = 1.0f / (float)C;
    for (size_t row = (size_t)blockIdx.x * RPB + rsub; row < rows; row += (size_t)gridDim.x * RPB) {
        const float *xr = x + row * C;
        float v[8];
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = lg + e * G;
            v[e] = c < C ? xr[c] : 0.f;
            s += v[e];
        }
        const float mu = group_sum<G>(s) * invC;
        float q = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = lg + e * G;
            const float d = c < C ? v[e] - mu : 0.f;
            q = fmaf(d, d, q);
        }
        const float rs = 1.0f / sqrtf(group_sum<G>(q) * invC + eps);
        unsigned short *yr = y + row * C;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = lg + e * G;
            if (c < C) yr[c] = f2bf(fmaf((v[e] - mu) * rs, gamma[c], beta[c]));
        }
        if (lg == 0) {
            mean[row] = mu;
            rstd[row] = rs;
        }
    }
}

// four normalised values of one quad: 8 bytes of bf16 or 16 bytes of float32 (sei_ln_fwd's fast path, round 5)
__device__ __forceinline__ void ln_store4(unsigned short *p, float a, float b, float c, float d) {
    uint2 w;
    w.x = (unsigned)f2bf(a) | ((unsigned)f2bf(b) << 16);
    w.y = (unsigned)f2bf(c) | ((unsigned)f2bf(d) << 16);
    *reinterpret_cast<uint2 *>(p) = w;
}
__device__ __forceinline__ void ln_store4(float *p, float a, float b, float c, float d) {
    *reinterpret_cast<float4 *>(p) = make_float4(a, b, c, d);
}

// Narrow rows with 16-byte lanes (C = 4 * L * NV, L a power of two <= 64, NV <= 2, i.e. C <= 512): L lanes own a row,
// NV float4 each; 256 / L rows per sweep. One 16-byte load and one 8-byte store per quad instead of four 4-byte
// loads and four 2-byte stores -- the scalar kernel above ran the 18.9-MB level-0 / level-1 tensors at 1.6 TB/s.
template <int L, int NV, typename OT = unsigned short>
__global__ __launch_bounds__(256) void ln_fwd_bf16_quad_kernel(const float *__restrict__ x,
                                                               const float *__restrict__ gamma,
                                                               const float *__restrict__ beta,
                                                               OT *__restrict__ y,
                                                               float *__restrict__ mean, float *__restrict__ rstd,
                                                               size_t rows, float eps) {
    constexpr int C = 4 * L * NV, RPB = 256 / L;
    const int lg = threadIdx.x % L, rsub = threadIdx.x / L;
    const float invC = 1.0f / (float)C;
    float4 gam[NV], bet[NV];
#pragma unroll
    for (int e = 0; e < NV; ++e) {
        gam[e] = *reinterpret_cast<const float4 *>(gamma + 4 * (lg + e * L));
        bet[e] = *reinterpret_cast<const float4 *>(beta + 4 * (lg + e * L));
    }
    for (size_t row = (size_t)blockIdx.x * RPB + rsub; row < rows; row += (size_t)gridDim.x * RPB) {
        float4 v[NV];
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < NV; ++e) {
            v[e] = *reinterpret_cast<const float4 *>(x + row * C + 4 * (lg + e * L));
            s += (v[e].x + v[e].y) + (v[e].z + v[e].w);
        }
        const float mu = group_sum<L>(s) * invC;
        float q = 0.f;
#pragma unroll
        for (int e = 0; e < NV; ++e) {
            const float dx = v[e].x - mu, dy = v[e].y - mu, dz = v[e].z - mu, dw = v[e].w - mu;
            q = fmaf(dx, dx, q); q = fmaf(dy, dy, q); q = fmaf(dz, dz, q); q = fmaf(dw, dw, q);
        }
        const float rs = 1.0f / sqrtf(group_sum<L>(q) * invC + eps);
#pragma unroll
        for (int e = 0; e < NV; ++e) {
            ln_store4(y + row * C + 4 * (lg + e * L), fmaf((v[e].x - mu) * rs, gam[e].x, bet[e].x),
                      fmaf((v[e].y - mu) * rs, gam[e].y, bet[e].y), fmaf((v[e].z - mu) * rs, gam[e].z, bet[e].z),
                      fmaf((v[e].w - mu) * rs, gam[e].w, bet[e].w));
        }
        if (lg == 0) {
            mean[row] = mu;
            rstd[row] = rs;
        }
    }
}

__global__ __launch_bounds__(256) void ln_fwd_bf16_wide_kernel(const float *__restrict__ x,
                                                               const float *__restrict__ gamma,
                                                               const float *__restrict__ beta,
                                                               unsigned short *__restrict__ y,
                                                               float *__restrict__ mean, float *__restrict__ rstd,
                                                               size_t rows, int C, float eps) {
    __shared__ float scratch[4];
    __shared__ float bc[2];
    const float invC = 1.0f / (float)C;
    for (size_t row = blockIdx.x; row < rows; row += gridDim.x) {
        const float *xr = x + row * C;
        float v[32];
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < 32; ++e) {
            const int c = threadIdx.x + e * 256;
            v[e] = c < C ? xr[c] : 0.f;
            s += v[e];
        }
        s = sei_block_sum<256>(s, scratch);
        if (threadIdx.x == 0) bc[0] = s * invC;
        __syncthreads();
        const float mu = bc[0];
        float q = 0.f;
#pragma unroll
        for (int e = 0; e < 32; ++e) {
            const int c = threadIdx.x + e * 256;
            const float d = c < C ? v[e] - mu : 0.f;
            q = fmaf(d, d, q);
        }
        q = sei_block_sum<256>(q, scratch);
        if (threadIdx.x == 0) bc[1] = 1.0f / sqrtf(q * invC + eps);
        __syncthreads();
        const float rs = bc[1];
        unsigned short *yr = y + row * C;
#pragma unroll
        for (int e = 0; e < 32; ++e) {
            const int c = threadIdx.x + e * 256;
            if (c < C) yr[c] = f2bf(fmaf((v[e] - mu) * rs, gamma[c], beta[c]));
        }
        if (threadIdx.x == 0) {
            mean[row] = mu;
            rstd[row] = rs;
        }
        __syncthreads();
    }
}

// Wide rows with 16-byte lanes: RW waves (1, 2 or 4) own a row, each lane up to 8 float4 (C <= 2048 * RW,
// C % 4 == 0); row data lives in registers (one HBM read), statistics by wave shuffles (+ one LDS exchange
// when RW > 1). 4 / RW rows per workgroup.
template <int RW, typename OT = unsigned short>
__global__ __launch_bounds__(256) void ln_fwd_bf16_vec_kernel(const float *__restrict__ x,
                                                              const float *__restrict__ gamma,
                                                              const float *__restrict__ beta,
                                                              OT *__restrict__ y,
                                                              float *__restrict__ mean, float *__restrict__ rstd,
                                                              size_t rows, int C, float eps) {
    __shared__ float part[2][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave % RW, rsub = wave / RW;                 // wave within the row, row within the workgroup
    constexpr int RPB = 4 / RW, STRIDE = 64 * RW;
    const int nq = C / 4;
    const float invC = 1.0f / (float)C;
    for (size_t row0 = (size_t)blockIdx.x * RPB; row0 < rows; row0 += (size_t)gridDim.x * RPB) {
        const size_t row = row0 + rsub;
        const bool live = row < rows;
        const float4 *xr = reinterpret_cast<const float4 *>(x + (live ? row : 0) * C);
        float4 v[8];
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int q = lane + 64 * wr + e * STRIDE;
            v[e] = (live && q < nq) ? xr[q] : make_float4(0.f, 0.f, 0.f, 0.f);
            s += (v[e].x + v[e].y) + (v[e].z + v[e].w);
        }
        s = sei_wave_sum(s);
        s = __shfl(s, 0, 64);
        if (RW > 1) {
            if (lane == 0) part[0][wave] = s;
            __syncthreads();
            s = 0.f;
#pragma unroll
            for (int k = 0; k < RW; ++k) s += part[0][rsub * RW + k];
        }
        const float mu = s * invC;
        float qq = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int q = lane + 64 * wr + e * STRIDE;
            if (q < nq) {
                const float dx = v[e].x - mu, dy = v[e].y - mu, dz = v[e].z - mu, dw = v[e].w - mu;
                qq = fmaf(dx, dx, qq); qq = fmaf(dy, dy, qq); qq = fmaf(dz, dz, qq); qq = fmaf(dw, dw, qq);
            }
        }
        qq = sei_wave_sum(qq);
        qq = __shfl(qq, 0, 64);
        if (RW > 1) {
            if (lane == 0) part[1][wave] = qq;
            __syncthreads();
            qq = 0.f;
#pragma unroll
            for (int k = 0; k < RW; ++k) qq += part[1][rsub * RW + k];
        }
        const float rs = 1.0f / sqrtf(qq * invC + eps);
        if (live) {
            OT *yr = y + row * C;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int q = lane + 64 * wr + e * STRIDE;
                if (q < nq) {
                    const float4 gm = reinterpret_cast<const float4 *>(gamma)[q], bt = reinterpret_cast<const float4 *>(beta)[q];
                    ln_store4(yr + 4 * (size_t)q, fmaf((v[e].x - mu) * rs, gm.x, bt.x), fmaf((v[e].y - mu) * rs, gm.y, bt.y),
                              fmaf((v[e].z - mu) * rs, gm.z, bt.z), fmaf((v[e].w - mu) * rs, gm.w, bt.w));
                }
            }
            if (lane == 0 && wr == 0) {
                mean[row] = mu;
                rstd[row] = rs;
            }
        }
        if (RW > 1) __syncthreads();                            // part[] is reused by the next row group
    }
}

__global__ __launch_bounds__(256) void colsum_bf16_kernel(const unsigned short *__restrict__ X,
                                                          float *__restrict__ out, size_t M, int N,
                                                          size_t rows_per_block) {
    extern __shared__ __attribute__((aligned(16))) float red[];
    const int cw = min(N, 256);
    const int rsubs = 256 / cw;
    const int cl = threadIdx.x % cw, rsub = threadIdx.x / cw;
    const size_t r0 = (size_t)blockIdx.x * rows_per_block, r1 = min(M, r0 + rows_per_block);
    const int c = blockIdx.y * cw + cl;
    float s = 0.f;
    if (c < N && rsub < rsubs)
        for (size_t r = r0 + rsub; r < r1; r += rsubs) s += bf2f(X[r * N + c]);
    if (rsub < rsubs) red[rsub * cw + cl] = s;
    __syncthreads();
    if (rsub == 0 && c < N) {
        float t = 0.f;
        for (int k = 0; k < rsubs; ++k) t += red[k * cw + cl];
        atomicAdd(out + c, t);
    }
}

// 16 bytes per lane: a lane owns 8 consecutive columns and walks rows (4 rows in flight); the 256/tpr row
// sub-groups of a workgroup are folded through LDS, then one float atomic per column per workgroup -- issued
// with CONSECUTIVE LANES ON CONSECUTIVE COLUMNS. (Letting each lane add its own 8 columns costs 8 atomic
// instructions that each touch every output cache line; the L2 serialises per instruction x line, and the
// kernel measured 226 us instead of 9 for the same 147k atomics.)
__global__ __launch_bounds__(256) void colsum_bf16_vec_kernel(const unsigned short *__restrict__ X,
                                                              float *__restrict__ out, size_t M, int N, int tpr,
                                                              size_t rows_per_block) {
    __shared__ float red[256 * 8];                              // [row sub-group][tpr * 8 columns]
    const int rsubs = 256 / tpr;
    const int cl = threadIdx.x % tpr, rsub = threadIdx.x / tpr;
    const int cg = blockIdx.y * tpr + cl;                       // column group: columns 8*cg .. 8*cg+7
    const size_t r0 = (size_t)blockIdx.x * rows_per_block, r1 = min(M, r0 + rows_per_block);
    float s[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) s[k] = 0.f;
    auto add = [&](const uint4 v) {
        s[0] += __uint_as_float(v.x << 16); s[1] += __uint_as_float(v.x & 0xffff0000u);
        s[2] += __uint_as_float(v.y << 16); s[3] += __uint_as_float(v.y & 0xffff0000u);
        s[4] += __uint_as_float(v.z << 16); s[5] += __uint_as_float(v.z & 0xffff0000u);
        s[6] += __uint_as_float(v.w << 16); s[7] += __uint_as_float(v.w & 0xffff0000u);
    };
    if (8 * cg < N) {
        const unsigned short *col = X + 8 * (size_t)cg;
        size_t r = r0 + rsub;
        for (; r + 3 * (size_t)rsubs < r1; r += 4 * (size_t)rsubs) {
            const uint4 v0 = *reinterpret_cast<const uint4 *>(col + r * N);
            const uint4 v1 = *reinterpret_cast<const uint4 *>(col + (r + rsubs) * N);
            const uint4 v2 = *reinterpret_cast<const uint4 *>(col + (r + 2 * (size_t)rsubs) * N);
            const uint4 v3 = *reinterpret_cast<const uint4 *>(col + (r + 3 * (size_t)rsubs) * N);
            add(v0); add(v1); add(v2); add(v3);
        }
        for (; r < r1; r += rsubs) add(*reinterpret_cast<const uint4 *>(col + r * N));
    }
    const int width = tpr * 8;                                  // columns of this workgroup
#pragma unroll
    for (int k = 0; k < 8; ++k) red[rsub * width + cl * 8 + k] = s[k];
    __syncthreads();
    const int c0 = blockIdx.y * width;
    for (int e = threadIdx.x; e < width; e += 256) {
        if (c0 + e >= N) break;
        float t = 0.f;
        for (int q = 0; q < rsubs; ++q) t += red[q * width + e];
        atomicAdd(out + c0 + e, t);
    }
}

template <int G>
int launch_ln16(const float *x, const float *gamma, const float *beta, unsigned short *y, float *mean, float *rstd,
                size_t rows, int C, float eps, hipStream_t s) {
    size_t grid = sei_ceil_div(rows, 256 / G);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(ln_fwd_bf16_group_kernel<G>, dim3((unsigned)grid), dim3(256), 0, s, x, gamma, beta, y, mean,
                       rstd, rows, C, eps);
    return sei_launch_status();
}

}  // namespace

extern "C" int sei_cast_bf16(const float *x, uint16_t *y, size_t n, void *stream) {
    SEI_REQUIRE(x && y && n > 0);
    SEI_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 7) == 0);
    size_t grid = sei_ceil_div(n / 4 + 1, 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(cast_bf16_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, x, y, n);
    return sei_launch_status();
}

// sei_transpose_bf16_many: the job table travels in the kernel arguments; a workgroup owns one 64 x 64 tile of one job
struct TransposeManyArgs {
    SeiTransposeJob job[SEI_TRANSPOSE_MAX_JOBS];
    int njobs;
};
__global__ __launch_bounds__(256) void transpose_many_kernel(TransposeManyArgs g) {
    __shared__ unsigned short tile[64][66];
    int j = 0, first = 0, tc = 0;
    for (; j < g.njobs; ++j) {                                  // (uniform: scalar loads from the argument block)
        tc = (g.job[j].C + 63) >> 6;
        const int tiles = tc * ((g.job[j].R + 63) >> 6);
        if ((int)blockIdx.x < first + tiles) break;
        first += tiles;
    }
    if (j >= g.njobs) return;
    const SeiTransposeJob &J = g.job[j];
    const int t = (int)blockIdx.x - first, r0 = (t / tc) * 64, c0 = (t % tc) * 64;
    for (int e = threadIdx.x; e < 64 * 64; e += 256) {
        const int r = e >> 6, c = e & 63;
        tile[r][c] = (r0 + r < J.R && c0 + c < J.C) ? J.src[(size_t)(r0 + r) * J.C + c0 + c] : (unsigned short)0;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 64 * 64; e += 256) {
        const int c = e >> 6, r = e & 63;
        if (r0 + r < J.R && c0 + c < J.C) J.dst[(size_t)(c0 + c) * J.R + r0 + r] = tile[r][c];
    }
}

extern "C" int sei_transpose_bf16_many(const SeiTransposeJob *jobs, int njobs, void *stream) {
    SEI_REQUIRE(jobs && njobs > 0 && njobs <= SEI_TRANSPOSE_MAX_JOBS);
    TransposeManyArgs g;
    size_t tiles = 0;
    for (int k = 0; k < njobs; ++k) {
        SEI_REQUIRE(jobs[k].src && jobs[k].dst && jobs[k].src != jobs[k].dst && jobs[k].R > 0 && jobs[k].C > 0);
        g.job[k] = jobs[k];
        tiles += sei_ceil_div((size_t)jobs[k].R, 64) * sei_ceil_div((size_t)jobs[k].C, 64);
    }
    g.njobs = njobs;
    SEI_REQUIRE(tiles < ((size_t)1 << 31));
    hipLaunchKernelGGL(transpose_many_kernel, dim3((unsigned)tiles), dim3(256), 0, (hipStream_t)stream, g);
    return sei_launch_status();
}

extern "C" int sei_weight_shadow_bf16(const float *w, uint16_t *w16, uint16_t *wt16, int R, int C, void *stream) {
    SEI_REQUIRE(w && (w16 || wt16) && R > 0 && C > 0);
    hipLaunchKernelGGL(cast_transpose_kernel<float>, dim3((unsigned)sei_ceil_div(C, 64), (unsigned)sei_ceil_div(R, 64)),
                       dim3(256), 0, (hipStream_t)stream, w, w16, wt16, R, C, R, (float *)nullptr);
    return sei_launch_status();
}

// Grid cap of the cast + column-sum kernel. Every workgroup ends in one float atomic per column, and those are what the launch
// waits for: tools/exp_cast.py, the step's ten casts with 512 / 1024 / 2048 workgroups: 145 / 232 / 346 us (and eight rows in
// flight per lane on 1024: 191 against 134 in the step).
#ifndef SEI_CAST_WGS
#define SEI_CAST_WGS 512
#endif
#ifndef SEI_CAST_PARTS_WGS
#define SEI_CAST_PARTS_WGS 2048
#endif
// rows per workgroup for a grid of at most `cap` workgroups
static size_t cast_rows_per_block(int R, int C, size_t cap, int *tpr_out, unsigned *col_blocks_out) {
    const int quads = C / 4;
    int tpr = 1;
    while (tpr < quads && tpr < 256) tpr <<= 1;
    const unsigned col_blocks = (unsigned)sei_ceil_div(quads, tpr);
    size_t rpb = (size_t)(256 / tpr) * 4;
    while (sei_ceil_div((size_t)R, rpb) * col_blocks > cap && rpb < (size_t)R) rpb *= 2;
    *tpr_out = tpr;
    *col_blocks_out = col_blocks;
    return rpb;
}
static int cast_colsum_launch(const float *x, uint16_t *x16, float *colsum, const float *row_weight, int R, int C,
                              void *stream, float *part = nullptr) {
    int tpr;
    unsigned col_blocks;
    const size_t rpb = cast_rows_per_block(R, C, part ? (size_t)SEI_CAST_PARTS_WGS : (size_t)SEI_CAST_WGS, &tpr, &col_blocks);
    hipLaunchKernelGGL(cast_colsum_kernel, dim3((unsigned)sei_ceil_div((size_t)R, rpb), col_blocks), dim3(256), 0,
                       (hipStream_t)stream, x, x16, colsum, (size_t)R, C, tpr, rpb, row_weight, part);
    return sei_launch_status();
}

// The cast with its column sums left as PARTIAL sums: part[g][c] = sum over row block g of (row_weight[r]) x[r][c], g <
// sei_cast_bf16_colsum_parts_count(R, C) -- for sei_fold_many (kind SEI_FOLD_SPLIT, ncol = split = C) at the end of the
// backward pass, with the pass's other partial sums. No atomics, so the grid is sized for bandwidth (2048 workgroups).
extern "C" size_t sei_cast_bf16_colsum_parts_count(int R, int C) {
    if (R <= 0 || C <= 0 || C % 4 != 0) return 0;
    int tpr;
    unsigned col_blocks;
    const size_t rpb = cast_rows_per_block(R, C, (size_t)SEI_CAST_PARTS_WGS, &tpr, &col_blocks);
    return sei_ceil_div((size_t)R, rpb);
}
extern "C" int sei_cast_bf16_colsum_parts(const float *x, uint16_t *x16, const float *row_weight, float *part, int R, int C,
                                          void *stream) {
    SEI_REQUIRE(x && x16 && part && R > 0 && C > 0 && C % 4 == 0);
    SEI_REQUIRE((reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(x16) & 7) == 0);
    return cast_colsum_launch(x, x16, nullptr, row_weight, R, C, stream, part);
}

// x (R, C) float32 -> x16 bf16 copy, and colsum[c] += sum_r row_weight[r] x[r][c] from the same pass (un-rounded input).
extern "C" int sei_cast_bf16_colsum_weighted(const float *x, uint16_t *x16, const float *row_weight, float *colsum, int R,
                                             int C, void *stream) {
    SEI_REQUIRE(x && x16 && row_weight && colsum && R > 0 && C > 0 && C % 4 == 0);
    SEI_REQUIRE((reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(x16) & 7) == 0);
    return cast_colsum_launch(x, x16, colsum, row_weight, R, C, stream);
}

extern "C" int sei_cast_transpose_bf16(const void *x, int x_is_bf16, uint16_t *x16, uint16_t *xt16, int R, int C,
                                       int ldt, float *colsum, void *stream) {
    SEI_REQUIRE(x && (x16 || xt16 || colsum) && R > 0 && C > 0 && ldt >= R);
    SEI_REQUIRE(!(x_is_bf16 && x16));
    if (!x_is_bf16 && x16 && !xt16 && C % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 &&
        (reinterpret_cast<uintptr_t>(x16) & 7) == 0)              // plain cast (+ column sums): streaming kernel
        return cast_colsum_launch((const float *)x, x16, colsum, nullptr, R, C, stream);
    // the grid covers ldt rows so that the zero padding of xt16 is written too
    dim3 grid((unsigned)sei_ceil_div(C, 64), (unsigned)sei_ceil_div(xt16 ? ldt : R, 64));
    if (x_is_bf16)
        hipLaunchKernelGGL(cast_transpose_kernel<unsigned short>, grid, dim3(256), 0, (hipStream_t)stream,
                           (const unsigned short *)x, x16, xt16, R, C, ldt, colsum);
    else
        hipLaunchKernelGGL(cast_transpose_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float *)x,
                           x16, xt16, R, C, ldt, colsum);
    return sei_launch_status();
}

// sei_ln_fwd's 16-byte-lane path (unet_kernels.hip calls it first): the kernels above with a float32 result. Returns -1 for
// shapes they do not take (the scalar-lane kernels of unet_kernels.hip then run). The Downsample LayerNorms of the U-Net
// (147456 x 32 ... 2304 x 2048 per step) ran at 1.6-2.1 TB/s on 4-byte lanes.
__attribute__((visibility("hidden"))) int sei_ln_fwd_f32_lanes16(const float *x, const float *gamma, const float *beta, float *y,
                                                                 float *mean, float *rstd, size_t rows, int C, float eps,
                                                                 hipStream_t s) {
    // (C < 32: the scalar lanes stay -- rows of 2-8 channels are ill-conditioned in float32 and the tiny goldens were taken
    // with that summation order)
    if (C % 4 != 0 || C < 32 || C > 8192 || ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(gamma) |
                                    reinterpret_cast<uintptr_t>(beta) | reinterpret_cast<uintptr_t>(y)) & 15) != 0)
        return -1;
    if (C > 512) {
        const int rw = C <= 2048 ? 1 : (C <= 4096 ? 2 : 4);
        size_t grid = sei_ceil_div(rows, (size_t)(4 / rw));
        if (grid > 8192) grid = 8192;
        if (rw == 1) hipLaunchKernelGGL((ln_fwd_bf16_vec_kernel<1, float>), dim3((unsigned)grid), dim3(256), 0, s, x, gamma, beta, y, mean, rstd, rows, C, eps);
        else if (rw == 2) hipLaunchKernelGGL((ln_fwd_bf16_vec_kernel<2, float>), dim3((unsigned)grid), dim3(256), 0, s, x, gamma, beta, y, mean, rstd, rows, C, eps);
        else hipLaunchKernelGGL((ln_fwd_bf16_vec_kernel<4, float>), dim3((unsigned)grid), dim3(256), 0, s, x, gamma, beta, y, mean, rstd, rows, C, eps);
        return sei_launch_status();
    }
    if (((C / 4) & (C / 4 - 1)) != 0) return -1;
    const int q = C / 4, nv = q > 64 ? q / 64 : 1, lanes = q / nv;
    size_t grid = sei_ceil_div(rows, (size_t)(256 / lanes));
    if (grid > 2048) grid = 2048;
#define SEI_LN_QUAD32(LL, NN)                                                                                          \
    hipLaunchKernelGGL((ln_fwd_bf16_quad_kernel<LL, NN, float>), dim3((unsigned)grid), dim3(256), 0, s, x, gamma, beta, y, \
                       mean, rstd, rows, eps);                                                                          \
    return sei_launch_status();
    switch (lanes * 100 + nv) {
        case 101: { SEI_LN_QUAD32(1, 1) }
        case 201: { SEI_LN_QUAD32(2, 1) }
        case 401: { SEI_LN_QUAD32(4, 1) }
        case 801: { SEI_LN_QUAD32(8, 1) }
        case 1601: { SEI_LN_QUAD32(16, 1) }
        case 3201: { SEI_LN_QUAD32(32, 1) }
        case 6401: { SEI_LN_QUAD32(64, 1) }
        case 6402: { SEI_LN_QUAD32(64, 2) }
        default: break;
    }
#undef SEI_LN_QUAD32
    return -1;
}

extern "C" int sei_ln_fwd_bf16(const float *x, const float *gamma, const float *beta, uint16_t *y, float *mean,
                               float *rstd, size_t rows, int C, float eps, void *stream) {
    SEI_REQUIRE(x && gamma && beta && y && mean && rstd && rows > 0 && C > 0);
    if (C > 8192) return SEI_ERR_TOO_LARGE;
    hipStream_t s = (hipStream_t)stream;
    if (C > 512 && C % 4 == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(gamma) |
                                   reinterpret_cast<uintptr_t>(beta)) & 15) == 0 &&
        (reinterpret_cast<uintptr_t>(y) & 7) == 0) {
        const int rw = C <= 2048 ? 1 : (C <= 4096 ? 2 : 4);
        size_t grid = sei_ceil_div(rows, (size_t)(4 / rw));
        if (grid > 8192) grid = 8192;
#define SEI_LN_VEC16(RW)                                                                                          \
    hipLaunchKernelGGL(ln_fwd_bf16_vec_kernel<RW>, dim3((unsigned)grid), dim3(256), 0, s, x, gamma, beta, y, mean, \
                       rstd, rows, C, eps);                                                                        \
    return sei_launch_status();
        if (rw == 1) { SEI_LN_VEC16(1) }
        if (rw == 2) { SEI_LN_VEC16(2) }
        SEI_LN_VEC16(4)
#undef SEI_LN_VEC16
    }
    if (C > 512) {
        size_t grid = rows < 8192 ? rows : 8192;
        hipLaunchKernelGGL(ln_fwd_bf16_wide_kernel, dim3((unsigned)grid), dim3(256), 0, s, x, gamma, beta, y, mean, rstd,
                           rows, C, eps);
        return sei_launch_status();
    }
    if (C % 4 == 0 && ((C / 4) & (C / 4 - 1)) == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(gamma) |
                                                         reinterpret_cast<uintptr_t>(beta)) & 15) == 0 &&
        (reinterpret_cast<uintptr_t>(y) & 7) == 0) {
        const int q = C / 4, nv = q > 64 ? q / 64 : 1, lanes = q / nv;
        size_t grid = sei_ceil_div(rows, (size_t)(256 / lanes));
        if (grid > 2048) grid = 2048;
#define SEI_LN_QUAD(LL, NN)                                                                                          \
    hipLaunchKernelGGL((ln_fwd_bf16_quad_kernel<LL, NN>), dim3((unsigned)grid), dim3(256), 0, s, x, gamma, beta, y, mean, \
                       rstd, rows, eps);                                                                              \
    return sei_launch_status();
        switch (lanes * 100 + nv) {
            case 101: { SEI_LN_QUAD(1, 1) }
            case 201: { SEI_LN_QUAD(2, 1) }
            case 401: { SEI_LN_QUAD(4, 1) }
            case 801: { SEI_LN_QUAD(8, 1) }
            case 1601: { SEI_LN_QUAD(16, 1) }
            case 3201: { SEI_LN_QUAD(32, 1) }
            case 6401: { SEI_LN_QUAD(64, 1) }
            case 6402: { SEI_LN_QUAD(64, 2) }
            default: break;
        }
#undef SEI_LN_QUAD
    }
    int gsz = 1;
    while (gsz < 64 && gsz < C) gsz <<= 1;
    switch (gsz) {
        case 1: return launch_ln16<1>(x, gamma, beta, y, mean, rstd, rows, C, eps, s);
        case 2: return launch_ln16<2>(x, gamma, beta, y, mean, rstd, rows, C, eps, s);
        case 4: return launch_ln16<4>(x, gamma, beta, y, mean, rstd, rows, C, eps, s);
        case 8: return launch_ln16<8>(x, gamma, beta, y, mean, rstd, rows, C, eps, s);
        case 16: return launch_ln16<16>(x, gamma, beta, y, mean, rstd, rows, C, eps, s);
        case 32: return launch_ln16<32>(x, gamma, beta, y, mean, rstd, rows, C, eps, s);
        default: return launch_ln16<64>(x, gamma, beta, y, mean, rstd, rows, C, eps, s);
    }
}

extern "C" int sei_colsum_bf16(const uint16_t *X, float *out, size_t M, int N, void *stream) {
    SEI_REQUIRE(X && out && M > 0 && N > 0);
    if (N % 8 == 0 && (reinterpret_cast<uintptr_t>(X) & 15) == 0) {
        const int groups = N / 8;
        int tpr = 1;
        while (tpr < groups && tpr < 256) tpr <<= 1;              // threads per row (power of two <= 256)
        const unsigned col_blocks = (unsigned)sei_ceil_div(groups, tpr);
        // ~320 workgroups: with 64 bytes in flight per thread that saturates HBM, and every workgroup costs one
        // serialised atomic per column
        size_t rpb = (size_t)(256 / tpr) * 4;
        while (sei_ceil_div(M, rpb) * col_blocks > 320 && rpb < M) rpb *= 2;
        hipLaunchKernelGGL(colsum_bf16_vec_kernel, dim3((unsigned)sei_ceil_div(M, rpb), col_blocks), dim3(256), 0,
                           (hipStream_t)stream, X, out, M, N, tpr, rpb);
        return sei_launch_status();
    }
    const int cw = N < 256 ? N : 256;
    const unsigned col_blocks = (unsigned)sei_ceil_div(N, cw);
    size_t rpb = 16;
    while (sei_ceil_div(M, rpb) * col_blocks > 2048 && rpb < M) rpb *= 2;
    const size_t lds = sizeof(float) * (size_t)(256 / cw) * cw;
    hipLaunchKernelGGL(colsum_bf16_kernel, dim3((unsigned)sei_ceil_div(M, rpb), col_blocks), dim3(256), lds,
                       (hipStream_t)stream, X, out, M, N, rpb);
    return sei_launch_status();
}
