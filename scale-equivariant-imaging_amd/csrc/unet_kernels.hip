// U-Net streaming kernels for gfx950 (reference: src/models/convolutional.py), NHWC float32.
//
//   (the depthwise 7x7 of ConvBlock.conv1 lives in dwconv_kernels.hip)
//   sei_ln_fwd / sei_ln_bwd       : LayerNorm over channels, eps 1e-6, biased variance     (:21-30)
//   sei_conv3x3_fwd / _bwd_weight : UNet.in_conv / out_conv, 3x3 'same'                    (:174-176)
//   sei_sepmap2                   : IdealDownsample / IdealUpsample as L1 X R1^T + L2 X R2^T (:54-133)
//   sei_colsum_f32                : bias gradients of the 1x1 convolutions
//   sei_adam_fused                : torch.optim.Adam step on a flat bucket (demo/train.py:157-186)
//
// All are HBM-bound streaming kernels: channels are the contiguous axis, so lanes map to channels and
// every global access is a coalesced 256-B wave row. Small parameter gradients are accumulated with
// float atomics after an in-block LDS reduction (gradients are accumulators by contract).
#include "sei_common.h"

namespace {

// =================================================================================================
// LayerNorm over the channel (contiguous) axis
// =================================================================================================
template <int G>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int off = G / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

constexpr int LN_THREADS = 256;
constexpr int LN_EPL = 8;   // max elements per lane in the group kernels (C <= G*LN_EPL)

// G lanes per row, C <= 8*G. Row data lives in registers (one HBM read), statistics by shuffles.
template <int G>
__global__ __launch_bounds__(LN_THREADS) void ln_fwd_group_kernel(
    const float *__restrict__ x, const float *__restrict__ gamma, const float *__restrict__ beta,
    float *__restrict__ y, float *__restrict__ mean, float *__restrict__ rstd, size_t rows, int C, float eps) {
    const int lg = threadIdx.x % G, rsub = threadIdx.x / G;
    constexpr int RPB = LN_THREADS / G;
    const float invC = 1.0f / (float)C;
    for (size_t row = (size_t)blockIdx.x * RPB + rsub; row < rows; row += (size_t)gridDim.x * RPB) {
        const float *xr = x + row * C;
        float v[LN_EPL];
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < LN_EPL; ++e) {
            const int c = lg + e * G;
            v[e] = c < C ? xr[c] : 0.f;
            s += v[e];
        }
        const float mu = group_sum<G>(s) * invC;
        float q = 0.f;
#pragma unroll
        for (int e = 0; e < LN_EPL; ++e) {
            const int c = lg + e * G;
            const float d = c < C ? v[e] - mu : 0.f;
            q = fmaf(d, d, q);
        }
        const float rs = 1.0f / sqrtf(group_sum<G>(q) * invC + eps);
        float *yr = y + row * C;
#pragma unroll
        for (int e = 0; e < LN_EPL; ++e) {
            const int c = lg + e * G;
            if (c < C) yr[c] = fmaf((v[e] - mu) * rs, gamma[c], beta[c]);
        }
        if (lg == 0) {
            mean[row] = mu;
            rstd[row] = rs;
        }
    }
}

// One workgroup per row for wide rows: element c = tid + k*256, up to LN_WIDE_EPT per thread.
constexpr int LN_WIDE_EPT = 32;   // C <= 8192
__global__ __launch_bounds__(LN_THREADS) void ln_fwd_wide_kernel(
    const float *__restrict__ x, const float *__restrict__ gamma, const float *__restrict__ beta,
    float *__restrict__ y, float *__restrict__ mean, float *__restrict__ rstd, size_t rows, int C, float eps) {
    __shared__ float scratch[LN_THREADS / 64];
    __shared__ float bc[2];
    const float invC = 1.0f / (float)C;
    for (size_t row = blockIdx.x; row < rows; row += gridDim.x) {
        const float *xr = x + row * C;
        float v[LN_WIDE_EPT];
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < LN_WIDE_EPT; ++e) {
            const int c = threadIdx.x + e * LN_THREADS;
            v[e] = c < C ? xr[c] : 0.f;
            s += v[e];
        }
        s = sei_block_sum<LN_THREADS>(s, scratch);
        if (threadIdx.x == 0) bc[0] = s * invC;
        __syncthreads();
        const float mu = bc[0];
        float q = 0.f;
#pragma unroll
        for (int e = 0; e < LN_WIDE_EPT; ++e) {
            const int c = threadIdx.x + e * LN_THREADS;
            const float d = c < C ? v[e] - mu : 0.f;
            q = fmaf(d, d, q);
        }
        q = sei_block_sum<LN_THREADS>(q, scratch);
        if (threadIdx.x == 0) bc[1] = 1.0f / sqrtf(q * invC + eps);
        __syncthreads();
        const float rs = bc[1];
        float *yr = y + row * C;
#pragma unroll
        for (int e = 0; e < LN_WIDE_EPT; ++e) {
            const int c = threadIdx.x + e * LN_THREADS;
            if (c < C) yr[c] = fmaf((v[e] - mu) * rs, gamma[c], beta[c]);
        }
        if (threadIdx.x == 0) {
            mean[row] = mu;
            rstd[row] = rs;
        }
        __syncthreads();
    }
}

// backward: gx = rstd * (g - mean(g) - xhat * mean(g*xhat)), g = gy*gamma;
//           ggamma[c] += sum_rows gy*xhat; gbeta[c] += sum_rows gy.
template <int G>
__global__ __launch_bounds__(LN_THREADS) void ln_bwd_group_kernel(
    const float *__restrict__ x, const float *__restrict__ gamma, const float *__restrict__ mean,
    const float *__restrict__ rstd, const float *__restrict__ gy, float *__restrict__ gx,
    float *__restrict__ ggamma, float *__restrict__ gbeta, size_t rows, int C) {
    extern __shared__ __attribute__((aligned(16))) float smem[];   // [2][C]
    const int lg = threadIdx.x % G, rsub = threadIdx.x / G;
    constexpr int RPB = LN_THREADS / G;
    for (int e = threadIdx.x; e < 2 * C; e += LN_THREADS) smem[e] = 0.f;
    __syncthreads();
    const float invC = 1.0f / (float)C;
    float gam[LN_EPL], dg[LN_EPL], db[LN_EPL];
#pragma unroll
    for (int e = 0; e < LN_EPL; ++e) {
        const int c = lg + e * G;
        gam[e] = c < C ? gamma[c] : 0.f;
        dg[e] = 0.f;
        db[e] = 0.f;
    }
    for (size_t row = (size_t)blockIdx.x * RPB + rsub; row < rows; row += (size_t)gridDim.x * RPB) {
        const float mu = mean[row], rs = rstd[row];
        const float *xr = x + row * C, *gr = gy + row * C;
        float xh[LN_EPL], g[LN_EPL];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int e = 0; e < LN_EPL; ++e) {
            const int c = lg + e * G;
            const float gyv = c < C ? gr[c] : 0.f;
            xh[e] = c < C ? (xr[c] - mu) * rs : 0.f;
            g[e] = gyv * gam[e];
            s1 += g[e];
            s2 = fmaf(g[e], xh[e], s2);
            dg[e] = fmaf(gyv, xh[e], dg[e]);
            db[e] += gyv;
        }
        s1 = group_sum<G>(s1) * invC;
        s2 = group_sum<G>(s2) * invC;
        float *gxr = gx + row * C;
#pragma unroll
        for (int e = 0; e < LN_EPL; ++e) {
            const int c = lg + e * G;
            if (c < C) gxr[c] = rs * (g[e] - s1 - xh[e] * s2);
        }
    }
#pragma unroll
    for (int e = 0; e < LN_EPL; ++e) {
        const int c = lg + e * G;
        if (c < C) {
            atomicAdd(&smem[c], dg[e]);
            atomicAdd(&smem[C + c], db[e]);
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += LN_THREADS) {
        atomicAdd(ggamma + c, smem[c]);
        atomicAdd(gbeta + c, smem[C + c]);
    }
}

__global__ __launch_bounds__(LN_THREADS) void ln_bwd_wide_kernel(
    const float *__restrict__ x, const float *__restrict__ gamma, const float *__restrict__ mean,
    const float *__restrict__ rstd, const float *__restrict__ gy, float *__restrict__ gx,
    float *__restrict__ ggamma, float *__restrict__ gbeta, size_t rows, int C) {
    __shared__ float scratch[LN_THREADS / 64];
    __shared__ float bc[2];
    const float invC = 1.0f / (float)C;
    float gam[LN_WIDE_EPT], dg[LN_WIDE_EPT], db[LN_WIDE_EPT];
#pragma unroll
    for (int e = 0; e < LN_WIDE_EPT; ++e) {
        const int c = threadIdx.x + e * LN_THREADS;
        gam[e] = c < C ? gamma[c] : 0.f;
        dg[e] = 0.f;
        db[e] = 0.f;
    }
    for (size_t row = blockIdx.x; row < rows; row += gridDim.x) {
        const float mu = mean[row], rs = rstd[row];
        const float *xr = x + row * C, *gr = gy + row * C;
        float xh[LN_WIDE_EPT], g[LN_WIDE_EPT];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int e = 0; e < LN_WIDE_EPT; ++e) {
            const int c = threadIdx.x + e * LN_THREADS;
            const float gyv = c < C ? gr[c] : 0.f;
            xh[e] = c < C ? (xr[c] - mu) * rs : 0.f;
            g[e] = gyv * gam[e];
            s1 += g[e];
            s2 = fmaf(g[e], xh[e], s2);
            dg[e] = fmaf(gyv, xh[e], dg[e]);
            db[e] += gyv;
        }
        s1 = sei_block_sum<LN_THREADS>(s1, scratch);
        s2 = sei_block_sum<LN_THREADS>(s2, scratch);
        if (threadIdx.x == 0) {
            bc[0] = s1 * invC;
            bc[1] = s2 * invC;
        }
        __syncthreads();
        const float m1 = bc[0], m2 = bc[1];
        float *gxr = gx + row * C;
#pragma unroll
        for (int e = 0; e < LN_WIDE_EPT; ++e) {
            const int c = threadIdx.x + e * LN_THREADS;
            if (c < C) gxr[c] = rs * (g[e] - m1 - xh[e] * m2);
        }
        __syncthreads();
    }
#pragma unroll
    for (int e = 0; e < LN_WIDE_EPT; ++e) {
        const int c = threadIdx.x + e * LN_THREADS;
        if (c < C) {
            atomicAdd(ggamma + c, dg[e]);
            atomicAdd(gbeta + c, db[e]);
        }
    }
}

// ---- LayerNorm backward, vectorised (C % 4 == 0) -----------------------------------------------------------
// Parameter gradients are written as per-workgroup partial rows into a workspace and folded by
// ln_bwd_fold_kernel in a fixed order (no atomics, bitwise reproducible).
//
// narrow rows (C = 4*G*NV, G a power of two <= 64): G lanes own one row, NV float4 each; 256/G rows per sweep.
template <int G, int NV>
__global__ __launch_bounds__(LN_THREADS) void ln_bwd_vec_kernel(
    const float *__restrict__ x, const float *__restrict__ gamma, const float *__restrict__ mean,
    const float *__restrict__ rstd, const float *__restrict__ gy, const float *__restrict__ res,
    float *__restrict__ gx, float *__restrict__ part, size_t rows) {
    constexpr int C = 4 * G * NV, RPB = LN_THREADS / G;
    __shared__ __attribute__((aligned(16))) float red[RPB * 2 * C];
    const int lg = threadIdx.x % G, rsub = threadIdx.x / G;
    const float invC = 1.0f / (float)C;
    float4 gam[NV], dg[NV], db[NV];
#pragma unroll
    for (int e = 0; e < NV; ++e) {
        gam[e] = *reinterpret_cast<const float4 *>(gamma + 4 * (lg + e * G));
        dg[e] = db[e] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (size_t row = (size_t)blockIdx.x * RPB + rsub; row < rows; row += (size_t)gridDim.x * RPB) {
        const float mu = mean[row], rs = rstd[row];
        float4 xh[NV], g[NV];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int e = 0; e < NV; ++e) {
            const size_t o = row * C + 4 * (lg + e * G);
            const float4 xv = *reinterpret_cast<const float4 *>(x + o);
            const float4 gv = *reinterpret_cast<const float4 *>(gy + o);
#define SEI_LN_BWD_LANE(f)                              \
    xh[e].f = (xv.f - mu) * rs;                         \
    g[e].f = gv.f * gam[e].f;                           \
    s1 += g[e].f;                                       \
    s2 = fmaf(g[e].f, xh[e].f, s2);                     \
    dg[e].f = fmaf(gv.f, xh[e].f, dg[e].f);             \
    db[e].f += gv.f;
            SEI_LN_BWD_LANE(x) SEI_LN_BWD_LANE(y) SEI_LN_BWD_LANE(z) SEI_LN_BWD_LANE(w)
#undef SEI_LN_BWD_LANE
        }
        s1 = group_sum<G>(s1) * invC;
        s2 = group_sum<G>(s2) * invC;
#pragma unroll
        for (int e = 0; e < NV; ++e) {
            float4 o4;
            o4.x = rs * (g[e].x - s1 - xh[e].x * s2);
            o4.y = rs * (g[e].y - s1 - xh[e].y * s2);
            o4.z = rs * (g[e].z - s1 - xh[e].z * s2);
            o4.w = rs * (g[e].w - s1 - xh[e].w * s2);
            if (res) {                                           // a second gradient of the same tensor (skip connection)
                const float4 r4 = *reinterpret_cast<const float4 *>(res + row * C + 4 * (lg + e * G));
                o4.x += r4.x; o4.y += r4.y; o4.z += r4.z; o4.w += r4.w;
            }
            *reinterpret_cast<float4 *>(gx + row * C + 4 * (lg + e * G)) = o4;
        }
    }
#pragma unroll
    for (int e = 0; e < NV; ++e) {
        *reinterpret_cast<float4 *>(red + (rsub * 2 + 0) * C + 4 * (lg + e * G)) = dg[e];
        *reinterpret_cast<float4 *>(red + (rsub * 2 + 1) * C + 4 * (lg + e * G)) = db[e];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 2 * C; e += LN_THREADS) {
        float s = 0.f;
#pragma unroll 4
        for (int r = 0; r < RPB; ++r) s += red[r * 2 * C + e];
        part[(size_t)blockIdx.x * 2 * C + e] = s;
    }
}

// wide rows in ONE pass (C = 1024 NV, NV <= 8; round 5): a workgroup takes whole rows -- a thread owns NV float4 of the row,
// 1 KiB apart, so x and gy are read ONCE (the two-pass form below reads both twice: 565 MB per 113-MB tensor where this one
// moves 340 MB + the partials) --, folds the row's two sums over its four waves through LDS, writes gx, and keeps the
// parameter-gradient partials of ITS rows in registers: part[workgroup][2 C] for sei_fold_many / ln_bwd_fold_kernel.
// The next row's loads are issued before the current row's sums are exchanged.
template <int NV>
__global__ __launch_bounds__(LN_THREADS) void ln_bwd_row_kernel(
    const float *__restrict__ x, const float *__restrict__ gamma, const float *__restrict__ mean,
    const float *__restrict__ rstd, const float *__restrict__ gy, const float *__restrict__ res,
    float *__restrict__ gx, float *__restrict__ part, size_t rows) {
    constexpr int C = 1024 * NV;
    __shared__ float red[2][2][LN_THREADS / 64];               // [row parity][sum][wave]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float invC = 1.0f / (float)C;
    float4 gam[NV], dg[NV], db[NV];
#pragma unroll
    for (int e = 0; e < NV; ++e) {
        gam[e] = *reinterpret_cast<const float4 *>(gamma + 4 * (threadIdx.x + e * LN_THREADS));
        dg[e] = db[e] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float4 xv[NV], gv[NV], xn[NV], gn[NV];
    size_t row = blockIdx.x;
    if (row < rows) {
#pragma unroll
        for (int e = 0; e < NV; ++e) {
            const size_t o = row * C + 4 * (threadIdx.x + e * LN_THREADS);
            xn[e] = *reinterpret_cast<const float4 *>(x + o);
            gn[e] = *reinterpret_cast<const float4 *>(gy + o);
        }
    }
    int parity = 0;
    for (; row < rows; row += gridDim.x, parity ^= 1) {
#pragma unroll
        for (int e = 0; e < NV; ++e) {
            xv[e] = xn[e];
            gv[e] = gn[e];
        }
        const size_t next = row + gridDim.x;
        if (next < rows) {
#pragma unroll
            for (int e = 0; e < NV; ++e) {
                const size_t o = next * C + 4 * (threadIdx.x + e * LN_THREADS);
                xn[e] = *reinterpret_cast<const float4 *>(x + o);
                gn[e] = *reinterpret_cast<const float4 *>(gy + o);
            }
        }
        const float mu = mean[row], rs = rstd[row];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int e = 0; e < NV; ++e) {
#define SEI_LN_ROW_LANE(f)                               \
    xv[e].f = (xv[e].f - mu) * rs;                       \
    dg[e].f = fmaf(gv[e].f, xv[e].f, dg[e].f);           \
    db[e].f += gv[e].f;                                  \
    gv[e].f *= gam[e].f;                                 \
    s1 += gv[e].f;                                       \
    s2 = fmaf(gv[e].f, xv[e].f, s2);
            SEI_LN_ROW_LANE(x) SEI_LN_ROW_LANE(y) SEI_LN_ROW_LANE(z) SEI_LN_ROW_LANE(w)
#undef SEI_LN_ROW_LANE
        }
        s1 = sei_wave_sum(s1);
        s2 = sei_wave_sum(s2);
        if (lane == 0) {
            red[parity][0][wave] = s1;
            red[parity][1][wave] = s2;
        }
        __syncthreads();                                         // (the other parity's words are free: two barriers back)
        s1 = s2 = 0.f;
#pragma unroll
        for (int k = 0; k < LN_THREADS / 64; ++k) {
            s1 += red[parity][0][k];
            s2 += red[parity][1][k];
        }
        s1 *= invC;
        s2 *= invC;
#pragma unroll
        for (int e = 0; e < NV; ++e) {
            const size_t o = row * C + 4 * (threadIdx.x + e * LN_THREADS);
            float4 o4;
            o4.x = rs * (gv[e].x - s1 - xv[e].x * s2);
            o4.y = rs * (gv[e].y - s1 - xv[e].y * s2);
            o4.z = rs * (gv[e].z - s1 - xv[e].z * s2);
            o4.w = rs * (gv[e].w - s1 - xv[e].w * s2);
            if (res) {
                const float4 r4 = *reinterpret_cast<const float4 *>(res + o);
                o4.x += r4.x; o4.y += r4.y; o4.z += r4.z; o4.w += r4.w;
            }
            *reinterpret_cast<float4 *>(gx + o) = o4;
        }
    }
    float *out = part + (size_t)blockIdx.x * 2 * C;
#pragma unroll
    for (int e = 0; e < NV; ++e) {
        *reinterpret_cast<float4 *>(out + 4 * (threadIdx.x + e * LN_THREADS)) = dg[e];
        *reinterpret_cast<float4 *>(out + C + 4 * (threadIdx.x + e * LN_THREADS)) = db[e];
    }
}

// wide rows, pass 1: stats[row] = (mean_c(gy*gamma), mean_c(gy*gamma*xhat)); one workgroup per row.
__global__ __launch_bounds__(LN_THREADS) void ln_bwd_rowstats_kernel(
    const float *__restrict__ x, const float *__restrict__ gamma, const float *__restrict__ mean,
    const float *__restrict__ rstd, const float *__restrict__ gy, float2 *__restrict__ stats, int C) {
    __shared__ float scratch[2][LN_THREADS / 64];
    const size_t row = blockIdx.x;
    const float mu = mean[row], rs = rstd[row];
    float s1 = 0.f, s2 = 0.f;
    for (int c = 4 * threadIdx.x; c < C; c += 4 * LN_THREADS) {
        const float4 xv = *reinterpret_cast<const float4 *>(x + row * C + c);
        const float4 gv = *reinterpret_cast<const float4 *>(gy + row * C + c);
        const float4 gm = *reinterpret_cast<const float4 *>(gamma + c);
        float g;
        g = gv.x * gm.x; s1 += g; s2 = fmaf(g, (xv.x - mu) * rs, s2);
        g = gv.y * gm.y; s1 += g; s2 = fmaf(g, (xv.y - mu) * rs, s2);
        g = gv.z * gm.z; s1 += g; s2 = fmaf(g, (xv.z - mu) * rs, s2);
        g = gv.w * gm.w; s1 += g; s2 = fmaf(g, (xv.w - mu) * rs, s2);
    }
    s1 = sei_wave_sum(s1);
    s2 = sei_wave_sum(s2);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
        scratch[0][wave] = s1;
        scratch[1][wave] = s2;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int k = 0; k < LN_THREADS / 64; ++k) {
            a += scratch[0][k];
            b += scratch[1][k];
        }
        const float invC = 1.0f / (float)C;
        stats[row] = make_float2(a * invC, b * invC);
    }
}

// wide rows, pass 2: a thread owns 4 consecutive channels and walks a chunk of rows (row scalars are
// wave-uniform); gx elementwise, the parameter-gradient partial of the chunk in registers.
__global__ __launch_bounds__(LN_THREADS) void ln_bwd_cols_kernel(
    const float *__restrict__ x, const float *__restrict__ gamma, const float *__restrict__ mean,
    const float *__restrict__ rstd, const float *__restrict__ gy, const float2 *__restrict__ stats,
    const float *__restrict__ res, float *__restrict__ gx, float *__restrict__ part, size_t rows, int C,
    int rows_per_chunk) {
    const int c = 4 * (blockIdx.x * LN_THREADS + threadIdx.x);
    if (c >= C) return;
    const float4 gm = *reinterpret_cast<const float4 *>(gamma + c);
    float4 dg = make_float4(0.f, 0.f, 0.f, 0.f), db = dg;
    const size_t r0 = (size_t)blockIdx.y * rows_per_chunk, r1 = min(rows, r0 + rows_per_chunk);
#pragma unroll 4
    for (size_t row = r0; row < r1; ++row) {
        const float mu = mean[row], rs = rstd[row];
        const float2 st = stats[row];
        const float4 xv = *reinterpret_cast<const float4 *>(x + row * C + c);
        const float4 gv = *reinterpret_cast<const float4 *>(gy + row * C + c);
        float4 o4;
#define SEI_LN_COL_LANE(f)                                   \
    {                                                        \
        const float xh = (xv.f - mu) * rs;                   \
        o4.f = rs * (gv.f * gm.f - st.x - xh * st.y);        \
        dg.f = fmaf(gv.f, xh, dg.f);                         \
        db.f += gv.f;                                        \
    }
        SEI_LN_COL_LANE(x) SEI_LN_COL_LANE(y) SEI_LN_COL_LANE(z) SEI_LN_COL_LANE(w)
#undef SEI_LN_COL_LANE
        if (res) {
            const float4 r4 = *reinterpret_cast<const float4 *>(res + row * C + c);
            o4.x += r4.x; o4.y += r4.y; o4.z += r4.z; o4.w += r4.w;
        }
        *reinterpret_cast<float4 *>(gx + row * C + c) = o4;
    }
    float *out = part + (size_t)blockIdx.y * 2 * C;
    *reinterpret_cast<float4 *>(out + c) = dg;
    *reinterpret_cast<float4 *>(out + C + c) = db;
}

// sei_fold_many: the folds of MANY reducing kernels in one launch (the LayerNorm / depthwise weight gradients of a whole
// backward pass: 34 + 18 launches of ~5 us per U-Net step, ~146 per SwinIR step). A job is one destination with up to
// three partial-sum arrays (the model calls of the step that share the parameter), folded one after the other into the
// running value exactly as the separate launches did: same slices, same order, bit-identical. A workgroup owns 64
// consecutive entries of one job and finds it by walking the job table in the kernel arguments.
struct FoldManyArgs {
    SeiFoldJob job[SEI_FOLD_MAX_JOBS];
    int njobs;
};
static_assert(sizeof(FoldManyArgs) <= 4096, "the job table travels in the kernel-argument block");
// (round 5: 16 slices x 64 lanes per workgroup, a lane owning FOUR consecutive entries where the job's rows are float4-able
// (ncol % 4 == 0, 16-byte aligned partial rows: every job of the U-Net step) and one entry otherwise. With 16 entries per
// 256-thread workgroup every wave-instruction touched four partial rows for 64 bytes each -- half of every line it fetched
// -- and the launch spent its time starting ~400 k waves of three loads each: 144 MB per U-Net step at 1.4 TB/s. Slices,
// strides and the order of every addition are unchanged: results are bit-identical to the 16-entry form.)
constexpr int FOLD_LANES = 64, FOLD_SLICES = 16, FOLD_NQ = 4;
__device__ __forceinline__ bool fold_vec4(const SeiFoldJob &J) {
    bool ok = (J.ncol & 3) == 0;
    for (int sg = 0; sg < J.nseg; ++sg) ok = ok && (reinterpret_cast<uintptr_t>(J.part[sg]) & 15) == 0;
    return ok;
}
__device__ __forceinline__ float *fold_dst(const SeiFoldJob &J, int e) {
    if (J.kind == SEI_FOLD_DWCONV7) {                            // e = t C + c -> gw[c][t], t < 49; bias gradient behind
        const int t = e / J.split, c = e - t * J.split;
        return t < 49 ? J.a + (size_t)c * 49 + t : (J.b ? J.b + c : nullptr);
    }                                                            // a | b | c, `split` entries each (c may be absent)
    return e < J.split ? J.a + e : e < 2 * J.split ? J.b + (e - J.split) : (J.c ? J.c + (e - 2 * J.split) : nullptr);
}
__global__ __launch_bounds__(FOLD_SLICES * FOLD_LANES) void fold_many_kernel(FoldManyArgs g) {
    __shared__ __attribute__((aligned(16))) float red[FOLD_SLICES][4 * FOLD_LANES * FOLD_NQ];
    int j = 0, first = 0;
    bool vec = false;
    for (; j < g.njobs; ++j) {                                  // (uniform: scalar loads from the argument block)
        vec = fold_vec4(g.job[j]);
        const int per = vec ? 4 * FOLD_LANES * FOLD_NQ : FOLD_LANES;
        const int wgs = (g.job[j].ncol + per - 1) / per;
        if ((int)blockIdx.x < first + wgs) break;
        first += wgs;
    }
    if (j >= g.njobs) return;
    const SeiFoldJob &J = g.job[j];
    const int el = threadIdx.x % FOLD_LANES, slice = threadIdx.x / FOLD_LANES;
    const int ncol = J.ncol;
    if (vec) {
        // FOLD_NQ strips of 256 entries per workgroup, every strip's loads issued before the first sum: with one 16-byte
        // load per lane in flight (a dozen partial rows per depthwise job) two resident workgroups kept 32 KB per CU in
        // the air and the launch ran at 1.4 TB/s whatever the access shape
        const int e0 = ((int)blockIdx.x - first) * (4 * FOLD_LANES * FOLD_NQ) + 4 * el;     // strip q: e0 + 256 q
        float4 total = make_float4(0.f, 0.f, 0.f, 0.f);      // running value of the strip this wave finishes (slice < FOLD_NQ)
        auto add = [](float4 &a, const float4 b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; };
        for (int sg = 0; sg < J.nseg; ++sg) {
            const float *part = J.part[sg];
            const int groups = J.groups[sg];
            float4 s0[FOLD_NQ], s1[FOLD_NQ], s2[FOLD_NQ], s3[FOLD_NQ];
#pragma unroll
            for (int q = 0; q < FOLD_NQ; ++q) s0[q] = s1[q] = s2[q] = s3[q] = make_float4(0.f, 0.f, 0.f, 0.f);
            int p = slice;
            for (; p + 48 < groups; p += 64) {
#pragma unroll
                for (int q = 0; q < FOLD_NQ; ++q) {
                    const int e = e0 + 4 * FOLD_LANES * q;
                    if (e < ncol) {
                        add(s0[q], *reinterpret_cast<const float4 *>(part + (size_t)p * ncol + e));
                        add(s1[q], *reinterpret_cast<const float4 *>(part + (size_t)(p + 16) * ncol + e));
                        add(s2[q], *reinterpret_cast<const float4 *>(part + (size_t)(p + 32) * ncol + e));
                        add(s3[q], *reinterpret_cast<const float4 *>(part + (size_t)(p + 48) * ncol + e));
                    }
                }
            }
            for (; p < groups; p += 16) {
#pragma unroll
                for (int q = 0; q < FOLD_NQ; ++q) {
                    const int e = e0 + 4 * FOLD_LANES * q;
                    if (e < ncol) add(s0[q], *reinterpret_cast<const float4 *>(part + (size_t)p * ncol + e));
                }
            }
            __syncthreads();                                     // (the last segment's read of red)
#pragma unroll
            for (int q = 0; q < FOLD_NQ; ++q) {
                float4 r;
                r.x = (s0[q].x + s1[q].x) + (s2[q].x + s3[q].x); r.y = (s0[q].y + s1[q].y) + (s2[q].y + s3[q].y);
                r.z = (s0[q].z + s1[q].z) + (s2[q].z + s3[q].z); r.w = (s0[q].w + s1[q].w) + (s2[q].w + s3[q].w);
                *reinterpret_cast<float4 *>(&red[slice][4 * FOLD_LANES * q + 4 * el]) = r;
            }
            __syncthreads();
            // the 16 slices' sums of strip q are folded by wave q (FOLD_NQ <= 16), in slice order
            if (slice < FOLD_NQ) {
                const int e = e0 + 4 * FOLD_LANES * slice;
                if (e < ncol) {
                    float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int k = 0; k < FOLD_SLICES; ++k)
                        add(sum, *reinterpret_cast<const float4 *>(&red[k][4 * FOLD_LANES * slice + 4 * el]));
                    // the running value takes the segments one by one, as the separate launches added them
                    float *d0 = fold_dst(J, e), *d1 = fold_dst(J, e + 1), *d2 = fold_dst(J, e + 2), *d3 = fold_dst(J, e + 3);
                    float4 &t = total;
                    if (sg == 0) {
                        t.x = d0 ? *d0 : 0.f; t.y = d1 ? *d1 : 0.f; t.z = d2 ? *d2 : 0.f; t.w = d3 ? *d3 : 0.f;
                    }
                    add(t, sum);
                    if (sg == J.nseg - 1) {
                        if (d0) *d0 = t.x;
                        if (d1) *d1 = t.y;
                        if (d2) *d2 = t.z;
                        if (d3) *d3 = t.w;
                    }
                }
            }
        }
        return;
    }
    const int e = ((int)blockIdx.x - first) * FOLD_LANES + el;
    float total = 0.f;
    for (int sg = 0; sg < J.nseg; ++sg) {
        const float *part = J.part[sg];
        const int groups = J.groups[sg];
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        if (e < ncol) {
            int p = slice;
            for (; p + 48 < groups; p += 64) {
                s0 += part[(size_t)p * ncol + e];
                s1 += part[(size_t)(p + 16) * ncol + e];
                s2 += part[(size_t)(p + 32) * ncol + e];
                s3 += part[(size_t)(p + 48) * ncol + e];
            }
            for (; p < groups; p += 16) s0 += part[(size_t)p * ncol + e];
        }
        __syncthreads();                                         // (the last segment's read of red)
        red[slice][el] = (s0 + s1) + (s2 + s3);
        __syncthreads();
        if (slice == 0 && e < ncol) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < FOLD_SLICES; ++k) s += red[k][el];
            float *dst = fold_dst(J, e);
            if (dst) {
                if (sg == 0) total = *dst;
                total += s;
                if (sg == J.nseg - 1) *dst = total;
            }
        }
    }
}

// fold: ggamma[c] += sum_p part[p][c]; gbeta[c] += sum_p part[p][C + c]   (16 entries x 16 slices per workgroup)
__global__ __launch_bounds__(256) void ln_bwd_fold_kernel(const float *__restrict__ part, int nparts, int C,
                                                          float *__restrict__ ggamma,
                                                          float *__restrict__ gbeta) {
    __shared__ float red[16][16];
    const int el = threadIdx.x & 15, slice = threadIdx.x >> 4;
    const int e = blockIdx.x * 16 + el;
    const size_t stride = (size_t)2 * C;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (e < 2 * C) {
        int p = slice;
        for (; p + 48 < nparts; p += 64) {
            s0 += part[(size_t)p * stride + e];
            s1 += part[(size_t)(p + 16) * stride + e];
            s2 += part[(size_t)(p + 32) * stride + e];
            s3 += part[(size_t)(p + 48) * stride + e];
        }
        for (; p < nparts; p += 16) s0 += part[(size_t)p * stride + e];
    }
    red[slice][el] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (slice == 0 && e < 2 * C) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) s += red[k][el];
        if (e < C) ggamma[e] += s;
        else gbeta[e - C] += s;
    }
}

// =================================================================================================
// 3x3 convolution with small channel counts (in_conv 3->hidden, out_conv hidden->3)
// =================================================================================================
constexpr int C3_THREADS = 256;

__device__ __forceinline__ size_t img_index(int nchw, int b, int c, int i, int j, int C, int H, int W) {
    return nchw ? (((size_t)b * C + c) * H + i) * W + j : (((size_t)b * H + i) * W + j) * C + c;
}

// y[p, co] = bias[co] + sum_{ci,ky,kx} wq(co,ci,ky,kx) * x[p + (ky-1, kx-1), ci]  (+ res[p, co])
// transposed=0: wq = w[co][ci][ky][kx]                    (forward)
// transposed=1: wq = w[ci][co][2-ky][2-kx], w is (Cin_of_fwd=Cout here ... ) i.e. the data gradient:
//               the caller passes Cin = forward Cout and Cout = forward Cin.
__global__ __launch_bounds__(C3_THREADS) void conv3x3_kernel(
    const float *__restrict__ x, const float *__restrict__ w, const float *__restrict__ bias,
    const float *__restrict__ res, float *__restrict__ y, int B, int H, int W, int Cin, int Cout,
    int nchw_in, int nchw_out, int transposed) {
    extern __shared__ __attribute__((aligned(16))) float sw[];   // [ci][tap][co]
    const int nw = Cin * Cout * 9;
    for (int e = threadIdx.x; e < nw; e += C3_THREADS) {
        const int co = e % Cout, t = (e / Cout) % 9, ci = e / (Cout * 9);
        const int ky = t / 3, kx = t % 3;
        sw[e] = transposed ? w[(((size_t)ci * Cout + co) * 3 + (2 - ky)) * 3 + (2 - kx)]
                           : w[(((size_t)co * Cin + ci) * 3 + ky) * 3 + kx];
    }
    __syncthreads();
    const size_t total = (size_t)B * H * W * Cout;
    for (size_t idx = (size_t)blockIdx.x * C3_THREADS + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * C3_THREADS) {
        // thread order follows the OUTPUT layout so stores coalesce
        int b, i, j, co;
        if (nchw_out) {
            j = (int)(idx % W); i = (int)((idx / W) % H); co = (int)((idx / ((size_t)W * H)) % Cout);
            b = (int)(idx / ((size_t)W * H * Cout));
        } else {
            co = (int)(idx % Cout); j = (int)((idx / Cout) % W); i = (int)((idx / ((size_t)Cout * W)) % H);
            b = (int)(idx / ((size_t)Cout * W * H));
        }
        float a = bias ? bias[co] : 0.f;
        for (int ky = 0; ky < 3; ++ky) {
            const int ii = i + ky - 1;
            if (ii < 0 || ii >= H) continue;
            for (int kx = 0; kx < 3; ++kx) {
                const int jj = j + kx - 1;
                if (jj < 0 || jj >= W) continue;
                const int t = ky * 3 + kx;
                for (int ci = 0; ci < Cin; ++ci)
                    a = fmaf(sw[(ci * 9 + t) * Cout + co], x[img_index(nchw_in, b, ci, ii, jj, Cin, H, W)], a);
            }
        }
        const size_t o = img_index(nchw_out, b, co, i, j, Cout, H, W);
        if (res) a += res[o];
        y[o] = a;
    }
}

// gw[co][ci][ky][kx] += sum_p gy[p,co] * x[p + (ky-1,kx-1), ci];  gb[co] += sum_p gy[p,co]
// Each thread owns a set of (co,ci,tap) outputs and walks this block's pixel range.
__global__ __launch_bounds__(C3_THREADS) void conv3x3_bwd_weight_kernel(
    const float *__restrict__ x, const float *__restrict__ gy, float *__restrict__ gw,
    float *__restrict__ gb, int B, int H, int W, int Cin, int Cout, int nchw_x, int nchw_gy,
    int pix_per_block) {
    const size_t npix = (size_t)B * H * W;
    const size_t p0 = (size_t)blockIdx.x * pix_per_block;
    const size_t p1 = min(npix, p0 + pix_per_block);
    const int nw = Cin * Cout * 9;
    for (int e = threadIdx.x; e < nw + Cout; e += C3_THREADS) {
        float acc = 0.f;
        if (e < nw) {
            const int kx = e % 3, ky = (e / 3) % 3, ci = (e / 9) % Cin, co = e / (9 * Cin);
            for (size_t p = p0; p < p1; ++p) {
                const int j = (int)(p % W), i = (int)((p / W) % H), b = (int)(p / ((size_t)W * H));
                const int ii = i + ky - 1, jj = j + kx - 1;
                if (ii < 0 || ii >= H || jj < 0 || jj >= W) continue;
                acc = fmaf(gy[img_index(nchw_gy, b, co, i, j, Cout, H, W)],
                           x[img_index(nchw_x, b, ci, ii, jj, Cin, H, W)], acc);
            }
            atomicAdd(gw + e, acc);
        } else if (gb) {
            const int co = e - nw;
            for (size_t p = p0; p < p1; ++p) {
                const int j = (int)(p % W), i = (int)((p / W) % H), b = (int)(p / ((size_t)W * H));
                acc += gy[img_index(nchw_gy, b, co, i, j, Cout, H, W)];
            }
            atomicAdd(gb + co, acc);
        }
    }
}

// Pixel-per-thread form for the two real cases (3 -> hidden and hidden -> 3): one thread owns one output
// pixel and all COUT accumulators; every input value is loaded once and multiplied into the COUT
// accumulators with weights read as LDS broadcasts. Stores follow the output layout.
template <int COUT>
__global__ __launch_bounds__(C3_THREADS) void conv3x3_pix_kernel(
    const float *__restrict__ x, const float *__restrict__ w, const float *__restrict__ bias,
    const float *__restrict__ res, float *__restrict__ y, int B, int H, int W, int Cin, int nchw_in,
    int nchw_out, int transposed) {
    extern __shared__ __attribute__((aligned(16))) float sw[];   // [ci][tap][co]
    const int nw = Cin * COUT * 9;
    for (int e = threadIdx.x; e < nw; e += C3_THREADS) {
        const int co = e % COUT, t = (e / COUT) % 9, ci = e / (COUT * 9);
        const int ky = t / 3, kx = t % 3;
        sw[e] = transposed ? w[(((size_t)ci * COUT + co) * 3 + (2 - ky)) * 3 + (2 - kx)]
                           : w[(((size_t)co * Cin + ci) * 3 + ky) * 3 + kx];
    }
    __syncthreads();
    const size_t npix = (size_t)B * H * W;
    for (size_t p = (size_t)blockIdx.x * C3_THREADS + threadIdx.x; p < npix; p += (size_t)gridDim.x * C3_THREADS) {
        const int j = (int)(p % W), i = (int)((p / W) % H), b = (int)(p / ((size_t)W * H));
        float acc[COUT];
#pragma unroll
        for (int co = 0; co < COUT; ++co) acc[co] = bias ? bias[co] : 0.f;
        for (int ky = 0; ky < 3; ++ky) {
            const int ii = i + ky - 1;
            if (ii < 0 || ii >= H) continue;
            for (int kx = 0; kx < 3; ++kx) {
                const int jj = j + kx - 1;
                if (jj < 0 || jj >= W) continue;
                const int t = ky * 3 + kx;
                if (nchw_in) {
                    const float *xp = x + ((size_t)b * Cin * H + ii) * W + jj;
                    for (int ci = 0; ci < Cin; ++ci) {
                        const float v = xp[(size_t)ci * H * W];
                        const float *wr = sw + (ci * 9 + t) * COUT;
#pragma unroll
                        for (int co = 0; co < COUT; ++co) acc[co] = fmaf(wr[co], v, acc[co]);
                    }
                } else {
                    const float *xp = x + (((size_t)b * H + ii) * W + jj) * Cin;
                    for (int ci = 0; ci < Cin; ++ci) {
                        const float v = xp[ci];
                        const float *wr = sw + (ci * 9 + t) * COUT;
#pragma unroll
                        for (int co = 0; co < COUT; ++co) acc[co] = fmaf(wr[co], v, acc[co]);
                    }
                }
            }
        }
        if (nchw_out) {
#pragma unroll
            for (int co = 0; co < COUT; ++co) {
                const size_t o = (((size_t)b * COUT + co) * H + i) * W + j;
                y[o] = res ? acc[co] + res[o] : acc[co];
            }
        } else {
            float *yp = y + p * COUT;
            const float *rp = res ? res + p * COUT : nullptr;
#pragma unroll
            for (int co = 0; co < COUT; ++co) yp[co] = rp ? acc[co] + rp[co] : acc[co];
        }
    }
}

// Image (CS <= 4 channels, either layout) -> the 32 hidden channels in NHWC, and (transposed = 1) the data gradient of a
// hidden -> image convolution: FOUR lanes per pixel, eight output channels each. The pixel-per-thread form above stores
// its 32 values as 32 four-byte stores 128 bytes apart from lane to lane (0.06 of the HBM rate, all of it store issue);
// here a pixel's 128 output bytes leave as 4 x 2 sixteen-byte stores of neighbouring lanes, the 9 x CS input values of
// a pixel are loads shared by its four lanes, the weights are LDS reads of [ci][tap][32] rows (two float4 per lane).
template <int CS>
__global__ __launch_bounds__(C3_THREADS) void conv3x3_small_to_c32_kernel(
    const float *__restrict__ x, const float *__restrict__ w, const float *__restrict__ bias,
    const float *__restrict__ res, float *__restrict__ y, int B, int H, int W, int nchw_in, int transposed) {
    __shared__ __attribute__((aligned(16))) float sw[CS * 9 * 32];   // [ci][tap][co]
    for (int e = threadIdx.x; e < CS * 9 * 32; e += C3_THREADS) {
        const int co = e & 31, t = (e >> 5) % 9, ci = e / (32 * 9);
        sw[e] = transposed ? w[((size_t)ci * 32 + co) * 9 + (8 - t)] : w[((size_t)co * CS + ci) * 9 + t];
    }
    __syncthreads();
    const int q = threadIdx.x & 3;
    float4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = b0;
    if (bias) {
        b0 = *reinterpret_cast<const float4 *>(bias + 8 * q);
        b1 = *reinterpret_cast<const float4 *>(bias + 8 * q + 4);
    }
    const size_t npix = (size_t)B * H * W;
    for (size_t p = ((size_t)blockIdx.x * C3_THREADS + threadIdx.x) >> 2; p < npix; p += ((size_t)gridDim.x * C3_THREADS) >> 2) {
        const int j = (int)(p % W), i = (int)((p / W) % H), b = (int)(p / ((size_t)W * H));
        float4 a0 = b0, a1 = b1;
        // (round 5: ONE 64-bit pixel address, then 32-bit tap / channel offsets -- img_index() per tap and channel was three
        // 64-bit multiplies each: a PMC pass counted 697 VALU instructions per wave around its 216 FMAs)
        const float *px = x + (nchw_in ? ((size_t)b * CS * H + i) * W + j : (((size_t)b * H + i) * W + j) * CS);
        const int pstr = nchw_in ? 1 : CS, cstr = nchw_in ? H * W : 1;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int ii = i + ky - 1;
            if (ii < 0 || ii >= H) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int jj = j + kx - 1;
                if (jj < 0 || jj >= W) continue;
                const float *tp = px + ((ky - 1) * W + (kx - 1)) * pstr;
#pragma unroll
                for (int ci = 0; ci < CS; ++ci) {
                    const float v = tp[ci * cstr];
                    const float4 w0 = *reinterpret_cast<const float4 *>(sw + (ci * 9 + ky * 3 + kx) * 32 + 8 * q);
                    const float4 w1 = *reinterpret_cast<const float4 *>(sw + (ci * 9 + ky * 3 + kx) * 32 + 8 * q + 4);
                    a0.x = fmaf(w0.x, v, a0.x); a0.y = fmaf(w0.y, v, a0.y); a0.z = fmaf(w0.z, v, a0.z); a0.w = fmaf(w0.w, v, a0.w);
                    a1.x = fmaf(w1.x, v, a1.x); a1.y = fmaf(w1.y, v, a1.y); a1.z = fmaf(w1.z, v, a1.z); a1.w = fmaf(w1.w, v, a1.w);
                }
            }
        }
        float *yp = y + p * 32 + 8 * q;
        if (res) {
            const float4 r0 = *reinterpret_cast<const float4 *>(res + p * 32 + 8 * q);
            const float4 r1 = *reinterpret_cast<const float4 *>(res + p * 32 + 8 * q + 4);
            a0.x += r0.x; a0.y += r0.y; a0.z += r0.z; a0.w += r0.w;
            a1.x += r1.x; a1.y += r1.y; a1.z += r1.z; a1.w += r1.w;
        }
        *reinterpret_cast<float4 *>(yp) = a0;
        *reinterpret_cast<float4 *>(yp + 4) = a1;
    }
}

// Weight gradient, tiled: a workgroup stages P pixels of gy (P x Cout) and of the im2col'ed input
// (P x Cin*9) in LDS, each thread owns a few of the Cout*Cin*9 outputs and reduces over the P pixels
// out of LDS; one float atomic per output per workgroup.
__global__ __launch_bounds__(C3_THREADS) void conv3x3_bwd_weight_tiled_kernel(
    const float *__restrict__ x, const float *__restrict__ gy, float *__restrict__ gw,
    float *__restrict__ gb, int B, int H, int W, int Cin, int Cout, int nchw_x, int nchw_gy, int P,
    int tiles_per_block) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int KC = Cin * 9;
    float *sG = sm;                 // [P][Cout]
    float *sX = sm + P * Cout;      // [P][KC]
    const size_t npix = (size_t)B * H * W;
    const int nout = Cout * KC;
    constexpr int MAXO = 8;         // outputs per thread (nout <= 8*256)
    float acc[MAXO];
#pragma unroll
    for (int q = 0; q < MAXO; ++q) acc[q] = 0.f;
    float accb = 0.f;
    for (int tile = 0; tile < tiles_per_block; ++tile) {
        const size_t p0 = ((size_t)blockIdx.x * tiles_per_block + tile) * P;
        if (p0 >= npix) break;
        const int np = (int)min((size_t)P, npix - p0);
        __syncthreads();
        for (int e = threadIdx.x; e < np * Cout; e += C3_THREADS) {
            const int pp = e / Cout, co = e % Cout;
            const size_t p = p0 + pp;
            const int j = (int)(p % W), i = (int)((p / W) % H), b = (int)(p / ((size_t)W * H));
            sG[e] = gy[img_index(nchw_gy, b, co, i, j, Cout, H, W)];
        }
        for (int e = threadIdx.x; e < np * KC; e += C3_THREADS) {
            const int pp = e / KC, k = e % KC;
            const int ci = k / 9, t = k % 9;
            const size_t p = p0 + pp;
            const int j = (int)(p % W), i = (int)((p / W) % H), b = (int)(p / ((size_t)W * H));
            const int ii = i + t / 3 - 1, jj = j + t % 3 - 1;
            sX[e] = (ii >= 0 && ii < H && jj >= 0 && jj < W) ? x[img_index(nchw_x, b, ci, ii, jj, Cin, H, W)] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < MAXO; ++q) {
            const int e = threadIdx.x + q * C3_THREADS;
            if (e < nout) {
                const int co = e / KC, k = e % KC;
                float a = acc[q];
                for (int pp = 0; pp < np; ++pp) a = fmaf(sG[pp * Cout + co], sX[pp * KC + k], a);
                acc[q] = a;
            }
        }
        if (gb && threadIdx.x < Cout)
            for (int pp = 0; pp < np; ++pp) accb += sG[pp * Cout + threadIdx.x];
    }
#pragma unroll
    for (int q = 0; q < MAXO; ++q) {
        const int e = threadIdx.x + q * C3_THREADS;
        if (e < nout) atomicAdd(gw + e, acc[q]);      // e = (co*Cin + ci)*9 + t: the torch weight layout
    }
    if (gb && threadIdx.x < Cout) atomicAdd(gb + threadIdx.x, accb);
}

// -------------------------------------------------------------------------------------------------
// Lane-per-channel forms for the default network (hidden = 32): the 32 lanes of a half-wave are the 32
// hidden channels, so every access to the NHWC hidden tensor is a coalesced 128-byte row (the pixel-per-thread
// kernels above read it with a 128-byte stride ACROSS lanes: 64 cache lines per load). A half-wave walks a
// run of C3L_RUN pixels of one image row with a 3x3 register window sliding by one column per pixel.
// -------------------------------------------------------------------------------------------------
constexpr int C3L_RUN = 16;

__device__ __forceinline__ float half_wave_sum(float v) {        // over the 32 lanes of this half-wave
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// forward hidden(32, NHWC) -> CS <= 4 channels: y[p, co] = bias[co] + sum_{t, ci} w[co][ci][t] x[p + t, ci] (+ res);
// transposed = 1: the data gradient of a CS -> 32 convolution (w is THAT convolution's (32, CS, 3, 3) weight, read with
// flipped taps and swapped channel roles): gx[p, c] = sum_{t, co} w[co][c][8 - t] gy[p + t, co]
template <int CS>
__global__ __launch_bounds__(C3_THREADS) void conv3x3_c32_to_small_kernel(
    const float *__restrict__ x, const float *__restrict__ w, const float *__restrict__ bias,
    const float *__restrict__ res, float *__restrict__ y, int B, int H, int W, int nchw_out, int nruns_row,
    int total_runs, int transposed) {
    const int ci = threadIdx.x & 31, grp = threadIdx.x >> 5;
    float wr[CS][9];
#pragma unroll
    for (int co = 0; co < CS; ++co)
#pragma unroll
        for (int t = 0; t < 9; ++t)
            wr[co][t] = transposed ? w[((size_t)ci * CS + co) * 9 + (8 - t)] : w[((size_t)co * 32 + ci) * 9 + t];
    for (int run = blockIdx.x * (C3_THREADS / 32) + grp; run < total_runs; run += gridDim.x * (C3_THREADS / 32)) {
        const int jr = run % nruns_row, bi = run / nruns_row;
        const int i = bi % H, b = bi / H;
        const int j0 = jr * C3L_RUN, jn = min(C3L_RUN, W - j0);
        const float *xb = x + (size_t)b * H * W * 32 + ci;
        auto load = [&](int ii, int jj) -> float {
            return (ii >= 0 && ii < H && jj >= 0 && jj < W) ? xb[((size_t)ii * W + jj) * 32] : 0.f;
        };
        float win[3][3];                                  // win[ky][slot], slot rotates with the column
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            win[ky][0] = load(i + ky - 1, j0 - 1);
            win[ky][1] = load(i + ky - 1, j0);
        }
        // this lane's (input channel's) share of every output of the run first: part[jl * CS + co]
        float part[C3L_RUN * CS];
#pragma unroll
        for (int jb = 0; jb < C3L_RUN + 2; jb += 3) {
#pragma unroll
            for (int s3 = 0; s3 < 3; ++s3) {
                const int jl = jb + s3;
                if (jl < C3L_RUN) {
                    if (jl < jn) {
#pragma unroll
                        for (int ky = 0; ky < 3; ++ky) win[ky][(s3 + 2) % 3] = load(i + ky - 1, j0 + jl + 1);
                    }
#pragma unroll
                    for (int co = 0; co < CS; ++co) {
                        float a = 0.f;
#pragma unroll
                        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                            for (int kx = 0; kx < 3; ++kx) a = fmaf(wr[co][ky * 3 + kx], win[ky][(s3 + kx) % 3], a);
                        part[jl * CS + co] = a;
                    }
                }
            }
        }
        // ... then ONE transposing butterfly over the 32 lanes for all C3L_RUN * CS sums (round 5): at offset 16, 8, 4, 2 a
        // lane keeps one half of its values and hands the other half to its partner, so the value count halves with the
        // lane distance -- 8 CS + 4 CS + 2 CS + CS exchanges and a last plain stage of CS instead of 5 per sum (240 -> 48 at
        // CS = 3; a PMC pass had 30 of the 45 VALU and all 15 LDS instructions per pixel step in the per-sum butterflies).
        // Every sum is added in the same tree as before: bit-identical. Afterwards lanes 2 jl and 2 jl + 1 both hold the CS
        // outputs of pixel jl.
#define SEI_C3_STAGE(OFF, NV)                                                                   \
        {                                                                                       \
            const bool up = (ci & (OFF)) != 0;                                                  \
            _Pragma("unroll") for (int k = 0; k < (NV) / 2; ++k) {                              \
                const float send = up ? part[k] : part[k + (NV) / 2];                           \
                const float keep = up ? part[k + (NV) / 2] : part[k];                           \
                part[k] = keep + __shfl_xor(send, (OFF), 64);                                   \
            }                                                                                   \
        }
        SEI_C3_STAGE(16, C3L_RUN * CS)
        SEI_C3_STAGE(8, C3L_RUN * CS / 2)
        SEI_C3_STAGE(4, C3L_RUN * CS / 4)
        SEI_C3_STAGE(2, C3L_RUN * CS / 8)
#undef SEI_C3_STAGE
#pragma unroll
        for (int co = 0; co < CS; ++co) part[co] += __shfl_xor(part[co], 1, 64);
        {
            const int jl = ci >> 1;                       // this lane pair's pixel of the run
            if (jl < jn) {
#pragma unroll
                for (int co = 0; co < CS; ++co) {
                    if ((co & 1) != (ci & 1)) continue;   // the pair's two lanes share the CS stores
                    const size_t o = img_index(nchw_out, b, co, i, j0 + jl, CS, H, W);
                    const float v = part[co] + (bias ? bias[co] : 0.f);
                    y[o] = res ? v + res[o] : v;
                }
            }
        }
    }
}

// weight gradient with the 32-channel tensor on the lanes:
//   SMALL_IS_OUT = true : x hidden (NHWC, 32), gy small (CS channels, either layout): gw[co][lane][t], lane = ci
//   SMALL_IS_OUT = false: gy hidden (NHWC, 32), x small (CS channels, either layout): gw[lane][ci][t], lane = co
// Per workgroup: 8 half-wave partials folded through LDS, then one float atomic per output.
template <int CS, bool SMALL_IS_OUT>
__global__ __launch_bounds__(C3_THREADS) void conv3x3_wgrad_c32_kernel(
    const float *__restrict__ x, const float *__restrict__ gy, float *__restrict__ gw, float *__restrict__ gb,
    int B, int H, int W, int nchw_small, int nruns_row, int total_runs) {
    __shared__ float red[C3_THREADS / 32][CS * 9 + CS][32];
    const int lane = threadIdx.x & 31, grp = threadIdx.x >> 5;
    float acc[CS][9];
#pragma unroll
    for (int c = 0; c < CS; ++c)
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[c][t] = 0.f;
    float accb[CS];                                      // bias gradient: SMALL_IS_OUT -> per co (lane 0 only);
#pragma unroll                                           //                else -> accb[0] per lane (= co)
    for (int c = 0; c < CS; ++c) accb[c] = 0.f;
    const float *big = SMALL_IS_OUT ? x : gy, *small = SMALL_IS_OUT ? gy : x;
    for (int run = blockIdx.x * (C3_THREADS / 32) + grp; run < total_runs; run += gridDim.x * (C3_THREADS / 32)) {
        const int jr = run % nruns_row, bi = run / nruns_row;
        const int i = bi % H, b = bi / H;
        const int j0 = jr * C3L_RUN, jn = min(C3L_RUN, W - j0);
        if (SMALL_IS_OUT) {
            // window of x (hidden, per lane); gy values of the pixel are uniform over the lanes
            const float *xb = big + (size_t)b * H * W * 32 + lane;
            auto load = [&](int ii, int jj) -> float {
                return (ii >= 0 && ii < H && jj >= 0 && jj < W) ? xb[((size_t)ii * W + jj) * 32] : 0.f;
            };
            float win[3][3];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                win[ky][0] = load(i + ky - 1, j0 - 1);
                win[ky][1] = load(i + ky - 1, j0);
            }
#pragma unroll
            for (int jb = 0; jb < C3L_RUN + 2; jb += 3) {
#pragma unroll
                for (int s3 = 0; s3 < 3; ++s3) {
                    const int jl = jb + s3;
                    if (jl < C3L_RUN && jl < jn) {
#pragma unroll
                        for (int ky = 0; ky < 3; ++ky) win[ky][(s3 + 2) % 3] = load(i + ky - 1, j0 + jl + 1);
#pragma unroll
                        for (int co = 0; co < CS; ++co) {
                            const float g = small[img_index(nchw_small, b, co, i, j0 + jl, CS, H, W)];
                            accb[co] += g;
#pragma unroll
                            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                                for (int kx = 0; kx < 3; ++kx)
                                    acc[co][ky * 3 + kx] = fmaf(g, win[ky][(s3 + kx) % 3], acc[co][ky * 3 + kx]);
                        }
                    }
                }
            }
        } else {
            // gy (hidden, per lane = co) at the pixel; the 3x3xCS window of x is uniform over the lanes
            const float *gb_ = big + ((size_t)(b * H + i) * W) * 32 + lane;
            auto load = [&](int c, int ii, int jj) -> float {
                return (ii >= 0 && ii < H && jj >= 0 && jj < W) ? small[img_index(nchw_small, b, c, ii, jj, CS, H, W)]
                                                                : 0.f;
            };
            float win[CS][3][3];
#pragma unroll
            for (int c = 0; c < CS; ++c)
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    win[c][ky][0] = load(c, i + ky - 1, j0 - 1);
                    win[c][ky][1] = load(c, i + ky - 1, j0);
                }
#pragma unroll
            for (int jb = 0; jb < C3L_RUN + 2; jb += 3) {
#pragma unroll
                for (int s3 = 0; s3 < 3; ++s3) {
                    const int jl = jb + s3;
                    if (jl < C3L_RUN && jl < jn) {
#pragma unroll
                        for (int c = 0; c < CS; ++c)
#pragma unroll
                            for (int ky = 0; ky < 3; ++ky) win[c][ky][(s3 + 2) % 3] = load(c, i + ky - 1, j0 + jl + 1);
                        const float g = gb_[(size_t)(j0 + jl) * 32];
                        accb[0] += g;
#pragma unroll
                        for (int c = 0; c < CS; ++c)
#pragma unroll
                            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                                for (int kx = 0; kx < 3; ++kx)
                                    acc[c][ky * 3 + kx] = fmaf(g, win[c][ky][(s3 + kx) % 3], acc[c][ky * 3 + kx]);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int c = 0; c < CS; ++c) {
#pragma unroll
        for (int t = 0; t < 9; ++t) red[grp][c * 9 + t][lane] = acc[c][t];
        red[grp][CS * 9 + c][lane] = accb[c];
    }
    __syncthreads();
    // consecutive threads own consecutive gw addresses (torch layout gw[co][ci][t]): contiguous float atomics --
    // lane-strided ones cost one L2 round per instruction and cache line, serialised over all workgroups
    constexpr int NOUT = CS * 32 * 9;
    for (int o = threadIdx.x; o < NOUT; o += C3_THREADS) {
        const int t = o % 9, rest = o / 9;
        int c, l;                                         // c: the small channel, l: the lane (hidden channel)
        if (SMALL_IS_OUT) {                               // o = (co * 32 + ci) * 9 + t, co = c, ci = l
            l = rest & 31;
            c = rest >> 5;
        } else {                                          // o = (co * CS + ci) * 9 + t, co = l, ci = c
            c = rest % CS;
            l = rest / CS;
        }
        float sum = 0.f;
#pragma unroll
        for (int gI = 0; gI < C3_THREADS / 32; ++gI) sum += red[gI][c * 9 + t][l];
        atomicAdd(gw + o, sum);
    }
    if (gb) {
        const int nb = SMALL_IS_OUT ? CS : 32;
        if ((int)threadIdx.x < nb) {
            float sum = 0.f;
#pragma unroll
            for (int gI = 0; gI < C3_THREADS / 32; ++gI)
                sum += SMALL_IS_OUT ? red[gI][CS * 9 + threadIdx.x][0] : red[gI][CS * 9][threadIdx.x];
            atomicAdd(gb + threadIdx.x, sum);
        }
    }
}

// Weight gradients of the network's two end convolutions on the matrix cores, exact float32 (round 5).
//   hidden tensor BIG (NHWC, 32 channels), small tensor SMALL (CS <= 3 channels, either layout):
//     SMALL_IS_OUT = false (in_conv, x = SMALL, gy = BIG):  gw[co][ci][t] = sum_p BIG[p][co] SMALL[p + off(t)][ci]
//     SMALL_IS_OUT = true  (out_conv, x = BIG, gy = SMALL): gw[co][ci][t] = sum_q BIG[q][ci] SMALL[q - off(t)][co]
//   i.e. one 32 x 32 product  G[m][n] = sum_pixels BIG[pixel][m] P[pixel][n]  with n = (small channel, tap) <= 27 columns of
//   shifted SMALL values (zero outside the image), reduced over all B H W pixels: v_mfma_f32_32x32x2_f32 with two pixels per
//   instruction -- lane (m = lane % 32, k = lane / 32) reads BIG as whole 128-byte pixel rows, lane (n, k) gathers its own
//   shifted SMALL value (a few MB: L2). Column 27 of P is 1 for in_conv: the bias gradient sum_p gy[p][co] falls out of the
//   same product; for out_conv it is the sum of the centre-tap column's values, kept per lane.
// The lane-per-channel kernels above spent 61-68 us per launch (27 FMAs + ~10 loads per pixel and half-wave, 512 workgroups
// adding 864 atomics each onto the same 864 addresses); here a wave owns a contiguous pixel range, a workgroup folds its eight
// accumulator tiles through LDS and 256 workgroups add.
// part != NULL: no atomics -- workgroup g leaves its sums as row g of part ([workgroups][9 * 32 * CS + bias entries], the
// weight gradient in torch's layout followed by the bias gradient) for sei_fold_many (SEI_FOLD_SPLIT, split = 288 CS).
template <bool SMALL_IS_OUT>
__global__ __launch_bounds__(512) void conv3x3_wgrad_mfma_kernel(const float *__restrict__ big, const float *__restrict__ sm,
                                                                 float *__restrict__ gw, float *__restrict__ gb, int B, int H,
                                                                 int W, int CS, int nchw_small, int pix_per_wave,
                                                                 float *__restrict__ part) {
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    __shared__ float red[8][1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = lane & 31, kk = lane >> 5;                    // A: row m = n; B: column n; both: pixel k = kk of the pair
    // this lane's column of P: small channel cn and tap (dy, dx); column 27: ones (in_conv's bias gradient); above: zeros
    const int cn = n / 9, tn = n - 9 * cn;
    const bool col_real = n < 9 * CS, col_ones = !SMALL_IS_OUT && n == 27;
    const int sgn = SMALL_IS_OUT ? -1 : 1;
    const int dy = sgn * (tn / 3 - 1), dx = sgn * (tn % 3 - 1);
    const long long npix = (long long)B * H * W;
    const long long p_begin = ((long long)blockIdx.x * 8 + wave) * pix_per_wave;
    const long long p_end = p_begin + pix_per_wave < npix ? p_begin + pix_per_wave : npix;
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    float bsum = 0.f;                                            // out_conv: the centre-tap column's running sum
    constexpr int U = 8;                                         // MFMAs (pixel pairs) per batch of loads
    // this lane's pixel (b, i, j): decomposed ONCE, then advanced by two per MFMA (a 64-bit division per pixel and lane made
    // the first version of this kernel VALU-bound: 47 us)
    long long p = p_begin + kk;
    int j = (int)(p % W);
    long long rest = p / W;
    int i = (int)(rest % H), b = (int)(rest / H);
    auto load_batch = [&](float (&av)[U], float (&bv)[U]) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool live = p < p_end;
            av[u] = live ? big[(size_t)p * 32 + n] : 0.f;
            const int ii = i + dy, jj = j + dx;
            const bool inside = live && col_real && ii >= 0 && ii < H && jj >= 0 && jj < W;
            const float v = inside ? sm[img_index(nchw_small, b, cn, ii, jj, CS, H, W)] : 0.f;
            bv[u] = (col_ones && live) ? 1.f : v;
            if (SMALL_IS_OUT && tn == 4) bsum += v;              // (dy, dx) = (0, 0): SMALL[q][cn] itself
            p += 2;
            j += 2;
            while (j >= W) {                                     // (W = 1: two rows per step)
                j -= W;
                if (++i == H) {
                    i = 0;
                    ++b;
                }
            }
        }
    };
    auto mfma_batch = [&](const float (&av)[U], const float (&bv)[U]) {
#pragma unroll
        for (int u = 0; u < U; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u], acc, 0, 0, 0);
    };
    // two batches in registers: the loads of the next 16 pixels are in flight under the MFMAs of these (a wave has ~7
    // batches in all: with load -> wait -> multiply in sequence it spent its time waiting: 32 us per launch)
    float a0[U], b0[U], a1[U], b1[U];
    load_batch(a0, b0);
    for (long long p0 = p_begin; p0 < p_end; p0 += 4 * U) {
        load_batch(a1, b1);                                      // (past the end: all lanes dead, zeros)
        mfma_batch(a0, b0);
        load_batch(a0, b0);
        mfma_batch(a1, b1);
    }
    // accumulator element e of lane (n, kk): G[m = 8 (e / 4) + 4 kk + e % 4][n]
#pragma unroll
    for (int e = 0; e < 16; ++e) red[wave][(8 * (e >> 2) + 4 * kk + (e & 3)) * 32 + n] = acc[e];
    __shared__ float bred[8][64];
    bred[wave][lane] = bsum;
    __syncthreads();
    for (int o = threadIdx.x; o < 1024; o += 512) {
        const int m = o >> 5, nn = o & 31;
        float t = 0.f;
#pragma unroll
        for (int w8 = 0; w8 < 8; ++w8) t += red[w8][o];
        const int nw = 288 * CS;                                 // entries of gw
        float *row = part ? part + (size_t)blockIdx.x * (nw + (SMALL_IS_OUT ? CS : 32)) : nullptr;
        if (nn < 9 * CS) {
            const int c = nn / 9, tap = nn - 9 * c;
            // in_conv: gw[co = m][ci = c][tap]; out_conv: gw[co = c][ci = m][tap]
            const size_t at = SMALL_IS_OUT ? ((size_t)c * 32 + m) * 9 + tap : ((size_t)m * CS + c) * 9 + tap;
            if (row) row[at] = t;
            else atomicAdd(gw + at, t);
        } else if (!SMALL_IS_OUT && nn == 27) {
            if (row) row[nw + m] = t;
            else if (gb) atomicAdd(gb + m, t);
        }
    }
    if (SMALL_IS_OUT && (int)threadIdx.x < CS) {                 // centre-tap lanes: n = 9 c + 4, both pixel halves, all waves
        float t = 0.f;
        const int c = threadIdx.x;
#pragma unroll
        for (int w8 = 0; w8 < 8; ++w8) t += bred[w8][9 * c + 4] + bred[w8][32 + 9 * c + 4];
        if (part) part[(size_t)blockIdx.x * (288 * CS + CS) + 288 * CS + c] = t;
        else if (gb) atomicAdd(gb + c, t);
    }
}

// =================================================================================================
// separable rank-2 spatial map, NHWC:  y[b,i',j',c] = sum_t sum_i L_t[i',i] sum_j R_t[j',j] x[b,i,j,c]
// pass W: T[b][t][i][j'][c] = sum_j R_t[j',j] x[b,i,j,c]      (workspace, 2*B*Hi*Wo*C floats)
// pass H: y[b,i',j',c]      = sum_t sum_i L_t[i',i] T[b][t][i][j'][c]
// =================================================================================================
// Register-blocked: a thread owns one (row, channel) and OT consecutive outputs along the mapped axis, so one
// activation load feeds 2*OT FMAs; the matrix entries are wave-uniform (blockIdx.y picks the output tile) and
// come through the scalar cache. Accumulation order per output is unchanged (j ascending; t=1 then t=2).
constexpr int SM_THREADS = 256;
constexpr int SM_LOADS = 6;            // activation loads issued together (extents are 3 * 2^k)

template <int OT>
__global__ __launch_bounds__(SM_THREADS) void sepmap_w_kernel(const float *__restrict__ x, float *__restrict__ T,
                                                              const float *__restrict__ R1,
                                                              const float *__restrict__ R2, size_t rows, int Hi,
                                                              int Wi, int Wo, int C) {
    const size_t idx = (size_t)blockIdx.x * SM_THREADS + threadIdx.x;      // (row = b*Hi + i, c)
    if (idx >= rows * C) return;
    const int jo0 = blockIdx.y * OT;
    const size_t r = idx / C;
    const int c = (int)(idx - r * C);
    const float *xr = x + r * Wi * C + c;
    const float *r1[OT], *r2[OT];
#pragma unroll
    for (int q = 0; q < OT; ++q) {
        const int jo = min(jo0 + q, Wo - 1);                               // clamped rows are not stored
        r1[q] = R1 + (size_t)jo * Wi;
        r2[q] = R2 + (size_t)jo * Wi;
    }
    float a1[OT], a2[OT];
#pragma unroll
    for (int q = 0; q < OT; ++q) a1[q] = a2[q] = 0.f;
    int j = 0;
    for (; j + SM_LOADS <= Wi; j += SM_LOADS) {                             // SM_LOADS loads in flight per thread
        float v[SM_LOADS];
#pragma unroll
        for (int u = 0; u < SM_LOADS; ++u) v[u] = xr[(size_t)(j + u) * C];
#pragma unroll
        for (int u = 0; u < SM_LOADS; ++u)
#pragma unroll
            for (int q = 0; q < OT; ++q) {
                a1[q] = fmaf(r1[q][j + u], v[u], a1[q]);
                a2[q] = fmaf(r2[q][j + u], v[u], a2[q]);
            }
    }
    for (; j < Wi; ++j) {
        const float v = xr[(size_t)j * C];
#pragma unroll
        for (int q = 0; q < OT; ++q) {
            a1[q] = fmaf(r1[q][j], v, a1[q]);
            a2[q] = fmaf(r2[q][j], v, a2[q]);
        }
    }
    const size_t b = r / Hi;
    const int i = (int)(r - b * Hi);
    const size_t plane = (size_t)Hi * Wo * C;
    float *o = T + (b * 2) * plane + ((size_t)i * Wo + jo0) * C + c;
#pragma unroll
    for (int q = 0; q < OT; ++q)
        if (jo0 + q < Wo) {
            o[(size_t)q * C] = a1[q];
            o[(size_t)q * C + plane] = a2[q];
        }
}

template <int OT>
__global__ __launch_bounds__(SM_THREADS) void sepmap_h_kernel(const float *__restrict__ T, float *__restrict__ y,
                                                              const float *__restrict__ L1,
                                                              const float *__restrict__ L2, int B, int Hi,
                                                              int Ho, size_t row) {
    const size_t idx = (size_t)blockIdx.x * SM_THREADS + threadIdx.x;      // (b, jc = j'*C + c)
    if (idx >= (size_t)B * row) return;
    const int io0 = blockIdx.y * OT;
    const size_t b = idx / row, jc = idx - b * row;
    const size_t plane = (size_t)Hi * row;
    const float *t1 = T + (b * 2) * plane + jc, *t2 = t1 + plane;
    const float *l1[OT], *l2[OT];
#pragma unroll
    for (int q = 0; q < OT; ++q) {
        const int io = min(io0 + q, Ho - 1);
        l1[q] = L1 + (size_t)io * Hi;
        l2[q] = L2 + (size_t)io * Hi;
    }
    float a[OT];
#pragma unroll
    for (int q = 0; q < OT; ++q) a[q] = 0.f;
    int i = 0;
    for (; i + SM_LOADS <= Hi; i += SM_LOADS) {
        float v1[SM_LOADS], v2[SM_LOADS];
#pragma unroll
        for (int u = 0; u < SM_LOADS; ++u) {
            v1[u] = t1[(size_t)(i + u) * row];
            v2[u] = t2[(size_t)(i + u) * row];
        }
#pragma unroll
        for (int u = 0; u < SM_LOADS; ++u)
#pragma unroll
            for (int q = 0; q < OT; ++q) {
                a[q] = fmaf(l1[q][i + u], v1[u], a[q]);
                a[q] = fmaf(l2[q][i + u], v2[u], a[q]);
            }
    }
    for (; i < Hi; ++i) {
        const float v1 = t1[(size_t)i * row], v2 = t2[(size_t)i * row];
#pragma unroll
        for (int q = 0; q < OT; ++q) {
            a[q] = fmaf(l1[q][i], v1, a[q]);
            a[q] = fmaf(l2[q][i], v2, a[q]);
        }
    }
    float *o = y + (b * Ho + io0) * row + jc;
#pragma unroll
    for (int q = 0; q < OT; ++q)
        if (io0 + q < Ho) o[(size_t)q * row] = a[q];
}

// ---- the same two passes with the matrices packed for scalar loads ------------------------------------------
// RW[j][j'][t] (t fastest, j' padded to SM_PAD) and LH[i][t][i'] (i' fastest, padded): the 2*OT (resp. OT) matrix
// entries one activation value is multiplied with are CONTIGUOUS, so they arrive as two or three wide scalar
// loads and feed packed FMAs directly. With separate (out, in) row-major matrices the same loop spent four scalar
// moves / lane spills per packed FMA gathering them (112 s_mov + 164 v_readlane/v_writelane per 72 v_pk_fma_f32).
// Accumulation order per output is unchanged.
constexpr int SM_PAD = 24;

template <int OT>
__global__ __launch_bounds__(SM_THREADS) void sepmap_w_packed_kernel(const float *__restrict__ x, float *__restrict__ T,
                                                                     const float *__restrict__ RW, size_t rows,
                                                                     int Hi, int Wi, int Wo, int wo_pad, int C) {
    const size_t idx = (size_t)blockIdx.x * SM_THREADS + threadIdx.x;      // (row = b*Hi + i, c)
    if (idx >= rows * C) return;
    const int jo0 = blockIdx.y * OT;
    const size_t r = idx / C;
    const int c = (int)(idx - r * C);
    const float *xr = x + r * Wi * C + c;
    const float *rw0 = RW + (size_t)jo0 * 2;
    const size_t ldw = (size_t)wo_pad * 2;
    float a1[OT], a2[OT];
#pragma unroll
    for (int q = 0; q < OT; ++q) a1[q] = a2[q] = 0.f;
    int j = 0;
    for (; j + SM_LOADS <= Wi; j += SM_LOADS) {
        float v[SM_LOADS];
#pragma unroll
        for (int u = 0; u < SM_LOADS; ++u) v[u] = xr[(size_t)(j + u) * C];
#pragma unroll
        for (int u = 0; u < SM_LOADS; ++u) {
            const float *rw = rw0 + (size_t)(j + u) * ldw;
#pragma unroll
            for (int q = 0; q < OT; ++q) {
                a1[q] = fmaf(rw[2 * q], v[u], a1[q]);
                a2[q] = fmaf(rw[2 * q + 1], v[u], a2[q]);
            }
        }
    }
    for (; j < Wi; ++j) {
        const float v = xr[(size_t)j * C];
        const float *rw = rw0 + (size_t)j * ldw;
#pragma unroll
        for (int q = 0; q < OT; ++q) {
            a1[q] = fmaf(rw[2 * q], v, a1[q]);
            a2[q] = fmaf(rw[2 * q + 1], v, a2[q]);
        }
    }
    const size_t b = r / Hi;
    const int i = (int)(r - b * Hi);
    const size_t plane = (size_t)Hi * Wo * C;
    float *o = T + (b * 2) * plane + ((size_t)i * Wo + jo0) * C + c;
#pragma unroll
    for (int q = 0; q < OT; ++q)
        if (jo0 + q < Wo) {
            o[(size_t)q * C] = a1[q];
            o[(size_t)q * C + plane] = a2[q];
        }
}

template <int OT>
__global__ __launch_bounds__(SM_THREADS) void sepmap_h_packed_kernel(const float *__restrict__ T, float *__restrict__ y,
                                                                     const float *__restrict__ LH, int B, int Hi,
                                                                     int Ho, int ho_pad, size_t row) {
    const size_t idx = (size_t)blockIdx.x * SM_THREADS + threadIdx.x;      // (b, jc = j'*C + c)
    if (idx >= (size_t)B * row) return;
    const int io0 = blockIdx.y * OT;
    const size_t b = idx / row, jc = idx - b * row;
    const size_t plane = (size_t)Hi * row;
    const float *t1 = T + (b * 2) * plane + jc, *t2 = t1 + plane;
    const float *lh0 = LH + io0;
    float a[OT];
#pragma unroll
    for (int q = 0; q < OT; ++q) a[q] = 0.f;
    int i = 0;
    for (; i + SM_LOADS <= Hi; i += SM_LOADS) {
        float v1[SM_LOADS], v2[SM_LOADS];
#pragma unroll
        for (int u = 0; u < SM_LOADS; ++u) {
            v1[u] = t1[(size_t)(i + u) * row];
            v2[u] = t2[(size_t)(i + u) * row];
        }
#pragma unroll
        for (int u = 0; u < SM_LOADS; ++u) {
            const float *l1 = lh0 + (size_t)(i + u) * 2 * ho_pad, *l2 = l1 + ho_pad;
#pragma unroll
            for (int q = 0; q < OT; ++q) {
                a[q] = fmaf(l1[q], v1[u], a[q]);
                a[q] = fmaf(l2[q], v2[u], a[q]);
            }
        }
    }
    for (; i < Hi; ++i) {
        const float v1 = t1[(size_t)i * row], v2 = t2[(size_t)i * row];
        const float *l1 = lh0 + (size_t)i * 2 * ho_pad, *l2 = l1 + ho_pad;
#pragma unroll
        for (int q = 0; q < OT; ++q) {
            a[q] = fmaf(l1[q], v1, a[q]);
            a[q] = fmaf(l2[q], v2, a[q]);
        }
    }
    float *o = y + (b * Ho + io0) * row + jc;
#pragma unroll
    for (int q = 0; q < OT; ++q)
        if (io0 + q < Ho) o[(size_t)q * row] = a[q];
}

// =================================================================================================
// column sums and Adam
// =================================================================================================
__global__ __launch_bounds__(256) void colsum_kernel(const float *__restrict__ X, const float *__restrict__ wrow,
                                                     float *__restrict__ out, size_t M, int N,
                                                     size_t rows_per_block) {
    // blockIdx.y tiles the columns (cw = min(N,256) per block, lanes = consecutive columns),
    // blockIdx.x tiles the rows; the 256/cw row sub-groups of a block are reduced through LDS.
    extern __shared__ __attribute__((aligned(16))) float red[];   // [rsubs][cw]
    const int cw = min(N, 256);
    const int rsubs = 256 / cw;
    const int cl = threadIdx.x % cw, rsub = threadIdx.x / cw;
    const size_t r0 = (size_t)blockIdx.x * rows_per_block, r1 = min(M, r0 + rows_per_block);
    const int c = blockIdx.y * cw + cl;
    float s = 0.f;
    if (c < N && rsub < rsubs)
        for (size_t r = r0 + rsub; r < r1; r += rsubs) s += wrow ? wrow[r] * X[r * N + c] : X[r * N + c];
    if (rsub < rsubs) red[rsub * cw + cl] = s;
    __syncthreads();
    if (rsub == 0 && c < N) {
        float t = 0.f;
        for (int k = 0; k < rsubs; ++k) t += red[k * cw + cl];
        atomicAdd(out + c, t);
    }
}

// The same sums with 16-byte lanes (N % 4 == 0): tpr threads cover a row's quads, 256 / tpr rows per sweep, eight
// sweeps in flight. (The scalar kernel above ran the 32-column level-0 gradient, 18.9 MB, at 0.8 TB/s.)
__global__ __launch_bounds__(256) void colsum_vec_kernel(const float *__restrict__ X, const float *__restrict__ wrow,
                                                         float *__restrict__ out, size_t M, int N, int tpr,
                                                         size_t rows_per_block) {
    __shared__ float4 red[256];
    const int cl = threadIdx.x % tpr, rsub = threadIdx.x / tpr, rsubs = 256 / tpr;
    const int q = blockIdx.y * tpr + cl;                             // quad of columns
    const bool live = 4 * q < N;
    const size_t r0 = (size_t)blockIdx.x * rows_per_block, r1 = min(M, r0 + rows_per_block);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live) {
        size_t r = r0 + rsub;
        for (; r + 7 * (size_t)rsubs < r1; r += 8 * (size_t)rsubs) {
            float4 v[8];
            float w[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                v[u] = *reinterpret_cast<const float4 *>(X + (r + (size_t)u * rsubs) * N + 4 * q);
                w[u] = wrow ? wrow[r + (size_t)u * rsubs] : 1.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                s.x = fmaf(w[u], v[u].x, s.x); s.y = fmaf(w[u], v[u].y, s.y);
                s.z = fmaf(w[u], v[u].z, s.z); s.w = fmaf(w[u], v[u].w, s.w);
            }
        }
        for (; r < r1; r += rsubs) {
            const float4 v = *reinterpret_cast<const float4 *>(X + r * N + 4 * q);
            const float w = wrow ? wrow[r] : 1.f;
            s.x = fmaf(w, v.x, s.x); s.y = fmaf(w, v.y, s.y); s.z = fmaf(w, v.z, s.z); s.w = fmaf(w, v.w, s.w);
        }
    }
    red[threadIdx.x] = s;
    __syncthreads();
    if (rsub == 0 && live) {
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int k = 0; k < rsubs; ++k) {
            const float4 a = red[k * tpr + cl];
            t.x += a.x; t.y += a.y; t.z += a.z; t.w += a.w;
        }
        atomicAdd(out + 4 * q + 0, t.x);
        atomicAdd(out + 4 * q + 1, t.y);
        atomicAdd(out + 4 * q + 2, t.z);
        atomicAdd(out + 4 * q + 3, t.w);
    }
}

__device__ __forceinline__ float grad_value(const float *g, size_t i) { return g[i]; }
__device__ __forceinline__ float grad_value(const unsigned short *g, size_t i) {
    return __uint_as_float((unsigned)g[i] << 16);
}

template <typename G>
__global__ __launch_bounds__(256) void adam_kernel(float *__restrict__ p, const G *__restrict__ g,
                                                   float *__restrict__ m, float *__restrict__ v, size_t n,
                                                   float beta1, float beta2, float eps, float wd,
                                                   float step_size, float inv_bc2_sqrt, float gscale,
                                                   unsigned short *__restrict__ p16) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float mi = m[i], vi = v[i];
        const float pn = sei_adam_element(p[i], grad_value(g, i) * gscale, mi, vi, beta1, beta2, eps, wd, step_size,
                                      inv_bc2_sqrt);
        m[i] = mi;
        v[i] = vi;
        p[i] = pn;
        if (p16) {                                  // bf16 shadow of the updated weight (throughput mode)
            const __bf16 b = (__bf16)pn;
            p16[i] = __builtin_bit_cast(unsigned short, b);
        }
    }
}

// The same update on 4 consecutive elements per thread: 16-byte accesses on every float32 stream, 8-byte on
// the bf16 ones (identical per-element arithmetic, so bit-identical to adam_kernel).
__device__ __forceinline__ float4 grad_quad(const float *g, size_t q) { return reinterpret_cast<const float4 *>(g)[q]; }
__device__ __forceinline__ float4 grad_quad(const unsigned short *g, size_t q) {
    const uint2 r = reinterpret_cast<const uint2 *>(g)[q];
    return make_float4(__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u), __uint_as_float(r.y << 16),
                       __uint_as_float(r.y & 0xffff0000u));
}
template <typename G>
__global__ __launch_bounds__(256) void adam_vec_kernel(float *__restrict__ p, const G *__restrict__ g,
                                                       float *__restrict__ m, float *__restrict__ v, size_t nquads,
                                                       float beta1, float beta2, float eps, float wd,
                                                       float step_size, float inv_bc2_sqrt, float gscale,
                                                       unsigned short *__restrict__ p16) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    auto update = [&](size_t q, const float4 pq, float4 mq, float4 vq, const float4 gq) {
        float4 o;
        o.x = sei_adam_element(pq.x, gq.x * gscale, mq.x, vq.x, beta1, beta2, eps, wd, step_size, inv_bc2_sqrt);
        o.y = sei_adam_element(pq.y, gq.y * gscale, mq.y, vq.y, beta1, beta2, eps, wd, step_size, inv_bc2_sqrt);
        o.z = sei_adam_element(pq.z, gq.z * gscale, mq.z, vq.z, beta1, beta2, eps, wd, step_size, inv_bc2_sqrt);
        o.w = sei_adam_element(pq.w, gq.w * gscale, mq.w, vq.w, beta1, beta2, eps, wd, step_size, inv_bc2_sqrt);
        reinterpret_cast<float4 *>(m)[q] = mq;
        reinterpret_cast<float4 *>(v)[q] = vq;
        reinterpret_cast<float4 *>(p)[q] = o;
        if (p16) {
            const __bf16 b0 = (__bf16)o.x, b1 = (__bf16)o.y, b2 = (__bf16)o.z, b3 = (__bf16)o.w;
            uint2 w;
            w.x = (unsigned)__builtin_bit_cast(unsigned short, b0) | ((unsigned)__builtin_bit_cast(unsigned short, b1) << 16);
            w.y = (unsigned)__builtin_bit_cast(unsigned short, b2) | ((unsigned)__builtin_bit_cast(unsigned short, b3) << 16);
            reinterpret_cast<uint2 *>(p16)[q] = w;
        }
    };
    size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; q + stride < nquads; q += 2 * stride) {          // two quads per stream in flight
        const size_t q2 = q + stride;
        const float4 pa = reinterpret_cast<float4 *>(p)[q], pb = reinterpret_cast<float4 *>(p)[q2];
        const float4 ma = reinterpret_cast<float4 *>(m)[q], mb = reinterpret_cast<float4 *>(m)[q2];
        const float4 va = reinterpret_cast<float4 *>(v)[q], vb = reinterpret_cast<float4 *>(v)[q2];
        const float4 ga = grad_quad(g, q), gb = grad_quad(g, q2);
        update(q, pa, ma, va, ga);
        update(q2, pb, mb, vb, gb);
    }
    if (q < nquads)
        update(q, reinterpret_cast<float4 *>(p)[q], reinterpret_cast<float4 *>(m)[q], reinterpret_cast<float4 *>(v)[q],
               grad_quad(g, q));
}

inline unsigned capped_grid(size_t work_items, int per_block, unsigned cap) {
    size_t g = sei_ceil_div(work_items, (size_t)per_block);
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (unsigned)g;
}

}  // namespace

// -------------------------------------------------------------------------------------------------
namespace {
template <int G>
int launch_ln_fwd(const float *x, const float *gamma, const float *beta, float *y, float *mean, float *rstd,
                  size_t rows, int C, float eps, hipStream_t s) {
    const unsigned grid = capped_grid(rows, LN_THREADS / G, 4096);
    hipLaunchKernelGGL(ln_fwd_group_kernel<G>, dim3(grid), dim3(LN_THREADS), 0, s, x, gamma, beta, y, mean, rstd,
                       rows, C, eps);
    return sei_launch_status();
}
template <int G>
int launch_ln_bwd(const float *x, const float *gamma, const float *mean, const float *rstd, const float *gy,
                  float *gx, float *ggamma, float *gbeta, size_t rows, int C, hipStream_t s) {
    // few blocks, many rows each: keeps the global atomics per column low
    const unsigned grid = capped_grid(rows, (LN_THREADS / G) * 4, 1024);
    hipLaunchKernelGGL(ln_bwd_group_kernel<G>, dim3(grid), dim3(LN_THREADS), sizeof(float) * 2 * C, s, x, gamma,
                       mean, rstd, gy, gx, ggamma, gbeta, rows, C);
    return sei_launch_status();
}
inline int ln_group(int C) {
    int g = 1;
    while (g < 64 && g < C) g <<= 1;   // consecutive lanes = consecutive channels
    return g;
}
}  // namespace

__attribute__((visibility("hidden"))) int sei_ln_fwd_f32_lanes16(const float *x, const float *gamma, const float *beta, float *y,
                                                                 float *mean, float *rstd, size_t rows, int C, float eps,
                                                                 hipStream_t s);                  // bf16_support.hip

extern "C" int sei_ln_fwd(const float *x, const float *gamma, const float *beta, float *y, float *mean,
                          float *rstd, size_t rows, int C, float eps, void *stream) {
    SEI_REQUIRE(x && gamma && beta && y && mean && rstd && rows > 0 && C > 0);
    if (C > LN_WIDE_EPT * LN_THREADS) return SEI_ERR_TOO_LARGE;
    hipStream_t s = (hipStream_t)stream;
    {   // 16-byte lanes where the shape allows (C = 4 * 2^k up to 512, any multiple of 4 above)
        const int rc = sei_ln_fwd_f32_lanes16(x, gamma, beta, y, mean, rstd, rows, C, eps, s);
        if (rc >= 0) return rc;
    }
    if (C > 64 * LN_EPL) {
        hipLaunchKernelGGL(ln_fwd_wide_kernel, dim3(capped_grid(rows, 1, 8192)), dim3(LN_THREADS), 0, s, x, gamma,
                           beta, y, mean, rstd, rows, C, eps);
        return sei_launch_status();
    }
    switch (ln_group(C)) {
        case 1: return launch_ln_fwd<1>(x, gamma, beta, y, mean, rstd, rows, C, eps, s);
        case 2: return launch_ln_fwd<2>(x, gamma, beta, y, mean, rstd, rows, C, eps, s);
        case 4: return launch_ln_fwd<4>(x, gamma, beta, y, mean, rstd, rows, C, eps, s);
        case 8: return launch_ln_fwd<8>(x, gamma, beta, y, mean, rstd, rows, C, eps, s);
        case 16: return launch_ln_fwd<16>(x, gamma, beta, y, mean, rstd, rows, C, eps, s);
        case 32: return launch_ln_fwd<32>(x, gamma, beta, y, mean, rstd, rows, C, eps, s);
        default: return launch_ln_fwd<64>(x, gamma, beta, y, mean, rstd, rows, C, eps, s);
    }
}

namespace {
// launch plan of sei_ln_bwd; workspace = [stats: 2*rows floats (wide only)] [partials: nparts * 2C floats]
struct LnBwdPlan {
    int kind;                 // 0 legacy (atomics, no workspace), 1 narrow vectorised, 2 wide two-pass, 3 wide one-pass
    int G, NV;
    unsigned grid, col_blocks, chunks;
    int rows_per_chunk;
    size_t stats_floats, nparts;
};
inline LnBwdPlan ln_bwd_plan(size_t rows, int C) {
    LnBwdPlan p{};
    if (C % 4 != 0 || C < 8) return p;
    if (C <= 512) {
        const int q = C / 4;                                  // float4 per row
        if ((q & (q - 1)) != 0) return p;                     // power of two only
        p.NV = q > 64 ? q / 64 : 1;
        p.G = q / p.NV;
        p.kind = 1;
        const size_t sweeps = sei_ceil_div(rows, (size_t)(LN_THREADS / p.G));
        size_t cap = ((size_t)1 << 20) / (2 * (size_t)C);     // <= 1M partial floats
        if (cap > 512) cap = 512;                             // two workgroups per CU stream at full rate; fewer partials to fold
        p.grid = (unsigned)(sweeps < cap ? sweeps : cap);
        p.nparts = p.grid;
        return p;
    }
    if (C % 1024 == 0 && C <= 8192 && rows >= 64) {            // whole rows per workgroup, x and gy read once
        p.kind = 3;
        p.NV = C / 1024;
        const size_t cap = C >= 8192 ? 256 : 512;               // partial rows: 2 C floats each (<= 17 MB)
        p.grid = (unsigned)(rows < cap ? rows : cap);
        p.nparts = p.grid;
        return p;
    }
    p.kind = 2;
    p.col_blocks = (unsigned)sei_ceil_div((size_t)C, 4 * LN_THREADS);
    size_t chunks = 768 / p.col_blocks > 0 ? 768 / p.col_blocks : 1;       // ~768 workgroups in pass 2
    if (chunks > rows) chunks = rows;
    p.rows_per_chunk = (int)sei_ceil_div(rows, chunks);
    p.chunks = (unsigned)sei_ceil_div(rows, (size_t)p.rows_per_chunk);
    p.stats_floats = 2 * rows;
    p.nparts = p.chunks;
    return p;
}
}  // namespace

extern "C" size_t sei_ln_bwd_workspace(size_t rows, int C) {
    if (rows == 0 || C <= 0) return 0;
    const LnBwdPlan p = ln_bwd_plan(rows, C);
    return p.stats_floats + p.nparts * 2 * (size_t)C;
}

// Where the partial sums of sei_ln_bwd lie in its workspace ([parts][2 C] floats from this offset on), and how many
// there are (0: this shape adds with atomics and leaves nothing to fold).
extern "C" size_t sei_ln_bwd_part_offset(size_t rows, int C) {
    if (rows == 0 || C <= 0) return 0;
    return ln_bwd_plan(rows, C).stats_floats;
}
extern "C" size_t sei_ln_bwd_part_count(size_t rows, int C) {
    if (rows == 0 || C <= 0) return 0;
    const LnBwdPlan p = ln_bwd_plan(rows, C);
    return p.kind != 0 ? p.nparts : 0;
}

extern "C" int sei_ln_bwd(const float *x, const float *gamma, const float *mean, const float *rstd,
                          const float *gy, float *gx, float *ggamma, float *gbeta, size_t rows, int C,
                          float *work, size_t work_floats, void *stream) {
    return sei_ln_bwd_res(x, gamma, mean, rstd, gy, nullptr, gx, ggamma, gbeta, rows, C, work, work_floats, stream);
}

extern "C" int sei_ln_bwd_res(const float *x, const float *gamma, const float *mean, const float *rstd,
                              const float *gy, const float *res, float *gx, float *ggamma, float *gbeta, size_t rows,
                              int C, float *work, size_t work_floats, void *stream) {
    SEI_REQUIRE(x && gamma && mean && rstd && gy && gx && rows > 0 && C > 0);
    SEI_REQUIRE(!res || sei_ln_bwd_part_count(rows, C) > 0);       // the residual rides in the vectorised kernels only
    // ggamma = gbeta = NULL: the partial sums stay in `work` for sei_fold_many (shapes with sei_ln_bwd_part_count > 0)
    SEI_REQUIRE((ggamma != nullptr) == (gbeta != nullptr));
    SEI_REQUIRE(ggamma || sei_ln_bwd_part_count(rows, C) > 0);
    if (C > LN_WIDE_EPT * LN_THREADS) return SEI_ERR_TOO_LARGE;
    hipStream_t s = (hipStream_t)stream;
    const LnBwdPlan p = ln_bwd_plan(rows, C);
    if (p.kind != 0) {
        SEI_REQUIRE(work && work_floats >= p.stats_floats + p.nparts * 2 * (size_t)C);
        SEI_REQUIRE(rows < ((size_t)1 << 31));
        float *part = work + p.stats_floats;
        if (p.kind == 1) {
#define SEI_LN_VEC(GG, NN)                                                                                    \
    hipLaunchKernelGGL((ln_bwd_vec_kernel<GG, NN>), dim3(p.grid), dim3(LN_THREADS), 0, s, x, gamma, mean, rstd, \
                       gy, res, gx, part, rows);                                                              \
    break;
            switch (p.G * 100 + p.NV) {
                case 201: SEI_LN_VEC(2, 1)
                case 401: SEI_LN_VEC(4, 1)
                case 801: SEI_LN_VEC(8, 1)
                case 1601: SEI_LN_VEC(16, 1)
                case 3201: SEI_LN_VEC(32, 1)
                case 6401: SEI_LN_VEC(64, 1)
                case 6402: SEI_LN_VEC(64, 2)
                default: return SEI_ERR_BAD_ARG;
            }
#undef SEI_LN_VEC
        } else if (p.kind == 3) {
#define SEI_LN_ROW(NN)                                                                                        \
    hipLaunchKernelGGL((ln_bwd_row_kernel<NN>), dim3(p.grid), dim3(LN_THREADS), 0, s, x, gamma, mean, rstd, gy, \
                       res, gx, part, rows);                                                                  \
    break;
            switch (p.NV) {
                case 1: SEI_LN_ROW(1)
                case 2: SEI_LN_ROW(2)
                case 3: SEI_LN_ROW(3)
                case 4: SEI_LN_ROW(4)
                case 5: SEI_LN_ROW(5)
                case 6: SEI_LN_ROW(6)
                case 7: SEI_LN_ROW(7)
                default: SEI_LN_ROW(8)
            }
#undef SEI_LN_ROW
        } else {
            float2 *stats = reinterpret_cast<float2 *>(work);
            hipLaunchKernelGGL(ln_bwd_rowstats_kernel, dim3((unsigned)rows), dim3(LN_THREADS), 0, s, x, gamma, mean,
                               rstd, gy, stats, C);
            hipLaunchKernelGGL(ln_bwd_cols_kernel, dim3(p.col_blocks, p.chunks), dim3(LN_THREADS), 0, s, x, gamma,
                               mean, rstd, gy, (const float2 *)stats, res, gx, part, rows, C, p.rows_per_chunk);
        }
        if (ggamma)
            hipLaunchKernelGGL(ln_bwd_fold_kernel, dim3((unsigned)sei_ceil_div((size_t)2 * C, 16)), dim3(256), 0, s,
                               (const float *)part, (int)p.nparts, C, ggamma, gbeta);
        return sei_launch_status();
    }
    // legacy shapes (C not a multiple of 4, or not 4 * 2^k below 512): scalar lanes, float atomics
    if (C > 64 * LN_EPL) {
        hipLaunchKernelGGL(ln_bwd_wide_kernel, dim3(capped_grid(rows, 2, 512)), dim3(LN_THREADS), 0, s, x, gamma,
                           mean, rstd, gy, gx, ggamma, gbeta, rows, C);
        return sei_launch_status();
    }
    switch (ln_group(C)) {
        case 1: return launch_ln_bwd<1>(x, gamma, mean, rstd, gy, gx, ggamma, gbeta, rows, C, s);
        case 2: return launch_ln_bwd<2>(x, gamma, mean, rstd, gy, gx, ggamma, gbeta, rows, C, s);
        case 4: return launch_ln_bwd<4>(x, gamma, mean, rstd, gy, gx, ggamma, gbeta, rows, C, s);
        case 8: return launch_ln_bwd<8>(x, gamma, mean, rstd, gy, gx, ggamma, gbeta, rows, C, s);
        case 16: return launch_ln_bwd<16>(x, gamma, mean, rstd, gy, gx, ggamma, gbeta, rows, C, s);
        case 32: return launch_ln_bwd<32>(x, gamma, mean, rstd, gy, gx, ggamma, gbeta, rows, C, s);
        default: return launch_ln_bwd<64>(x, gamma, mean, rstd, gy, gx, ggamma, gbeta, rows, C, s);
    }
}

extern "C" int sei_conv3x3_fwd(const float *x, const float *w, const float *bias, const float *res, float *y,
                               int B, int H, int W, int Cin, int Cout, int nchw_in, int nchw_out,
                               int transposed, void *stream) {
    SEI_REQUIRE(x && w && y && x != y && B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0);
    const size_t lds = sizeof(float) * (size_t)Cin * Cout * 9;
    if (lds > 64 * 1024) return SEI_ERR_TOO_LARGE;
    const size_t npix = (size_t)B * H * W;
    const unsigned pgrid = capped_grid(npix, C3_THREADS, 8192);
    hipStream_t s = (hipStream_t)stream;
    if (Cin == 32 && Cout <= 4 && !nchw_in) {      // hidden -> image (or the data gradient of image -> hidden, whose
        // pixel-per-thread form took 8 ms per launch on the 192 x 192 grids of the x4 network): lanes = hidden channels
        const int nruns_row = (int)sei_ceil_div(W, C3L_RUN);
        const size_t runs = (size_t)B * H * nruns_row;
        SEI_REQUIRE(runs < ((size_t)1 << 31));
        const dim3 grid(capped_grid(runs, C3_THREADS / 32, 65535));
#define SEI_C3_LANES(CS)                                                                                            \
    hipLaunchKernelGGL(conv3x3_c32_to_small_kernel<CS>, grid, dim3(C3_THREADS), 0, s, x, w, bias, res, y, B, H, W, \
                       nchw_out ? 1 : 0, nruns_row, (int)runs, transposed ? 1 : 0);                                 \
    return sei_launch_status();
        switch (Cout) {
            case 1: SEI_C3_LANES(1)
            case 2: SEI_C3_LANES(2)
            case 3: SEI_C3_LANES(3)
            default: SEI_C3_LANES(4)
        }
#undef SEI_C3_LANES
    }
    if (Cout == 32 && Cin <= 4 && !nchw_out && (((uintptr_t)y | (uintptr_t)res | (uintptr_t)bias) & 15) == 0) {
        const unsigned grid4 = capped_grid(npix * 4, C3_THREADS, 16384);
#define SEI_C3_TO32(CS)                                                                                              \
    hipLaunchKernelGGL(conv3x3_small_to_c32_kernel<CS>, dim3(grid4), dim3(C3_THREADS), 0, s, x, w, bias, res, y, B, \
                       H, W, nchw_in ? 1 : 0, transposed ? 1 : 0);                                                  \
    return sei_launch_status();
        switch (Cin) {
            case 1: SEI_C3_TO32(1)
            case 2: SEI_C3_TO32(2)
            case 3: SEI_C3_TO32(3)
            default: SEI_C3_TO32(4)
        }
#undef SEI_C3_TO32
    }
#define SEI_C3_PIX(CO)                                                                                          \
    hipLaunchKernelGGL(conv3x3_pix_kernel<CO>, dim3(pgrid), dim3(C3_THREADS), lds, s, x, w, bias, res, y, B, H, \
                       W, Cin, nchw_in ? 1 : 0, nchw_out ? 1 : 0, transposed ? 1 : 0);                          \
    return sei_launch_status();
    switch (Cout) {
        case 3: SEI_C3_PIX(3)
        case 8: SEI_C3_PIX(8)
        case 16: SEI_C3_PIX(16)
        case 32: SEI_C3_PIX(32)
        default: break;
    }
#undef SEI_C3_PIX
    const size_t total = npix * Cout;
    hipLaunchKernelGGL(conv3x3_kernel, dim3(capped_grid(total, C3_THREADS, 4096)), dim3(C3_THREADS), lds, s, x, w,
                       bias, res, y, B, H, W, Cin, Cout, nchw_in ? 1 : 0, nchw_out ? 1 : 0, transposed ? 1 : 0);
    return sei_launch_status();
}

namespace {
inline bool c3_mfma_ok(size_t npix, int Cin, int Cout, int nchw_x, int nchw_gy) {
    const bool small_out = Cin == 32 && Cout >= 1 && Cout <= 3 && !nchw_x, small_in = Cout == 32 && Cin >= 1 && Cin <= 3 && !nchw_gy;
    return (small_out || small_in) && npix < ((size_t)1 << 40);
}
inline unsigned c3_mfma_grid(size_t npix, size_t workgroups, size_t &ppw) {
    ppw = sei_ceil_div(npix, workgroups * 8);                            // pixels per wave, eight waves per workgroup
    ppw = sei_ceil_div(ppw, 16) * 16;                                   // whole batches of 16 pixels
    return (unsigned)sei_ceil_div(npix, ppw * 8);
}
inline void c3_mfma_launch(const float *x, const float *gy, float *gw, float *gb, float *part, int B, int H, int W, int Cin,
                           int Cout, int nchw_x, int nchw_gy, unsigned grid, size_t ppw, hipStream_t st) {
    if (Cin == 32)
        hipLaunchKernelGGL(conv3x3_wgrad_mfma_kernel<true>, dim3(grid), dim3(512), 0, st, x, gy, gw, gb, B, H, W, Cout,
                           nchw_gy ? 1 : 0, (int)ppw, part);
    else
        hipLaunchKernelGGL(conv3x3_wgrad_mfma_kernel<false>, dim3(grid), dim3(512), 0, st, gy, x, gw, gb, B, H, W, Cin,
                           nchw_x ? 1 : 0, (int)ppw, part);
}
}  // namespace

// Two-stage form of sei_conv3x3_bwd_weight for the network's end convolutions (3 <-> 32 channels): _parts_count = how many
// partial rows the launch leaves (0: shape not served, use sei_conv3x3_bwd_weight), each Cout * Cin * 9 + Cout floats (the
// weight gradient in torch's layout, then the bias gradient); sei_fold_many adds them up (SEI_FOLD_SPLIT, split = Cout * Cin * 9).
// No atomics: 1024 workgroups instead of 256.
extern "C" size_t sei_conv3x3_bwd_weight_parts_count(int B, int H, int W, int Cin, int Cout, int nchw_x, int nchw_gy) {
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    const size_t npix = (size_t)B * H * W;
    if (!c3_mfma_ok(npix, Cin, Cout, nchw_x, nchw_gy)) return 0;
    size_t ppw;
    return c3_mfma_grid(npix, 1024, ppw);
}
extern "C" int sei_conv3x3_bwd_weight_parts(const float *x, const float *gy, float *part, int B, int H, int W, int Cin,
                                            int Cout, int nchw_x, int nchw_gy, void *stream) {
    SEI_REQUIRE(x && gy && part && B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0);
    const size_t npix = (size_t)B * H * W;
    SEI_REQUIRE(c3_mfma_ok(npix, Cin, Cout, nchw_x, nchw_gy));
    size_t ppw;
    const unsigned grid = c3_mfma_grid(npix, 1024, ppw);
    c3_mfma_launch(x, gy, nullptr, nullptr, part, B, H, W, Cin, Cout, nchw_x, nchw_gy, grid, ppw, (hipStream_t)stream);
    return sei_launch_status();
}

extern "C" int sei_conv3x3_bwd_weight(const float *x, const float *gy, float *gw, float *gb, int B, int H,
                                      int W, int Cin, int Cout, int nchw_x, int nchw_gy, void *stream) {
    SEI_REQUIRE(x && gy && gw && B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0);
    const size_t npix = (size_t)B * H * W;
    const int KC = Cin * 9;
    {   // the two real layers of the default network: lanes = the 32 hidden channels
        const bool small_out = Cin == 32 && Cout <= 4 && !nchw_x, small_in = Cout == 32 && Cin <= 4 && !nchw_gy;
        if (c3_mfma_ok(npix, Cin, Cout, nchw_x, nchw_gy)) {
            // (28 columns of the 32 x 32 product: three small channels x nine taps + the bias column)
            size_t ppw;
            const unsigned grid = c3_mfma_grid(npix, 256, ppw);        // one workgroup per CU: 256-way atomics per output
            c3_mfma_launch(x, gy, gw, gb, nullptr, B, H, W, Cin, Cout, nchw_x, nchw_gy, grid, ppw, (hipStream_t)stream);
            return sei_launch_status();
        }
        if (small_out || small_in) {
            const int nruns_row = (int)sei_ceil_div(W, C3L_RUN);
            const size_t runs = (size_t)B * H * nruns_row;
            SEI_REQUIRE(runs < ((size_t)1 << 31));
            // ~512 workgroups: every CU busy, few atomics per output
            const dim3 grid(capped_grid(runs, (C3_THREADS / 32) * (int)sei_ceil_div(runs, (size_t)512 * 8), 65535));
            hipStream_t st = (hipStream_t)stream;
            const int CSv = small_out ? Cout : Cin, lay = small_out ? (nchw_gy ? 1 : 0) : (nchw_x ? 1 : 0);
#define SEI_C3_WG(CS, OUT)                                                                                          \
    hipLaunchKernelGGL((conv3x3_wgrad_c32_kernel<CS, OUT>), grid, dim3(C3_THREADS), 0, st, x, gy, gw, gb, B, H, W, \
                       lay, nruns_row, (int)runs);                                                                  \
    return sei_launch_status();
            if (small_out) {
                switch (CSv) {
                    case 1: SEI_C3_WG(1, true)
                    case 2: SEI_C3_WG(2, true)
                    case 3: SEI_C3_WG(3, true)
                    default: SEI_C3_WG(4, true)
                }
            } else {
                switch (CSv) {
                    case 1: SEI_C3_WG(1, false)
                    case 2: SEI_C3_WG(2, false)
                    case 3: SEI_C3_WG(3, false)
                    default: SEI_C3_WG(4, false)
                }
            }
#undef SEI_C3_WG
        }
    }
    if (Cout * KC <= 8 * C3_THREADS) {
        int P = 128;
        while (P > 8 && (size_t)P * (Cout + KC) * sizeof(float) > 48 * 1024) P /= 2;
        int tpb = 1;                          // pixel tiles per workgroup: keep the atomics per output low
        while (sei_ceil_div(npix, (size_t)P * tpb) > 1024) tpb *= 2;
        const size_t lds = (size_t)P * (Cout + KC) * sizeof(float);
        hipLaunchKernelGGL(conv3x3_bwd_weight_tiled_kernel, dim3((unsigned)sei_ceil_div(npix, (size_t)P * tpb)),
                           dim3(C3_THREADS), lds, (hipStream_t)stream, x, gy, gw, gb, B, H, W, Cin, Cout,
                           nchw_x ? 1 : 0, nchw_gy ? 1 : 0, P, tpb);
        return sei_launch_status();
    }
    int ppb = 64;
    while (sei_ceil_div(npix, ppb) > 2048) ppb *= 2;
    hipLaunchKernelGGL(conv3x3_bwd_weight_kernel, dim3((unsigned)sei_ceil_div(npix, ppb)), dim3(C3_THREADS), 0,
                       (hipStream_t)stream, x, gy, gw, gb, B, H, W, Cin, Cout, nchw_x ? 1 : 0, nchw_gy ? 1 : 0,
                       ppb);
    return sei_launch_status();
}

extern "C" int sei_sepmap2(const float *x, float *y, int B, int Hi, int Wi, int Ho, int Wo, int C,
                           const float *L1, const float *R1, const float *L2, const float *R2, float *work,
                           size_t work_floats, void *stream) {
    SEI_REQUIRE(x && y && L1 && R1 && L2 && R2 && work && x != y);
    SEI_REQUIRE(B > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0 && C > 0);
    const size_t need = (size_t)2 * B * Hi * Wo * C;
    SEI_REQUIRE(work_floats >= need);
    hipStream_t s = (hipStream_t)stream;
    const size_t rows = (size_t)B * Hi, nw = rows * C, nh = (size_t)B * Wo * C;
    SEI_REQUIRE(sei_ceil_div(nw, SM_THREADS) < (1u << 31) && sei_ceil_div(nh, SM_THREADS) < (1u << 31));
    const dim3 gw((unsigned)sei_ceil_div(nw, SM_THREADS)), gh((unsigned)sei_ceil_div(nh, SM_THREADS));
    if (Wo % 12 == 0)
        hipLaunchKernelGGL(sepmap_w_kernel<12>, dim3(gw.x, Wo / 12), dim3(SM_THREADS), 0, s, x, work, R1, R2, rows, Hi,
                           Wi, Wo, C);
    else
        hipLaunchKernelGGL(sepmap_w_kernel<8>, dim3(gw.x, (unsigned)sei_ceil_div(Wo, 8)), dim3(SM_THREADS), 0, s, x,
                           work, R1, R2, rows, Hi, Wi, Wo, C);
    if (Ho % 12 == 0)
        hipLaunchKernelGGL(sepmap_h_kernel<12>, dim3(gh.x, Ho / 12), dim3(SM_THREADS), 0, s, (const float *)work, y, L1,
                           L2, B, Hi, Ho, (size_t)Wo * C);
    else
        hipLaunchKernelGGL(sepmap_h_kernel<8>, dim3(gh.x, (unsigned)sei_ceil_div(Ho, 8)), dim3(SM_THREADS), 0, s,
                           (const float *)work, y, L1, L2, B, Hi, Ho, (size_t)Wo * C);
    return sei_launch_status();
}

extern "C" int sei_sepmap2_packed(const float *x, float *y, int B, int Hi, int Wi, int Ho, int Wo, int C,
                                  const float *RW, const float *LH, float *work, size_t work_floats, void *stream) {
    SEI_REQUIRE(x && y && RW && LH && work && x != y);
    SEI_REQUIRE(B > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0 && C > 0);
    const size_t need = (size_t)2 * B * Hi * Wo * C;
    SEI_REQUIRE(work_floats >= need);
    hipStream_t s = (hipStream_t)stream;
    const int wo_pad = (int)sei_ceil_div(Wo, SM_PAD) * SM_PAD, ho_pad = (int)sei_ceil_div(Ho, SM_PAD) * SM_PAD;
    const size_t rows = (size_t)B * Hi, nw = rows * C, nh = (size_t)B * Wo * C;
    SEI_REQUIRE(sei_ceil_div(nw, SM_THREADS) < (1u << 31) && sei_ceil_div(nh, SM_THREADS) < (1u << 31));
    const dim3 gw((unsigned)sei_ceil_div(nw, SM_THREADS)), gh((unsigned)sei_ceil_div(nh, SM_THREADS));
    // 24 outputs per thread where the extent and the workgroup count allow: half the passes over the input (the H pass
    // re-reads the two-plane intermediate Ho / OT times, at the fine levels 75 MB each time): 24->48 at C = 128 87 -> 71 us,
    // 12->24 at C = 512 58 -> 49 us (tools/exp_sepmap.py)
    if (Wo % 24 == 0 && (size_t)gw.x * (Wo / 24) >= 1024)      // (fewer workgroups than that: 48->24 at C = 32 ran 50 vs 31 us)
        hipLaunchKernelGGL(sepmap_w_packed_kernel<24>, dim3(gw.x, Wo / 24), dim3(SM_THREADS), 0, s, x, work, RW, rows, Hi,
                           Wi, Wo, wo_pad, C);
    else if (Wo % 12 == 0)
        hipLaunchKernelGGL(sepmap_w_packed_kernel<12>, dim3(gw.x, Wo / 12), dim3(SM_THREADS), 0, s, x, work, RW, rows, Hi,
                           Wi, Wo, wo_pad, C);
    else
        hipLaunchKernelGGL(sepmap_w_packed_kernel<8>, dim3(gw.x, (unsigned)sei_ceil_div(Wo, 8)), dim3(SM_THREADS), 0, s, x,
                           work, RW, rows, Hi, Wi, Wo, wo_pad, C);
    if (Ho % 24 == 0 && (size_t)gh.x * (Ho / 24) >= 1024)
        hipLaunchKernelGGL(sepmap_h_packed_kernel<24>, dim3(gh.x, Ho / 24), dim3(SM_THREADS), 0, s, (const float *)work, y,
                           LH, B, Hi, Ho, ho_pad, (size_t)Wo * C);
    else if (Ho % 12 == 0)
        hipLaunchKernelGGL(sepmap_h_packed_kernel<12>, dim3(gh.x, Ho / 12), dim3(SM_THREADS), 0, s, (const float *)work, y,
                           LH, B, Hi, Ho, ho_pad, (size_t)Wo * C);
    else
        hipLaunchKernelGGL(sepmap_h_packed_kernel<8>, dim3(gh.x, (unsigned)sei_ceil_div(Ho, 8)), dim3(SM_THREADS), 0, s,
                           (const float *)work, y, LH, B, Hi, Ho, ho_pad, (size_t)Wo * C);
    return sei_launch_status();
}

namespace {
int launch_colsum(const float *X, const float *wrow, float *out, size_t M, int N, void *stream) {
    if (N % 4 == 0 && (reinterpret_cast<uintptr_t>(X) & 15) == 0) {
        const int quads = N / 4;
        int tpr = 1;
        while (tpr < quads && tpr < 256) tpr <<= 1;
        const unsigned col_blocks = (unsigned)sei_ceil_div(quads, tpr);
        // ~128 workgroups with 128 bytes in flight per thread: every workgroup costs one atomic per column, and those
        // serialise per address (288 workgroups on 128 columns spent 30 us on a 19-MB tensor, most of it in the atomics)
        size_t rpb = (size_t)(256 / tpr) * 8;
        while (sei_ceil_div(M, rpb) * col_blocks > 128 && rpb < M) rpb *= 2;
        hipLaunchKernelGGL(colsum_vec_kernel, dim3((unsigned)sei_ceil_div(M, rpb), col_blocks), dim3(256), 0,
                           (hipStream_t)stream, X, wrow, out, M, N, tpr, rpb);
        return sei_launch_status();
    }
    const int cw = N < 256 ? N : 256;
    const unsigned col_blocks = (unsigned)sei_ceil_div(N, cw);
    // ~2048 workgroups in all: enough to fill 256 CUs, few enough to keep the atomics per column low
    size_t rpb = 16;
    while (sei_ceil_div(M, rpb) * col_blocks > 2048 && rpb < M) rpb *= 2;
    const size_t lds = sizeof(float) * (size_t)(256 / cw) * cw;
    hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)sei_ceil_div(M, rpb), col_blocks), dim3(256), lds,
                       (hipStream_t)stream, X, wrow, out, M, N, rpb);
    return sei_launch_status();
}
}  // namespace

extern "C" int sei_colsum_f32(const float *X, float *out, size_t M, int N, void *stream) {
    SEI_REQUIRE(X && out && M > 0 && N > 0);
    return launch_colsum(X, nullptr, out, M, N, stream);
}

extern "C" int sei_colsum_weighted_f32(const float *X, const float *row_weight, float *out, size_t M, int N,
                                       void *stream) {
    SEI_REQUIRE(X && row_weight && out && M > 0 && N > 0);
    return launch_colsum(X, row_weight, out, M, N, stream);
}

extern "C" int sei_adam_scalars(float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                                float *out6, void *stream) {
    (void)stream;                                    // host arithmetic only; the argument keeps the call convention
    SEI_REQUIRE(out6 && step > 0);
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    out6[0] = beta1; out6[1] = beta2; out6[2] = eps; out6[3] = weight_decay;
    out6[4] = (float)((double)lr / bc1);
    out6[5] = (float)(1.0 / sqrt(bc2));
    return 0;
}

namespace {
struct Six { float v[6]; };
__global__ void store_six_kernel(float *dst, Six s) {
    if (threadIdx.x < 6) dst[threadIdx.x] = s.v[threadIdx.x];
}
}  // namespace

// The same six scalars into a DEVICE array, as arguments of a one-wave kernel: ordered on the stream like any other launch
// (a host buffer copied asynchronously could be overwritten for step t+1 before the copy of step t has run -- the host
// runs ahead of a queue of replayed graphs).
extern "C" int sei_adam_scalars_to_device(float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                                          float *dev6, void *stream) {
    SEI_REQUIRE(dev6 && step > 0);
    Six s;
    const int rc = sei_adam_scalars(lr, beta1, beta2, eps, weight_decay, step, s.v, nullptr);
    if (rc != 0) return rc;
    hipLaunchKernelGGL(store_six_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, dev6, s);
    return sei_launch_status();
}

extern "C" int sei_adam_fused(float *param, const void *grad, int grad_is_bf16, float *exp_avg, float *exp_avg_sq,
                              size_t n, float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                              float grad_scale, uint16_t *param_bf16, void *stream) {
    SEI_REQUIRE(param && grad && exp_avg && exp_avg_sq && n > 0 && step > 0);
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    const float step_size = (float)((double)lr / bc1);
    const float inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
    hipStream_t s = (hipStream_t)stream;
    // 16-byte-aligned streams take the 4-wide kernel; a ragged tail (and unaligned views) the scalar one.
    // Grid: one two-quad iteration per thread (up to 2^20 workgroups) -- measured 3.17 ms for the 645 M-parameter
    // bucket against 3.43 ms with 8192 looping workgroups (tools/exp_adam.py).
    const size_t gsz = grad_is_bf16 ? 2 : 4;
    const bool aligned = ((reinterpret_cast<uintptr_t>(param) | reinterpret_cast<uintptr_t>(exp_avg) |
                           reinterpret_cast<uintptr_t>(exp_avg_sq)) & 15) == 0 &&
                         (reinterpret_cast<uintptr_t>(grad) & (4 * gsz - 1)) == 0 &&
                         (!param_bf16 || (reinterpret_cast<uintptr_t>(param_bf16) & 7) == 0);
    const size_t nq = aligned ? n / 4 : 0, done = 4 * nq;
#define SEI_ADAM(KERNEL, G, COUNT, OFF)                                                                              \
    hipLaunchKernelGGL(KERNEL<G>, dim3(capped_grid(COUNT, 256 * 2, 1u << 20)), dim3(256), 0, s, param + (OFF),      \
                       reinterpret_cast<const G *>(grad) + (OFF), exp_avg + (OFF), exp_avg_sq + (OFF), COUNT, beta1, \
                       beta2, eps, weight_decay, step_size, inv_bc2_sqrt, grad_scale,                                \
                       param_bf16 ? param_bf16 + (OFF) : nullptr)
    if (nq > 0) {
        if (grad_is_bf16) SEI_ADAM(adam_vec_kernel, unsigned short, nq, 0);
        else SEI_ADAM(adam_vec_kernel, float, nq, 0);
    }
    if (done < n) {
        const size_t rest = n - done;
        if (grad_is_bf16) SEI_ADAM(adam_kernel, unsigned short, rest, done);
        else SEI_ADAM(adam_kernel, float, rest, done);
    }
#undef SEI_ADAM
    return sei_launch_status();
}


extern "C" int sei_fold_many(const SeiFoldJob *jobs, int njobs, void *stream) {
    SEI_REQUIRE(jobs && njobs > 0 && njobs <= SEI_FOLD_MAX_JOBS);
    FoldManyArgs g;
    size_t wgs = 0;
    for (int j = 0; j < njobs; ++j) {
        const SeiFoldJob &J = jobs[j];
        SEI_REQUIRE(J.a && J.ncol > 0 && J.split > 0 && J.nseg >= 1 && J.nseg <= 3);
        SEI_REQUIRE(J.kind == SEI_FOLD_SPLIT || J.kind == SEI_FOLD_DWCONV7);
        if (J.kind == SEI_FOLD_DWCONV7) SEI_REQUIRE(J.ncol == 50 * J.split);
        else SEI_REQUIRE(J.ncol <= 3 * J.split && (J.ncol <= J.split || J.b));
        for (int sg = 0; sg < J.nseg; ++sg) SEI_REQUIRE(J.part[sg] && J.groups[sg] > 0);
        for (int k = 0; k < j; ++k) SEI_REQUIRE(jobs[k].a != J.a);      // one job per destination: no two workgroups add to one address
        g.job[j] = J;
        bool vec = (J.ncol & 3) == 0;                           // as fold_vec4 in the kernel
        for (int sg = 0; sg < J.nseg; ++sg) vec = vec && (reinterpret_cast<uintptr_t>(J.part[sg]) & 15) == 0;
        wgs += sei_ceil_div((size_t)J.ncol, vec ? 4 * FOLD_LANES * FOLD_NQ : FOLD_LANES);
    }
    g.njobs = njobs;
    SEI_REQUIRE(wgs < ((size_t)1 << 31));
    hipLaunchKernelGGL(fold_many_kernel, dim3((unsigned)wgs), dim3(FOLD_SLICES * FOLD_LANES), 0, (hipStream_t)stream, g);
    return sei_launch_status();
}
