// Streaming kernels of the SwinIR throughput (bf16) path on gfx950: everything between the GEMMs of
// models/_swin_ops16.py. Tokens stay float32 (B*H*W, C = 180) between blocks, as in the f32 path; GEMM operands are
// bf16 with the channel count padded to a multiple of 64 (192) and exact zeros in the pad, so that every operand is
// whole 16-byte LDS-DMA chunks for sei_gemm_bf16nt and the attention heads are 32 wide for the MFMA kernel.
//
//   sei_pack / sei_unpack_add   the model's float32 parameter bucket <-> the padded / permuted GEMM layouts, through
//                               an int32 index map (one launch each per step for the whole model)
//   sei_ln_fwd_bf16_pad         LayerNorm over C channels -> bf16 rows of ldy >= C (zeros in the pad)
//   sei_ln_bwd_pad              its backward from a float32 gradient with row stride ldg, plus an optional residual
//                               gradient: gx = LN'(gy) + res
//   sei_cast_pad_bf16           float32 (M, C) -> bf16 (M, ldy), optional per-row factor (stochastic depth) and the
//                               column sums of the scaled rows (the bias gradient) in the same pass
//   sei_pad_nhwc_bf16           sei_pad_nhwc with bf16 output and channel padding
#include "sei_common.h"

namespace {

__device__ __forceinline__ unsigned short f2bf(float v) {
    const __bf16 b = (__bf16)v;
    return __builtin_bit_cast(unsigned short, b);
}

inline unsigned stream_grid(size_t items, int per_block) {
    size_t g = sei_ceil_div(items, (size_t)per_block);
    return (unsigned)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

__global__ __launch_bounds__(256) void pack_kernel(const float *__restrict__ src, const int *__restrict__ map,
                                                    void *__restrict__ dst, size_t n, int to_bf16) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int m = map[i];
        const float v = m >= 0 ? src[m] : 0.f;
        if (to_bf16) reinterpret_cast<unsigned short *>(dst)[i] = f2bf(v);
        else reinterpret_cast<float *>(dst)[i] = v;
    }
}

__global__ __launch_bounds__(256) void unpack_add_kernel(const float *__restrict__ src, const int *__restrict__ map,
                                                          float *__restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int m = map[i];
        if (m >= 0) dst[m] += src[i];                      // every parameter element appears at most once in a map
    }
}

// one wave per row, lane l holds channels 4l .. 4l+3 (C <= 256, C % 4 == 0)
__global__ __launch_bounds__(256) void ln_fwd_bf16_pad_kernel(const float *__restrict__ x, const float *__restrict__ gamma,
                                                               const float *__restrict__ beta,
                                                               unsigned short *__restrict__ y, float *__restrict__ mean,
                                                               float *__restrict__ rstd, size_t rows, int C, int ldy,
                                                               float eps, int ones_col) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool live = 4 * lane < C;
    // ones_col: padding column C holds 1.0 -- the weight-gradient GEMM that reduces over these rows then leaves the
    // column sums of its other operand (= the layer's bias gradient) in column C of its output, for nothing
    const unsigned short pad0 = (ones_col && 4 * lane == C) ? (unsigned short)0x3F80 : (unsigned short)0;
    float4 gm = make_float4(0.f, 0.f, 0.f, 0.f), bt = gm;
    if (live) {
        gm = reinterpret_cast<const float4 *>(gamma)[lane];
        bt = reinterpret_cast<const float4 *>(beta)[lane];
    }
    const float invC = 1.0f / (float)C;
    for (size_t row = (size_t)blockIdx.x * 4 + wave; row < rows; row += (size_t)gridDim.x * 4) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (live) v = reinterpret_cast<const float4 *>(x + row * C)[lane];
        float s = (v.x + v.y) + (v.z + v.w);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
        const float mu = s * invC;
        const float dx = v.x - mu, dy = v.y - mu, dz = v.z - mu, dw = v.w - mu;
        float q = live ? (dx * dx + dy * dy) + (dz * dz + dw * dw) : 0.f;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) q += __shfl_xor(q, off, 64);
        const float rs = rsqrtf(q * invC + eps);
        if (lane == 0) {
            mean[row] = mu;
            rstd[row] = rs;
        }
        if (4 * lane < ldy) {
            ushort4 o = make_ushort4(pad0, 0, 0, 0);
            if (live) {
                o.x = f2bf(dx * rs * gm.x + bt.x);
                o.y = f2bf(dy * rs * gm.y + bt.y);
                o.z = f2bf(dz * rs * gm.z + bt.z);
                o.w = f2bf(dw * rs * gm.w + bt.w);
            }
            reinterpret_cast<ushort4 *>(y + row * ldy)[lane] = o;
        }
    }
}

__global__ __launch_bounds__(256) void ln_bwd_pad_kernel(const float *__restrict__ x, const float *__restrict__ gamma,
                                                          const float *__restrict__ mean, const float *__restrict__ rstd,
                                                          const float *__restrict__ gy, const float *__restrict__ res,
                                                          float *__restrict__ gx, float *__restrict__ part,
                                                          size_t rows, int C, int ldg) {
    __shared__ float4 red[2][4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool live = 4 * lane < C;
    float4 gm = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live) gm = reinterpret_cast<const float4 *>(gamma)[lane];
    float4 ag = make_float4(0.f, 0.f, 0.f, 0.f), ab = ag;
    const float invC = 1.0f / (float)C;
    for (size_t row = (size_t)blockIdx.x * 4 + wave; row < rows; row += (size_t)gridDim.x * 4) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f), g = v;
        if (live) {
            v = reinterpret_cast<const float4 *>(x + row * C)[lane];
            g = reinterpret_cast<const float4 *>(gy + row * ldg)[lane];
        }
        const float mu = mean[row], rs = rstd[row];
        const float4 xh = make_float4((v.x - mu) * rs, (v.y - mu) * rs, (v.z - mu) * rs, (v.w - mu) * rs);
        if (live) {
            ag.x += g.x * xh.x; ag.y += g.y * xh.y; ag.z += g.z * xh.z; ag.w += g.w * xh.w;
            ab.x += g.x; ab.y += g.y; ab.z += g.z; ab.w += g.w;
        }
        const float4 t = make_float4(g.x * gm.x, g.y * gm.y, g.z * gm.z, g.w * gm.w);
        float s1 = live ? (t.x + t.y) + (t.z + t.w) : 0.f;
        float s2 = live ? (t.x * xh.x + t.y * xh.y) + (t.z * xh.z + t.w * xh.w) : 0.f;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            s1 += __shfl_xor(s1, off, 64);
            s2 += __shfl_xor(s2, off, 64);
        }
        s1 *= invC;
        s2 *= invC;
        if (live) {
            float4 o = make_float4(rs * (t.x - s1 - xh.x * s2), rs * (t.y - s1 - xh.y * s2),
                                   rs * (t.z - s1 - xh.z * s2), rs * (t.w - s1 - xh.w * s2));
            if (res) {
                const float4 r = reinterpret_cast<const float4 *>(res + row * C)[lane];
                o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
            }
            reinterpret_cast<float4 *>(gx + row * C)[lane] = o;
        }
    }
    red[0][wave][lane] = ag;
    red[1][wave][lane] = ab;
    __syncthreads();
    if (wave == 0 && live) {
        float4 a = red[0][0][lane], b = red[1][0][lane];
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            const float4 a2 = red[0][w][lane], b2 = red[1][w][lane];
            a.x += a2.x; a.y += a2.y; a.z += a2.z; a.w += a2.w;
            b.x += b2.x; b.y += b2.y; b.z += b2.z; b.w += b2.w;
        }
        // per-workgroup partial sums [workgroup][2][C]: folded by fold_partials_kernel (no atomics: 2048 workgroups
        // adding into the same 2 C addresses serialised in L2 and tripled this kernel's time)
        float *dst = part + (size_t)blockIdx.x * 2 * C;
        reinterpret_cast<float4 *>(dst)[lane] = a;
        reinterpret_cast<float4 *>(dst + C)[lane] = b;
    }
}

// out[c] += sum over g of part[g][c], c < ncol (ncol = 2 C for LayerNorm: gamma then beta, contiguous outputs not
// required: out_a for c < split, out_b for the rest)
// A workgroup owns 16 consecutive columns and splits the groups over 16 interleaved slices, four independent loads
// per slice in flight; the slices meet in LDS in a fixed order (bitwise reproducible). (First version: 64 columns per
// workgroup, four slices of 256-512 DEPENDENT loads each -- 64 us per fold, 12 % of the SwinIR step.)
__global__ __launch_bounds__(256) void fold_partials_kernel(const float *__restrict__ part, int groups, int ncol,
                                                             int split, float *__restrict__ out_a,
                                                             float *__restrict__ out_b, float *__restrict__ out_c = nullptr) {
    __shared__ float red[16][16];
    const int el = threadIdx.x & 15, slice = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + el;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (c < ncol) {
        int g = slice;
        for (; g + 48 < groups; g += 64) {
            s0 += part[(size_t)g * ncol + c];
            s1 += part[(size_t)(g + 16) * ncol + c];
            s2 += part[(size_t)(g + 32) * ncol + c];
            s3 += part[(size_t)(g + 48) * ncol + c];
        }
        for (; g < groups; g += 16) s0 += part[(size_t)g * ncol + c];
    }
    red[slice][el] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (slice == 0 && c < ncol) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) s += red[k][el];
        if (c < split) out_a[c] += s;
        else if (c < 2 * split) out_b[c - split] += s;
        else if (out_c) out_c[c - 2 * split] += s;          // (three sums per row: token_gemm.hip's LayerNorm epilogue)
    }
}

// one wave per row, 4 rows per workgroup pass; lane l converts channels 4l..4l+3
__global__ __launch_bounds__(256) void cast_pad_bf16_kernel(const float *__restrict__ x, const float *__restrict__ scale,
                                                             unsigned short *__restrict__ y, float *__restrict__ colsum,
                                                             size_t rows, int C, int ldy) {
    __shared__ float4 red[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool live = 4 * lane < C;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (size_t row = (size_t)blockIdx.x * 4 + wave; row < rows; row += (size_t)gridDim.x * 4) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (live) {
            v = reinterpret_cast<const float4 *>(x + row * C)[lane];
            if (scale) {
                const float f = scale[row];
                v.x *= f; v.y *= f; v.z *= f; v.w *= f;
            }
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        if (4 * lane < ldy)
            reinterpret_cast<ushort4 *>(y + row * ldy)[lane] = make_ushort4(f2bf(v.x), f2bf(v.y), f2bf(v.z), f2bf(v.w));
    }
    if (colsum) {                                          // here: per-workgroup partials [workgroup][C]
        red[wave][lane] = acc;
        __syncthreads();
        if (wave == 0 && live) {
            float4 a = red[0][lane];
#pragma unroll
            for (int w = 1; w < 4; ++w) {
                const float4 a2 = red[w][lane];
                a.x += a2.x; a.y += a2.y; a.z += a2.z; a.w += a2.w;
            }
            reinterpret_cast<float4 *>(colsum + (size_t)blockIdx.x * C)[lane] = a;
        }
    }
}

// ones_col: channel C (the first padding channel) holds 1.0 in EVERY row -- the weight gradient's reduction over the grid
// then leaves the column sums of the other operand there (the bias gradient: see sei_pad_nhwc_bf16_ones)
__global__ __launch_bounds__(256) void pad_nhwc_bf16_kernel(const float *__restrict__ src, unsigned short *__restrict__ dst,
                                                             int B, int H, int W, int C, int Cp, int guard, int ones_col) {
    const int Hp = H + 2, Wp = W + 2;
    const size_t rows = (size_t)B * Hp * Wp + 2 * (size_t)guard;
    const int c4 = Cp / 4;
    const size_t total = rows * c4;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const size_t r = e / c4;
        const int c = (int)(e - r * c4);
        ushort4 o = make_ushort4(ones_col && 4 * c == C ? (unsigned short)0x3f80 : (unsigned short)0, 0, 0, 0);
        if (4 * c < C && r >= (size_t)guard && r < rows - guard) {
            const size_t p = r - guard;
            const int x = (int)(p % Wp), y = (int)((p / Wp) % Hp);
            const size_t b = p / ((size_t)Wp * Hp);
            if (x >= 1 && x <= W && y >= 1 && y <= H) {
                const float4 v = reinterpret_cast<const float4 *>(src + ((b * H + (y - 1)) * W + (x - 1)) * C)[c];
                o = make_ushort4(f2bf(v.x), f2bf(v.y), f2bf(v.z), f2bf(v.w));
            }
        }
        reinterpret_cast<ushort4 *>(dst)[e] = o;
    }
}

}  // namespace

extern "C" int sei_pack(const float *src, const int *map, void *dst, size_t n, int to_bf16, void *stream) {
    SEI_REQUIRE(src && map && dst && n > 0);
    hipLaunchKernelGGL(pack_kernel, dim3(stream_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, src, map, dst, n, to_bf16);
    return sei_launch_status();
}

extern "C" int sei_unpack_add(const float *src, const int *map, float *dst, size_t n, void *stream) {
    SEI_REQUIRE(src && map && dst && n > 0);
    hipLaunchKernelGGL(unpack_add_kernel, dim3(stream_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, src, map, dst, n);
    return sei_launch_status();
}

extern "C" int sei_ln_fwd_bf16_pad(const float *x, const float *gamma, const float *beta, uint16_t *y, float *mean,
                                   float *rstd, size_t rows, int C, int ldy, float eps, int ones_col, void *stream) {
    SEI_REQUIRE(x && gamma && beta && y && mean && rstd && rows > 0 && C > 0 && C % 4 == 0 && C <= 256);
    SEI_REQUIRE(!ones_col || ldy > C);
    SEI_REQUIRE(ldy >= C && ldy % 4 == 0 && ldy <= 256 && (((uintptr_t)x | (uintptr_t)gamma | (uintptr_t)beta) & 15) == 0 &&
                ((uintptr_t)y & 7) == 0);
    hipLaunchKernelGGL(ln_fwd_bf16_pad_kernel, dim3(stream_grid(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, gamma,
                       beta, y, mean, rstd, rows, C, ldy, eps, ones_col);
    return sei_launch_status();
}

// [groups][3][C] partial sums -> a[c] += .., b[c] += .., c3[c] += .. (c3 may be null: the third sum is dropped)
int sei_fold_partials3(const float *part, int groups, int C, float *a, float *b, float *c3, hipStream_t s) {
    const int ncol = 3 * C;
    hipLaunchKernelGGL(fold_partials_kernel, dim3((unsigned)sei_ceil_div(ncol, 16)), dim3(256), 0, s, part, groups, ncol, C, a,
                       b, c3);
    return sei_launch_status();
}

constexpr unsigned PART_GROUPS = 1024;                     // workgroups (= partial rows) of the two reducing kernels

extern "C" size_t sei_swin_partials_floats(int C) { return C > 0 ? (size_t)PART_GROUPS * 2 * (size_t)C : 0; }

extern "C" int sei_ln_bwd_pad(const float *x, const float *gamma, const float *mean, const float *rstd, const float *gy,
                              const float *res, float *gx, float *ggamma, float *gbeta, size_t rows, int C, int ldg,
                              float *work, size_t work_floats, void *stream) {
    SEI_REQUIRE(x && gamma && mean && rstd && gy && gx && ggamma && gbeta && rows > 0 && C > 0 && C % 4 == 0 && C <= 256);
    SEI_REQUIRE(ldg >= C && ldg % 4 == 0 && work && work_floats >= (size_t)PART_GROUPS * 2 * C);
    SEI_REQUIRE((((uintptr_t)x | (uintptr_t)gamma | (uintptr_t)gy | (uintptr_t)res | (uintptr_t)gx | (uintptr_t)work) & 15) == 0);
    size_t g = sei_ceil_div(rows, 4);
    if (g > PART_GROUPS) g = PART_GROUPS;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(ln_bwd_pad_kernel, dim3((unsigned)g), dim3(256), 0, s, x, gamma, mean, rstd, gy, res, gx, work, rows,
                       C, ldg);
    hipLaunchKernelGGL(fold_partials_kernel, dim3((unsigned)sei_ceil_div(2 * C, 16)), dim3(256), 0, s,
                       (const float *)work, (int)g, 2 * C, C, ggamma, gbeta);
    return sei_launch_status();
}

extern "C" int sei_cast_pad_bf16(const float *x, const float *row_scale, uint16_t *y, float *colsum, size_t rows,
                                 int C, int ldy, float *work, size_t work_floats, void *stream) {
    SEI_REQUIRE(x && y && rows > 0 && C > 0 && C % 4 == 0 && C <= 256 && ldy >= C && ldy % 4 == 0 && ldy <= 256);
    SEI_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 7) == 0);
    if (colsum) SEI_REQUIRE(work && work_floats >= (size_t)PART_GROUPS * C && ((uintptr_t)work & 15) == 0);
    size_t g = sei_ceil_div(rows, 4);
    if (g > PART_GROUPS) g = PART_GROUPS;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(cast_pad_bf16_kernel, dim3((unsigned)g), dim3(256), 0, s, x, row_scale, y, colsum ? work : nullptr,
                       rows, C, ldy);
    if (colsum)
        hipLaunchKernelGGL(fold_partials_kernel, dim3((unsigned)sei_ceil_div(C, 16)), dim3(256), 0, s, (const float *)work,
                           (int)g, C, C, colsum, colsum);
    return sei_launch_status();
}

extern "C" int sei_pad_nhwc_bf16_ones(const float *x, uint16_t *xp, int B, int H, int W, int C, int Cp, int guard_rows, int ones_col,
                                 void *stream) {
    SEI_REQUIRE((!ones_col || Cp > C) && x && xp && B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && Cp >= C && Cp % 4 == 0 && guard_rows >= 0);
    SEI_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)xp & 7) == 0);
    const size_t items = ((size_t)B * (H + 2) * (W + 2) + 2 * (size_t)guard_rows) * (Cp / 4);
    hipLaunchKernelGGL(pad_nhwc_bf16_kernel, dim3(stream_grid(items, 256)), dim3(256), 0, (hipStream_t)stream, x, xp, B,
                       H, W, C, Cp, guard_rows, ones_col);
    return sei_launch_status();
}

extern "C" int sei_pad_nhwc_bf16(const float *x, uint16_t *xp, int B, int H, int W, int C, int Cp, int guard_rows,
                                 void *stream) {
    return sei_pad_nhwc_bf16_ones(x, xp, B, H, W, C, Cp, guard_rows, 0, stream);
}
