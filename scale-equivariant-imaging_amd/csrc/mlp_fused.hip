// Fused pointwise MLP of a ConvBlock for the SHALLOW levels of the U-Net (C = 32, 128) in throughput (bf16) mode:
//     out = res_scale * x + conv3(gelu(conv2(h2)))        (reference: src/models/convolutional.py:40-51)
// where conv2 (C -> 4C) and conv3 (4C -> C) are 1x1 convolutions = GEMMs over the (pixels, channels) view.
//
// Why: at these levels the two GEMMs hold 0.2 % of the step's FLOPs but were 1.8 ms of it, because the 4C-wide hidden
// activation crossed HBM five times per pass (f32 pre-GELU h3 and bf16 gelu(h3) written by conv2, h4 read by conv3,
// h3 and h4 read again by the backward pass): 1344 B per pixel forward against 320 B of true inputs and outputs.
// Here the hidden activation never leaves the registers:
//
//   forward    per wave and 32-pixel tile: H^T = W2 h2^T as 32x32 MFMA tiles with the hidden unit on the accumulator
//              ROWS and the pixel on the lane, bias + GELU in registers, and the tile is re-used directly as the A
//              operand of the second product (cdna_hip_programming.md section 3, "an accumulator tile as the next
//              MFMA's operand"): out += (H^T tile)^T W3^T sums over the tile's rows = hidden units. No LDS at all.
//   backward   recomputes h3 = conv2(h2) the same way (bit-identical to the forward), G^T = W3^T go^T,
//              gh3 = G * gelu'(h3), gh2 = gh3 W2 through the same accumulator-as-operand step, and writes what
//              the two weight-gradient GEMMs (sei_gemm_bf16nt_dw2, merged across the step's model calls) need --
//              bf16 go, gelu(h3) and gh3 -- from a second pair of products with the operands swapped (pixels on the
//              accumulator rows), whose stores are contiguous along the hidden dimension. The bias gradients are
//              summed in registers and leave as a few float atomics per workgroup.
//
// The same values as the unfused path: h3 in f32, gelu / gelu' by sei_gelu_bf16out / sei_dgelu_bf16out, h4 and gh3
// rounded to bf16 once.
#include "sei_common.h"

namespace {

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x4 = __attribute__((ext_vector_type(4))) __bf16;
using f32x16 = __attribute__((ext_vector_type(16))) float;

__device__ __forceinline__ f32x16 mfma(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ int acc_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

// rows r0 .. r0+31 of a row-major bf16 matrix with `ld` columns as the A operand (row = lane & 31) or as the B operand
// of its transpose (column = lane & 31), k-step s: k = 16 s + 8 (lane >> 5) + 0..7
__device__ __forceinline__ bf16x8 row_frag(const unsigned short *m, int ld, int r0, int s, int lane) {
    return *reinterpret_cast<const bf16x8 *>(m + (size_t)(r0 + (lane & 31)) * ld + 16 * s + 8 * (lane >> 5));
}
// B operand [k][col] = m[col0 + col][k0 + k] for a k-step whose A operand is an accumulator tile converted in place:
// element e of lane half h is k = 8 (e >> 2) + 4 h + (e & 3): two 8-byte pieces of row col0 + (lane & 31)
__device__ __forceinline__ bf16x8 perm_frag(const unsigned short *m, int ld, int col0, int k0, int lane) {
    const unsigned short *p = m + (size_t)(col0 + (lane & 31)) * ld + k0 + 4 * (lane >> 5);
    const bf16x4 lo = *reinterpret_cast<const bf16x4 *>(p), hi = *reinterpret_cast<const bf16x4 *>(p + 8);
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}
__device__ __forceinline__ bf16x8 acc_frag(const f32x16 &x, int s) {
    bf16x8 a;
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] = (__bf16)x[8 * s + e];
    return a;
}
__device__ __forceinline__ unsigned short f2bf(float v) {
    const __bf16 b = (__bf16)v;
    return __builtin_bit_cast(unsigned short, b);
}

constexpr int WAVES = 4;

// ---------------------------------------------------------------------------------------------------------------
template <int C, int RT>
__global__ __launch_bounds__(64 * WAVES) void mlp_fwd_kernel(const unsigned short *__restrict__ h2,
                                                              const unsigned short *__restrict__ W2,
                                                              const float *__restrict__ b2,
                                                              const unsigned short *__restrict__ W3,
                                                              const float *__restrict__ b3, const float *__restrict__ x,
                                                              float res_scale, float *__restrict__ out, int M) {
    constexpr int KS = C / 16, CT = C / 32, HT = 4 * C / 32;
    const int lane = threadIdx.x & 63;
    const int wave_global = blockIdx.x * WAVES + (threadIdx.x >> 6), wave_count = gridDim.x * WAVES;
    const int ntile = (M + 32 * RT - 1) / (32 * RT);
    for (int tile = wave_global; tile < ntile; tile += wave_count) {
        const int row0 = tile * 32 * RT;
        bf16x8 f[RT][KS];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const int r = min(row0 + 32 * rt + (lane & 31), M - 1);
#pragma unroll
            for (int s = 0; s < KS; ++s)
                f[rt][s] = *reinterpret_cast<const bf16x8 *>(h2 + (size_t)r * C + 16 * s + 8 * (lane >> 5));
        }
        f32x16 acc[RT][CT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) acc[rt][ct] = f32x16{0};
#pragma unroll 1
        for (int ht = 0; ht < HT; ++ht) {
            f32x16 hT[RT];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) hT[rt] = f32x16{0};
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const bf16x8 w = row_frag(W2, C, 32 * ht, s, lane);
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) hT[rt] = mfma(w, f[rt][s], hT[rt]);
            }
            float bias[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) bias[r] = b2[32 * ht + acc_row(r, lane)];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int r = 0; r < 16; ++r) hT[rt][r] = sei_gelu_bf16out(hT[rt][r] + bias[r]);
#pragma unroll
            for (int s = 0; s < 2; ++s) {
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    const bf16x8 w = perm_frag(W3, 4 * C, 32 * ct, 32 * ht + 16 * s, lane);
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) acc[rt][ct] = mfma(acc_frag(hT[rt], s), w, acc[rt][ct]);
                }
            }
        }
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const int c = 32 * ct + (lane & 31);
                const float bc = b3[c];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = row0 + 32 * rt + acc_row(r, lane);
                    if (row < M) {
                        const size_t o = (size_t)row * C + c;
                        out[o] = fmaf(res_scale, x[o], acc[rt][ct][r] + bc);
                    }
                }
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------
template <int C, int RT>
__global__ __launch_bounds__(64 * WAVES) void mlp_bwd_kernel(const float *__restrict__ go,
                                                              const unsigned short *__restrict__ h2,
                                                              const unsigned short *__restrict__ W2,
                                                              const float *__restrict__ b2,
                                                              const unsigned short *__restrict__ W3T,
                                                              const unsigned short *__restrict__ W2T,
                                                              float *__restrict__ gh2, unsigned short *__restrict__ go16,
                                                              unsigned short *__restrict__ h4,
                                                              unsigned short *__restrict__ gh3, float *__restrict__ db3,
                                                              float *__restrict__ db2, int M) {
    constexpr int KS = C / 16, CT = C / 32, HT = 4 * C / 32;
    __shared__ float red_b2[4 * C], red_b3[C];
    for (int i = threadIdx.x; i < 4 * C; i += 64 * WAVES) red_b2[i] = 0.f;
    for (int i = threadIdx.x; i < C; i += 64 * WAVES) red_b3[i] = 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave_global = blockIdx.x * WAVES + (threadIdx.x >> 6), wave_count = gridDim.x * WAVES;
    const int ntile = (M + 32 * RT - 1) / (32 * RT);
    float sum_b3[KS][8];
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int e = 0; e < 8; ++e) sum_b3[s][e] = 0.f;
    for (int tile = wave_global; tile < ntile; tile += wave_count) {
        const int row0 = tile * 32 * RT;
        bf16x8 f[RT][KS], g[RT][KS];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const int rr = row0 + 32 * rt + (lane & 31);
            const bool live = rr < M;
            const int r = min(rr, M - 1);
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const size_t o = (size_t)r * C + 16 * s + 8 * (lane >> 5);
                f[rt][s] = *reinterpret_cast<const bf16x8 *>(h2 + o);
                const float4 a = *reinterpret_cast<const float4 *>(go + o), b = *reinterpret_cast<const float4 *>(go + o + 4);
                const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
                bf16x8 t;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float ve = live ? v[e] : 0.f;
                    t[e] = (__bf16)ve;
                    sum_b3[s][e] += ve;
                }
                g[rt][s] = t;
                if (live) *reinterpret_cast<bf16x8 *>(go16 + o) = t;
            }
        }
        f32x16 acc[RT][CT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) acc[rt][ct] = f32x16{0};
#pragma unroll 1
        for (int ht = 0; ht < HT; ++ht) {
            bf16x8 w2f[KS], w3f[KS];
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                w2f[s] = row_frag(W2, C, 32 * ht, s, lane);
                w3f[s] = row_frag(W3T, C, 32 * ht, s, lane);
            }
            // orientation 1: hidden unit on the accumulator rows -> gh2 = gh3 W2 (reduction over hidden)
            {
                float bias[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) bias[r] = b2[32 * ht + acc_row(r, lane)];
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    f32x16 hT = {0}, gT = {0};
#pragma unroll
                    for (int s = 0; s < KS; ++s) {
                        hT = mfma(w2f[s], f[rt][s], hT);
                        gT = mfma(w3f[s], g[rt][s], gT);
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) gT[r] *= sei_dgelu_bf16out(hT[r] + bias[r]);
#pragma unroll
                    for (int s = 0; s < 2; ++s)
#pragma unroll
                        for (int ct = 0; ct < CT; ++ct)
                            acc[rt][ct] = mfma(acc_frag(gT, s), perm_frag(W2T, 4 * C, 32 * ct, 32 * ht + 16 * s, lane),
                                               acc[rt][ct]);
                }
            }
            // orientation 2: pixel on the accumulator rows, hidden unit on the lane -> contiguous stores of gelu(h3), gh3
            {
                const int hid = 32 * ht + (lane & 31);
                const float bias = b2[hid];
                float colsum = 0.f;
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    f32x16 h = {0}, gg = {0};
#pragma unroll
                    for (int s = 0; s < KS; ++s) {
                        h = mfma(f[rt][s], w2f[s], h);
                        gg = mfma(g[rt][s], w3f[s], gg);
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = row0 + 32 * rt + acc_row(r, lane);
                        const float h3 = h[r] + bias;
                        float cdf, pdf;
                        sei_phi_pdf_bf16out(h3, cdf, pdf);
                        const float gv = gg[r] * fmaf(h3, pdf, cdf);
                        if (row < M) {
                            const size_t o = (size_t)row * (4 * C) + hid;
                            h4[o] = f2bf(h3 * cdf);
                            gh3[o] = f2bf(gv);
                            colsum += gv;
                        }
                    }
                }
                colsum += __shfl_xor(colsum, 32, 64);
                if (lane < 32) atomicAdd(&red_b2[hid], colsum);
            }
        }
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const int c = 32 * ct + (lane & 31);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = row0 + 32 * rt + acc_row(r, lane);
                    if (row < M) gh2[(size_t)row * C + c] = acc[rt][ct][r];
                }
            }
    }
    // bias gradient of conv3: every lane holds sums of its pixels for channels 16 s + 8 (lane >> 5) + e
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float v = sum_b3[s][e];
#pragma unroll
            for (int off = 16; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
            if ((lane & 31) == 0) atomicAdd(&red_b3[16 * s + 8 * (lane >> 5) + e], v);
        }
    __syncthreads();
    for (int i = threadIdx.x; i < 4 * C; i += 64 * WAVES) atomicAdd(db2 + i, red_b2[i]);
    for (int i = threadIdx.x; i < C; i += 64 * WAVES) atomicAdd(db3 + i, red_b3[i]);
}

inline unsigned mlp_grid(int M, int rows_per_wave) {
    const size_t tiles = sei_ceil_div((size_t)M, (size_t)rows_per_wave);
    size_t g = sei_ceil_div(tiles, (size_t)WAVES);
    return (unsigned)(g < 1 ? 1 : (g > 1024 ? 1024 : g));
}

}  // namespace

extern "C" int sei_mlp_fused_fwd(const uint16_t *h2, const uint16_t *W2, const float *b2, const uint16_t *W3,
                                 const float *b3, const float *x, float res_scale, float *out, int M, int C,
                                 void *stream) {
    SEI_REQUIRE(h2 && W2 && b2 && W3 && b3 && x && out && M > 0 && (C == 32 || C == 128));
    SEI_REQUIRE((((uintptr_t)h2 | (uintptr_t)W2 | (uintptr_t)W3) & 15) == 0);
    hipStream_t s = (hipStream_t)stream;
    if (C == 32)
        hipLaunchKernelGGL((mlp_fwd_kernel<32, 2>), dim3(mlp_grid(M, 64)), dim3(64 * WAVES), 0, s, h2, W2, b2, W3, b3, x,
                           res_scale, out, M);
    else
        hipLaunchKernelGGL((mlp_fwd_kernel<128, 1>), dim3(mlp_grid(M, 32)), dim3(64 * WAVES), 0, s, h2, W2, b2, W3, b3, x,
                           res_scale, out, M);
    return sei_launch_status();
}

extern "C" int sei_mlp_fused_bwd(const float *go, const uint16_t *h2, const uint16_t *W2, const float *b2,
                                 const uint16_t *W3T, const uint16_t *W2T, float *gh2, uint16_t *go16, uint16_t *h4,
                                 uint16_t *gh3, float *db3, float *db2, int M, int C, void *stream) {
    SEI_REQUIRE(go && h2 && W2 && b2 && W3T && W2T && gh2 && go16 && h4 && gh3 && db3 && db2 && M > 0 &&
                (C == 32 || C == 128));
    SEI_REQUIRE((((uintptr_t)go | (uintptr_t)h2 | (uintptr_t)W2 | (uintptr_t)W3T | (uintptr_t)W2T | (uintptr_t)go16) & 15) == 0);
    hipStream_t s = (hipStream_t)stream;
    if (C == 32)
        hipLaunchKernelGGL((mlp_bwd_kernel<32, 1>), dim3(mlp_grid(M, 32)), dim3(64 * WAVES), 0, s, go, h2, W2, b2, W3T, W2T,
                           gh2, go16, h4, gh3, db3, db2, M);
    else
        hipLaunchKernelGGL((mlp_bwd_kernel<128, 1>), dim3(mlp_grid(M, 32)), dim3(64 * WAVES), 0, s, go, h2, W2, b2, W3T,
                           W2T, gh2, go16, h4, gh3, db3, db2, M);
    return sei_launch_status();
}
