// sei_gemm_f32: every 1x1 convolution of the U-Net (reference src/models/convolutional.py:40,42,106,143)
// and their data/weight gradients, as a row-major GEMM on the exact-f32 matrix cores of gfx950.
//
//   D[M,N] = op(A)[M,K] * op(B)[K,N]  (+ fused epilogue)
//
// MFMA: v_mfma_f32_32x32x2_f32 (f32 in / f32 accumulate, 64 FLOP/clk/SIMD = 157 TFLOP/s chip peak). One workgroup =
// WM x WN waves, each wave owns TM x TN accumulator tiles of 32x32. K is walked in steps of BK = 32: the next A/B tiles
// are prefetched from HBM/L2 into registers while the current ones are consumed from LDS (issue-early / write-late).
//
// The MFMA takes k = 0 from lanes 0-31 and k = 1 from lanes 32-63.  Which k of the tile those are is free as long as A and
// B agree, so step kk of a tile feeds k = kk from the low lane half and k = BK/2 + kk from the high half: a lane then needs
// BK/2 CONSECUTIVE k of its row.  A K-contiguous global operand is therefore kept as it lies, s[o][k] with a row stride of
// BK + 4 floats (ds_write_b128 in, ds_read_b128 out, 36-float stride = conflict-free for 16-lane b128 groups); an
// M/N-contiguous one is stored k-major, s[k][o] with a row pad of 4, and read with consecutive-lane ds_read_b32.
//
// Roofline: compute-bound on the f32 MFMA pipe for the U-Net shapes (arithmetic intensity >= 64
// FLOP/B at the 128x128 tile); algorithmic FLOPs = 2*M*N*K per call.
#include "sei_common.h"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

#ifndef SEI_F32_BK
#define SEI_F32_BK 32
#endif
constexpr int BK = SEI_F32_BK;
struct GemmArgs {
    const float *A, *B;
    float *D;
    int M, N, K;
    int epilogue;
    const float *bias, *R1, *R2;
    float *D2;
    int splitk;       // >1: K is split over blockIdx.z and partial tiles are atomically added to D
    int k_per_split;  // multiple of BK
    int batch;        // blockIdx.z / splitk indexes a batch of independent GEMMs
    long long strideA, strideB, strideD;
};

// ---- global -> register tile loads ---------------------------------------------------------------
// Tile of ROWS "outer" indices (m or n) by BK k-indices, as NV float4 per thread.
// KCONTIG: element (o, k) at base[o*ld + k]  (ld = K);  else at base[k*ld + o]  (ld = M or N).
template <int ROWS, int NT, bool KCONTIG>
struct TileLoader {
    static constexpr int NV = (ROWS * BK / 4 + NT - 1) / NT;

    __device__ __forceinline__ static void load(float4 (&v)[NV], const float *__restrict__ base, int ld,
                                                int o0, int k0, int o_lim, int k_lim, bool aligned) {
        // Interior k-tiles of an aligned operand take unconditional float4 loads: an outer index past the edge is clamped
        // onto the last valid row (or the last aligned group of four) -- what it fetches only ever reaches accumulator rows /
        // columns that the epilogue does not store.  The test is wave-uniform, so the main loop carries one scalar branch
        // instead of a divergent region per load.
        if (aligned && k0 + BK <= k_lim && o_lim >= 4 && (ROWS * BK / 4) % NT == 0) {
#pragma unroll
            for (int it = 0; it < NV; ++it) {
                const int f = threadIdx.x + it * NT;
                if (KCONTIG) {
                    const int o = min(o0 + f / (BK / 4), o_lim - 1), k = k0 + ((f % (BK / 4)) << 2);
                    v[it] = *reinterpret_cast<const float4 *>(base + (size_t)o * ld + k);
                } else {
                    const int k = k0 + f / (ROWS / 4), o = min(o0 + ((f % (ROWS / 4)) << 2), o_lim - 4);
                    v[it] = *reinterpret_cast<const float4 *>(base + (size_t)k * ld + o);
                }
            }
            return;
        }
#pragma unroll
        for (int it = 0; it < NV; ++it) {
            const int f = threadIdx.x + it * NT;
            float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
            if (f < ROWS * BK / 4) {
                if (KCONTIG) {
                    const int o = o0 + f / (BK / 4), k = k0 + ((f % (BK / 4)) << 2);
                    if (o < o_lim) {
                        const float *p = base + (size_t)o * ld + k;
                        if (aligned && k + 3 < k_lim) {
                            r = *reinterpret_cast<const float4 *>(p);
                        } else {
                            if (k < k_lim) r.x = p[0];
                            if (k + 1 < k_lim) r.y = p[1];
                            if (k + 2 < k_lim) r.z = p[2];
                            if (k + 3 < k_lim) r.w = p[3];
                        }
                    }
                } else {
                    const int k = k0 + f / (ROWS / 4), o = o0 + ((f % (ROWS / 4)) << 2);
                    if (k < k_lim) {
                        const float *p = base + (size_t)k * ld + o;
                        if (aligned && o + 3 < o_lim) {
                            r = *reinterpret_cast<const float4 *>(p);
                        } else {
                            if (o < o_lim) r.x = p[0];
                            if (o + 1 < o_lim) r.y = p[1];
                            if (o + 2 < o_lim) r.z = p[2];
                            if (o + 3 < o_lim) r.w = p[3];
                        }
                    }
                }
            }
            v[it] = r;
        }
    }

    // LDS image: s[o][k] for a K-contiguous source, s[k][o] otherwise (see the file header)
    static constexpr int LD = KCONTIG ? BK + 4 : ROWS + 4;
    static constexpr int SIZE = KCONTIG ? ROWS * LD : BK * LD;

    __device__ __forceinline__ static void store(const float4 (&v)[NV], float *__restrict__ s) {
#pragma unroll
        for (int it = 0; it < NV; ++it) {
            const int f = threadIdx.x + it * NT;
            if (f < ROWS * BK / 4) {
                if (KCONTIG) {
                    const int o = f / (BK / 4), k = (f % (BK / 4)) << 2;
                    *reinterpret_cast<float4 *>(s + o * LD + k) = v[it];
                } else {
                    const int k = f / (ROWS / 4), o = (f % (ROWS / 4)) << 2;
                    *reinterpret_cast<float4 *>(s + k * LD + o) = v[it];
                }
            }
        }
    }

    // MFMA operand values of outer index o for the steps 4q .. 4q+3 of the tile, lane half lh
    __device__ __forceinline__ static void fragment(float (&out)[4], const float *__restrict__ s, int o, int lh, int q) {
        if (KCONTIG) {
            const float4 t = *reinterpret_cast<const float4 *>(s + o * LD + lh * (BK / 2) + 4 * q);
            out[0] = t.x, out[1] = t.y, out[2] = t.z, out[3] = t.w;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) out[e] = s[(lh * (BK / 2) + 4 * q + e) * LD + o];
        }
    }
};

template <int TM, int TN, int WM, int WN, bool TRANSA, bool TRANSB>
__global__ __launch_bounds__(WM *WN * 64) void gemm_f32_kernel(GemmArgs g) {
    constexpr int NT = WM * WN * 64;
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
    using LA = TileLoader<BM, NT, !TRANSA>;   // A (M,K) row-major is K-contiguous when !TRANSA
    using LB = TileLoader<BN, NT, TRANSB>;    // B (N,K) row-major is K-contiguous when TRANSB
    __shared__ __attribute__((aligned(16))) float As[LA::SIZE];
    __shared__ __attribute__((aligned(16))) float Bs[LB::SIZE];

    const int zb = blockIdx.z / g.splitk, zs = blockIdx.z - zb * g.splitk;
    const float *__restrict__ A = g.A + (size_t)zb * g.strideA;
    const float *__restrict__ B = g.B + (size_t)zb * g.strideB;
    float *__restrict__ D = g.D + (size_t)zb * g.strideD;
    const int M = g.M, N = g.N, K = g.K;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int k_begin = zs * g.k_per_split;
    const int k_end = min(K, k_begin + g.k_per_split);

    const int lda = TRANSA ? M : K, ldb = TRANSB ? K : N;
    const bool a_al = ((lda & 3) == 0) && ((reinterpret_cast<uintptr_t>(A) & 15) == 0);
    const bool b_al = ((ldb & 3) == 0) && ((reinterpret_cast<uintptr_t>(B) & 15) == 0);

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, lh = lane >> 5;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float4 ra[LA::NV], rb[LB::NV];
    LA::load(ra, A, lda, m0, k_begin, M, k_end, a_al);
    LB::load(rb, B, ldb, n0, k_begin, N, k_end, b_al);

    // One LDS image per operand, two barriers per k-tile; the second workgroup of the CU fills the matrix pipe across them.
    // (Two images and one barrier, the next tile written in front of the last quarter of the MFMAs, measured SLOWER on
    // every U-Net shape -- 33.6 vs 27.9 ms over tools/bench_gemm.py's list -- and is not kept.)
    for (int k0 = k_begin; k0 < k_end; k0 += BK) {
        __syncthreads();                 // previous tile fully consumed
        LA::store(ra, As);
        LB::store(rb, Bs);
        __syncthreads();
        if (k0 + BK < k_end) {           // prefetch the next tile under this tile's MFMAs
            LA::load(ra, A, lda, m0, k0 + BK, M, k_end, a_al);
            LB::load(rb, B, ldb, n0, k0 + BK, N, k_end, b_al);
        }
#pragma unroll
        for (int q = 0; q < BK / 8; ++q) {
            float a[TM][4], b[TN][4];
#pragma unroll
            for (int i = 0; i < TM; ++i) LA::fragment(a[i], As, wm * (32 * TM) + 32 * i + li, lh, q);
#pragma unroll
            for (int j = 0; j < TN; ++j) LB::fragment(b[j], Bs, wn * (32 * TN) + 32 * j + li, lh, q);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], b[j][e], acc[i][j], 0, 0, 0);
        }
    }

    // ---- epilogue: C/D map col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5) ---------------------
    const int epi = g.epilogue;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + wn * (32 * TN) + 32 * j + li;
            if (col >= N) continue;
            const float bias = (epi == SEI_EPI_BIAS || epi == SEI_EPI_BIAS_GELU || epi == SEI_EPI_BIAS_RES ||
                                epi == SEI_EPI_BIAS_ROWSCALE || epi == SEI_EPI_BIAS_SCALE_RES) ? g.bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * (32 * TM) + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (row >= M) continue;
                const size_t o = (size_t)row * N + col;
                float v = acc[i][j][r];
                if (g.splitk > 1) {
                    atomicAdd(D + o, v);
                    continue;
                }
                switch (epi) {
                    case SEI_EPI_BIAS: v += bias; break;
                    case SEI_EPI_BIAS_GELU:
                        v += bias;
                        g.D2[o + (size_t)zb * g.strideD] = sei_gelu(v);
                        break;
                    case SEI_EPI_BIAS_RES:
                        v += bias;
                        v += g.R1[o];
                        if (g.R2) v += g.R2[o];
                        break;
                    case SEI_EPI_MUL_DGELU: v *= sei_dgelu(g.R1[o]); break;
                    case SEI_EPI_ACCUM: v += D[o]; break;
                    case SEI_EPI_BIAS_ROWSCALE: v += bias * g.R1[row]; break;
                    case SEI_EPI_BIAS_SCALE_RES: v = g.R2[o] + g.R1[row] * (v + bias); break;
                    default: break;
                }
                D[o] = v;
            }
        }
    }
}

template <int TM, int TN, bool TA, bool TB, int WM = 2, int WN = 2>
int launch(const GemmArgs &g, hipStream_t s) {
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
    dim3 grid((unsigned)sei_ceil_div(g.N, BN), (unsigned)sei_ceil_div(g.M, BM), (unsigned)(g.splitk * g.batch));
    hipLaunchKernelGGL((gemm_f32_kernel<TM, TN, WM, WN, TA, TB>), grid, dim3(WM * WN * 64), 0, s, g);
    return sei_launch_status();
}

template <int TM, int TN>
int dispatch_layout(const GemmArgs &g, int ta, int tb, hipStream_t s) {
    if (!ta && tb) return launch<TM, TN, false, true>(g, s);
    if (!ta && !tb) return launch<TM, TN, false, false>(g, s);
    if (ta && !tb) return launch<TM, TN, true, false>(g, s);
    return launch<TM, TN, true, true>(g, s);
}

// 96 x 128 tiles (one row of four waves, three 32 x 32 accumulator tiles each): the bottleneck level's activation
// matrices have 288 or 576 rows (9 pixels x 32 or 64 crops), which 128-row tiles cover with 25 % / 10 % of their MFMAs
// on rows that do not exist.
int dispatch_layout_96(const GemmArgs &g, int ta, int tb, hipStream_t s) {
    if (!ta && tb) return launch<3, 1, false, true, 1, 4>(g, s);
    if (!ta && !tb) return launch<3, 1, false, false, 1, 4>(g, s);
    if (ta && !tb) return launch<3, 1, true, false, 1, 4>(g, s);
    return launch<3, 1, true, true, 1, 4>(g, s);
}

inline size_t tiles(int M, int N, int bm, int bn) { return sei_ceil_div(M, bm) * sei_ceil_div(N, bn); }

}  // namespace

// Tile and split-K choice: fill the 256 CUs (>= ~2 workgroups each where the problem allows) with the
// largest tile; when even 64x64 tiles leave the chip idle and K is long (weight gradients: K = pixels),
// split K and combine with float atomics (gradients are accumulated, so D is already an accumulator).
extern "C" int sei_gemm_f32_ex(const float *A, const float *B, float *D, int M, int N, int K, int transA,
                               int transB, int epilogue, const float *bias, const float *R1,
                               const float *R2, float *D2, int batch, long long strideA, long long strideB,
                               long long strideD, int allow_splitk, void *stream) {
    SEI_REQUIRE(A && B && D && M > 0 && N > 0 && K > 0 && batch > 0);
    SEI_REQUIRE(epilogue >= SEI_EPI_NONE && epilogue <= SEI_EPI_BIAS_SCALE_RES);
    if (epilogue == SEI_EPI_BIAS || epilogue == SEI_EPI_BIAS_GELU || epilogue == SEI_EPI_BIAS_RES ||
        epilogue == SEI_EPI_BIAS_ROWSCALE || epilogue == SEI_EPI_BIAS_SCALE_RES)
        SEI_REQUIRE(bias);
    if (epilogue == SEI_EPI_BIAS_GELU) SEI_REQUIRE(D2);
    if (epilogue == SEI_EPI_BIAS_RES || epilogue == SEI_EPI_MUL_DGELU || epilogue == SEI_EPI_BIAS_ROWSCALE ||
        epilogue == SEI_EPI_BIAS_SCALE_RES)
        SEI_REQUIRE(R1);
    if (epilogue == SEI_EPI_BIAS_SCALE_RES) SEI_REQUIRE(R2);
    GemmArgs g;
    g.A = A; g.B = B; g.D = D; g.M = M; g.N = N; g.K = K; g.epilogue = epilogue;
    g.bias = bias; g.R1 = R1; g.R2 = R2; g.D2 = D2;
    g.batch = batch; g.strideA = strideA; g.strideB = strideB; g.strideD = strideD;
    g.splitk = 1;
    g.k_per_split = (int)(sei_ceil_div(K, BK) * BK);

    const size_t want = 512;
    int tm = 1, tn = 1;
    if (tiles(M, N, 128, 128) * batch >= want) { tm = 2; tn = 2; }
    else if (tiles(M, N, 64, 128) * batch >= want) { tm = 1; tn = 2; }
    const size_t t = tiles(M, N, 64 * tm, 64 * tn) * batch;
    if (allow_splitk && epilogue == SEI_EPI_ACCUM && t < 256 && K >= 16 * BK) {
        size_t sk = sei_ceil_div(want, t);
        const size_t max_sk = (size_t)K / (8 * BK);       // at least 8 k-tiles per split
        if (sk > max_sk) sk = max_sk;
        if (sk > 1) {
            g.k_per_split = (int)(sei_ceil_div(sei_ceil_div(K, sk), BK) * BK);
            g.splitk = (int)sei_ceil_div(K, g.k_per_split);
        }
    }
    hipStream_t s = (hipStream_t)stream;
    // rows in whole 96-row tiles but not in whole 128-row ones, and enough tiles to fill the chip: no padded rows
    if (tm == 2 && g.splitk == 1 && M % 96 == 0 && M % 128 != 0 && tiles(M, N, 96, 128) * batch >= want)
        return dispatch_layout_96(g, transA, transB, s);
    if (tm == 2) return dispatch_layout<2, 2>(g, transA, transB, s);
    if (tn == 2) return dispatch_layout<1, 2>(g, transA, transB, s);
    return dispatch_layout<1, 1>(g, transA, transB, s);
}

extern "C" int sei_gemm_f32(const float *A, const float *B, float *D, int M, int N, int K, int transA,
                            int transB, int epilogue, const float *bias, const float *R1, const float *R2,
                            float *D2, void *stream) {
    return sei_gemm_f32_ex(A, B, D, M, N, K, transA, transB, epilogue, bias, R1, R2, D2, 1, 0, 0, 0, 1, stream);
}
