// sei_gemm_f32: every 1x1 convolution of the U-Net (reference src/models/convolutional.py:40,42,106,143)
// and their data/weight gradients, as a row-major GEMM on the exact-f32 matrix cores of gfx950.
//
//   D[M,N] = op(A)[M,K] * op(B)[K,N]  (+ fused epilogue)
//
// MFMA: v_mfma_f32_32x32x2_f32 (f32 in / f32 accumulate, bit-exact fmaf chain, 64 FLOP/clk/SIMD =
// 157 TFLOP/s chip peak). One workgroup = WM x WN waves, each wave owns TM x TN accumulator tiles of
// 32x32. K is walked in steps of BK = 16: the next A/B tiles are prefetched from HBM/L2 into
// registers while the current ones are consumed from LDS (issue-early / write-late staging).
//
// LDS image is k-major for both operands -- As[k][m], Bs[k][n] with a row pad of 4 floats -- so the
// MFMA fragment read (lane l supplies A[m = l&31][k = l>>5], B[k = l>>5][n = l&31]) is a
// consecutive-lane ds_read_b32: conflict-free for either global layout. A K-contiguous global
// operand is transposed on its way into LDS (float4 global load -> 4 scalar LDS writes, 2-way bank
// aliasing which ds_write_b32 absorbs); an M/N-contiguous one is stored with ds_write_b128.
//
// Roofline: compute-bound on the f32 MFMA pipe for the U-Net shapes (arithmetic intensity >= 64
// FLOP/B at the 128x128 tile); algorithmic FLOPs = 2*M*N*K per call.
#include "sei_common.h"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int BK = 16;
constexpr int PAD = 4;

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
#pragma unroll
        for (int it = 0; it < NV; ++it) {
            const int f = threadIdx.x + it * NT;
            float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
            if (f < ROWS * BK / 4) {
                if (KCONTIG) {
                    const int o = o0 + (f >> 2), k = k0 + ((f & 3) << 2);
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

    // registers -> LDS image s[k][o], row stride ROWS + PAD
    __device__ __forceinline__ static void store(const float4 (&v)[NV], float *__restrict__ s) {
        constexpr int LD = ROWS + PAD;
#pragma unroll
        for (int it = 0; it < NV; ++it) {
            const int f = threadIdx.x + it * NT;
            if (f < ROWS * BK / 4) {
                if (KCONTIG) {
                    const int o = f >> 2, k = (f & 3) << 2;
                    s[(k + 0) * LD + o] = v[it].x;
                    s[(k + 1) * LD + o] = v[it].y;
                    s[(k + 2) * LD + o] = v[it].z;
                    s[(k + 3) * LD + o] = v[it].w;
                } else {
                    const int k = f / (ROWS / 4), o = (f % (ROWS / 4)) << 2;
                    *reinterpret_cast<float4 *>(s + k * LD + o) = v[it];
                }
            }
        }
    }
};

template <int TM, int TN, int WM, int WN, bool TRANSA, bool TRANSB>
__global__ __launch_bounds__(WM *WN * 64) void gemm_f32_kernel(GemmArgs g) {
    constexpr int NT = WM * WN * 64;
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
    constexpr int LDA = BM + PAD, LDB = BN + PAD;
    __shared__ __attribute__((aligned(16))) float As[BK * LDA];
    __shared__ __attribute__((aligned(16))) float Bs[BK * LDB];

    using LA = TileLoader<BM, NT, !TRANSA>;   // A (M,K) row-major is K-contiguous when !TRANSA
    using LB = TileLoader<BN, NT, TRANSB>;    // B (N,K) row-major is K-contiguous when TRANSB

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

    for (int k0 = k_begin; k0 < k_end; k0 += BK) {
        __syncthreads();                 // previous tile fully consumed
        LA::store(ra, As);
        LB::store(rb, Bs);
        __syncthreads();
        if (k0 + BK < k_end) {           // prefetch the next tile under this tile's MFMAs
            LA::load(ra, A, lda, m0, k0 + BK, M, k_end, a_al);
            LB::load(rb, B, ldb, n0, k0 + BK, N, k_end, b_al);
        }
        const float *as = As + lh * LDA + wm * (32 * TM) + li;
        const float *bs = Bs + lh * LDB + wn * (32 * TN) + li;
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
            float a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = as[(2 * kk) * LDA + 32 * i];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = bs[(2 * kk) * LDB + 32 * j];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
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

template <int TM, int TN, bool TA, bool TB>
int launch(const GemmArgs &g, hipStream_t s) {
    constexpr int WM = 2, WN = 2;
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
    if (tm == 2) return dispatch_layout<2, 2>(g, transA, transB, s);
    if (tn == 2) return dispatch_layout<1, 2>(g, transA, transB, s);
    return dispatch_layout<1, 1>(g, transA, transB, s);
}

extern "C" int sei_gemm_f32(const float *A, const float *B, float *D, int M, int N, int K, int transA,
                            int transB, int epilogue, const float *bias, const float *R1, const float *R2,
                            float *D2, void *stream) {
    return sei_gemm_f32_ex(A, B, D, M, N, K, transA, transB, epilogue, bias, R1, R2, D2, 1, 0, 0, 0, 1, stream);
}
