// sei_gemm_bf16: the same GEMM contract as sei_gemm_f32 (1x1 convolutions of the U-Net and their
// gradients, reference src/models/convolutional.py:40,42,106,143) on the bf16 matrix cores of gfx950.
//
//   D[M,N] (f32) = op(A)[M,K] * op(B)[K,N],  A and B are FLOAT32 in HBM, rounded to bf16 (RNE,
//   v_cvt_pk_bf16_f32) on their way into LDS; products accumulate in f32 (v_mfma_f32_32x32x16_bf16).
//
// This is the throughput mode (BASELINE.json configs quote bf16); parity claims are made in f32.
//
// LDS image, both operands: [kblk = 4][rows + 2][8 bf16]  -- 16-byte cells holding 8 consecutive k of one
// row. The MFMA fragment of lane l (row l&31, k-half l>>5) is ONE cell: a ds_read_b128 whose 16-lane
// groups cover 256 contiguous bytes (conflict-free); the +2 row pad makes the ds_write_b128 of the
// k-contiguous loader conflict-free as well (cell index kq*(rows+2)+row -> 8 distinct 16-B slots).
// Two loaders fill it from either global layout:
//   KC (k contiguous):     2 x float4 along k            -> 1 cell
//   OC (outer contiguous): 8 x float2 (8 k-rows, 2 cols) -> 2 cells (register transpose)
// Double-buffered LDS, one barrier per 32-deep k-tile, next tile's global loads in flight under the MFMAs.
#include "sei_common.h"

namespace {

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int BK = 32;
constexpr int KBLK = BK / 8;

struct GemmArgs {
    const void *A, *B;        // float or bf16 (uint16_t) per the kernel's SA / SB template types
    float *D;
    int M, N, K;
    int epilogue;
    const float *bias, *R1, *R2;
    float *D2;
    int splitk, k_per_split, batch;
    long long strideA, strideB, strideD;
};

__device__ __forceinline__ bf16x8 pack8(const float (&v)[8]) {
    bf16x8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = (__bf16)v[j];
    return r;
}
__device__ __forceinline__ float widen(float v) { return v; }
__device__ __forceinline__ float widen(unsigned short v) { return __uint_as_float((unsigned)v << 16); }

// 8 consecutive source elements -> 8 floats (vector loads when aligned)
__device__ __forceinline__ void load8(const float *p, float (&v)[8]) {
    const float4 a = *reinterpret_cast<const float4 *>(p), b = *reinterpret_cast<const float4 *>(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void load8(const unsigned short *p, float (&v)[8]) {
    const uint4 a = *reinterpret_cast<const uint4 *>(p);
    v[0] = __uint_as_float(a.x << 16); v[1] = __uint_as_float(a.x & 0xffff0000u);
    v[2] = __uint_as_float(a.y << 16); v[3] = __uint_as_float(a.y & 0xffff0000u);
    v[4] = __uint_as_float(a.z << 16); v[5] = __uint_as_float(a.z & 0xffff0000u);
    v[6] = __uint_as_float(a.w << 16); v[7] = __uint_as_float(a.w & 0xffff0000u);
}
__device__ __forceinline__ void load2(const float *p, float &x0, float &x1) {
    const float2 t = *reinterpret_cast<const float2 *>(p);
    x0 = t.x; x1 = t.y;
}
__device__ __forceinline__ void load2(const unsigned short *p, float &x0, float &x1) {
    const unsigned t = *reinterpret_cast<const unsigned *>(p);
    x0 = __uint_as_float(t << 16); x1 = __uint_as_float(t & 0xffff0000u);
}
template <typename T> struct VecAlign;                 // element alignment needed by load8 / load2
template <> struct VecAlign<float> { static constexpr int K8 = 4, O2 = 2; };
template <> struct VecAlign<unsigned short> { static constexpr int K8 = 8, O2 = 2; };

// ---- k-contiguous operand: element (o, k) at base[o*ld + k] ------------------------------------------
template <int ROWS, int NT, typename T>
struct LoaderKC {
    static constexpr int UNITS = ROWS * KBLK;            // one unit = one cell (8 k of one row)
    static constexpr int NU = (UNITS + NT - 1) / NT;
    float v[NU][8];

    __device__ __forceinline__ static bool aligned_for(const T *base, int ld) {
        return (ld % VecAlign<T>::K8 == 0) && ((reinterpret_cast<uintptr_t>(base) & 15) == 0);
    }

    __device__ __forceinline__ void load(const T *__restrict__ base, int ld, int o0, int k0, int o_lim,
                                         int k_lim, bool aligned) {
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int f = threadIdx.x + u * NT;
            const int o = o0 + (f >> 2), k = k0 + ((f & 3) << 3);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[u][j] = 0.f;
            if (f < UNITS && o < o_lim && k < k_lim) {
                const T *p = base + (size_t)o * ld + k;
                if (aligned && k + 7 < k_lim) {
                    load8(p, v[u]);
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        if (k + j < k_lim) v[u][j] = widen(p[j]);
                }
            }
        }
    }

    __device__ __forceinline__ void store(bf16x8 *__restrict__ s) const {
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int f = threadIdx.x + u * NT;
            if (f < UNITS) s[(f & 3) * (ROWS + 2) + (f >> 2)] = pack8(v[u]);
        }
    }
};

// ---- outer-contiguous operand: element (o, k) at base[k*ld + o] --------------------------------------
template <int ROWS, int NT, typename T>
struct LoaderOC {
    static constexpr int UNITS = (ROWS / 2) * KBLK;      // one unit = 8 k-rows x 2 columns
    static constexpr int NU = (UNITS + NT - 1) / NT;
    float v[NU][2][8];

    __device__ __forceinline__ static bool aligned_for(const T *base, int ld) {
        return (ld % VecAlign<T>::O2 == 0) && ((reinterpret_cast<uintptr_t>(base) & (2 * sizeof(T) - 1)) == 0);
    }

    __device__ __forceinline__ void load(const T *__restrict__ base, int ld, int o0, int k0, int o_lim,
                                         int k_lim, bool aligned) {
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int f = threadIdx.x + u * NT;
            const int o = o0 + ((f % (ROWS / 2)) << 1), k = k0 + ((f / (ROWS / 2)) << 3);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float x0 = 0.f, x1 = 0.f;
                if (f < UNITS && k + j < k_lim && o < o_lim) {
                    const T *p = base + (size_t)(k + j) * ld + o;
                    if (aligned && o + 1 < o_lim) {
                        load2(p, x0, x1);
                    } else {
                        x0 = widen(p[0]);
                        if (o + 1 < o_lim) x1 = widen(p[1]);
                    }
                }
                v[u][0][j] = x0;
                v[u][1][j] = x1;
            }
        }
    }

    __device__ __forceinline__ void store(bf16x8 *__restrict__ s) const {
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int f = threadIdx.x + u * NT;
            if (f < UNITS) {
                const int cell = (f / (ROWS / 2)) * (ROWS + 2) + ((f % (ROWS / 2)) << 1);
                s[cell] = pack8(v[u][0]);
                s[cell + 1] = pack8(v[u][1]);
            }
        }
    }
};

template <int ROWS, int NT, bool KCONTIG, typename T>
struct LoaderSel { using type = LoaderKC<ROWS, NT, T>; };
template <int ROWS, int NT, typename T>
struct LoaderSel<ROWS, NT, false, T> { using type = LoaderOC<ROWS, NT, T>; };

template <int TM, int TN, int WM, int WN, bool TRANSA, bool TRANSB, typename SA, typename SB>
__global__ __launch_bounds__(WM *WN * 64) void gemm_bf16_kernel(GemmArgs g) {
    constexpr int NT = WM * WN * 64;
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
    constexpr int CA = KBLK * (BM + 2), CB = KBLK * (BN + 2);     // cells per stage
    __shared__ __attribute__((aligned(16))) bf16x8 smem[2 * (CA + CB)];
    // stage s: A cells at smem + s*(CA+CB), B cells right behind them
    auto stageA = [&](int st) { return smem + st * (CA + CB); };
    auto stageB = [&](int st) { return smem + st * (CA + CB) + CA; };

    using LA = typename LoaderSel<BM, NT, !TRANSA, SA>::type;
    using LB = typename LoaderSel<BN, NT, TRANSB, SB>::type;

    const int zb = blockIdx.z / g.splitk, zs = blockIdx.z - zb * g.splitk;
    const SA *__restrict__ A = static_cast<const SA *>(g.A) + (size_t)zb * g.strideA;
    const SB *__restrict__ B = static_cast<const SB *>(g.B) + (size_t)zb * g.strideB;
    float *__restrict__ D = g.D + (size_t)zb * g.strideD;
    const int M = g.M, N = g.N, K = g.K;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int k_begin = zs * g.k_per_split;
    const int k_end = min(K, k_begin + g.k_per_split);
    const int lda = TRANSA ? M : K, ldb = TRANSB ? K : N;
    const bool a_al = LA::aligned_for(A, lda), b_al = LB::aligned_for(B, ldb);

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

    LA la;
    LB lb;
    la.load(A, lda, m0, k_begin, M, k_end, a_al);
    lb.load(B, ldb, n0, k_begin, N, k_end, b_al);
    la.store(stageA(0));
    lb.store(stageB(0));
    __syncthreads();

    int cur = 0;
    for (int k0 = k_begin; k0 < k_end; k0 += BK) {
        const bool more = k0 + BK < k_end;
        if (more) {
            la.load(A, lda, m0, k0 + BK, M, k_end, a_al);
            lb.load(B, ldb, n0, k0 + BK, N, k_end, b_al);
        }
        const bf16x8 *as = stageA(cur) + lh * (BM + 2) + wm * (32 * TM) + li;
        const bf16x8 *bs = stageB(cur) + lh * (BN + 2) + wn * (32 * TN) + li;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = as[(2 * s) * (BM + 2) + 32 * i];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = bs[(2 * s) * (BN + 2) + 32 * j];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (more) {
            la.store(stageA(cur ^ 1));
            lb.store(stageB(cur ^ 1));
        }
        __syncthreads();
        cur ^= 1;
    }

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

template <int TM, int TN, int WM, int WN, bool TA, bool TB, typename SA, typename SB>
int launch(const GemmArgs &g, hipStream_t s) {
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
    dim3 grid((unsigned)sei_ceil_div(g.N, BN), (unsigned)sei_ceil_div(g.M, BM), (unsigned)(g.splitk * g.batch));
    hipLaunchKernelGGL((gemm_bf16_kernel<TM, TN, WM, WN, TA, TB, SA, SB>), grid, dim3(WM * WN * 64), 0, s, g);
    return sei_launch_status();
}

template <int TM, int TN, int WM, int WN, typename SA, typename SB>
int dispatch_layout(const GemmArgs &g, int ta, int tb, hipStream_t s) {
    if (!ta && tb) return launch<TM, TN, WM, WN, false, true, SA, SB>(g, s);
    if (!ta && !tb) return launch<TM, TN, WM, WN, false, false, SA, SB>(g, s);
    if (ta && !tb) return launch<TM, TN, WM, WN, true, false, SA, SB>(g, s);
    return launch<TM, TN, WM, WN, true, true, SA, SB>(g, s);
}

template <int TM, int TN, int WM, int WN>
int dispatch_types(const GemmArgs &g, int ta, int tb, int a16, int b16, hipStream_t s) {
    using H = unsigned short;
    if (a16 && b16) return dispatch_layout<TM, TN, WM, WN, H, H>(g, ta, tb, s);
    if (a16) return dispatch_layout<TM, TN, WM, WN, H, float>(g, ta, tb, s);
    if (b16) return dispatch_layout<TM, TN, WM, WN, float, H>(g, ta, tb, s);
    return dispatch_layout<TM, TN, WM, WN, float, float>(g, ta, tb, s);
}

inline size_t tiles(int M, int N, int bm, int bn) { return sei_ceil_div(M, bm) * sei_ceil_div(N, bn); }

}  // namespace

// a_is_bf16 / b_is_bf16: the operand is stored as bf16 (uint16_t) instead of float32.
extern "C" int sei_gemm_bf16_mixed(const void *A, int a_is_bf16, const void *B, int b_is_bf16, float *D, int M,
                                   int N, int K, int transA, int transB, int epilogue, const float *bias,
                                   const float *R1, const float *R2, float *D2, int batch, long long strideA,
                                   long long strideB, long long strideD, int allow_splitk, void *stream) {
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
        const size_t max_sk = (size_t)K / (8 * BK);
        if (sk > max_sk) sk = max_sk;
        if (sk > 1) {
            g.k_per_split = (int)(sei_ceil_div(sei_ceil_div(K, sk), BK) * BK);
            g.splitk = (int)sei_ceil_div(K, g.k_per_split);
        }
    }
    hipStream_t s = (hipStream_t)stream;
    if (tm == 2) return dispatch_types<2, 2, 2, 2>(g, transA, transB, a_is_bf16, b_is_bf16, s);
    if (tn == 2) return dispatch_types<1, 2, 2, 2>(g, transA, transB, a_is_bf16, b_is_bf16, s);
    return dispatch_types<1, 1, 2, 2>(g, transA, transB, a_is_bf16, b_is_bf16, s);
}

extern "C" int sei_gemm_bf16_ex(const float *A, const float *B, float *D, int M, int N, int K, int transA,
                                int transB, int epilogue, const float *bias, const float *R1,
                                const float *R2, float *D2, int batch, long long strideA, long long strideB,
                                long long strideD, int allow_splitk, void *stream) {
    return sei_gemm_bf16_mixed(A, 0, B, 0, D, M, N, K, transA, transB, epilogue, bias, R1, R2, D2, batch, strideA,
                               strideB, strideD, allow_splitk, stream);
}
