// Ideal{Down,Up}sample on the matrix cores at LARGE extents (96 ... 256 pixels: the x4 network's fine levels, the
// un-cropped 256 x 256 series), bf16 throughput mode, gfx950.
// (reference: src/models/convolutional.py:54-92,113-133 -- the FFT "ideal" resamplers = the real separable rank-2 map
//  y[b,:,:,c] = L1 X R1^T + L2 X R2^T, models/_mats.py)
//
// csrc/sepmap_mfma.hip keeps one image x 16 channels of the intermediate T in LDS, which stops at 64-pixel extents; the
// f32 FMA kernels that served everything larger are bound by the FMA issue rate (23 ms of the x4 network's 155-ms step at
// 0.12 of the HBM peak). Both products of the map are the SAME batched GEMM with a tiny constant matrix:
//
//     out[r] (M x N) = A (M x K) . X[r] (K x N),     X[r] a contiguous row-major block (N contiguous), r = 0 .. batch - 1
//
//   pass W   r = (b, i): X[r] = x[b, i, :, :] (Wi x C, float32), A = [R1; R2] (2 Wo x Wi)   -> T[b, i, t, jo, c] in bf16
//   pass H   r = b:      X[r] = T[b] viewed as (2 Hi x Wo C): row 2 i + t,                  A[io][2 i + t] = L_t[io][i] (Ho x 2 Hi)
//                                                                                           -> y[b, io, jo, c] in float32
//
// so one kernel does both: A lives in REGISTERS (each wave owns up to three 16-row tiles of A for every k: <= 36 fragments,
// bf16 head + bf16 remainder so that the operator itself stays exact to ~2^-17 and only the activations are rounded, as in
// sepmap_mfma.hip), a persistent workgroup walks (r, 64-column tile) items, stages the item's (K x 64) block of X in LDS as
// [64 k][64 n] bf16 images (float32 is converted on the way), reads it back with the transposing LDS read as the MFMA's A
// operand -- so a lane's accumulator holds FOUR CONSECUTIVE n of one output row: 8- / 16-byte stores -- and prefetches the
// next item's block into registers under the MFMAs. x is read once, T written and read once, y written once.
#include "sei_common.h"

namespace {

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x4 = __attribute__((ext_vector_type(4))) __bf16;
using f32x4 = __attribute__((ext_vector_type(4))) float;
typedef short v4s __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int sb_swz(int row) { return (((row >> 1) & 1) << 1) | (((row >> 3) & 1) << 2); }

struct CmatArgs {
    const void *X;              // (batch, K, N) float32 (IN32) or bf16
    void *out;                  // (batch, M, N) bf16 (OUT16) or float32
    const unsigned short *Ah, *Al;   // (Mp, KP) bf16 head / remainder, rows padded to whole 16-row tiles, K to whole 32s
    int M, K, N, KP;
    int batch;
    int mgroups;                // M is cut into groups of <= waves * MTW tiles; an item = (r, n-tile, m-group)
    int ntiles;                 // 64-column tiles of N (the last one may be 16 / 32 / 48 wide)
};

// KS = KP / 32 k-steps, MTW = 16-row tiles of A per wave; the workgroup has blockDim.x / 64 waves.
template <int KS, int MTW, bool IN32, bool OUT16>
__global__ __launch_bounds__(512) void cmat_gemm_kernel(CmatArgs g) {
    constexpr int NIMG = (KS + 1) / 2;                              // [64 k][64 n] images per buffer
    constexpr int BUF = NIMG * 8192;
    __shared__ __attribute__((aligned(1024))) char smem[2 * BUF];
    const int tid = threadIdx.x, lane = tid & 63, l16 = lane & 15, lg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), nwaves = blockDim.x >> 6, nthreads = blockDim.x;
    const int mtiles = (g.M + 15) >> 4, per_group = nwaves * MTW;

    // zero both buffers once: rows K .. KP - 1 meet zero columns of A but must not hold NaN patterns, and a narrow last
    // n-tile leaves columns unwritten
    for (int e = tid; e < 2 * BUF / 16; e += nthreads) reinterpret_cast<uint4 *>(smem)[e] = make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();

    const int items = g.batch * g.ntiles * g.mgroups;
    const size_t esz_in = IN32 ? 4 : 2;
    // ---- staging: thread t owns 16-byte pieces p = t, t + nthreads, ... of the (K x 64) block: row k = p / 8 (16 float32
    //      per... ) -- IN32: a piece is 4 float32 (16 B in, 8 B out): 16 pieces per row; bf16: 8 elements: 8 pieces per row
    constexpr int PPR = IN32 ? 16 : 8;                              // pieces per 64-column row
    constexpr int MAXP = (KS * 32 * PPR + 383) / 384;               // pieces per thread at the smallest workgroup (6 waves)
    uint4 pre[MAXP];
    auto item_of = [&](int it, int &r, int &nt, int &mg) {
        mg = it % g.mgroups;
        const int q = it / g.mgroups;
        nt = q % g.ntiles;
        r = q / g.ntiles;
    };
    auto prefetch = [&](int it) {
        int r, nt, mg;
        item_of(it, r, nt, mg);
        const int ncols = min(64, g.N - 64 * nt);
        const char *base = reinterpret_cast<const char *>(g.X) + ((size_t)r * g.K * g.N + (size_t)64 * nt) * esz_in;
#pragma unroll
        for (int e = 0; e < MAXP; ++e) {
            const int p = tid + e * nthreads;
            const int k = p / PPR, c = p % PPR;                     // row, piece within the row
            const int col = c * (IN32 ? 4 : 8);
            if (k < g.K && col < ncols)
                pre[e] = *reinterpret_cast<const uint4 *>(base + ((size_t)k * g.N + col) * esz_in);
            else
                pre[e] = make_uint4(0u, 0u, 0u, 0u);
        }
    };
    auto commit = [&](char *buf) {                                  // registers -> the swizzled bf16 images
#pragma unroll
        for (int e = 0; e < MAXP; ++e) {
            const int p = tid + e * nthreads;
            const int k = p / PPR, c = p % PPR;
            if (k >= KS * 32) continue;
            char *row = buf + (k >> 6) * 8192 + (k & 63) * 128;
            if constexpr (IN32) {
                const int ch = (c >> 1) ^ sb_swz(k & 63);           // 16-byte chunk = 8 bf16 = two pieces
                const f32x4 v = __builtin_bit_cast(f32x4, pre[e]);
                bf16x4 o;
#pragma unroll
                for (int q = 0; q < 4; ++q) o[q] = (__bf16)v[q];
                *reinterpret_cast<bf16x4 *>(row + ch * 16 + (c & 1) * 8) = o;
            } else {
                const int ch = c ^ sb_swz(k & 63);
                *reinterpret_cast<uint4 *>(row + ch * 16) = pre[e];
            }
        }
    };
    // ---- fragments of X^T: 8 k (32 ks + 8 lg + j) of column l16 of 16-column block blk
    const int tq = l16 >> 2, tp = l16 & 3;
    const int rm_lane = 128 * (8 * lg + tq) + 16 * ((tp >> 1) ^ sb_swz(8 * lg + tq)) + 8 * (tp & 1);
    auto frag = [&](const char *img, int blk, int ks) -> bf16x8 {
        const int base = rm_lane ^ (32 * blk);
        const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) v4s *)(img + base + 128 * 32 * ks));
        const v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) v4s *)(img + base + 128 * (32 * ks + 4)));
        return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    };

    // ---- A fragments of this wave, per m-group (reloaded when the group changes: L2 hits, a few KB)
    bf16x8 ah[MTW][KS], al[MTW][KS];
    int loaded_group = -1;
    auto load_a = [&](int mg) {
#pragma unroll
        for (int t = 0; t < MTW; ++t) {
            const int mt = min(mg * per_group + wave * MTW + t, mtiles - 1);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const size_t o = (size_t)(16 * mt + l16) * g.KP + 32 * ks + 8 * lg;
                ah[t][ks] = *reinterpret_cast<const bf16x8 *>(g.Ah + o);
                al[t][ks] = *reinterpret_cast<const bf16x8 *>(g.Al + o);
            }
        }
        loaded_group = mg;
    };

    int it = blockIdx.x;
    if (it < items) prefetch(it);
    int cur = 0;
    for (; it < items; it += gridDim.x) {
        int r, nt, mg;
        item_of(it, r, nt, mg);
        char *buf = smem + cur * BUF;
        commit(buf);
        if (mg != loaded_group) load_a(mg);
        __syncthreads();                                            // the block is in LDS (and the A fragments have landed)
        if (it + (int)gridDim.x < items) prefetch(it + gridDim.x);  // next item's loads fly under the MFMAs
        const int ncols = min(64, g.N - 64 * nt), nblk = (ncols + 15) >> 4;
        f32x4 acc[MTW][4];
#pragma unroll
        for (int t = 0; t < MTW; ++t)
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) acc[t][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            bf16x8 fx[4];
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) fx[nb] = frag(buf + (ks >> 1) * 8192, nb, ks & 1);
#pragma unroll
            for (int t = 0; t < MTW; ++t)
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) {
                    if (nb < nblk) {                                // uniform
                        acc[t][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fx[nb], ah[t][ks], acc[t][nb], 0, 0, 0);
                        acc[t][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fx[nb], al[t][ks], acc[t][nb], 0, 0, 0);
                    }
                }
        }
        // accumulator: column (lane l16) = row m of the output, registers = n = 16 nb + 4 lg .. + 3
#pragma unroll
        for (int t = 0; t < MTW; ++t) {
            const int mt = mg * per_group + wave * MTW + t;
            const int m = 16 * mt + l16;
            if (mt >= mtiles || m >= g.M) continue;
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                if (nb >= nblk) continue;
                const size_t o = ((size_t)r * g.M + m) * g.N + 64 * nt + 16 * nb + 4 * lg;
                if constexpr (OUT16) {
                    bf16x4 v;
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = (__bf16)acc[t][nb][q];
                    *reinterpret_cast<bf16x4 *>(reinterpret_cast<unsigned short *>(g.out) + o) = v;
                } else {
                    *reinterpret_cast<f32x4 *>(reinterpret_cast<float *>(g.out) + o) = acc[t][nb];
                }
            }
        }
        cur ^= 1;                                                   // the next commit writes the other buffer: readers of
    }                                                               // this one are past the barrier of the next iteration
}

// bf16 head + remainder of the two pass matrices, zero-padded: AW = [R1; R2] (2 Wo x Wi), AH[io][2 i + t] = L_t[io][i].
__global__ __launch_bounds__(256) void sepmap_big_pack_kernel(const float *__restrict__ L1, const float *__restrict__ R1,
                                                              const float *__restrict__ L2, const float *__restrict__ R2,
                                                              unsigned short *__restrict__ out, int Hi, int Wi, int Ho,
                                                              int Wo, int MWp, int KWp, int MHp, int KHp) {
    const size_t nW = (size_t)MWp * KWp, nH = (size_t)MHp * KHp;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < nW + nH; e += (size_t)gridDim.x * 256) {
        float v = 0.f;
        size_t hi_at, lo_at;
        if (e < nW) {
            const int k = (int)(e % KWp), m = (int)(e / KWp);
            if (m < 2 * Wo && k < Wi) v = (m < Wo ? R1 : R2)[(size_t)(m < Wo ? m : m - Wo) * Wi + k];
            hi_at = e;
            lo_at = nW + e;
        } else {
            const size_t f = e - nW;
            const int k = (int)(f % KHp), m = (int)(f / KHp);
            if (m < Ho && k < 2 * Hi) v = ((k & 1) ? L2 : L1)[(size_t)m * Hi + (k >> 1)];
            hi_at = 2 * nW + f;
            lo_at = 2 * nW + nH + f;
        }
        const __bf16 h = (__bf16)v;
        const __bf16 l = (__bf16)(v - (float)h);
        out[hi_at] = __builtin_bit_cast(unsigned short, h);
        out[lo_at] = __builtin_bit_cast(unsigned short, l);
    }
}

struct BigPlan {
    int MWp, KWp, MHp, KHp;      // padded matrix extents of the two passes
    bool ok;
};
inline BigPlan big_plan(int Hi, int Wi, int Ho, int Wo, int C) {
    BigPlan p;
    p.MWp = (2 * Wo + 15) / 16 * 16;
    p.KWp = (Wi + 31) / 32 * 32;
    p.MHp = (Ho + 15) / 16 * 16;
    p.KHp = (2 * Hi + 31) / 32 * 32;
    // an extent beyond 64 on either side (csrc/sepmap_mfma.hip serves the rest; its 48 -> 96 calls measured no faster than
    // the f32 kernels: 455 us against 230 here), k-steps the kernel is instantiated for, 16-byte rows everywhere
    // (an extent OF 64 counts too: 64 -> 32 and 32 -> 64 overflow the one-workgroup kernel's LDS and went to the f32
    // kernels -- 1.3-3.8 ms per launch in the un-cropped 256-pixel series; the caller asks sei_sepmap2_bf16 first)
    p.ok = (Hi >= 64 || Wi >= 64 || Ho >= 64 || Wo >= 64) && Hi >= 32 && Wi >= 32 && Hi <= 256 && Wi <= 256 && Ho >= 16 && Wo >= 16 && Ho <= 512 && Wo <= 512 && C % 8 == 0 &&
           p.KWp / 32 <= 16 && p.KHp / 32 <= 16 && C % 16 == 0;
    return p;
}

template <bool IN32, bool OUT16>
int cmat_launch(CmatArgs &g, hipStream_t s) {
    const int KS = g.KP / 32;
    const int mtiles = (g.M + 15) / 16;
    // tiles of A per wave: the fewest that put an m-group on <= 8 waves while the fragments fit the registers
    // (KS * MTW <= 18 -> 36 fragments = 144 registers); M beyond 8 * MTW tiles is cut into m-groups (X re-read from L2)
    int MTW = 1;
    while (MTW < 3 && (mtiles + MTW - 1) / MTW > 8 && KS * (MTW + 1) <= 18) ++MTW;
    g.mgroups = (mtiles + 8 * MTW - 1) / (8 * MTW);
    const int tiles_per_group = (mtiles + g.mgroups - 1) / g.mgroups;
    int waves = (tiles_per_group + MTW - 1) / MTW;
    if (waves < 6) waves = 6;           // the staging registers are sized for >= 384 threads: idle waves where M is small
    g.ntiles = (g.N + 63) / 64;
    const long long items = (long long)g.batch * g.ntiles * g.mgroups;
    if (items <= 0 || items >= (1ll << 31)) return SEI_ERR_BAD_ARG;
    const unsigned grid = (unsigned)(items < 256 ? items : 256);
#define SB_CASE(KSV, MTV)                                                                                              \
    if (KS == KSV && MTW == MTV) {                                                                                     \
        hipLaunchKernelGGL((cmat_gemm_kernel<KSV, MTV, IN32, OUT16>), dim3(grid), dim3(64 * waves), 0, s, g);          \
        return sei_launch_status();                                                                                    \
    }
    SB_CASE(1, 1) SB_CASE(1, 2) SB_CASE(1, 3) SB_CASE(2, 1) SB_CASE(2, 2) SB_CASE(2, 3) SB_CASE(3, 1) SB_CASE(3, 2) SB_CASE(3, 3) SB_CASE(4, 1) SB_CASE(4, 2) SB_CASE(4, 3)
    SB_CASE(6, 1) SB_CASE(6, 2) SB_CASE(6, 3) SB_CASE(8, 1) SB_CASE(8, 2) SB_CASE(12, 1) SB_CASE(16, 1)
#undef SB_CASE
    return SEI_ERR_BAD_ARG;
}

}  // namespace

extern "C" size_t sei_sepmap2_big_eligible(int B, int Hi, int Wi, int Ho, int Wo, int C) {
    if (B <= 0) return 0;
    const BigPlan p = big_plan(Hi, Wi, Ho, Wo, C);
    if (!p.ok) return 0;
    const int ksw = p.KWp / 32, ksh = p.KHp / 32;
    auto built = [](int ks) { return ks == 1 || ks == 2 || ks == 3 || ks == 4 || ks == 6 || ks == 8 || ks == 12 || ks == 16; };
    return built(ksw) && built(ksh) ? 1 : 0;
}

// uint16 elements of the packed matrices of a map / of the bf16 intermediate T of a call
extern "C" size_t sei_sepmap2_big_pack_elems(int Hi, int Wi, int Ho, int Wo) {
    const BigPlan p = big_plan(Hi, Wi, Ho, Wo, 16);
    return p.ok ? 2 * ((size_t)p.MWp * p.KWp + (size_t)p.MHp * p.KHp) : 0;
}
extern "C" size_t sei_sepmap2_big_work_elems(int B, int Hi, int Wi, int Ho, int Wo, int C) {
    return (size_t)B * Hi * 2 * Wo * C;
}

extern "C" int sei_sepmap2_big_pack(const float *L1, const float *R1, const float *L2, const float *R2, uint16_t *packed,
                                    int Hi, int Wi, int Ho, int Wo, void *stream) {
    SEI_REQUIRE(L1 && R1 && L2 && R2 && packed);
    const BigPlan p = big_plan(Hi, Wi, Ho, Wo, 16);
    if (!p.ok) return SEI_ERR_BAD_ARG;
    hipLaunchKernelGGL(sepmap_big_pack_kernel, dim3(64), dim3(256), 0, (hipStream_t)stream, L1, R1, L2, R2, packed, Hi, Wi, Ho,
                       Wo, p.MWp, p.KWp, p.MHp, p.KHp);
    return sei_launch_status();
}

// y[b,:,:,c] = L1 X R1^T + L2 X R2^T with bf16-rounded activations on the matrix cores, any extent sei_sepmap2_big_eligible
// takes; `packed` from sei_sepmap2_big_pack, `work` >= sei_sepmap2_big_work_elems uint16 (the bf16 intermediate).
extern "C" int sei_sepmap2_big(const float *x, float *y, int B, int Hi, int Wi, int Ho, int Wo, int C, const uint16_t *packed,
                               uint16_t *work, void *stream) {
    SEI_REQUIRE(x && y && packed && work && x != y);
    SEI_REQUIRE((((uintptr_t)x | (uintptr_t)y | (uintptr_t)packed | (uintptr_t)work) & 15) == 0);
    if (!sei_sepmap2_big_eligible(B, Hi, Wi, Ho, Wo, C)) return SEI_ERR_BAD_ARG;
    const BigPlan p = big_plan(Hi, Wi, Ho, Wo, C);
    const size_t nW = (size_t)p.MWp * p.KWp, nH = (size_t)p.MHp * p.KHp;
    hipStream_t s = (hipStream_t)stream;
    CmatArgs w;
    w.X = x; w.out = work; w.Ah = packed; w.Al = packed + nW;
    w.M = 2 * Wo; w.K = Wi; w.N = C; w.KP = p.KWp; w.batch = B * Hi;
    int rc = cmat_launch<true, true>(w, s);
    if (rc != SEI_OK) return rc;
    CmatArgs h;
    h.X = work; h.out = y; h.Ah = packed + 2 * nW; h.Al = packed + 2 * nW + nH;
    h.M = Ho; h.K = 2 * Hi; h.N = Wo * C; h.KP = p.KHp; h.batch = B;
    return cmat_launch<false, false>(h, s);
}
