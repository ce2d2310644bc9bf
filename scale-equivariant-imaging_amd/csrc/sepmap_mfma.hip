// Ideal{Down,Up}sample on the matrix cores (bf16 throughput mode), gfx950.
// (reference: src/models/convolutional.py:54-92,113-133 -- the FFT "ideal" resamplers, which are exactly the real
//  separable rank-2 map  y[b,:,:,c] = L1 X R1^T + L2 X R2^T  (models/_mats.py))
//
// The f32 kernels (sepmap_*_packed_kernel, unet_kernels.hip) evaluate the two dense products on packed f32 FMAs and
// are bound by the FMA issue rate (1.36 G FMAs per fine-level call = 35 us at one wave-instruction per 4 cycles).
// Per image row the W product is a small GEMM  T_t[i] (Wo x C) = R_t (Wo x Wi) . X_i (Wi x C)  and per output column
// the H product is  Y[:, jo, :] (Ho x C) = [L1 L2] (Ho x 2 Hi) . [T1; T2][:, jo, :] (2 Hi x C): batches of GEMMs with
// K = 12 ... 48 and N = the channel count, i.e. v_mfma_f32_16x16x32_bf16 work with the CONSTANT matrix as the A operand
// and 16 channels on the lanes' columns.
//
// One workgroup (12 waves, one per CU: T fills LDS) walks items = one image x 16 channels:
//   pass W: a wave takes image rows i = wave, wave + 12, ...; its B fragments (8 consecutive input columns of one
//           channel per lane) come straight from global memory (buffer loads, f32 -> bf16 in registers: every element
//           of x is loaded once), the A fragments (rows of R) from LDS; the result tile has the output column on the
//           accumulator rows and the channel on the lane, and goes to LDS as bf16 in the layout the next product
//           reads K-contiguous:  T[t][jo][c][i];
//   pass H: a wave owns ONE 16-row tile of Ho and every n-th output column jo: A = T[t][jo][c][i..] (one ds_read_b128
//           per fragment), B = rows of [L1 L2], held in registers for the whole launch; f32 (or bf16) results straight
//           to y (64-byte segments: 16 channels).
// Round 5 (tools/exp_sepmap_mfma.py; 96 x 24 x 24 x 128 -> 48 and its transpose, us per launch: 52.0 / 66.8 before):
//   * items in XCD order (each XCD walks a contiguous eighth: the two 64-byte halves of a line meet in one L2)  49.3 / 59.7
//   * the next item's first rows are loaded before pass H (which only reads LDS and stores y), as buffer loads off
//     16 constant lane offsets (the 64-bit addresses took 96 VGPRs)                                           46.7 / 47.0
//   * 16, then 12 waves per workgroup with pass H's constant operand hoisted into registers                  38.9 / 38.1
//   * barriers that settle LDS only (s_waitcnt lgkmcnt(0); s_barrier) and the reduction depths as template
//     arguments (the compiler's s_waitcnt vmcnt(0) at every join of the ks branches is gone)                 35.7 / 35.6
//   = 4.0 TB/s of x + y. With every HBM access taken out the launch still took 27 us: what is left is the
//   latency of the short dependent chains (LDS read -> 2-4 MFMAs -> LDS write / store) at 3 waves per SIMD.
// The matrices are split into a bf16 head and a bf16 remainder (two MFMAs per product): the operator itself stays
// exact to ~2^-17 and only the ACTIVATIONS are rounded to bf16 (x before the W product, T between the products) -- the
// rounding every GEMM operand of this mode already has. Matrix-core time is negligible (a few hundred MFMAs per
// workgroup); the kernel streams x once and y once.
// Eligible: 12 <= Hi, Wi <= 64 with the T image of one workgroup inside LDS, C % 16 == 0 (12 -> 24 at 512 channels, 64
// images: 47.8 us on the f32 kernels, 30.6 here since round 5); everything else stays on sei_sepmap2_small (extents <= 8),
// sei_sepmap2_big (the x4 network's fine levels) or the f32 kernels.
#include "sei_common.h"

namespace {

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int SMM_WAVES = 12, SMM_THREADS = 64 * SMM_WAVES;
constexpr int SMM_NC = 16;                        // channels per workgroup = the MFMA's N
constexpr int SMM_PADK = 8;                       // bf16 elements of padding per K-contiguous LDS row (bank spread)
constexpr int SMM_LDS = 152 * 1024;

__device__ __forceinline__ unsigned short smm_f2bf(float v) {
    const __bf16 b = (__bf16)v;
    return __builtin_bit_cast(unsigned short, b);
}
__device__ __forceinline__ float smm_bf2f(unsigned short u) {
    return __builtin_bit_cast(float, (unsigned)u << 16);
}

struct SmmGeom {
    int B, Hi, Wi, Ho, Wo, C;
    int HiP, WiP;          // K extents padded to 32
    int HoT, WoT;          // 16-row tiles of the outputs
    int ldR, ldL, ldT;     // LDS row strides in elements (K extent + SMM_PADK)
    int offRlo, offL, offLlo, offT;   // element offsets of the LDS sections (R head at 0)
    int out16;             // y is bf16 (sei_sepmap2_bf16_out16): rounded once from the float32 accumulator
};

// The four matrices in the kernel's LDS image (sei_sepmap2_bf16_pack, once per map): bf16 head + remainder, zero-padded
// to 16-row tiles x (K padded to 32, + SMM_PADK): [R head | R remainder | L head | L remainder], each [t][row][k].
__global__ __launch_bounds__(256) void sepmap_pack_kernel(const float *__restrict__ L1, const float *__restrict__ R1,
                                                          const float *__restrict__ L2, const float *__restrict__ R2,
                                                          unsigned short *__restrict__ out, SmmGeom g) {
    const int nR = 2 * g.WoT * 16 * g.ldR, nL = 2 * g.HoT * 16 * g.ldL;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < nR + nL; e += gridDim.x * 256) {
        float v = 0.f;
        int hi_at, lo_at;
        if (e < nR) {
            const int k = e % g.ldR, row = (e / g.ldR) % (g.WoT * 16), t = e / (g.ldR * g.WoT * 16);
            if (row < g.Wo && k < g.Wi) v = (t ? R2 : R1)[(size_t)row * g.Wi + k];
            hi_at = e;
            lo_at = g.offRlo + e;
        } else {
            const int f = e - nR;
            const int k = f % g.ldL, row = (f / g.ldL) % (g.HoT * 16), t = f / (g.ldL * g.HoT * 16);
            if (row < g.Ho && k < g.Hi) v = (t ? L2 : L1)[(size_t)row * g.Hi + k];
            hi_at = g.offL + f;
            lo_at = g.offLlo + f;
        }
        const unsigned short h = smm_f2bf(v);
        out[hi_at] = h;
        out[lo_at] = smm_f2bf(v - smm_bf2f(h));
    }
}

template <int ksW, int ksH>
__global__ __launch_bounds__(SMM_THREADS) void sepmap_mfma_kernel(const float *__restrict__ x, float *__restrict__ y,
                                                                   const unsigned short *__restrict__ mats, SmmGeom g,
                                                                   int items) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[SMM_LDS / 2];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lc = lane & 15, lg = lane >> 4;                  // MFMA column / row-in-tile, and the k group
    const int ctiles = g.C / SMM_NC;

    // ---- once per workgroup: the packed matrices -> LDS (16-byte copies); T's K padding (i >= Hi) zeroed -- it meets
    //      zero matrix columns, but uninitialised LDS may hold NaN patterns; pass W never writes there
    unsigned short *Rh = lds, *Rl = lds + g.offRlo, *Lh = lds + g.offL, *Ll = lds + g.offLlo, *T = lds + g.offT;
    {
        const int n16 = g.offT / 8;                             // the matrix sections, in 16-byte pieces
        for (int e = threadIdx.x; e < n16; e += SMM_THREADS)
            reinterpret_cast<uint4 *>(lds)[e] = reinterpret_cast<const uint4 *>(mats)[e];
        const int first = g.Hi / 8 * 8;                         // (columns first .. Hi-1 are rewritten by every pass W)
        const int pad8 = (g.ldT - first) / 8;                   // 16-byte pieces from there to the end of a T row
        const int rows = 2 * g.Wo * SMM_NC;
        for (int e = threadIdx.x; e < rows * pad8; e += SMM_THREADS) {
            const int k = e % pad8, row = e / pad8;
            *reinterpret_cast<uint4 *>(T + (size_t)row * g.ldT + first + 8 * k) = make_uint4(0u, 0u, 0u, 0u);
        }
    }
    __syncthreads();

    // Between the passes only LDS has to be settled. __syncthreads() is a workgroup fence first: s_waitcnt vmcnt(0) in
    // front of the s_barrier, i.e. every wave sat out the write latency of its y stores and the whole prefetch of the next
    // image at each barrier (stamped: pass H of the 48 -> 24 map 16 k cycles for two iterations of arithmetic).
    auto lds_barrier = [] { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    // Pass H's share of a wave: ONE 16-row tile of the output rows (h_ht) and every h_nk-th output column from h_k on.
    // Its operand from the constant matrices -- rows of [L1 L2], head and remainder -- does not depend on the column or
    // the item: it is read from LDS once, here, and stays in registers (8 - 16 fragments). Re-read per (column, tile)
    // pair it was two thirds of the pass's LDS traffic (6 KB of ds_read_b128 per pair against 2 KB for T's fragment:
    // 860 KB per item on the 24 -> 48 map), and with the loads of x hidden behind pass H (below) LDS was what the
    // kernel waited for: 30.7 of its 40.3 us remained with every load and store of HBM taken out.
    const int h_ht = wave % g.HoT;                             // (HoT <= 8 < SMM_WAVES: every tile has its waves)
    const int h_k = wave / g.HoT, h_nk = (SMM_WAVES - wave % g.HoT + g.HoT - 1) / g.HoT;
    constexpr bool HOIST_LO = ksW + ksH < 4;                   // (48 - 64 pixel inputs both ways: the remainders would spill)
    bf16x8 Lf[2][2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int lrow = (t * g.HoT * 16 + h_ht * 16 + lc) * g.ldL + lg * 8 + ks * 32;
            if (ks < ksH) {
                Lf[t][ks][0] = *reinterpret_cast<const bf16x8 *>(Lh + lrow);
                Lf[t][ks][1] = HOIST_LO ? *reinterpret_cast<const bf16x8 *>(Ll + lrow) : bf16x8{};
            } else {
                Lf[t][ks][0] = Lf[t][ks][1] = bf16x8{};
            }
        }
    // Item order: a workgroup's 16 channels are 64 bytes of every pixel -- HALF a 128-byte line, whose other half belongs
    // to the next channel tile. Workgroups go to the 8 XCDs round-robin (blockIdx % 8), each with its own L2: with the
    // plain order item = blockIdx the two halves were fetched by two XCDs and every line of x crossed the fabric twice
    // (profiles/r05_b: 3.1 bytes fetched per byte written, where the maps of a step read as much as they write). Each XCD
    // walks its own contiguous eighth of the items instead, so neighbouring channel tiles run side by side on one L2.
    const int chunk = (items + 7) / 8, its = 8 * chunk;
    auto item_at = [&](int it) { return (it & 7) * chunk + (it >> 3); };
    auto advance = [&](int it) {                                // the next position of this workgroup that holds an item
        while (it < its && item_at(it) >= items) it += gridDim.x;
        return it;
    };
    // A lane's 8 (16) input columns of row i, one channel: 4-byte BUFFER loads -- the item's slice of x as the resource,
    // 16 lane offsets that never change (VGPRs, set once), the row offset in an SGPR. As 48 separately computed 64-bit
    // addresses the three rows in flight took 96 address registers (239 VGPRs, spills once the prefetch below was added),
    // and the address temporaries aliased load destinations still in flight (an s_waitcnt vmcnt in front of the address
    // arithmetic). Columns j >= Wi get an offset beyond the resource's num_records (an image is < 2^31 bytes, smm_plan):
    // the buffer load returns 0 for them, as a masked load would -- no Inf * 0 = NaN against the zero padding of R's rows.
    unsigned offs[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int j = (e >> 3) * 32 + lg * 8 + (e & 7);
        offs[e] = j < g.Wi ? (unsigned)((j * g.C + lc) * 4) : 0x7ffffff0u;
    }
    const unsigned row_bytes = (unsigned)g.Wi * g.C * 4, slice_bytes = (unsigned)g.Hi * row_bytes;
    typedef __amdgpu_buffer_rsrc_t rsrc_t;
    auto load_row = [&](rsrc_t xr, int i, float (&v)[16]) {
        const unsigned rb = (unsigned)i * row_bytes;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, offs[e], rb, 0));
        if constexpr (ksW > 1) {
#pragma unroll
            for (int e = 8; e < 16; ++e)
                v[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, offs[e], rb, 0));
        } else {
#pragma unroll
            for (int e = 8; e < 16; ++e) v[e] = 0.f;
        }
    };
    auto x_of = [&](int item) {                                 // (wave-uniform) the item's image, from its first channel on
        const int b = item / ctiles, c0 = (item - b * ctiles) * SMM_NC;
        float *base = const_cast<float *>(x) + (size_t)b * g.Hi * g.Wi * g.C + c0;
        return __builtin_amdgcn_make_buffer_rsrc(base, 0, slice_bytes - (unsigned)c0 * 4, 0x00020000);
    };
    // Two rows in flight per wave (rows wave, wave + SMM_WAVES of the item; row + 2 SMM_WAVES follows as a buffer comes free):
    // loaded for the FIRST item here, for every
    // later one right before the previous item's pass H -- that pass only reads LDS and stores y, so the next image's
    // loads run under it (with 24-row images: all of them).
    float va[16], vb[16];
    int it = advance(blockIdx.x);
    if (it < its) {
        const rsrc_t xb = x_of(item_at(it));
        if (wave < g.Hi) load_row(xb, wave, va);
        if (wave + SMM_WAVES < g.Hi) load_row(xb, wave + SMM_WAVES, vb);
    }
    while (it < its) {
    const int item = item_at(it);
    const int b = item / ctiles, c0 = (item - b * ctiles) * SMM_NC;
    // ---- pass W: T[t][jo][c][i] = sum_j R_t[jo][j] x[b][i][j][c0 + c] --------------------------------------------
    const rsrc_t xb = x_of(item);
    auto w_row = [&](int i, const float (&v)[16]) {
        bf16x8 bx[2];
#pragma unroll
        for (int e = 0; e < 16; ++e) bx[e >> 3][e & 7] = (__bf16)v[e];
        for (int t = 0; t < 2; ++t) {
            for (int jt = 0; jt < g.WoT; ++jt) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                const int arow = (t * g.WoT * 16 + jt * 16 + lc) * g.ldR + lg * 8;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    if (ks < ksW) {
                        const bf16x8 ah = *reinterpret_cast<const bf16x8 *>(Rh + arow + ks * 32);
                        const bf16x8 al = *reinterpret_cast<const bf16x8 *>(Rl + arow + ks * 32);
                        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bx[ks], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bx[ks], acc, 0, 0, 0);
                    }
                }
                // accumulator: row (= jo within the tile) 4 * lg + r, column (= channel) lc
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int jo = jt * 16 + 4 * lg + r;
                    if (jo < g.Wo) T[((size_t)(t * g.Wo + jo) * SMM_NC + lc) * g.ldT + i] = smm_f2bf(acc[r]);
                }
            }
        }
    };
    {   // the first two rows are in va / vb; rows i + 32 follow as their buffers come free
        int i = wave;
        while (i < g.Hi) {
            w_row(i, va);
            if (i + 2 * SMM_WAVES < g.Hi) load_row(xb, i + 2 * SMM_WAVES, va);
            i += SMM_WAVES;
            if (i >= g.Hi) break;
            w_row(i, vb);
            if (i + 2 * SMM_WAVES < g.Hi) load_row(xb, i + 2 * SMM_WAVES, vb);
            i += SMM_WAVES;
        }
    }
    lds_barrier();
    it = advance(it + gridDim.x);
    if (it < its) {                                             // the next item's first rows: in flight during pass H
        const rsrc_t xn = x_of(item_at(it));
        if (wave < g.Hi) load_row(xn, wave, va);
        if (wave + SMM_WAVES < g.Hi) load_row(xn, wave + SMM_WAVES, vb);
    }

    // ---- pass H: y[b][io][jo][c0 + c] = sum_t sum_i L_t[io][i] T[t][jo][c][i] --------------------------------------
    // Operand roles swapped against pass W: A = T[t][jo][c][i..] (row = channel), B = L_t[io][i..] (column = io), so the
    // accumulator holds FOUR CONSECUTIVE CHANNELS of one output pixel per lane: one 16-byte store instead of four
    // 4-byte ones. Two (jo, io-tile) items per iteration: independent chains, so the LDS reads and MFMAs of one cover
    // the latencies of the other.
    const size_t ybase = (size_t)b * g.Ho * g.Wo * g.C + c0 + 4 * lg;
    float *yb = y + ybase;
    unsigned short *yb16 = reinterpret_cast<unsigned short *>(y) + ybase;
    auto h_col = [&](int jo, f32x4 &acc) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const size_t trow = ((size_t)(t * g.Wo + jo) * SMM_NC + lc) * g.ldT + lg * 8;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                if (ks < ksH) {
                    const bf16x8 at = *reinterpret_cast<const bf16x8 *>(T + trow + ks * 32);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(at, Lf[t][ks][0], acc, 0, 0, 0);
                    const bf16x8 lo = HOIST_LO ? Lf[t][ks][1]
                                               : *reinterpret_cast<const bf16x8 *>(
                                                     Ll + (t * g.HoT * 16 + h_ht * 16 + lc) * g.ldL + lg * 8 + ks * 32);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(at, lo, acc, 0, 0, 0);
                }
            }
        }
    };
    auto h_store = [&](int jo, const f32x4 &acc) {
        const int io = h_ht * 16 + lc;                            // accumulator: column = io, rows = channels 4 lg .. 4 lg + 3
        if (io < g.Ho) {
            if (g.out16) {
                ushort4 h;
                h.x = smm_f2bf(acc[0]); h.y = smm_f2bf(acc[1]); h.z = smm_f2bf(acc[2]); h.w = smm_f2bf(acc[3]);
                *reinterpret_cast<ushort4 *>(yb16 + ((size_t)io * g.Wo + jo) * g.C) = h;
            } else {
                *reinterpret_cast<float4 *>(yb + ((size_t)io * g.Wo + jo) * g.C) = make_float4(acc[0], acc[1], acc[2], acc[3]);
            }
        }
    };
    {
        for (int jo = h_k; jo < g.Wo; jo += 2 * h_nk) {
            f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
            const int jo1 = jo + h_nk;
            h_col(jo, a0);
            if (jo1 < g.Wo) h_col(jo1, a1);
            h_store(jo, a0);
            if (jo1 < g.Wo) h_store(jo1, a1);
        }
    }
    lds_barrier();                                              // T is rewritten by the next item's pass W
    }
}

inline bool smm_plan(int B, int Hi, int Wi, int Ho, int Wo, int C, SmmGeom &g) {
    if (B <= 0 || C <= 0 || C % SMM_NC != 0) return false;
    // (extents <= 8: sei_sepmap2_small)
    if (Hi < 12 || Wi < 12 || Hi > 64 || Wi > 64 || Ho < 1 || Wo < 1 || Ho > 128 || Wo > 128) return false;
    g.B = B; g.Hi = Hi; g.Wi = Wi; g.Ho = Ho; g.Wo = Wo; g.C = C; g.out16 = 0;
    g.HiP = (Hi + 31) / 32 * 32;
    g.WiP = (Wi + 31) / 32 * 32;
    if (g.HiP > 64 || g.WiP > 64) return false;
    g.HoT = (Ho + 15) / 16;
    g.WoT = (Wo + 15) / 16;
    g.ldR = g.WiP + SMM_PADK;
    g.ldL = g.HiP + SMM_PADK;
    g.ldT = g.HiP + SMM_PADK;
    const size_t nR = (size_t)2 * g.WoT * 16 * g.ldR, nL = (size_t)2 * g.HoT * 16 * g.ldL;
    const size_t nT = (size_t)2 * Wo * SMM_NC * g.ldT;
    g.offRlo = (int)nR;
    g.offL = (int)(2 * nR);
    g.offLlo = (int)(2 * nR + nL);
    g.offT = (int)(2 * nR + 2 * nL);
    if ((size_t)B * (C / SMM_NC) >= ((size_t)1 << 31)) return false;
    if ((size_t)Hi * Wi * C * 4 >= ((size_t)1 << 31)) return false;   // an image is one buffer resource (32-bit offsets)
    return (2 * nR + 2 * nL + nT) * 2 <= (size_t)SMM_LDS;
}

}  // namespace

// 1 when sei_sepmap2_bf16 serves this shape, else 0 (the caller then takes sei_sepmap2_packed).
extern "C" size_t sei_sepmap2_bf16_eligible(int B, int Hi, int Wi, int Ho, int Wo, int C) {
    SmmGeom g;
    return smm_plan(B, Hi, Wi, Ho, Wo, C, g) ? 1 : 0;
}

// Elements (uint16) of the packed matrix image of a map (0 when no batch / channel count makes the shape eligible).
extern "C" size_t sei_sepmap2_bf16_pack_elems(int Hi, int Wi, int Ho, int Wo) {
    SmmGeom g;
    return smm_plan(1, Hi, Wi, Ho, Wo, SMM_NC, g) ? (size_t)g.offT : 0;
}

// L1, L2: (Ho, Hi), R1, R2: (Wo, Wi) float32 row-major -> the kernel's bf16 head + remainder image (once per map).
extern "C" int sei_sepmap2_bf16_pack(const float *L1, const float *R1, const float *L2, const float *R2, uint16_t *packed,
                                     int Hi, int Wi, int Ho, int Wo, void *stream) {
    SEI_REQUIRE(L1 && R1 && L2 && R2 && packed);
    SmmGeom g;
    if (!smm_plan(1, Hi, Wi, Ho, Wo, SMM_NC, g)) return SEI_ERR_BAD_ARG;
    hipLaunchKernelGGL(sepmap_pack_kernel, dim3(32), dim3(256), 0, (hipStream_t)stream, L1, R1, L2, R2, packed, g);
    return sei_launch_status();
}

// y[b,:,:,c] = L1 X R1^T + L2 X R2^T with bf16-rounded activations on the matrix cores (bf16 throughput mode); `packed`
// from sei_sepmap2_bf16_pack for the same (Hi, Wi, Ho, Wo), 16-byte aligned. SEI_ERR_BAD_ARG when not eligible.
static int smm_launch(const float *x, float *y, int out16, int B, int Hi, int Wi, int Ho, int Wo, int C,
                      const uint16_t *packed, void *stream) {
    SEI_REQUIRE(x && y && packed && x != y && (reinterpret_cast<uintptr_t>(packed) & 15) == 0);
    SEI_REQUIRE((reinterpret_cast<uintptr_t>(y) & 15) == 0);
    SmmGeom g;
    if (!smm_plan(B, Hi, Wi, Ho, Wo, C, g)) return SEI_ERR_BAD_ARG;
    g.out16 = out16;
    const int items = B * (C / SMM_NC);
    const int grid = items < 256 ? items : 256;                 // one resident workgroup per CU walks its items
    auto go = [&](auto kernel) {
        hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(SMM_THREADS), 0, (hipStream_t)stream, x, y, packed, g, items);
    };
    const int ksW = g.WiP / 32, ksH = g.HiP / 32;              // 32-wide reduction steps of the two passes: 1 or 2 each
    if (ksW == 1 && ksH == 1) go(sepmap_mfma_kernel<1, 1>);
    else if (ksW == 1) go(sepmap_mfma_kernel<1, 2>);
    else if (ksH == 1) go(sepmap_mfma_kernel<2, 1>);
    else go(sepmap_mfma_kernel<2, 2>);
    return sei_launch_status();
}

extern "C" int sei_sepmap2_bf16(const float *x, float *y, int B, int Hi, int Wi, int Ho, int Wo, int C,
                                const uint16_t *packed, void *stream) {
    return smm_launch(x, y, 0, B, Hi, Wi, Ho, Wo, C, packed, stream);
}

// The same map with a bf16 result (the float32 accumulator rounded once, to nearest even): what a 1x1 convolution behind
// the resampler reads as its GEMM operand -- no float32 copy of the resampled tensor, no cast pass.
extern "C" int sei_sepmap2_bf16_out16(const float *x, uint16_t *y16, int B, int Hi, int Wi, int Ho, int Wo, int C,
                                      const uint16_t *packed, void *stream) {
    return smm_launch(x, reinterpret_cast<float *>(y16), 1, B, Hi, Wi, Ho, Wo, C, packed, stream);
}
