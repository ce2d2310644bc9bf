// sei_gemm_bf16nt / sei_gemm_bf16nt_dw2: the large 1x1-convolution GEMMs of the U-Net in throughput (bf16) mode.
//
//   D[M,N] = op(A) op(B)      A, B: bf16 in HBM, read as the model stores them; f32 accumulation
//   outputs: D32 (float) and/or D16 (bf16), fused epilogues as sei_gemm_f32 (+ BIAS_ROWSCALE).
//
//   forward           Y  = X W^T     X (M,K) and W (N,K) both K-contiguous
//   data gradient     dX = dY W      W read REDUCTION-MAJOR (b_rmajor): its rows are the reduction index
//   weight gradient   dW = dY^T X    both operands reduction-major (a_rmajor, b_rmajor); sei_gemm_bf16nt_dw2 runs
//                                    the reduction over the (dY, X) pairs of two model calls in one launch
//
// Structure (cdna_hip_programming.md section 5, "minimum 2-phase" loop):
//   * 8 waves, block tile 128x128 (2 workgroups per CU) / 192x256 / 96x256, BK = 64.
//   * HBM/L2 -> LDS by global_load_lds_dwordx4: each wave-instruction ("piece") moves 1 KiB of full lines, no
//     VGPR staging. K-contiguous operand: LDS image [row][128 B], 16-byte chunk c of row r at chunk position
//     c ^ ((r>>1)&7), fragment = one ds_read_b128. Reduction-major operand: image [64 k-rows][256 B = 128
//     columns], chunk swizzle swz_rmajor(row), fragment = two ds_read_b64_tr_b16 (the transposing LDS read).
//     Because the DMA destination is lane-linear, each swizzle is applied to the per-lane SOURCE address and
//     again on the fragment read; SQ_LDS_BANK_CONFLICT = 0 on every variant.
//   * double-buffered LDS, one barrier per k-tile (NSTAGE = 2); NSTAGE > 2 = ring with counted vmcnt and raw
//     barriers (tuning variants: measured no better than two co-resident 2-stage workgroups).
//   * tile order: each XCD walks a contiguous range of a band-major order (see the kernel prologue).
//   * narrow outputs split K over workgroups (zero-fill kernel + float atomics), split chosen by rounds.
//   * gemm_bf16pp.h: an opt-in 256x256 ping-pong schedule on the same LDS images.
//
// Requirements checked by the host entries: K % 8 == 0, leading dimensions % 8 == 0, 16-byte aligned A and B.
#include "sei_common.h"

namespace {

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f32x16 = __attribute__((ext_vector_type(16))) float;
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void glb_void;

typedef short v4s __attribute__((ext_vector_type(4)));
typedef short v8s __attribute__((ext_vector_type(8)));

constexpr int BK = 64;
constexpr int ROW_BYTES = 128;
constexpr int NWAVES = 8, NT = 512;

// 16 bytes of zeros: the DMA source of every lane whose chunk lies past the K (or row) edge.
__device__ __attribute__((aligned(16))) unsigned short g_zero_chunk[8] = {0, 0, 0, 0, 0, 0, 0, 0};

struct NtArgs {
    const unsigned short *A, *B;
    float *D32;
    unsigned short *D16;
    int M, N, K, lda, ldb;     // lda / ldb: row strides in elements of A / B as stored
    int epilogue;
    const float *bias, *R1, *R2;
    unsigned short *D2_16;     // BIAS_GELU: gelu(D) in bf16
    int splitk, k_per_split;
    int tiles_m, tiles_n;
    int tiles_per_xcd, band;   // tile order: see the kernel prologue
    // second reduction segment of reduction-major operands: rows k >= k_seg come from A2 / B2 (row k - k_seg)
    const unsigned short *A2, *B2;
    int k_seg;
    // host-side only (the kernels never read them): the caller's explicit tile / band choice of the _ex entry
    // points; 0 = the dispatcher decides. Per call, so the library holds no mutable state.
    int force_tile, force_band;
    // quadrant kernel, unsplit launches whose result is rounded to bf16 (sei_gemm_bf16nt_colsum): colsum[n] += sum_m of the
    // ROUNDED results D16[m][n] -- the bias gradient b2 = column sums of gh3 = (dY W3) gelu'(h3) without a pass over gh3.
    float *colsum;
    // host-side only: sei_gemm_bf16nt_plan. Non-null: the dispatcher writes (family, BM, BN, K splits) there and launches
    // NOTHING -- the schedule a call with these shapes would take, for tests that pin the timed launch set.
    unsigned long long *plan;
    // 3x3 convolution on a zero-bordered grid (sei_gemm_bf16nt_conv): K = 9 * conv_cin; k-tile kt reads A rows
    // shifted by conv_off[tap], tap = k / conv_cin, at column k - tap * conv_cin. conv_cin = 0: a plain GEMM.
    int conv_cin;
    int conv_off[9];
    // Adam applied in the epilogue (sei_gemm_bf16nt_dw2_adam, ADAM instantiations only): the accumulator IS the
    // complete gradient of element (m, n); parameter, both moments and the bf16 shadow share D's (M, N) layout.
    // adam_h (device): beta1, beta2, eps, weight_decay, step_size, 1/sqrt(bias_correction2) of this step.
    float *adam_p, *adam_m, *adam_v;
    unsigned short *adam_p16;
    const float *adam_h;
    // A batch of GEMMs in one launch that differ in the ROW at which a reduction-major B starts and in their output
    // (sei_gemm_bf16nt_dw2_taps: the nine taps of a 3x3 convolution's weight gradient, every one the same dY against the
    // padded input grid shifted by the tap): batch t reads B / B2 from row batch_row[t] on and writes D32 + t * batch_ld.
    int batch;
    int batch_row[9];
    long long batch_ld;
    // sei_gemm_bf16nt_conv_unpad (row-patch epilogue only): the M rows are the pixels of a zero-bordered (H + 2) x (W + 2)
    // grid per image; only interior rows are written, to row (b H + y - 1) W + x - 1 of D32 / R1 (the NHWC tensors),
    // with LeakyReLU(0.01) before the residual when unpad_act. unpad_W = 0: rows as they are.
    int unpad_H, unpad_W, unpad_act;
    // Split-K through slabs (quadrant kernel, sei_gemm_bf16nt_ws): every K slice of a tile stores its accumulators into its
    // slab of the caller's workspace (write-through), draws a ticket from the tile's counter, and the slice that draws the
    // last one adds the other slabs to its registers and runs the WHOLE epilogue (any epilogue: the sum is complete) --
    // no zero-fill launch, no float atomics (1.3 TB/s chip-wide: 29 us for a 288 x 128 tile's 147 KB per workgroup).
    // ws_cnt: one counter per tile, zero between launches (the last arriver puts it back); ws_slab: tiles x splits slabs
    // of BM x BN floats in accumulator order. nullptr: split launches add their tiles with float atomics as before.
    float *ws_slab;
    unsigned *ws_cnt;
    // host-side only: the caller's workspace as handed over, and an explicit number of K slices (0 = the dispatcher's)
    void *ws;
    size_t ws_bytes;
    int force_splitk;
};

typedef float nt_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 nt_load4(const float *p) {
    const nt_f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const nt_f32x4 *>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void nt_store4(float *p, const float4 v) {
    nt_f32x4 t;
    t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
    __builtin_nontemporal_store(t, reinterpret_cast<nt_f32x4 *>(p));
}

// (LDS-DMA pieces go out through sei_common.h's dma16_base / dma16_lane: inline asm, see there)

__device__ __forceinline__ unsigned short f2bf(float v) {
    const __bf16 b = (__bf16)v;
    return __builtin_bit_cast(unsigned short, b);
}

// ---- operand stored K-contiguous: element (o, k) at G[o*ld + k]; LDS image [o][128 B], chunk swizzle (r>>1)&7
// One 1-KiB wave-instruction ("piece") of a ROWS-row tile: piece q covers rows 8q .. 8q+7.
template <int ROWS>
__device__ __forceinline__ void stage_piece(const unsigned short *__restrict__ G, int ld, int row0, int row_lim,
                                            int k0, int k_lim, char *lds_tile, int q, int lane) {
    if (q < ROWS / 8) {                      // wave-uniform
        const int row = 8 * q + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);      // logical chunk this lane must fetch
        const int grow = min(row0 + row, row_lim - 1);     // rows past the edge re-read the last row
        const int k = k0 + 8 * c;
        const unsigned short *src = k < k_lim ? G + (size_t)grow * ld + k : g_zero_chunk;
        dma16_lane(src, lds_tile + q * 1024);
    }
}
template <int ROWS>
__device__ __forceinline__ void stage_tile(const unsigned short *__restrict__ G, int ld, int row0, int row_lim,
                                           int k0, int k_lim, char *lds_tile, int wave, int lane) {
#pragma unroll
    for (int q0 = 0; q0 < ROWS / 8; q0 += NWAVES)
        stage_piece<ROWS>(G, ld, row0, row_lim, k0, k_lim, lds_tile, q0 + wave, lane);
}

// ---- operand stored reduction-major: element (o, k) at G[k*ld + o]; LDS image [64 k-rows][256 B = 128 o],
// 16-byte chunk ch of row r at chunk position ch ^ swz(r) (cdna_hip_programming.md T10, image (b)); consumed
// with ds_read_b64_tr_b16. Rows past K and chunks past the o edge are fed from the zero chunk.
__device__ __forceinline__ int swz_rmajor(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }

__device__ __forceinline__ void stage_piece_rmajor(const unsigned short *__restrict__ G,
                                                   const unsigned short *__restrict__ G2, int k_seg, int ld, int o0,
                                                   int o_lim, int k0, int k_lim, char *lds_tile, int q, int lane) {
    const int row = 4 * q + (lane >> 4);                  // 64 rows x 256 B = 16 pieces of 4 rows
    const int ch = (lane & 15) ^ swz_rmajor(row);
    const int k = k0 + row, o = o0 + 8 * ch;
    const unsigned short *rowp = k < k_seg ? G + (size_t)k * ld : G2 + (size_t)(k - k_seg) * ld;
    const unsigned short *src = (k < k_lim && o < o_lim) ? rowp + o : g_zero_chunk;
    dma16_lane(src, lds_tile + q * 1024);
}
__device__ __forceinline__ void stage_tile_rmajor(const unsigned short *__restrict__ G,
                                                  const unsigned short *__restrict__ G2, int k_seg, int ld, int o0,
                                                  int o_lim, int k0, int k_lim, char *lds_tile, int wave, int lane) {
#pragma unroll
    for (int q0 = 0; q0 < 16; q0 += NWAVES)
        stage_piece_rmajor(G, G2, k_seg, ld, o0, o_lim, k0, k_lim, lds_tile, q0 + wave, lane);
}

// fragment of a reduction-major operand: 8 k-values (k = 16s + 8h + j) of column c0 + (lane & 15)
__device__ __forceinline__ bf16x8 frag_rmajor(const char *tile, int s, int lane, int c0) {
    const int g = lane >> 4, i16 = lane & 15, q = i16 >> 2, p = i16 & 3;
    const int col0 = c0 + 16 * (g & 1);
    const int row = 16 * s + 8 * (g >> 1) + q;
    const int chunk = (col0 >> 3) + (p >> 1);
    auto addr = [&](int r) {
        return (__attribute__((address_space(3))) v4s *)(tile + 256 * r + 16 * (chunk ^ swz_rmajor(r)) + 8 * (p & 1));
    };
    const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(addr(row));
    const v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(addr(row + 4));
    const v8s both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, both);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {          // counted wait: at most N LDS-DMA loads still in flight
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (N == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
    else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else static_assert(N < 0, "add the immediate");
}

// the row-patch epilogue (ROWEPI == 4) needs one wave row of the tile (32 TM rows x BN floats) inside the operand stages
template <int TM, int TN, int WM, int WN, int NSTAGE>
constexpr bool patch_fits() {
    return (NSTAGE * (32 * TM * WM + 32 * TN * WN) * ROW_BYTES) / (32 * TN * WN * 4) >= 32 * TM;
}

template <int TM, int TN, int WM, int WN, bool ARM, bool BRM, int NSTAGE = 2, int ROWEPI = 0>
__global__ __launch_bounds__(NT, (NSTAGE == 1 ? 3 : 1)) void gemm_bf16nt_kernel(NtArgs g) {
    static_assert(WM * WN == NWAVES, "8 waves per workgroup");
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
    static_assert(!ARM || BM == 128, "a reduction-major A tile is 64 x 128");
    static_assert(!BRM || BN % 128 == 0, "a reduction-major B tile is BN/128 images of 64 k-rows x 128 columns");
    constexpr int STAGE = (BM + BN) * ROW_BYTES;
    __shared__ __attribute__((aligned(1024))) char smem[NSTAGE * STAGE];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, lh = lane >> 5;

    // ---- tile / split assignment --------------------------------------------------------------
    // Workgroups are dealt round-robin to the 8 XCDs (each with its own 4 MB L2), so XCD x runs blocks
    // x, x+8, x+16, ... in order. Those blocks walk a CONTIGUOUS range of a band-major tile order (bands of
    // `band` tile columns, row by row inside a band): the ~32-64 tiles an XCD has in flight then form a
    // near-square patch sharing A rows and B columns through its L2. (A strip order -- one tile column per
    // XCD -- measured 52 % L2 hits on a 4096^3 GEMM: 16x the unique bytes through the fabric.)
    // Launches with few tiles (tiles_per_xcd == 0: one-tile weight gradients split 200-fold over K) use the
    // plain order instead -- consecutive blocks = consecutive tiles, then the next K split -- so that the splits
    // of a tile spread over all XCDs (with the XCD order every working block had bid % 8 == 0: one XCD).
    int bid = blockIdx.x;
    // (the argument block itself is never written: a modified copy would leave the scalar registers for scratch memory)
    // ROWEPI == 3 marks the batched instantiation (plain epilogue): everywhere else the three pointers below ARE the
    // arguments, and the compiler keeps the register budget of the unbatched kernel (the one-stage weight-gradient loop
    // went from 78 to 102 VGPRs, i.e. from three workgroups per CU to two, with a run-time test here).
    const unsigned short *Bp = g.B, *B2p = g.B2;
    float *D32p = g.D32;
    if constexpr (ROWEPI == 3) {
        const int per = gridDim.x / g.batch, bt = bid / per;
        bid -= bt * per;
        const ptrdiff_t shift = (ptrdiff_t)g.batch_row[bt] * g.ldb;
        Bp += shift;
        B2p += shift;
        D32p += (size_t)bt * g.batch_ld;
    }
    int zs, ord;
    if (g.tiles_per_xcd == 0) {
        const int ntile = g.tiles_m * g.tiles_n;
        zs = bid / ntile;
        ord = bid - zs * ntile;
    } else {
        const int per_split = 8 * g.tiles_per_xcd;
        zs = bid / per_split;
        bid -= zs * per_split;
        ord = (bid & 7) * g.tiles_per_xcd + (bid >> 3);
        if ((bid >> 3) >= g.tiles_per_xcd || ord >= g.tiles_m * g.tiles_n) return;   // block-uniform, before any barrier
    }
    int tm_i, tn_i;
    {
        const int band_tiles = g.tiles_m * g.band;
        const int b = ord / band_tiles, r = ord - b * band_tiles;
        const int width = min(g.band, g.tiles_n - b * g.band);
        tm_i = r / width;
        tn_i = b * g.band + (r - tm_i * width);
    }
    const int m0 = tm_i * BM, n0 = tn_i * BN;
    const int M = g.M, N = g.N, K = g.K;
    const int k_begin = zs * g.k_per_split;
    const int k_end = min(K, k_begin + g.k_per_split);

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- epilogue inputs --------------------------------------------------------------------------
    // The auxiliary values (running gradient for ACCUM, residuals, GELU' input) are independent of the main
    // loop and could be loaded before it. MEASURED AND REJECTED: holding them costs 32-64 VGPRs, which takes
    // the 128x128 tile from two resident workgroups per CU to one (101 -> 158 VGPRs), and the lost overlap
    // between co-resident blocks outweighs the hidden latency (bench 1000 -> 883 images/s). Kept behind a
    // constant for tiles that are register-rich.
    constexpr bool PREFETCH_AUX = false;
    const int epi = g.epilogue;
    const bool lead = zs == 0;          // with split-K, split 0 carries bias / residual terms
    const bool split = g.splitk > 1;
    const float *aux1p = nullptr, *aux2p = nullptr;
    if (epi == SEI_EPI_BIAS_RES && (!split || lead)) { aux1p = g.R1; aux2p = g.R2; }
    else if (epi == SEI_EPI_BIAS_SCALE_RES) aux1p = g.R2;     // D = R2 + R1[row] * (acc + bias); never split
    else if (epi == SEI_EPI_MUL_DGELU) aux1p = g.R1;
    else if (epi == SEI_EPI_ACCUM && !split) aux1p = D32p;
    float a1[TM][TN][16], a2[TM][TN][16];
    auto gather_aux = [&](int i, int j) {
        const int col = n0 + wn * (32 * TN) + 32 * j + li;
        const int row_base = m0 + wm * (32 * TM) + 32 * i + 4 * lh;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = row_base + (r & 3) + 8 * (r >> 2);
            const bool ok = col < N && row < M;
            const size_t o = (size_t)row * N + col;
            a1[i][j][r] = (ok && aux1p) ? aux1p[o] : 0.f;
            a2[i][j][r] = (ok && aux2p) ? aux2p[o] : 0.f;
        }
    };
    if constexpr (PREFETCH_AUX) {
        if (aux1p) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) gather_aux(i, j);
        } else {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) a1[i][j][r] = a2[i][j][r] = 0.f;
        }
    }

    // Staging. A k-tile that lies inside [k_begin, k_end) -- and, for a reduction-major operand, inside one of
    // its two K segments -- is issued from per-lane byte offsets fixed for the whole launch and a uniform base
    // pointer (rows / columns past the edge are clamped to the last valid one: they only feed outputs that are
    // never stored). Measured on the 256x256 schedule: the per-piece 64-bit lane arithmetic and bounds selects of
    // the general path cost 20 % of the loop. Everything else (K tails, the tile that straddles k_seg) takes the
    // general path below.
    constexpr int NPA = ARM ? 2 : (BM + 63) / 64, NPB = BN / 64;               // pieces per wave (1 KiB each)
    static_assert(BN % 64 == 0, "whole pieces per wave");
    unsigned off_a[NPA], off_b[NPB];
#pragma unroll
    for (int e = 0; e < NPA; ++e) {
        const int q = wave + NWAVES * e;
        if constexpr (ARM) {
            const int row = 4 * q + (lane >> 4), ch = (lane & 15) ^ swz_rmajor(row);
            off_a[e] = (unsigned)row * (unsigned)g.lda * 2u + 2u * (unsigned)min(8 * ch, M - 8 - m0);
        } else {
            const int row = 8 * q + (lane >> 3), c = (lane & 7) ^ ((row >> 1) & 7);
            off_a[e] = (unsigned)min(row, M - 1 - m0) * (unsigned)g.lda * 2u + 16u * c;
        }
    }
#pragma unroll
    for (int e = 0; e < NPB; ++e) {
        const int q = wave + NWAVES * e;
        if constexpr (BRM) {
            const int im = q >> 4, qq = q & 15;                // 16 pieces per 128-column image
            const int row = 4 * qq + (lane >> 4), ch = (lane & 15) ^ swz_rmajor(row);
            off_b[e] = (unsigned)row * (unsigned)g.ldb * 2u +
                       2u * (unsigned)max(min(128 * im + 8 * ch, N - 8 - n0), 0);
        } else {
            const int row = 8 * q + (lane >> 3), c = (lane & 7) ^ ((row >> 1) & 7);
            off_b[e] = (unsigned)min(row, N - 1 - n0) * (unsigned)g.ldb * 2u + 16u * c;
        }
    }
    auto rm_base = [&](const unsigned short *X1, const unsigned short *X2, int ld, int o0, int k0) -> const char * {
        if (k0 + BK <= g.k_seg && k0 + BK <= k_end) return reinterpret_cast<const char *>(X1 + (size_t)k0 * ld + o0);
        if (k0 >= g.k_seg && k0 + BK <= k_end) return reinterpret_cast<const char *>(X2 + (size_t)(k0 - g.k_seg) * ld + o0);
        return nullptr;
    };
    auto stage = [&](int k0, char *dst) {
        const char *ab, *bb;
        if constexpr (ARM) ab = rm_base(g.A, g.A2, g.lda, m0, k0);
        else if (g.conv_cin > 0) {                         // uniform per k-tile: the tap's row shift (K % BK == 0)
            const int tap = k0 / g.conv_cin;
            ab = reinterpret_cast<const char *>(g.A + ((ptrdiff_t)m0 + g.conv_off[tap]) * (ptrdiff_t)g.lda +
                                                (k0 - tap * g.conv_cin));
        }
        else ab = k0 + BK <= k_end ? reinterpret_cast<const char *>(g.A + (size_t)m0 * g.lda + k0) : nullptr;
        if (ab) {
#pragma unroll
            for (int e = 0; e < NPA; ++e)
                if (ARM || wave + NWAVES * e < BM / 8)         // wave-uniform (a 96-row tile has 12 pieces)
                    dma16_base(ab, off_a[e], dst + (wave + NWAVES * e) * 1024);
        } else {
            if constexpr (ARM) stage_tile_rmajor(g.A, g.A2, g.k_seg, g.lda, m0, M, k0, k_end, dst, wave, lane);
            else stage_tile<BM>(g.A, g.lda, m0, M, k0, k_end, dst, wave, lane);
        }
        if constexpr (BRM) bb = rm_base(Bp, B2p, g.ldb, n0, k0);
        else bb = k0 + BK <= k_end ? reinterpret_cast<const char *>(Bp + (size_t)n0 * g.ldb + k0) : nullptr;
        if (bb) {
#pragma unroll
            for (int e = 0; e < NPB; ++e)
                dma16_base(bb, off_b[e], dst + BM * ROW_BYTES + (wave + NWAVES * e) * 1024);
        } else if constexpr (BRM) {
#pragma unroll
            for (int im = 0; im < BN / 128; ++im)           // 16 KB image per 128 columns
                stage_tile_rmajor(Bp, B2p, g.k_seg, g.ldb, n0 + 128 * im, N, k0, k_end,
                                  dst + BM * ROW_BYTES + im * 16384, wave, lane);
        } else {
            stage_tile<BN>(Bp, g.ldb, n0, N, k0, k_end, dst + BM * ROW_BYTES, wave, lane);
        }
    };
    const int sw = (li >> 1) & 7;                              // swizzle key of this lane's rows
    const int a_off = (wm * 32 * TM + li) * ROW_BYTES;
    const int b_off = BM * ROW_BYTES + (wn * 32 * TN + li) * ROW_BYTES;

    auto compute = [&](const char *now) {
#pragma unroll
        for (int s = 0; s < BK / 16; ++s) {
            const int pc = ((2 * s + lh) ^ sw) * 16;
            bf16x8 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                if constexpr (ARM) a[i] = frag_rmajor(now, s, lane, wm * 32 * TM + 32 * i);
                else a[i] = *reinterpret_cast<const bf16x8 *>(now + a_off + i * 32 * ROW_BYTES + pc);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if constexpr (BRM) {
                    const int c0 = wn * 32 * TN + 32 * j;
                    b[j] = frag_rmajor(now + BM * ROW_BYTES + (c0 >> 7) * 16384, s, lane, c0 & 127);
                }
                else b[j] = *reinterpret_cast<const bf16x8 *>(now + b_off + j * 32 * ROW_BYTES + pc);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    };

    if constexpr (NSTAGE == 1) {
        // one stage, nothing overlapped inside the workgroup: 32 KB of LDS and <= 84 VGPRs let THREE workgroups
        // share a CU and cover for one another (the guide's "step-3" structure)
        for (int k0 = k_begin; k0 < k_end; k0 += BK) {
            stage(k0, smem);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            compute(smem);
            __syncthreads();
        }
    } else if constexpr (NSTAGE == 2) {
        // two stages, one workgroup's DMA in flight under its own MFMAs; relies on a second resident
        // workgroup per CU for overlap (128x128 tile)
        stage(k_begin, smem);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int cur = 0;
        for (int k0 = k_begin; k0 < k_end; k0 += BK) {
            if (k0 + BK < k_end) stage(k0 + BK, smem + (cur ^ 1) * STAGE);
            compute(smem + cur * STAGE);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            cur ^= 1;
        }
    } else {
        // Ring of NSTAGE slots, NSTAGE-1 tiles of LDS-DMA in flight (cdna_hip_programming.md T3+T4): counted
        // vmcnt (never 0 in steady state) and a raw s_barrier, so the DMA queue never drains at a barrier.
        // Per wave and tile the DMA count is uniform: PER = (BM + BN) / 64 instructions.
        constexpr int PER = (BM + BN) / 64;
        static_assert((BM + BN) % 64 == 0, "uniform DMA count per wave");
        const int nt = (k_end - k_begin + BK - 1) / BK;
#pragma unroll
        for (int t = 0; t < NSTAGE - 1; ++t)
            if (t < nt) stage(k_begin + t * BK, smem + t * STAGE);
        int slot = 0;
        for (int t = 0; t < nt; ++t) {
            // tile t must have landed: everything issued after it may still be in flight
            if (t + NSTAGE - 2 < nt) wait_vmcnt<(NSTAGE - 2) * PER>();
            else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();          // every wave's share of tile t is in LDS; slot t-1 is free
            const int tn = t + NSTAGE - 1;
            if (tn < nt) {
                int ns = slot + NSTAGE - 1;
                if (ns >= NSTAGE) ns -= NSTAGE;
                stage(k_begin + tn * BK, smem + ns * STAGE);
            }
            compute(smem + slot * STAGE);
            if (++slot == NSTAGE) slot = 0;
        }
        wait_vmcnt<0>();
    }

    // ---- epilogue: optimizer step on the finished gradient tile (never split over K) ---------------------------
    // The tile goes through LDS in two halves of 64 rows x 128 columns (32 KB, the operand stage that the loop has
    // finished with): accumulators in, whole rows out, so that every access to the four parameter-sized streams is
    // 16 bytes per lane on 512 contiguous bytes of a row (8 per lane for the bf16 shadow) -- 26 bytes move per output
    // element and nothing else bounds this launch.
    if constexpr (ROWEPI == 2) {                                     // the gradient itself, as bf16 quads on whole rows
        static_assert(TM == 2 && TN == 1 && WM == 2 && WN == 4, "the row patch below is laid out for the 128 x 128 tile");
        float *patch = reinterpret_cast<float *>(smem);              // [64][128]
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            __builtin_amdgcn_s_barrier();
            if (wm == h) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        patch[(32 * i + 4 * lh + (r & 3) + 8 * (r >> 2)) * 128 + 32 * wn + li] = acc[i][0][r];
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int idx = threadIdx.x + NT * k, r = idx >> 5, c4 = idx & 31;
                const int row = m0 + 64 * h + r, col = n0 + 4 * c4;
                if (row >= M || col >= N) continue;                  // N % 8 == 0: a quad is all in or all out
                const float4 v = *reinterpret_cast<const float4 *>(patch + r * 128 + 4 * c4);
                uint2 w;
                w.x = (unsigned)f2bf(v.x) | ((unsigned)f2bf(v.y) << 16);
                w.y = (unsigned)f2bf(v.z) | ((unsigned)f2bf(v.w) << 16);
                *reinterpret_cast<uint2 *>(g.D16 + (size_t)row * N + col) = w;
            }
        }
        return;
    }
    if constexpr (ROWEPI == 1) {
        static_assert(TM == 2 && TN == 1 && WM == 2 && WN == 4, "the row patch below is laid out for the 128 x 128 tile");
        const float beta1 = g.adam_h[0], beta2 = g.adam_h[1], eps = g.adam_h[2], wd = g.adam_h[3];
        const float step_size = g.adam_h[4], inv_bc2_sqrt = g.adam_h[5];
        float *patch = reinterpret_cast<float *>(smem);              // [64][128]
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            __builtin_amdgcn_s_barrier();                            // operands (h = 0) / the previous half are done with
            if (wm == h) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        patch[(32 * i + 4 * lh + (r & 3) + 8 * (r >> 2)) * 128 + 32 * wn + li] = acc[i][0][r];
            }
            __syncthreads();
#pragma unroll 1
          for (int kk = 0; kk < 4; kk += 2) {                        // two quads at a time: registers for 3 workgroups / CU
            float4 gq[2], pq[2], mq[2], vq[2];
            size_t off[2];
            bool ok[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {                            // 6 independent 16-byte loads per lane in flight
                const int idx = threadIdx.x + NT * (kk + k), r = idx >> 5, c4 = idx & 31;
                const int row = m0 + 64 * h + r, col = n0 + 4 * c4;
                ok[k] = row < M && col < N;                          // N % 8 == 0: a quad is all in or all out
                off[k] = ok[k] ? (size_t)row * N + col : 0;
                gq[k] = *reinterpret_cast<const float4 *>(patch + r * 128 + 4 * c4);
                // streamed once per step: non-temporal, so that they do not push the GEMM operands out of L2 / MALL
                pq[k] = nt_load4(g.adam_p + off[k]);
                mq[k] = nt_load4(g.adam_m + off[k]);
                vq[k] = nt_load4(g.adam_v + off[k]);
            }
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                if (!ok[k]) continue;
                float4 o;
                o.x = sei_adam_element(pq[k].x, gq[k].x, mq[k].x, vq[k].x, beta1, beta2, eps, wd, step_size, inv_bc2_sqrt);
                o.y = sei_adam_element(pq[k].y, gq[k].y, mq[k].y, vq[k].y, beta1, beta2, eps, wd, step_size, inv_bc2_sqrt);
                o.z = sei_adam_element(pq[k].z, gq[k].z, mq[k].z, vq[k].z, beta1, beta2, eps, wd, step_size, inv_bc2_sqrt);
                o.w = sei_adam_element(pq[k].w, gq[k].w, mq[k].w, vq[k].w, beta1, beta2, eps, wd, step_size, inv_bc2_sqrt);
                nt_store4(g.adam_m + off[k], mq[k]);
                nt_store4(g.adam_v + off[k], vq[k]);
                nt_store4(g.adam_p + off[k], o);
                if (g.adam_p16) {
                    uint2 w;
                    w.x = (unsigned)f2bf(o.x) | ((unsigned)f2bf(o.y) << 16);
                    w.y = (unsigned)f2bf(o.z) | ((unsigned)f2bf(o.w) << 16);
                    typedef unsigned nt_u32x2 __attribute__((ext_vector_type(2)));
                    nt_u32x2 t;
                    t.x = w.x; t.y = w.y;
                    __builtin_nontemporal_store(t, reinterpret_cast<nt_u32x2 *>(g.adam_p16 + off[k]));
                }
            }
          }
        }
        return;
    }

    // ---- epilogue through an LDS row patch (ROWEPI == 4: unsplit launches with N % 4 == 0 and aligned arrays) -----------
    // The accumulators go through the operand stages (free after the loop) in passes of whole wave rows, and every
    // thread then owns quads of a row: residual / GELU' inputs arrive as 16-byte loads on contiguous row segments, float32
    // results leave as 16-byte and bf16 results as 8-byte stores. The element-wise epilogue below issues one 4- or 2-byte
    // access per value and lane and is bound by that instruction count wherever K is short.
    if constexpr ((ROWEPI == 4 || ROWEPI == 5) && patch_fits<TM, TN, WM, WN, NSTAGE>()) {     // 5: + the unpad row remap
        constexpr int ROWS_WM = 32 * TM;
        constexpr int LDS_ROWS = (NSTAGE * STAGE) / (BN * 4);
        constexpr int GW = LDS_ROWS >= WM * ROWS_WM ? WM : (LDS_ROWS >= 2 * ROWS_WM && WM >= 2 ? 2 : 1);
        static_assert(LDS_ROWS >= ROWS_WM && WM % GW == 0, "one wave row of the tile fits the operand stages");
        constexpr int RP = GW * ROWS_WM, QPR = BN / 4, TOTAL = RP * QPR;
        float *patch = reinterpret_cast<float *>(smem);
        const bool to_bf16_only = g.D16 != nullptr && D32p == nullptr;
#pragma unroll 1
        for (int pass = 0; pass < WM / GW; ++pass) {
            __builtin_amdgcn_s_barrier();                            // operands / the previous pass are done with
            if (wm / GW == pass) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            patch[((wm % GW) * ROWS_WM + 32 * i + 4 * lh + (r & 3) + 8 * (r >> 2)) * BN + wn * (32 * TN) +
                                  32 * j + li] = acc[i][j][r];
            }
            __syncthreads();
            // quads per thread and round: two (loads of both in flight before the first store); one in the one-stage loop,
            // whose three workgroups per CU live on 85 VGPRs
            constexpr int U = NSTAGE == 1 ? 1 : 2;
#pragma unroll 1
            for (int base = threadIdx.x; base < TOTAL; base += U * NT) {
                float4 v[U], r1[U], r2[U];
                size_t off[U];
                int rowi[U];
                bool ok[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int idx = base + u * NT;
                    const int r = idx / QPR, c4 = idx - r * QPR;
                    int row = m0 + pass * RP + r;
                    const int col = n0 + 4 * c4;
                    ok[u] = idx < TOTAL && row < M && col < N;                  // N % 4 == 0: a quad is all in or all out
                    if constexpr (ROWEPI == 5) {                                // padded-grid pixel -> NHWC row, border rows dropped
                        const int Wp = g.unpad_W + 2, HWp = (g.unpad_H + 2) * Wp;
                        const int b = row / HWp, rem = row - b * HWp, yy = rem / Wp, xx = rem - yy * Wp;
                        ok[u] = ok[u] && yy >= 1 && yy <= g.unpad_H && xx >= 1 && xx <= g.unpad_W;
                        row = (b * g.unpad_H + yy - 1) * g.unpad_W + xx - 1;
                    }
                    off[u] = ok[u] ? (size_t)row * N + col : 0;
                    rowi[u] = ok[u] ? row : 0;
                    v[u] = *reinterpret_cast<const float4 *>(patch + (idx < TOTAL ? r * BN + 4 * c4 : 0));
                    r1[u] = r2[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (epi == SEI_EPI_BIAS || epi == SEI_EPI_BIAS_GELU || epi == SEI_EPI_BIAS_RES ||
                        epi == SEI_EPI_BIAS_ROWSCALE || epi == SEI_EPI_BIAS_SCALE_RES) {
                        const int cb = ok[u] ? col : 0;
                        float4 b = make_float4(g.bias[cb], g.bias[cb + 1], g.bias[cb + 2], g.bias[cb + 3]);
                        if (epi == SEI_EPI_BIAS_ROWSCALE) {
                            const float sc = g.R1[rowi[u]];
                            b.x *= sc; b.y *= sc; b.z *= sc; b.w *= sc;
                        }
                        v[u].x += b.x; v[u].y += b.y; v[u].z += b.z; v[u].w += b.w;
                    }
                    if (epi == SEI_EPI_BIAS_RES || epi == SEI_EPI_MUL_DGELU)
                        r1[u] = *reinterpret_cast<const float4 *>(g.R1 + off[u]);
                    else if (epi == SEI_EPI_ACCUM) r1[u] = *reinterpret_cast<const float4 *>(D32p + off[u]);
                    if (epi == SEI_EPI_BIAS_SCALE_RES || (epi == SEI_EPI_BIAS_RES && g.R2))
                        r2[u] = *reinterpret_cast<const float4 *>(g.R2 + off[u]);
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (!ok[u]) continue;
                    float4 o = v[u];
                    if (epi == SEI_EPI_MUL_DGELU) {
                        if (to_bf16_only) {
                            o.x *= sei_dgelu_bf16out(r1[u].x); o.y *= sei_dgelu_bf16out(r1[u].y);
                            o.z *= sei_dgelu_bf16out(r1[u].z); o.w *= sei_dgelu_bf16out(r1[u].w);
                        } else {
                            o.x *= sei_dgelu(r1[u].x); o.y *= sei_dgelu(r1[u].y);
                            o.z *= sei_dgelu(r1[u].z); o.w *= sei_dgelu(r1[u].w);
                        }
                    } else if (epi == SEI_EPI_BIAS_SCALE_RES) {
                        const float sc = g.R1[rowi[u]];
                        o.x = fmaf(o.x, sc, r2[u].x); o.y = fmaf(o.y, sc, r2[u].y);
                        o.z = fmaf(o.z, sc, r2[u].z); o.w = fmaf(o.w, sc, r2[u].w);
                    } else {
                        if (ROWEPI == 5 && g.unpad_act) {
                            o.x = o.x > 0.f ? o.x : 0.01f * o.x; o.y = o.y > 0.f ? o.y : 0.01f * o.y;
                            o.z = o.z > 0.f ? o.z : 0.01f * o.z; o.w = o.w > 0.f ? o.w : 0.01f * o.w;
                        }
                        o.x += r1[u].x + r2[u].x; o.y += r1[u].y + r2[u].y;
                        o.z += r1[u].z + r2[u].z; o.w += r1[u].w + r2[u].w;
                    }
                    if (epi == SEI_EPI_BIAS_GELU) {
                        uint2 w;
                        w.x = (unsigned)f2bf(sei_gelu_bf16out(o.x)) | ((unsigned)f2bf(sei_gelu_bf16out(o.y)) << 16);
                        w.y = (unsigned)f2bf(sei_gelu_bf16out(o.z)) | ((unsigned)f2bf(sei_gelu_bf16out(o.w)) << 16);
                        *reinterpret_cast<uint2 *>(g.D2_16 + off[u]) = w;
                    }
                    if (D32p) *reinterpret_cast<float4 *>(D32p + off[u]) = o;
                    if (g.D16) {
                        uint2 w;
                        w.x = (unsigned)f2bf(o.x) | ((unsigned)f2bf(o.y) << 16);
                        w.y = (unsigned)f2bf(o.z) | ((unsigned)f2bf(o.w) << 16);
                        *reinterpret_cast<uint2 *>(g.D16 + off[u]) = w;
                    }
                }
            }
        }
        return;
    }

    // ---- epilogue -----------------------------------------------------------------------------------
    // Per 32x32 accumulator tile: the auxiliary values are already in registers (prefetched above) or are
    // gathered first (16 independent loads in flight); then compute and store. Interleaving the loads with
    // the stores would serialise them.
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + wn * (32 * TN) + 32 * j + li;
            const bool col_ok = col < N;
            const int row_base = m0 + wm * (32 * TM) + 32 * i + 4 * lh;
            float bias = (col_ok && (!split || lead) &&
                          (epi == SEI_EPI_BIAS || epi == SEI_EPI_BIAS_GELU || epi == SEI_EPI_BIAS_RES ||
                           epi == SEI_EPI_BIAS_ROWSCALE || epi == SEI_EPI_BIAS_SCALE_RES))
                             ? g.bias[col] : 0.f;
            if constexpr (!PREFETCH_AUX) gather_aux(i, j);
            if (epi == SEI_EPI_BIAS_ROWSCALE) {          // D = acc + bias[n] * R1[m]   (R1: one value per row)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = row_base + (r & 3) + 8 * (r >> 2);
                    a1[i][j][r] = row < M ? bias * g.R1[row] : 0.f;
                }
                bias = 0.f;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row_base + (r & 3) + 8 * (r >> 2);
                if (!col_ok || row >= M) continue;
                const size_t o = (size_t)row * N + col;
                float v = acc[i][j][r] + bias;
                if (split) {
                    atomicAdd(D32p + o, v + a1[i][j][r] + a2[i][j][r]);
                    continue;
                }
                if (epi == SEI_EPI_MUL_DGELU)
                    v *= (g.D16 && !D32p) ? sei_dgelu_bf16out(a1[i][j][r]) : sei_dgelu(a1[i][j][r]);
                else if (epi == SEI_EPI_BIAS_SCALE_RES) v = fmaf(v, g.R1[row], a1[i][j][r]);
                else v += a1[i][j][r] + a2[i][j][r];
                if (epi == SEI_EPI_BIAS_GELU) g.D2_16[o] = f2bf(sei_gelu_bf16out(v));
                if (D32p) D32p[o] = v;
                if (g.D16) g.D16[o] = f2bf(v);
            }
        }
    }
}

// Zero-fill of a split-K output. A kernel (not hipMemsetAsync): a memset issued from inside the library was
// observed not to be replayed reliably as part of a captured hipGraph, which left stale partial sums.
__global__ __launch_bounds__(256) void zero_fill_kernel(float *__restrict__ p, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t n4 = n / 4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride)
        reinterpret_cast<float4 *>(p)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (size_t i = n4 * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = 0.f;
}

#ifdef SEI_TUNING
// Tools-only build (make tuning -> libsei_hip_tuning.so): process-wide defaults for the tile / band arguments and a forced
// K-split count, so that the experiment scripts under tools/ can steer launches. Not in libsei_hip.so.
int g_tuning_tile = 0, g_tuning_band = 0, g_tuning_splitk = 0;
#include "gemm_bf16pp.h"  // 256 x 256 ping-pong schedule on the same LDS images (tools-only build)
#endif
#include "gemm_bf16pq.h"

template <int TM, int TN, int WM, int WN, bool ARM = false, bool BRM = false, int NSTAGE = 2, int ROWEPI = 0>
int launch_nt(NtArgs &g, hipStream_t s) {
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
    g.tiles_m = (int)sei_ceil_div(g.M, BM);
    g.tiles_n = (int)sei_ceil_div(g.N, BN);
    const size_t tiles = (size_t)g.tiles_m * g.tiles_n;
    const size_t nbatch = g.batch > 1 ? (size_t)g.batch : 1;
    SEI_REQUIRE(tiles * nbatch < ((size_t)1 << 27));
    {
        constexpr size_t STAGE_BYTES = NSTAGE * (size_t)(BM + BN) * ROW_BYTES;
        const double conc = 32.0 * (STAGE_BYTES <= 40 * 1024 ? 3 : (STAGE_BYTES <= 80 * 1024 ? 2 : 1));   // tiles in flight per XCD
        int band = g.force_band > 0 ? g.force_band : (int)(sqrt(conc * BM / BN) + 0.5);   // square patch in elements
        if (band < 1) band = 1;
        if (band > g.tiles_n) band = g.tiles_n;
        g.band = band;
        g.tiles_per_xcd = tiles < 64 ? 0 : (int)sei_ceil_div(tiles, 8);      // 0 = plain order (few tiles)
    }
    g.splitk = 1;
    g.k_per_split = g.K;
    const bool splittable = (g.epilogue == SEI_EPI_NONE || g.epilogue == SEI_EPI_BIAS ||
                             g.epilogue == SEI_EPI_BIAS_RES || g.epilogue == SEI_EPI_ACCUM) && !g.unpad_W;
    if ((ROWEPI == 0 || ROWEPI == 3) && splittable && g.D32 && !g.D16 && g.K >= 8 * BK) {
        // Wave-quantisation-aware split: a launch takes ceil(tiles*sk / slots) rounds of workgroups, each
        // round costing (k-tiles per split + a fixed prologue/epilogue/atomics overhead); pick the cheapest sk.
        constexpr size_t STAGE_BYTES = NSTAGE * (size_t)(BM + BN) * ROW_BYTES;
        const size_t slots = 256 * (STAGE_BYTES <= 40 * 1024 ? 3 : (STAGE_BYTES <= 80 * 1024 ? 2 : 1));   // resident workgroups
        const size_t ktiles = sei_ceil_div(g.K, BK);
        // up to 256 splits: a one-tile weight gradient of the shallow levels (128 x 32 outputs, K = 221,184 =
        // 3456 k-tiles) capped at 16 splits kept 16 CUs busy for 147 us; its atomics are contiguous and few
        const size_t max_sk = ktiles / 4 < 256 ? ktiles / 4 : 256;
        const double overhead = 6.0 + (g.epilogue == SEI_EPI_ACCUM ? 0.0 : 2.0);   // in k-tile units
        // a split launch that does not accumulate into a running gradient pays a zero-fill launch and its atomics on top:
        // ~32 k-tiles' worth (round 4, tools/exp_skinny.py: 9216 x 128 x 512 24 -> 12.6 us and 2304 x 512 x 2048 37 -> 28 us
        // unsplit, while 288 / 576 x 2048 x 8192 still want 8 / 6 splits)
        const double split_cost = g.epilogue == SEI_EPI_ACCUM ? 2.0 : 32.0;
        double best = 1e30;
        size_t best_sk = 1;
        for (size_t sk = 1; sk <= max_sk; ++sk) {
            const double rounds = (double)sei_ceil_div(tiles * nbatch * sk, slots);
            const double cost = rounds * ((double)sei_ceil_div(ktiles, sk) + overhead + (sk > 1 ? split_cost : 0.0));
            if (cost < best * 0.97) {            // prefer fewer splits unless the gain is real
                best = cost;
                best_sk = sk;
            }
        }
#ifdef SEI_TUNING
        if (g_tuning_splitk > 0) best_sk = (size_t)g_tuning_splitk < ktiles ? (size_t)g_tuning_splitk : ktiles;
#endif
        if (best_sk > 1) {
            g.k_per_split = (int)(sei_ceil_div(ktiles, best_sk) * BK);
            g.splitk = (int)sei_ceil_div(g.K, g.k_per_split);
        }
    }
    if (g.plan) {
        *g.plan = (1ull << 48) | ((unsigned long long)BM << 32) | ((unsigned long long)BN << 16) | (unsigned)g.splitk;
        return SEI_OK;
    }
    if (g.splitk > 1 && g.epilogue != SEI_EPI_ACCUM) {      // ACCUM adds into the running gradient as it is
        if (nbatch > 1) return SEI_ERR_BAD_ARG;             // (batched launches accumulate or run unsplit)
        const size_t n = (size_t)g.M * g.N;
        size_t zg = sei_ceil_div(n / 4 + 1, 256);
        if (zg > 2048) zg = 2048;
        if ((reinterpret_cast<uintptr_t>(g.D32) & 15) != 0) return SEI_ERR_BAD_ARG;
        hipLaunchKernelGGL(zero_fill_kernel, dim3((unsigned)zg), dim3(256), 0, s, g.D32, n);
    }
    const size_t per_split = g.tiles_per_xcd ? 8 * (size_t)g.tiles_per_xcd : tiles;
    if constexpr (ROWEPI == 0 && patch_fits<TM, TN, WM, WN, NSTAGE>()) {
        // unsplit launches whose rows can be moved as aligned quads take the row-patch epilogue
        const auto al = [](const void *p, uintptr_t m) { return (reinterpret_cast<uintptr_t>(p) & m) == 0; };
        const bool r1_rows = g.epilogue == SEI_EPI_BIAS_ROWSCALE || g.epilogue == SEI_EPI_BIAS_SCALE_RES;
        if (g.splitk == 1 && g.N % 4 == 0 && al(g.D32, 15) && al(g.D16, 7) && al(g.D2_16, 7) && (r1_rows || al(g.R1, 15)) &&
            al(g.R2, 15) && !g.force_band) {
            if constexpr (!ARM && !BRM && NSTAGE == 2) {             // (the convolutions' two tile shapes)
                if (g.unpad_W) {
                    hipLaunchKernelGGL((gemm_bf16nt_kernel<TM, TN, WM, WN, ARM, BRM, NSTAGE, 5>), dim3((unsigned)per_split),
                                       dim3(NT), 0, s, g);
                    return sei_launch_status();
                }
            }
            hipLaunchKernelGGL((gemm_bf16nt_kernel<TM, TN, WM, WN, ARM, BRM, NSTAGE, 4>), dim3((unsigned)per_split), dim3(NT),
                               0, s, g);
            return sei_launch_status();
        }
    }
    if (g.unpad_W) return SEI_ERR_BAD_ARG;                  // the row remap lives in the row-patch epilogue only
    hipLaunchKernelGGL((gemm_bf16nt_kernel<TM, TN, WM, WN, ARM, BRM, NSTAGE, ROWEPI>),
                       dim3((unsigned)(per_split * g.splitk * nbatch)),
                       dim3(NT), 0, s, g);
    return sei_launch_status();
}

}  // namespace

#ifdef SEI_TUNING
extern "C" int sei_debug_set_nt_tile(int code) {
    if (code >= 1000) {                // 1000 + n: force n K-splits on splittable launches (1000 = the cost model)
        g_tuning_splitk = code - 1000;
        return SEI_OK;
    }
    if (code >= 100) {                 // 100 + band: force the band width of the tile order (100 = automatic)
        g_tuning_band = code - 100;
        return SEI_OK;
    }
    g_tuning_tile = code;
    return SEI_OK;
}
#endif

// Quadrant schedule (gemm_bf16pq.h) or not, and which tile: returns 0, or 10 * RF + NF.
//   * launches that do not split K (bf16 / GELU / GELU' outputs: the expanding 1x1 convolutions and their data
//     gradients) take it when 288- or 256-row tiles fill the chip (tools/exp_pq.py: 2304 x 8192 x 2048
//     124 -> 83 us, 576 x 32768 x 8192 495 -> 246 us, 288 x 32768 x 8192 249 -> 170 us);
//   * split-K launches (f32 outputs of the contracting convolutions) only with a long reduction and a wide
//     output (576 x 8192 x 32768: 441 -> 288 us; 2304 x 2048 x 8192: 130 -> 104 us); shorter ones lose to the
//     128 x 128 loop (9216 x 512 x 2048: 44 vs 55 us), whose three workgroups per CU hide the atomics.
int pq_choose(const NtArgs &g, bool would_split) {
    const int M = g.M, N = g.N, K = g.K;
    const int rf = M % 288 == 0 ? 9 : ((M % 256 == 0 || M >= 4096) ? 8 : 0);
    // K >= 128: even two k-tiles per tile pay, because the epilogue moves whole rows (36864 x 512 x 128 with
    // GELU 36 -> 28 us, its data gradient 49 -> 32 us); the 128 x 128 loop stores element by element
    if (g.force_tile != 0 || !rf || K < 32 || N < 128 || !pq_eligible(g)) return 0;
    const size_t tm = sei_ceil_div(M, 32 * rf);
    const size_t t4 = N >= 256 ? tm * sei_ceil_div(N, 256) : 0, t2 = tm * sei_ceil_div(N, 128);
    auto fills = [](size_t t) { return (double)t / (double)(sei_ceil_div(t, 256) * 256) >= 0.8; };
    if (K < 128 && (would_split || N > 128)) return 0;          // one-k-tile launches: only the C = 32 expanding convolutions
    if (N < 512 && would_split) return 0;                       // narrow f32 outputs: no gain measured (36864 x 128 x 512)
    // enough whole tiles to fill the chip: the launch does not split K whatever its epilogue allows
    if (t4 >= 192 && fills(t4)) return 10 * rf + 4;
    if (t2 >= 192 && t4 < 192 && fills(t2)) return 10 * rf + 2;
    if (would_split) {
        if (K < 8192 || N < 2048 || (M < 2304 && N < 8192)) return 0;
        // (256-column tiles from 96 of them on: the joint backward's 3456 x 2048 x 8192 runs 153 us there against 192 on
        // 128-column tiles, while 2304 rows -- 64 tiles -- prefer the narrow ones: 104 vs 124, tools/exp_joint_rows.py)
        return 10 * rf + ((t4 >= 96 || (t4 >= 64 && N >= 8192)) ? 4 : 2);
    }
    if (t2 >= 128 && t4 < 192) return 10 * rf + 2;             // half the chip on 288-row tiles beats 96 tiles of 192x256
    return 0;
}

// With a split-K workspace (sei_gemm_bf16nt_ws): the float32-output launches above that split K -- and the ones the rules
// above leave on the 128 x 128 loop because their atomics cost more than the split bought -- choose tile width AND slice
// count together from a cost model in microseconds, fitted to tools/exp_splitk_slabs.py (batch 32 shapes, MI355X):
//   per round of 256 workgroups: k-tiles per slice x c + o (+ e0 + e1 (S - 1) when split: slab store, ticket, slab reads),
//   c = 0.95 / 1.5 us per k-tile of a 128- / 256-column tile fed from L2 / MALL (1.05 / 1.75 when the weight is too large
//   to stay there and streams from HBM), o = 10 / 16, e0 + e1 (S - 1) = 5 + 3 (S - 1) / 8 + 5 (S - 1).
// Model -> measured (us; today's launch in brackets): 2304 x 2048 x 8192: 128 columns, 2 slices 79 -> 80 (102, atomics);
// 1152 x 2048 x 8192: 128, 4: 54 -> 54 (70 on the 128 x 128 loop); 3456 x 2048 x 8192: 256, 2: 125 -> 131 (154); 4608 x 512 x
// 2048: 128, 3: 32 -> 30 (34); 576 x 8192 x 32768: 256, 4: 263 -> 269 (283); 864 x 8192 x 32768: 128, 4: 475 -> 455 (470).
// Returns 10 * RF + NF and sets g.force_splitk, or 0 (no workspace, not such a launch: the rules above decide).
int pq_choose_slabs(NtArgs &g, bool would_split) {
    const int M = g.M, N = g.N, K = g.K;
    if (!g.ws || !would_split || g.epilogue == SEI_EPI_ACCUM || g.force_tile != 0 || g.force_splitk != 0) return 0;
    if (K < 2048 || K % BK != 0 || N < 512 || !pq_eligible(g)) return 0;
    const int rf = M % 288 == 0 ? 9 : ((M % 256 == 0 || M >= 4096) ? 8 : 0);
    if (!rf) return 0;
    const size_t tm = sei_ceil_div(M, 32 * rf), kt = (size_t)K / BK;
    const bool from_hbm = (double)N * (double)K * 2.0 >= 128.0 * 1048576.0;
    double best = 1e30;
    int best_nf = 0, best_sk = 1;
    for (int nf = 2; nf <= 4; nf += 2) {
        const size_t tiles = tm * sei_ceil_div(N, 64 * nf);
        const double c = nf == 2 ? (from_hbm ? 1.05 : 0.95) : (from_hbm ? 1.75 : 1.5);
        const double o = nf == 2 ? 10.0 : 16.0, e0 = nf == 2 ? 5.0 : 8.0, e1 = nf == 2 ? 3.0 : 5.0;
        const size_t slab = (size_t)32 * rf * 64 * nf * 4;
        for (size_t sk = 1; sk <= 16 && sk * 4 <= kt; ++sk) {
            if (sk > 1 && (tiles > 4096 || 16384 + tiles * sk * slab > g.ws_bytes || sk * slab >= ((size_t)1 << 31))) break;
            const double rounds = (double)sei_ceil_div(tiles * sk, 256);
            const double cost = rounds * ((double)sei_ceil_div(kt, sk) * c + o + (sk > 1 ? e0 + e1 * (double)(sk - 1) : 0.0));
            if (cost < best * 0.98) {
                best = cost;
                best_nf = nf;
                best_sk = (int)sk;
            }
        }
    }
    if (!best_nf) return 0;
    // One row tile against a weight that streams from HBM (288 x 8192 x 32768, the B-row bottleneck convolution): the model's
    // per-k-tile constants do not hold once all 256 CUs pull on the weight (tools/exp_l4_slices.py, us: 128 columns x 3 / 4
    // slices 202 / 213, 256 columns x 6 / 7 / 8 slices 193 / 190 / 203) -- 256-column tiles on ~224 workgroups
    if (from_hbm && tm == 1 && N >= 4096) {
        const size_t tiles4 = sei_ceil_div(N, 256), slab4 = (size_t)32 * rf * 256 * 4;
        size_t sk = 224 / tiles4;
        if (sk * 4 > kt) sk = kt / 4;
        if (sk >= 2 && 16384 + tiles4 * sk * slab4 <= g.ws_bytes) {
            best_nf = 4;
            best_sk = (int)sk;
        }
    }
    g.force_splitk = best_sk;
    return 10 * rf + best_nf;
}

// Weight gradients (both operands reduction-major). Rounds 1-4 kept them all on the 128 x 128 loop: the quadrant schedule
// tied at K = 3456 and lost at K = 864 -- measured while the compiler drained vmcnt in front of every transposing read
// (sei_common.h, dma16_*). Re-measured with the inline-asm DMA (tools/exp_dw_pq.py, float32 store): 2048 x 8192 x 3456
// 138 -> 124 us, 8192 x 2048 x 3456 137 -> 116 us on 256 x 256 tiles (930-1000 TFLOP/s); K = 864 still ties (639 vs 619 us:
// 13 k-tiles per tile do not amortise one workgroup's unoverlapped epilogue) and 256 x 128 tiles lose everywhere. So: whole
// 256 x 256 tiles that fill the chip and a reduction of at least 2048.
int pq_choose_rr(const NtArgs &g) {
    if (g.force_tile != 0 || g.K < 2048 || g.M % 256 != 0 || g.N % 256 != 0 || !pq_eligible(g, true)) return 0;
    const size_t t4 = (size_t)(g.M / 256) * (size_t)(g.N / 256);
    if (t4 < 192 || (double)t4 / (double)(sei_ceil_div(t4, 256) * 256) < 0.8) return 0;
    return 84;
}

// The weight-gradient GEMMs whose epilogue applies the Adam step stay on the 128 x 128 loop: two workgroups per CU overlap
// one's 26-byte-per-element epilogue with the other's main loop (tools/exp_dw_adam_pq.py: 8192 x 32768 x 864 1288 us =
// 5.5 TB/s against 1537 us on 256 x 256 quadrant tiles, 2048 x 8192 x 3456 178 against 182 us). The quadrant kernel's Adam
// epilogue stays reachable through sei_gemm_bf16nt_dw2_adam_ex (tile 30 / 33) and is held to the loop's results by a test.
// (The two co-resident workgroups run in phase -- both in the main loop, then both in the epilogue. Starting every CU's
// second workgroup late by 7 - 40 us, so that one's epilogue falls under the other's main loop, only ADDS the delay:
// tools/exp_adam_stagger.py, 2048 x 8192 x 3456 184 -> 188 ... 210 us, 8192 x 32768 x 864 1409 -> 1412 ... 1434. The main loop
// of a 128 x 128 tile moves 1.77 MB of operands through the CU's global -> LDS path against the epilogue's 0.43 MB: both
// phases are bound by that same path, so there is nothing to hide one under the other.)
bool pq_adam_auto(const NtArgs &) { return false; }

extern "C" int sei_colsum_bf16(const uint16_t *X, float *out, size_t M, int N, void *stream);     // bf16_support.hip

static int nt_entry(const uint16_t *A, int lda, int a_rmajor, const uint16_t *B, int ldb, int b_rmajor,
                    float *D32, uint16_t *D16, int M, int N, int K, int epilogue, const float *bias,
                    const float *R1, const float *R2, uint16_t *D2_16, int tile, int band,
                    void *stream, unsigned long long *plan, float *colsum = nullptr, void *ws = nullptr,
                    size_t ws_bytes = 0, int splitk = 0);

// The column sums ride in the quadrant kernel's epilogue where the launch takes it unsplit with a bf16 result; any other
// schedule is followed by the column-sum kernel over the result (same quantity, one more launch).
static int nt_entry_colsum(const uint16_t *A, int lda, int a_rmajor, const uint16_t *B, int ldb, int b_rmajor,
                           uint16_t *D16, int M, int N, int K, int epilogue, const float *R1, float *colsum, void *stream,
                           void *ws = nullptr, size_t ws_bytes = 0, int tile = 0, int splitk = 0, int band = 0) {
    unsigned long long plan = 0;
    int rc = nt_entry(A, lda, a_rmajor, B, ldb, b_rmajor, nullptr, D16, M, N, K, epilogue, nullptr, R1, nullptr, nullptr, tile, band,
                      stream, &plan, nullptr, ws, ws_bytes, splitk);
    if (rc != SEI_OK) return rc;
    // (unsplit, or split through slabs: the last arriver's epilogue sees the complete tile)
    const bool rides = (plan >> 48) == 2 && ((plan & 0xFFFF) == 1 || (plan & 0x8000) != 0) && N % 4 == 0 &&
                       (reinterpret_cast<uintptr_t>(colsum) & 15) == 0;
    rc = nt_entry(A, lda, a_rmajor, B, ldb, b_rmajor, nullptr, D16, M, N, K, epilogue, nullptr, R1, nullptr, nullptr, tile, band,
                  stream, nullptr, rides ? colsum : nullptr, ws, ws_bytes, splitk);
    if (rc != SEI_OK || rides) return rc;
    return sei_colsum_bf16(D16, colsum, (size_t)M, N, stream);
}

static int nt_entry(const uint16_t *A, int lda, int a_rmajor, const uint16_t *B, int ldb, int b_rmajor,
                    float *D32, uint16_t *D16, int M, int N, int K, int epilogue, const float *bias,
                    const float *R1, const float *R2, uint16_t *D2_16, int tile, int band,
                    void *stream, unsigned long long *plan, float *colsum, void *ws, size_t ws_bytes, int splitk) {
    SEI_REQUIRE(A && B && (D32 || D16) && M > 0 && N > 0 && K > 0 && splitk >= 0);
    SEI_REQUIRE(K % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0);
    SEI_REQUIRE(lda >= (a_rmajor ? M : K) && ldb >= (b_rmajor ? N : K));
    if (a_rmajor) SEI_REQUIRE(M % 8 == 0);
    if (b_rmajor) SEI_REQUIRE(N % 8 == 0);
    SEI_REQUIRE((((uintptr_t)A | (uintptr_t)B) & 15) == 0);
    SEI_REQUIRE(epilogue == SEI_EPI_NONE || epilogue == SEI_EPI_BIAS || epilogue == SEI_EPI_BIAS_GELU ||
                epilogue == SEI_EPI_BIAS_RES || epilogue == SEI_EPI_MUL_DGELU || epilogue == SEI_EPI_ACCUM ||
                epilogue == SEI_EPI_BIAS_ROWSCALE || epilogue == SEI_EPI_BIAS_SCALE_RES);
    if (epilogue == SEI_EPI_BIAS || epilogue == SEI_EPI_BIAS_GELU || epilogue == SEI_EPI_BIAS_RES ||
        epilogue == SEI_EPI_BIAS_ROWSCALE || epilogue == SEI_EPI_BIAS_SCALE_RES)
        SEI_REQUIRE(bias);
    if (epilogue == SEI_EPI_BIAS_ROWSCALE) SEI_REQUIRE(R1);
    if (epilogue == SEI_EPI_BIAS_SCALE_RES) SEI_REQUIRE(R1 && R2 && D32 && !a_rmajor && !b_rmajor);
    if (epilogue == SEI_EPI_BIAS_GELU) SEI_REQUIRE(D2_16);
    if (epilogue == SEI_EPI_BIAS_RES || epilogue == SEI_EPI_MUL_DGELU) SEI_REQUIRE(R1);
    if (epilogue == SEI_EPI_ACCUM) SEI_REQUIRE(D32 && !D16);
    NtArgs g; g.plan = nullptr; g.colsum = nullptr; g.unpad_H = 0; g.unpad_W = 0; g.unpad_act = 0;
    g.ws_slab = nullptr; g.ws_cnt = nullptr; g.ws = nullptr; g.ws_bytes = 0; g.force_splitk = 0;
    g.A = A; g.B = B; g.D32 = D32; g.D16 = D16; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb;
    g.epilogue = epilogue; g.bias = bias; g.R1 = R1; g.R2 = R2; g.D2_16 = D2_16;
    g.A2 = A; g.B2 = B; g.k_seg = K;                       // one reduction segment
    g.conv_cin = 0;
    g.batch = 1;
    g.plan = plan;
    g.colsum = colsum;
    g.ws = ws; g.ws_bytes = ws ? ws_bytes : 0; g.force_splitk = splitk;
    SEI_REQUIRE(tile >= 0 && band >= 0);
    if (epilogue == SEI_EPI_BIAS_SCALE_RES && tile == 0)                // the quadrant kernel has no such epilogue
        tile = (N > 128 && N <= 192) ? 6 : 1;
#ifdef SEI_TUNING
    if (tile == 0) tile = g_tuning_tile;
    if (band == 0) band = g_tuning_band;
#endif
    g.force_tile = tile;
    g.force_band = band;
    hipStream_t s = (hipStream_t)stream;
    const bool would_split = (epilogue == SEI_EPI_NONE || epilogue == SEI_EPI_BIAS || epilogue == SEI_EPI_BIAS_RES ||
                              epilogue == SEI_EPI_ACCUM) && D32 && !D16;
    if (tile >= 30 && tile < 40 && a_rmajor && b_rmajor && pq_eligible(g, true)) {
        if (tile == 30) return launch_pq<8, 4, true, true>(g, s);
        if (tile == 33) return launch_pq<8, 2, true, true>(g, s);
    }
    if (tile >= 30 && tile < 40 && !a_rmajor && pq_eligible(g)) {   // quadrant schedule
        if (b_rmajor) {
            switch (tile) {
                case 30: return launch_pq<8, 4, false, true>(g, s);
                case 31: return launch_pq<9, 4, false, true>(g, s);
                case 32: return launch_pq<9, 2, false, true>(g, s);
                case 33: return launch_pq<8, 2, false, true>(g, s);
                case 38: return launch_pq<9, 2, false, true, 0, 2>(g, s);      // 288 x 128 on TWO LDS stages (round 4's)
                case 39: return launch_pq<8, 2, false, true, 0, 2>(g, s);
                default: break;
            }
        } else {
            switch (tile) {
                case 30: return launch_pq<8, 4>(g, s);
                case 31: return launch_pq<9, 4>(g, s);
                case 32: return launch_pq<9, 2>(g, s);
                case 33: return launch_pq<8, 2>(g, s);
                case 38: return launch_pq<9, 2, false, false, 0, 2>(g, s);
                case 39: return launch_pq<8, 2, false, false, 0, 2>(g, s);
#ifdef SEI_TUNING
                case 34: return launch_pq<8, 4, false, false, 2>(g, s);      // ablations: timing only, wrong results
                case 35: return launch_pq<8, 4, false, false, 7>(g, s);
#endif
                default: break;
            }
        }
    }
#ifdef SEI_TUNING
    if (tile == 20) {                            // 256 x 256 ping-pong schedule (tuning aid)
        if (a_rmajor && b_rmajor) return launch_pp<true, true>(g, s);
        if (a_rmajor) return launch_pp<true, false>(g, s);
        if (b_rmajor) return launch_pp<false, true>(g, s);
        return launch_pp<false, false>(g, s);
    }
    if (!a_rmajor && !b_rmajor) {
        switch (tile) {                          // ring-pipelined candidates (tuning aid)
            case 11: return launch_nt<2, 1, 2, 4, false, false, 4>(g, s);      // 128 x 128, 4 stages
            case 12: return launch_nt<4, 1, 2, 4, false, false, 3>(g, s);      // 256 x 128, 3 stages
            case 13: return launch_nt<2, 2, 2, 4, false, false, 3>(g, s);      // 128 x 256, 3 stages
            case 14: return launch_nt<2, 1, 2, 4, false, false, 3>(g, s);      // 128 x 128, 3 stages
            default: break;
        }
    }
#endif
    if (tile == 15 && !a_rmajor && !b_rmajor) return launch_nt<2, 1, 2, 4, false, false, 1>(g, s);   // 1 stage
    if (tile == 16 && a_rmajor && b_rmajor) return launch_nt<2, 2, 2, 4, true, true>(g, s);   // 128 x 256
    if (tile == 17 && a_rmajor && b_rmajor) return launch_nt<2, 2, 2, 4, true, true, 1>(g, s);   // 128 x 256, 1 stage
    if (tile == 15) {
        if (a_rmajor && b_rmajor) return launch_nt<2, 1, 2, 4, true, true, 1>(g, s);
        if (b_rmajor) return launch_nt<2, 1, 2, 4, false, true, 1>(g, s);
    }
    // a reduction-major operand fixes the tile at 128 x 128 (its LDS rows are 256 bytes = 128 columns)
    if (a_rmajor && b_rmajor) {
        switch (pq_choose_rr(g)) {
            case 84: return launch_pq<8, 4, true, true>(g, s);
            case 82: return launch_pq<8, 2, true, true>(g, s);
            default: break;
        }
        // short reductions (the bottleneck weight gradient: K = 864 over 16,384 tiles) are prologue + epilogue
        // bound: one LDS stage and three workgroups per CU (1030 -> 800 us); long ones want the double buffer
        if (K <= 1024 && tile != 1) return launch_nt<2, 1, 2, 4, true, true, 1>(g, s);
        return launch_nt<2, 1, 2, 4, true, true>(g, s);
    }
    if (a_rmajor) return launch_nt<2, 1, 2, 4, true, false>(g, s);
    // (slabs serve every epilogue: also the convolution behind the downsampler, float32 out with its bias scaled per row)
    const int slab_code = pq_choose_slabs(g, would_split || (epilogue == SEI_EPI_BIAS_ROWSCALE && D32 && !D16));
    if (b_rmajor) {
        switch (slab_code ? slab_code : pq_choose(g, would_split)) {
            case 94: return launch_pq<9, 4, false, true>(g, s);
            case 92: return launch_pq<9, 2, false, true>(g, s);
            case 84: return launch_pq<8, 4, false, true>(g, s);
            case 82: return launch_pq<8, 2, false, true>(g, s);
            default: break;
        }
        // skinny data gradients (the bottleneck levels): 192 x 256 with two 128-column images of the weight -- but not
        // the float32 ones that split K (Downsample's convolution between the two deepest levels, 288 / 576 x 2048 x 8192:
        // 65 / 75 us there against 40 / 55 on 128 x 128 tiles with 8 / 6 splits, tools/exp_skinny.py)
        // (M <= 144 against a 32768-wide weight -- the reference's default batch of 8: 128 x 128 tiles, 256 of them, stream it
        // faster than 128 tiles of 192 x 256: 189 / 157 us against 259 / 243 at 144 / 72 rows, tools/exp_b8.py)
        if (tile != 1 && N >= 2048 && K >= 2048 && M <= 768 && !would_split && !(M <= 144 && N >= 16384))
            return launch_nt<3, 2, 2, 4, false, true>(g, s);
        return launch_nt<2, 1, 2, 4, false, true>(g, s);
    }
    switch (tile) {
        case 6: return launch_nt<1, 3, 4, 2>(g, s);      // 128 x 192 (SwinIR: N = 180 / 360 / 576)
        case 1: return launch_nt<2, 1, 2, 4>(g, s);      // 128 x 128
        case 2: return launch_nt<2, 2, 2, 4>(g, s);      // 128 x 256
        case 3: return launch_nt<3, 2, 2, 4>(g, s);      // 192 x 256
        case 5: return launch_nt<3, 1, 1, 8>(g, s);      //  96 x 256
#ifdef SEI_TUNING
        case 4: return launch_nt<4, 2, 2, 4>(g, s);      // 256 x 256 (tuning aid)
#endif
        default: break;
    }
    // Tile choice (measured on MI355X with the band-major tile order, tools/exp_tiles576.py): 128x128 runs two
    // workgroups per CU, which hides each block's load latency and epilogue behind the other's MFMAs, and is
    // the best or within a few % of the best wherever there are many row tiles (M = 1152 / 2304 x 8192 x 2048:
    // 110 vs 144 us for 192x256). Skinny outputs (the 8192-channel bottleneck: M = 288 or 576 rows against
    // N = 8192 / 32768) are weight-streaming: there 192x256 wins (M = 288: 221 vs 309 us), padding included,
    // and 96x256 never does. A 256x256 tile on this loop needs the rolled epilogue of gemm_bf16pp.h.
    switch (slab_code ? slab_code : pq_choose(g, would_split)) {
        case 94: return launch_pq<9, 4>(g, s);
        case 92: return launch_pq<9, 2>(g, s);
        case 84: return launch_pq<8, 4>(g, s);
        case 82: return launch_pq<8, 2>(g, s);
        default: break;
    }
    // (not the convolution behind the downsampler, which cannot split K: 288 / 576 x 8192 x 2048 run 56 / 48 us on 64
    // tiles of 192 x 256 and 32 / 41 us on 128 x 128, tools/exp_skinny.py)
    // (and not M <= 144 against 32768 columns: 197 / 156 us on 128 x 128 against 236 / 210, tools/exp_b8.py)
    if (N >= 2048 && K >= 2048 && M <= 768 && epilogue != SEI_EPI_BIAS_ROWSCALE && !(M <= 144 && N >= 16384))
        return launch_nt<3, 2, 2, 4>(g, s);                                              // 192 x 256
    // short reductions are all prologue and epilogue: one LDS stage (32 KB) and <= 84 VGPRs put three workgroups
    // on a CU instead of two (36864 x 512 x 128: 34 -> 27 us; 9216 x 2048 x 512: 50 -> 44 us; loses from K ~ 2048)
    if (K <= 1024 && tile != 1) return launch_nt<2, 1, 2, 4, false, false, 1>(g, s);
    return launch_nt<2, 1, 2, 4>(g, s);                                                  // 128 x 128
}

extern "C" int sei_gemm_bf16nt_ex(const uint16_t *A, int lda, int a_rmajor, const uint16_t *B, int ldb, int b_rmajor,
                                  float *D32, uint16_t *D16, int M, int N, int K, int epilogue, const float *bias,
                                  const float *R1, const float *R2, uint16_t *D2_16, int tile, int band,
                                  void *stream) {
    return nt_entry(A, lda, a_rmajor, B, ldb, b_rmajor, D32, D16, M, N, K, epilogue, bias, R1, R2, D2_16, tile, band,
                    stream, nullptr);
}

extern "C" int sei_gemm_bf16nt(const uint16_t *A, int lda, int a_rmajor, const uint16_t *B, int ldb, int b_rmajor,
                               float *D32, uint16_t *D16, int M, int N, int K, int epilogue, const float *bias,
                               const float *R1, const float *R2, uint16_t *D2_16, void *stream) {
    return nt_entry(A, lda, a_rmajor, B, ldb, b_rmajor, D32, D16, M, N, K, epilogue, bias, R1, R2, D2_16, 0, 0, stream,
                    nullptr);
}

// D16 = (A op(B)) gelu'(R1) rounded to bf16 (SEI_EPI_MUL_DGELU) or the plain product (SEI_EPI_NONE), and colsum[n] += the
// column sums of the rounded result: the data gradient gh3 of a ConvBlock's conv3 together with conv2's bias gradient.
extern "C" int sei_gemm_bf16nt_colsum(const uint16_t *A, int lda, int a_rmajor, const uint16_t *B, int ldb, int b_rmajor,
                                      uint16_t *D16, int M, int N, int K, int epilogue, const float *R1, float *colsum,
                                      void *stream) {
    SEI_REQUIRE(D16 && colsum && (epilogue == SEI_EPI_MUL_DGELU || epilogue == SEI_EPI_NONE));
    return nt_entry_colsum(A, lda, a_rmajor, B, ldb, b_rmajor, D16, M, N, K, epilogue, R1, colsum, stream);
}

// sei_gemm_bf16nt / sei_gemm_bf16nt_colsum with a split-K workspace (NtArgs::ws_slab): `ws` = ws_bytes of device memory,
// 256-byte aligned, whose first 16 KiB are ZERO before the first call and are left zero by every call (tile counters; the
// rest needs no initialisation), used by ONE stream at a time. Launches of the quadrant kernel that split K then meet in
// slabs instead of float atomics, and launches whose epilogue the atomics could not serve (bf16 / GELU / GELU' results,
// riding column sums) may split too. ws = nullptr: exactly sei_gemm_bf16nt_ex / sei_gemm_bf16nt_colsum. colsum != nullptr
// asks for sei_gemm_bf16nt_colsum's semantics (D16 only, SEI_EPI_MUL_DGELU or SEI_EPI_NONE). tile / band as sei_gemm_bf16nt_ex;
// splitk > 0 asks for that many K slices where the schedule can split at all (experiments, tests), 0 = automatic.
extern "C" int sei_gemm_bf16nt_ws(const uint16_t *A, int lda, int a_rmajor, const uint16_t *B, int ldb, int b_rmajor,
                                  float *D32, uint16_t *D16, int M, int N, int K, int epilogue, const float *bias,
                                  const float *R1, const float *R2, uint16_t *D2_16, float *colsum, void *ws,
                                  size_t ws_bytes, int tile, int band, int splitk, void *stream) {
    SEI_REQUIRE(tile >= 0 && band >= 0 && splitk >= 0 && (ws == nullptr || ws_bytes >= 16384));
    if (colsum) {
        SEI_REQUIRE(D16 && !D32 && (epilogue == SEI_EPI_MUL_DGELU || epilogue == SEI_EPI_NONE));
        return nt_entry_colsum(A, lda, a_rmajor, B, ldb, b_rmajor, D16, M, N, K, epilogue, R1, colsum, stream, ws, ws_bytes,
                               tile, splitk, band);
    }
    return nt_entry(A, lda, a_rmajor, B, ldb, b_rmajor, D32, D16, M, N, K, epilogue, bias, R1, R2, D2_16, tile, band, stream,
                    nullptr, nullptr, ws, ws_bytes, splitk);
}

// The schedule sei_gemm_bf16nt would take for these shapes, launching nothing (host arithmetic only, no GPU needed):
// (family << 48) | (tile rows << 32) | (tile columns << 16) | K splits, family 1 = the 128 x 128 loop's kernel
// (gemm_bf16nt_kernel, whatever its tile), 2 = the quadrant schedule (gemm_bf16pq_kernel); 0 = arguments the entry point
// would refuse. Operands are taken as densely packed and 16-byte aligned, which is what models/_ops.py passes.
static size_t nt_plan(int a_rmajor, int b_rmajor, int out_f32, int out_bf16, int M, int N, int K, int epilogue,
                      size_t ws_bytes);
extern "C" size_t sei_gemm_bf16nt_plan(int a_rmajor, int b_rmajor, int out_f32, int out_bf16, int M, int N, int K,
                                       int epilogue) {
    return nt_plan(a_rmajor, b_rmajor, out_f32, out_bf16, M, N, K, epilogue, 0);
}
// ... and the schedule sei_gemm_bf16nt_ws would take with a workspace of ws_bytes (bit 15 of the split count set: the K
// slices meet in slabs).
extern "C" size_t sei_gemm_bf16nt_plan_ws(int a_rmajor, int b_rmajor, int out_f32, int out_bf16, int M, int N, int K,
                                          int epilogue, size_t ws_bytes) {
    return nt_plan(a_rmajor, b_rmajor, out_f32, out_bf16, M, N, K, epilogue, ws_bytes);
}
static size_t nt_plan(int a_rmajor, int b_rmajor, int out_f32, int out_bf16, int M, int N, int K, int epilogue,
                      size_t ws_bytes) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    const uint16_t *fake16 = reinterpret_cast<const uint16_t *>(uintptr_t(1) << 20);      // never dereferenced
    float *fake32 = reinterpret_cast<float *>(uintptr_t(1) << 20);
    const bool with_bias = epilogue == SEI_EPI_BIAS || epilogue == SEI_EPI_BIAS_GELU || epilogue == SEI_EPI_BIAS_RES ||
                           epilogue == SEI_EPI_BIAS_ROWSCALE || epilogue == SEI_EPI_BIAS_SCALE_RES;
    const bool with_r1 = epilogue == SEI_EPI_BIAS_RES || epilogue == SEI_EPI_MUL_DGELU ||
                         epilogue == SEI_EPI_BIAS_ROWSCALE || epilogue == SEI_EPI_BIAS_SCALE_RES;
    unsigned long long plan = 0;
    const int rc = nt_entry(fake16, a_rmajor ? M : K, a_rmajor, fake16, b_rmajor ? N : K, b_rmajor,
                            out_f32 ? fake32 : nullptr, out_bf16 ? const_cast<uint16_t *>(fake16) : nullptr, M, N, K,
                            epilogue, with_bias ? fake32 : nullptr, with_r1 ? fake32 : nullptr,
                            epilogue == SEI_EPI_BIAS_SCALE_RES ? fake32 : nullptr,
                            epilogue == SEI_EPI_BIAS_GELU ? const_cast<uint16_t *>(fake16) : nullptr, 0, 0, nullptr, &plan,
                            nullptr, ws_bytes ? fake32 : nullptr, ws_bytes, 0);
    return rc == SEI_OK ? (size_t)plan : 0;
}

extern "C" int sei_gemm_bf16nt_dw2_ex(const uint16_t *A1, const uint16_t *A2, int lda, const uint16_t *B1,
                                      const uint16_t *B2, int ldb, float *D32, int M, int N, int K1, int K2,
                                      int accumulate, int tile, void *stream) {
    SEI_REQUIRE(A1 && A2 && B1 && B2 && D32 && M > 0 && N > 0 && K1 > 0 && K2 > 0);
    SEI_REQUIRE(M % 8 == 0 && N % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0 && lda >= M && ldb >= N);
    SEI_REQUIRE((K1 + K2) % 8 == 0);
    SEI_REQUIRE((((uintptr_t)A1 | (uintptr_t)A2 | (uintptr_t)B1 | (uintptr_t)B2) & 15) == 0);
    NtArgs g; g.plan = nullptr; g.colsum = nullptr; g.unpad_H = 0; g.unpad_W = 0; g.unpad_act = 0;
    g.ws_slab = nullptr; g.ws_cnt = nullptr; g.ws = nullptr; g.ws_bytes = 0; g.force_splitk = 0;
    g.A = A1; g.B = B1; g.A2 = A2; g.B2 = B2; g.k_seg = K1;
    g.D32 = D32; g.D16 = nullptr; g.M = M; g.N = N; g.K = K1 + K2; g.lda = lda; g.ldb = ldb;
    g.epilogue = accumulate ? SEI_EPI_ACCUM : SEI_EPI_NONE;
    g.bias = nullptr; g.R1 = nullptr; g.R2 = nullptr; g.D2_16 = nullptr;
    g.conv_cin = 0;
    g.batch = 1;
    SEI_REQUIRE(tile >= 0);
    g.force_band = 0;
#ifdef SEI_TUNING
    if (tile == 0) tile = g_tuning_tile;
    g.force_band = g_tuning_band;
#endif
    g.force_tile = tile;
#ifdef SEI_TUNING
    if (tile == 36) return launch_pq<8, 4, true, true, 2>(g, (hipStream_t)stream);   // ablations: timing only
    if (tile == 37) return launch_pq<8, 4, true, true, 7>(g, (hipStream_t)stream);
#endif
    if (tile == 30 && pq_eligible(g, true)) return launch_pq<8, 4, true, true>(g, (hipStream_t)stream);   // tuning aid
    if (tile == 33 && pq_eligible(g, true)) return launch_pq<8, 2, true, true>(g, (hipStream_t)stream);
    switch (pq_choose_rr(g)) {
        case 84: return launch_pq<8, 4, true, true>(g, (hipStream_t)stream);
        case 82: return launch_pq<8, 2, true, true>(g, (hipStream_t)stream);
        default: break;
    }
    if (K1 + K2 <= 1024) return launch_nt<2, 1, 2, 4, true, true, 1>(g, (hipStream_t)stream);   // as sei_gemm_bf16nt
    return launch_nt<2, 1, 2, 4, true, true>(g, (hipStream_t)stream);
}

extern "C" int sei_gemm_bf16nt_dw2_adam_ex(const uint16_t *A1, const uint16_t *A2, int lda, const uint16_t *B1,
                                           const uint16_t *B2, int ldb, float *param, float *exp_avg, float *exp_avg_sq,
                                           uint16_t *param_bf16, const float *hyper, int M, int N, int K1, int K2,
                                           int tile, void *stream) {
    SEI_REQUIRE(A1 && A2 && B1 && B2 && param && exp_avg && exp_avg_sq && hyper && M > 0 && N > 0 && K1 > 0 && K2 > 0);
    SEI_REQUIRE(M % 8 == 0 && N % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0 && lda >= M && ldb >= N);
    SEI_REQUIRE((K1 + K2) % 8 == 0);
    SEI_REQUIRE((((uintptr_t)A1 | (uintptr_t)A2 | (uintptr_t)B1 | (uintptr_t)B2) & 15) == 0);
    SEI_REQUIRE((((uintptr_t)param | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) == 0 &&
                ((uintptr_t)param_bf16 & 7) == 0);              // the epilogue moves whole quads
    NtArgs g; g.plan = nullptr; g.colsum = nullptr; g.unpad_H = 0; g.unpad_W = 0; g.unpad_act = 0;
    g.ws_slab = nullptr; g.ws_cnt = nullptr; g.ws = nullptr; g.ws_bytes = 0; g.force_splitk = 0;
    g.A = A1; g.B = B1; g.A2 = A2; g.B2 = B2; g.k_seg = K1;
    g.D32 = param; g.D16 = nullptr; g.M = M; g.N = N; g.K = K1 + K2; g.lda = lda; g.ldb = ldb;
    g.epilogue = SEI_EPI_NONE;
    g.bias = nullptr; g.R1 = nullptr; g.R2 = nullptr; g.D2_16 = nullptr;
    g.conv_cin = 0;
    g.batch = 1;
    g.force_tile = 0; g.force_band = 0;
    g.adam_p = param; g.adam_m = exp_avg; g.adam_v = exp_avg_sq; g.adam_p16 = param_bf16; g.adam_h = hyper;
    SEI_REQUIRE(tile == 0 || tile == 1 || tile == 30 || tile == 33);
    // tile 30 / 33: the quadrant schedule's 256 x 256 / 256 x 128 tiles with the Adam epilogue (round 5): half the operand
    // bytes per output element of the 128 x 128 loop
    if ((tile == 30 || tile == 33 || (tile == 0 && pq_adam_auto(g))) && pq_eligible(g, true)) {
        if (tile == 33) return launch_pq<8, 2, true, true, 0, 2, true>(g, (hipStream_t)stream);
        return launch_pq<8, 4, true, true, 0, 2, true>(g, (hipStream_t)stream);
    }
    // (the double-buffered loop: with the Adam epilogue's 112 registers only two workgroups fit a CU, which is what
    // the one-stage loop was meant to beat with three; measured at K = 864: 1416-1421 us against 1426-1459)
    return launch_nt<2, 1, 2, 4, true, true, 2, 1>(g, (hipStream_t)stream);
}

extern "C" int sei_gemm_bf16nt_dw2_adam(const uint16_t *A1, const uint16_t *A2, int lda, const uint16_t *B1,
                                        const uint16_t *B2, int ldb, float *param, float *exp_avg, float *exp_avg_sq,
                                        uint16_t *param_bf16, const float *hyper, int M, int N, int K1, int K2,
                                        void *stream) {
    return sei_gemm_bf16nt_dw2_adam_ex(A1, A2, lda, B1, B2, ldb, param, exp_avg, exp_avg_sq, param_bf16, hyper, M, N, K1, K2,
                                       0, stream);
}

extern "C" int sei_gemm_bf16nt_dw2_bf16out(const uint16_t *A1, const uint16_t *A2, int lda, const uint16_t *B1,
                                           const uint16_t *B2, int ldb, uint16_t *D16, int M, int N, int K1, int K2,
                                           void *stream) {
    SEI_REQUIRE(A1 && A2 && B1 && B2 && D16 && M > 0 && N > 0 && K1 > 0 && K2 > 0);
    SEI_REQUIRE(M % 8 == 0 && N % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0 && lda >= M && ldb >= N);
    SEI_REQUIRE((K1 + K2) % 8 == 0);
    SEI_REQUIRE((((uintptr_t)A1 | (uintptr_t)A2 | (uintptr_t)B1 | (uintptr_t)B2) & 15) == 0 && ((uintptr_t)D16 & 7) == 0);
    NtArgs g; g.plan = nullptr; g.colsum = nullptr; g.unpad_H = 0; g.unpad_W = 0; g.unpad_act = 0;
    g.ws_slab = nullptr; g.ws_cnt = nullptr; g.ws = nullptr; g.ws_bytes = 0; g.force_splitk = 0;
    g.A = A1; g.B = B1; g.A2 = A2; g.B2 = B2; g.k_seg = K1;
    g.D32 = nullptr; g.D16 = D16; g.M = M; g.N = N; g.K = K1 + K2; g.lda = lda; g.ldb = ldb;
    g.epilogue = SEI_EPI_NONE;
    g.bias = nullptr; g.R1 = nullptr; g.R2 = nullptr; g.D2_16 = nullptr;
    g.conv_cin = 0;
    g.batch = 1;
    g.force_tile = 0; g.force_band = 0;
    g.adam_p = nullptr; g.adam_m = nullptr; g.adam_v = nullptr; g.adam_p16 = nullptr; g.adam_h = nullptr;
    if (K1 + K2 <= 1024) return launch_nt<2, 1, 2, 4, true, true, 1, 2>(g, (hipStream_t)stream);
    return launch_nt<2, 1, 2, 4, true, true, 2, 2>(g, (hipStream_t)stream);
}

extern "C" int sei_gemm_bf16nt_dw2_taps(const uint16_t *A1, const uint16_t *A2, int lda, const uint16_t *B1,
                                        const uint16_t *B2, int ldb, float *D32, int M, int N, int K1, int K2,
                                        int accumulate, int ntaps, const int *tap_rows, long long tap_ld, void *stream) {
    SEI_REQUIRE(A1 && A2 && B1 && B2 && D32 && tap_rows && M > 0 && N > 0 && K1 > 0 && K2 >= 0);
    SEI_REQUIRE(ntaps >= 1 && ntaps <= 9 && tap_ld >= (long long)M * N);
    SEI_REQUIRE(M % 8 == 0 && N % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0 && lda >= M && ldb >= N);
    SEI_REQUIRE((K1 + K2) % 8 == 0);
    SEI_REQUIRE((((uintptr_t)A1 | (uintptr_t)A2 | (uintptr_t)B1 | (uintptr_t)B2) & 15) == 0);
    NtArgs g; g.plan = nullptr; g.colsum = nullptr; g.unpad_H = 0; g.unpad_W = 0; g.unpad_act = 0;
    g.ws_slab = nullptr; g.ws_cnt = nullptr; g.ws = nullptr; g.ws_bytes = 0; g.force_splitk = 0;
    g.A = A1; g.B = B1; g.A2 = A2; g.B2 = B2; g.k_seg = K1;
    g.D32 = D32; g.D16 = nullptr; g.M = M; g.N = N; g.K = K1 + K2; g.lda = lda; g.ldb = ldb;
    g.epilogue = accumulate ? SEI_EPI_ACCUM : SEI_EPI_NONE;
    g.bias = nullptr; g.R1 = nullptr; g.R2 = nullptr; g.D2_16 = nullptr;
    g.conv_cin = 0;
    g.force_tile = 0; g.force_band = 0;
    g.batch = ntaps;
    for (int t = 0; t < 9; ++t) g.batch_row[t] = t < ntaps ? tap_rows[t] : 0;
    g.batch_ld = tap_ld;
    return launch_nt<2, 1, 2, 4, true, true, 2, 3>(g, (hipStream_t)stream);
}

extern "C" int sei_gemm_bf16nt_dw2(const uint16_t *A1, const uint16_t *A2, int lda, const uint16_t *B1,
                                   const uint16_t *B2, int ldb, float *D32, int M, int N, int K1, int K2,
                                   int accumulate, void *stream) {
    return sei_gemm_bf16nt_dw2_ex(A1, A2, lda, B1, B2, ldb, D32, M, N, K1, K2, accumulate, 0, stream);
}

// 3x3 convolution (stride 1, zero padding 1) of an NHWC batch as ONE implicit GEMM on the zero-bordered grid of
// sei_pad_nhwc_bf16: output row r (a pixel of the padded grid) = sum over taps t and channels c of
// Ap[r + row_off[t]][c] * B[n][t * cin_pad + c]. The im2col matrix exists only as LDS tiles: each k-tile of the
// LDS-DMA stream is the 64-channel slice of one tap, fetched from the SAME array at that tap's row shift.
extern "C" int sei_gemm_bf16nt_conv(const uint16_t *Ap, int cin_pad, const int *row_off9, const uint16_t *B, int ldb,
                                    float *D32, uint16_t *D16, int M, int N, int epilogue, const float *bias,
                                    void *stream) {
    SEI_REQUIRE(Ap && row_off9 && B && (D32 || D16) && M > 0 && N > 0 && cin_pad > 0 && cin_pad % BK == 0);
    SEI_REQUIRE(ldb >= 9 * cin_pad && ldb % 8 == 0 && (((uintptr_t)Ap | (uintptr_t)B) & 15) == 0);
    SEI_REQUIRE(epilogue == SEI_EPI_NONE || epilogue == SEI_EPI_BIAS);
    if (epilogue == SEI_EPI_BIAS) SEI_REQUIRE(bias);
    NtArgs g; g.plan = nullptr; g.colsum = nullptr; g.unpad_H = 0; g.unpad_W = 0; g.unpad_act = 0;
    g.ws_slab = nullptr; g.ws_cnt = nullptr; g.ws = nullptr; g.ws_bytes = 0; g.force_splitk = 0;
    g.batch = 1;
    g.A = Ap; g.B = B; g.D32 = D32; g.D16 = D16; g.M = M; g.N = N; g.K = 9 * cin_pad; g.lda = cin_pad; g.ldb = ldb;
    g.epilogue = epilogue; g.bias = bias; g.R1 = nullptr; g.R2 = nullptr; g.D2_16 = nullptr;
    g.A2 = Ap; g.B2 = B; g.k_seg = g.K;
    g.force_tile = 0; g.force_band = 0;
    g.conv_cin = cin_pad;
    for (int t = 0; t < 9; ++t) g.conv_off[t] = row_off9[t];
    hipStream_t s = (hipStream_t)stream;
    if (N > 128 && N <= 192) return launch_nt<1, 3, 4, 2>(g, s);               // 128 x 192: one column tile
    return launch_nt<2, 1, 2, 4>(g, s);
}

// The same convolution with sei_unpad_nhwc in its epilogue: y (Bimg, H, W, N) = [LeakyReLU](conv + bias) [+ res], the
// border rows of the grid are computed and dropped, the (R, N) intermediate never exists.
extern "C" int sei_gemm_bf16nt_conv_unpad(const uint16_t *Ap, int cin_pad, const int *row_off9, const uint16_t *B, int ldb,
                                          float *y, const float *res, int Bimg, int H, int W, int N, const float *bias,
                                          int act, void *stream) {
    SEI_REQUIRE(Ap && row_off9 && B && y && Bimg > 0 && H > 0 && W > 0 && N > 0 && N % 4 == 0 && cin_pad > 0 && cin_pad % BK == 0);
    SEI_REQUIRE(ldb >= 9 * cin_pad && ldb % 8 == 0 && (((uintptr_t)Ap | (uintptr_t)B) & 15) == 0);
    SEI_REQUIRE((((uintptr_t)y | (uintptr_t)res) & 15) == 0 && (act == 0 || act == 1));
    SEI_REQUIRE((size_t)Bimg * (H + 2) * (W + 2) < ((size_t)1 << 31));
    NtArgs g;
    g.plan = nullptr;
    g.colsum = nullptr;
    g.ws_slab = nullptr; g.ws_cnt = nullptr; g.ws = nullptr; g.ws_bytes = 0; g.force_splitk = 0;
    g.batch = 1;
    g.A = Ap; g.B = B; g.D32 = y; g.D16 = nullptr; g.M = Bimg * (H + 2) * (W + 2); g.N = N; g.K = 9 * cin_pad;
    g.lda = cin_pad; g.ldb = ldb;
    g.epilogue = res ? SEI_EPI_BIAS_RES : (bias ? SEI_EPI_BIAS : SEI_EPI_NONE);
    g.bias = bias; g.R1 = res; g.R2 = nullptr; g.D2_16 = nullptr;
    if (res && !bias) return SEI_ERR_BAD_ARG;               // (every residual convolution of the backbones has a bias)
    g.A2 = Ap; g.B2 = B; g.k_seg = g.K;
    g.force_tile = 0; g.force_band = 0;
    g.conv_cin = cin_pad;
    for (int t = 0; t < 9; ++t) g.conv_off[t] = row_off9[t];
    g.unpad_H = H; g.unpad_W = W; g.unpad_act = act;
    hipStream_t s = (hipStream_t)stream;
    if (N > 128 && N <= 192) return launch_nt<1, 3, 4, 2>(g, s);
    return launch_nt<2, 1, 2, 4>(g, s);
}
