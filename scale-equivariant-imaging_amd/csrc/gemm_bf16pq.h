// gemm_bf16pq_kernel: the "quadrant" schedule of the bf16 GEMM for the deep U-Net levels (included by
// gemm_bf16nt.hip). One workgroup per CU, 8 waves = 2 (rows) x 4 (columns), 16x16x32 MFMAs, tile
// (32 RF) x (64 NF) x 64:  RF = 9 -> 288 rows: every row count of this model is 288 * 2^j (9 * 2^j pixels x 32
// images), so 2304-, 576- and 288-row GEMMs cut into whole tiles and a 2304 x 8192 output is exactly 256 tiles;
// RF = 8 -> 256 rows for everything else. NF = 4 -> 256 columns, NF = 2 -> 128 (twice the tiles for short M).
//
// What differs from the 128 x 128 loop and from gemm_bf16pp_kernel (measured on 4096^3: removing the LDS-DMA
// issue from the ping-pong loop gave +22 %, removing the fragment reads +1 %, removing the barriers nothing):
//   * DMA issue costs instructions, not bandwidth. A piece is one global_load_lds whose per-lane byte offset is
//     fixed for the whole launch (row clamped to the matrix, chunk pre-swizzled) and whose base is a uniform
//     pointer that advances 128 B per k-tile: no 64-bit lane arithmetic and no bounds selects in the loop.
//     (K ranges are multiples of 64 here; the host falls back to the 128 x 128 kernel otherwise.)
//   * Half-tiles are cut by QUADRANT, not by wave group: A(q) holds the rows every wave needs for its row
//     quadrant q, B(q) the columns for column quadrant q. A half-tile is therefore read in ONE phase and free
//     again right after it, which lets the DMA run 5-6 phases (1.5 k-tiles) ahead inside two stages of LDS:
//         phase 1: read B(0), A(0)   MFMA (0,0)   issue A(1) of tile t+1
//         phase 2: read B(1)         MFMA (0,1)   issue B(0) of tile t+2
//         phase 3: read A(1)         MFMA (1,1)   issue A(0) of tile t+2
//         phase 4: --                MFMA (1,0)   issue B(1) of tile t+2;  counted vmcnt: tile t+1 landed
//     each phase = [reads + DMA issue] s_barrier [lgkmcnt(0); MFMAs] s_barrier; the wr = 1 waves run one barrier
//     behind, so their read segment coincides with the other group's MFMA segment.
//   * The epilogue goes through a wave-private LDS patch (16 rows x 64 NF/4 columns): accumulators in, whole
//     rows out, so every auxiliary load, store and atomic is 16 B per lane on full 128/256-B row segments
//     (a 16x16 accumulator stored as it stands is 32-B bf16 segments).
//
// Hazards, by count (segments: group 0 reads in segment 2p and computes in 2p+1, group 1 reads in 2p+1 and
// computes in 2p+2; a barrier event separates consecutive segments):
//   WAR  B(0) is re-staged one phase after its reads: those reads are retired BEFORE the first barrier of
//        phase 1 (lgkmcnt counted down to the A reads issued after them). Everything else is re-staged two
//        phases after its reads were retired by lgkmcnt(0).
//   RAW  the vmcnt of phase 4 leaves only the pieces issued in phases 2-4 in flight (a wave with a third A(0)
//        piece counts one more) and sits before the barrier event that ends segment 2*4+1; tile t+1 is first
//        read in the segment after it.
#pragma once

using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int N>
__device__ __forceinline__ void pq_wait_vmcnt() {
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if constexpr (N == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (N == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
    else if constexpr (N == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    else if constexpr (N == 11) asm volatile("s_waitcnt vmcnt(11)" ::: "memory");
    else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if constexpr (N == 13) asm volatile("s_waitcnt vmcnt(13)" ::: "memory");
    else if constexpr (N == 14) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
    else static_assert(N < 0, "add the immediate");
}
template <int N>
__device__ __forceinline__ void pq_wait_lgkmcnt() {
    if constexpr (N == 4) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
    else if constexpr (N == 5) asm volatile("s_waitcnt lgkmcnt(5)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
    else static_assert(N < 0, "add the immediate");
}

template <int RF, int NF>
struct PqGeom {
    static constexpr int QA0 = (RF + 1) / 2, QA1 = RF / 2, QB = NF / 2;     // fragments per quadrant
    static constexpr int BM = 32 * RF, BN = 64 * NF;
    static constexpr int A0_ROWS = 32 * QA0, A1_ROWS = 32 * QA1, B_ROWS = 64 * QB;
    static constexpr int OFF_A0 = 0, OFF_A1 = A0_ROWS * ROW_BYTES, OFF_B0 = OFF_A1 + A1_ROWS * ROW_BYTES,
                         OFF_B1 = OFF_B0 + B_ROWS * ROW_BYTES, STAGE = OFF_B1 + B_ROWS * ROW_BYTES;
    static constexpr int PA0 = A0_ROWS / 8, PA1 = A1_ROWS / 8, PB = B_ROWS / 8;     // 1-KiB pieces per half-tile
    static constexpr int NA0 = (PA0 + 7) / 8, NA1 = PA1 / 8, NB = PB / 8;            // pieces per wave (NA0: max)
    static_assert(PA1 % 8 == 0 && PB % 8 == 0, "uniform piece counts except A(0)");
    static constexpr int EPI_LDW = 16 * NF + 4;                                      // floats per patch row
    static_assert(8 * 16 * EPI_LDW * 4 <= 2 * STAGE, "epilogue patches fit the stages");
};

// 16-B chunk swizzle of a reduction-major half-tile ([64 k-rows][128 or 64 columns]): the fragment reads of 32
// lanes (ds_read_b64_tr_b16: k-rows 8 lg + q, two neighbouring chunks each) fall on 16 different chunk slots of
// the 256-B bank row. 256-B rows: swz_rmajor (gemm_bf16nt.hip); 128-B rows (two per bank row): below.
__device__ __forceinline__ int pq_swz_rm128(int row) { return (((row >> 1) & 1) << 1) | (((row >> 3) & 1) << 2); }

// BRM: B stored reduction-major (element (n, k) at B[k * ldb + n]: the weight as the forward pass stores it, used
// by the data gradient). Its half-tiles are [64 k-rows][columns of the quadrant] images read with
// ds_read_b64_tr_b16; the DMA base then advances 64 rows per k-tile.
// ARM: A stored reduction-major as well (element (m, k) at A[k * lda + m]): the weight gradient dW = dY^T X, both
// operands pixel-major; RF = 8 only. Both reduction-major: any K (zero rows past k_end), two K segments.
// NS = 3: the FREE-RUNNING schedule of the 128-column tiles (round 5). Why: with NF = 2 a quadrant phase carries 10 MFMAs
// per wave (160 cycles at 16 per v_mfma_f32_16x16x32_bf16) against a read segment of 12 ds_read_b128 plus ~1.6 LDS-DMA
// pieces at 60-185 cycles each; the two wave groups alternate read and MFMA segments behind 8 barriers per k-tile, so the
// k-tile takes 8 x max(read segment, MFMA segment) ~ 8 x 250 cycles of READ segments: 1.56 us per k-tile for 0.48 us of
// matrix work (288 x 32768 x 8192: 200 us, and a third LDS stage in the same schedule changed nothing: 207 us,
// tools/exp_ring3.py -- the loop is not short of bytes in flight). A 128-column tile's stage is 52 KB, so THREE fit (156 KB)
// and the fragments of half a k-tile are 44 registers, so TWO sets fit beside the 72 accumulator registers. That allows:
//   * ONE barrier per k-tile: before it every wave waits for its own pieces of tile t (vmcnt counted: the pieces of tile
//     t + 1 stay in flight) and for its last fragment reads of tile t - 1 (lgkmcnt(0): the data is in registers); after
//     it tile t has landed for everyone and the stage of tile t - 1 is free: tile t + 2 is issued into it, two k-tiles
//     ahead of its first read;
//   * per wave and k-tile: read half 0 of tile t into fragment set X, MFMAs on set Y (half 1 of tile t - 1, read before
//     the barrier), read half 1 of tile t into Y, MFMAs on X -- every read has 18 MFMAs (288 cycles) to land under;
//   * the two waves of a SIMD (wr = 0 / 1) issue their LDS-DMA pieces at different points of the iteration (right after
//     the barrier / after the first MFMA block), so one wave's DMA issue sits under the other's MFMAs.
// Tile order, LDS images, swizzles, piece tables, K tails and the epilogue are the quadrant schedule's.
// ADAM: the accumulator is the COMPLETE gradient of its element and the epilogue applies the optimizer step instead of
// storing it (parameter, both moments, bf16 shadow in D's layout; NtArgs::adam_*), as gemm_bf16nt_kernel's ROWEPI = 1.
template <int RF, int NF, bool ARM = false, bool BRM = false, int ABL = 0, int NS = 2, bool ADAM = false>
__global__ __launch_bounds__(NT) void gemm_bf16pq_kernel(NtArgs g) {
    using G = PqGeom<RF, NF>;
    constexpr int QA0 = G::QA0, QA1 = G::QA1, QB = G::QB;
    static_assert(NS == 2 || NS == 3, "two LDS stages (quadrant phases) or three (free-running schedule)");
    static_assert(NS * G::STAGE <= 160 * 1024, "the stages fit one CU's LDS");
    __shared__ __attribute__((aligned(1024))) char smem[NS * G::STAGE];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int l16 = lane & 15, lg = lane >> 4;

    // ---- tile / split assignment (band-major order per XCD, as gemm_bf16nt_kernel) -------------------------
    int bid = blockIdx.x;
    const int per_split = 8 * g.tiles_per_xcd;
    const int zs = bid / per_split;
    bid -= zs * per_split;
    const int ord = (bid & 7) * g.tiles_per_xcd + (bid >> 3);
    if ((bid >> 3) >= g.tiles_per_xcd || ord >= g.tiles_m * g.tiles_n) return;
    int tm_i, tn_i;
    {
        const int band_tiles = g.tiles_m * g.band;
        const int b = ord / band_tiles, r = ord - b * band_tiles;
        const int width = min(g.band, g.tiles_n - b * g.band);
        tm_i = r / width;
        tn_i = b * g.band + (r - tm_i * width);
    }
    const int m0 = tm_i * G::BM, n0 = tn_i * G::BN;
    const int M = g.M, N = g.N;
    const int k_begin = zs * g.k_per_split;
    const int k_end = min(g.K, k_begin + g.k_per_split);
    const int nt = (k_end - k_begin + BK - 1) / BK;           // the last k-tile may be partial (zero-filled)

    f32x4 acc[RF][NF];
#pragma unroll
    for (int i = 0; i < RF; ++i)
#pragma unroll
        for (int f = 0; f < NF; ++f) acc[i][f] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- DMA pieces: fixed per-lane byte offsets, uniform base per k-tile ---------------------------------------
    // K-contiguous operand: LDS row r of a half-tile -> (group = r / rows-per-group, rr) -> matrix row; chunk
    //   swizzle (r >> 1) & 7; offset = row * ld * 2 + 16 * chunk; the base advances 128 B per k-tile.
    // Reduction-major operand: a piece is 4 (256-B rows) or 8 (128-B rows) k-rows of the [64][columns] image;
    //   offset = k-row * ld * 2 + 2 * column; the base advances 64 rows per k-tile and moves to the second
    //   segment (A2 / B2: the other forward pass of a merged weight gradient) at k_seg. A k-tile that straddles
    //   k_seg or runs past k_end is issued by the slow path (per-lane row select, zero chunk past the end).
    constexpr int ARP = 256, BRP = G::B_ROWS * 2;             // row pitch of a reduction-major half
    static_assert(!ARM || RF == 8, "a reduction-major A half is 64 k-rows x 128 columns");
    auto rm_piece = [&](int p, int pitch, int &krow, int &ch) {
        const int cpr = pitch / 16;
        krow = (64 / cpr) * p + lane / cpr;
        ch = (lane % cpr) ^ (pitch == 256 ? swz_rmajor(krow) : pq_swz_rm128(krow));
    };
    auto a_col = [&](int p, int q, int &krow) -> int {        // ARM: first of this lane's 8 rows of D, from m0
        int ch;
        rm_piece(p, ARP, krow, ch);
        const int c = 8 * ch, grp = c / (16 * QA0), rr = c - grp * 16 * QA0;
        return min(grp * 16 * RF + q * 16 * QA0 + rr, M - 8 - m0);           // M % 8 == 0 (host-checked)
    };
    auto b_col = [&](int p, int q, int &krow) -> int {        // BRM: first of this lane's 8 columns of D, from n0
        int ch;
        rm_piece(p, BRP, krow, ch);
        const int c = 8 * ch, grp = c / (16 * QB), rr = c - grp * 16 * QB;
        return min(grp * 16 * NF + q * 16 * QB + rr, N - 8 - n0);            // N % 8 == 0 (host-checked)
    };
    auto a_off = [&](int p, int q) -> unsigned {              // piece p of A(q)
        if constexpr (ARM) {
            int krow;
            const int col = a_col(p, q, krow);
            return (unsigned)krow * (unsigned)g.lda * 2u + 2u * col;
        }
        const int r = 8 * p + (lane >> 3);
        const int per = 16 * (q ? QA1 : QA0);
        const int grp = r / per, rr = r - grp * per;
        int row = grp * 16 * RF + (q ? 16 * QA0 : 0) + rr;
        row = min(row, M - 1 - m0);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        return (unsigned)row * (unsigned)g.lda * 2u + 16u * c;
    };
    auto b_off = [&](int p, int q) -> unsigned {              // piece p of B(q)
        if constexpr (BRM) {
            int krow;
            const int col = b_col(p, q, krow);
            return (unsigned)krow * (unsigned)g.ldb * 2u + 2u * col;
        }
        const int r = 8 * p + (lane >> 3);
        const int grp = r / (16 * QB), rr = r - grp * 16 * QB;
        int col = grp * 16 * NF + q * 16 * QB + rr;
        col = min(col, N - 1 - n0);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        return (unsigned)col * (unsigned)g.ldb * 2u + 16u * c;
    };
    unsigned oa0[G::NA0], oa1[G::NA1], ob0[G::NB], ob1[G::NB];
#pragma unroll
    for (int e = 0; e < G::NA0; ++e) oa0[e] = a_off(min(wave + 8 * e, G::PA0 - 1), 0);
#pragma unroll
    for (int e = 0; e < G::NA1; ++e) oa1[e] = a_off(wave + 8 * e, 1);
#pragma unroll
    for (int e = 0; e < G::NB; ++e) {
        ob0[e] = b_off(wave + 8 * e, 0);
        ob1[e] = b_off(wave + 8 * e, 1);
    }
    // (its address once, not a GOT load per use: an s_load inside the loop also forces every LDS wait behind it to
    // lgkmcnt(0), scalar loads returning out of order)
    const unsigned short *zero_chunk = g_zero_chunk;
    asm volatile("" : "+s"(zero_chunk));
    const bool third = G::PA0 % 8 != 0 && wave < G::PA0 % 8;   // this wave issues NA0 pieces of A(0), else NA0 - 1
    // uniform base of k-tile u for a reduction-major operand; nullptr = slow path
    auto rm_base = [&](const unsigned short *X1, const unsigned short *X2, int ld, int o0, int k0) -> const char * {
        if (k0 + BK <= g.k_seg && k0 + BK <= k_end) return reinterpret_cast<const char *>(X1 + (size_t)k0 * ld + o0);
        if (k0 >= g.k_seg && k0 + BK <= k_end) return reinterpret_cast<const char *>(X2 + (size_t)(k0 - g.k_seg) * ld + o0);
        return nullptr;
    };
    auto rm_slow = [&](const unsigned short *X1, const unsigned short *X2, int ld, int o0, int k0, int krow, int col)
        -> const unsigned short * {
        const int k = k0 + krow;
        if (k >= k_end) return zero_chunk;
        return k < g.k_seg ? X1 + (size_t)k * ld + o0 + col : X2 + (size_t)(k - g.k_seg) * ld + o0 + col;
    };
    // K-contiguous operand, last k-tile of a K that is not a multiple of 64: chunks past k_end come from the zero chunk
    auto kc_src = [&](const char *base, unsigned off, int p, int k0) -> const void * {
        const int c = (lane & 7) ^ ((4 * (p & 1) + (lane >> 4)) & 7);         // logical chunk of this lane in piece p
        return k0 + 8 * c < k_end ? (const void *)(base + off) : (const void *)zero_chunk;
    };
    auto issue = [&](int u, int which) {                      // which: 0 = A(0), 1 = A(1), 2 = B(0), 3 = B(1)
        if constexpr (ABL & 2) { if (u > 1) return; }
        char *stage = smem + (NS == 2 ? (u & 1) : (u % NS)) * G::STAGE;
        const int k0 = k_begin + u * BK;
        const bool tail = k0 + BK > k_end;                    // wave-uniform
        if (which < 2) {
            const char *ap;
            if constexpr (ARM) ap = rm_base(g.A, g.A2, g.lda, m0, k0);
            else ap = reinterpret_cast<const char *>(g.A + (size_t)m0 * g.lda + k0);
            char *dst = stage + (which ? G::OFF_A1 : G::OFF_A0);
            if (ARM && ap == nullptr) {
#pragma unroll
                for (int e = 0; e < G::NA1; ++e) {            // ARM: NA0 == NA1, uniform
                    int krow;
                    const int col = a_col(wave + 8 * e, which, krow);
                    dma16_lane(rm_slow(g.A, g.A2, g.lda, m0, k0, krow, col), dst + (wave + 8 * e) * 1024);
                }
            } else if (!ARM && tail) {
#pragma unroll
                for (int e = 0; e < G::NA0; ++e) {
                    if (which ? e >= G::NA1 : (G::PA0 % 8 != 0 && e == G::NA0 - 1 && !third)) break;
                    dma16_lane(kc_src(ap, which ? oa1[e < G::NA1 ? e : 0] : oa0[e], wave + 8 * e, k0), dst + (wave + 8 * e) * 1024);
                }
            } else if (which == 0) {
#pragma unroll
                for (int e = 0; e < G::NA0; ++e) {
                    if (G::PA0 % 8 != 0 && e == G::NA0 - 1 && !third) break;
                    dma16_base(ap, oa0[e], dst + (wave + 8 * e) * 1024);
                }
            } else {
#pragma unroll
                for (int e = 0; e < G::NA1; ++e)
                    dma16_base(ap, oa1[e], dst + (wave + 8 * e) * 1024);
            }
        } else {
            const char *bp;
            if constexpr (BRM) bp = rm_base(g.B, g.B2, g.ldb, n0, k0);
            else bp = reinterpret_cast<const char *>(g.B + (size_t)n0 * g.ldb + k0);
            char *dst = stage + (which == 3 ? G::OFF_B1 : G::OFF_B0);
            if (BRM && bp == nullptr) {
#pragma unroll
                for (int e = 0; e < G::NB; ++e) {
                    int krow;
                    const int col = b_col(wave + 8 * e, which - 2, krow);
                    dma16_lane(rm_slow(g.B, g.B2, g.ldb, n0, k0, krow, col), dst + (wave + 8 * e) * 1024);
                }
            } else if (!BRM && tail) {
#pragma unroll
                for (int e = 0; e < G::NB; ++e)
                    dma16_lane(kc_src(bp, which == 3 ? ob1[e] : ob0[e], wave + 8 * e, k0), dst + (wave + 8 * e) * 1024);
            } else {
#pragma unroll
                for (int e = 0; e < G::NB; ++e)
                    dma16_base(bp, which == 3 ? ob1[e] : ob0[e], dst + (wave + 8 * e) * 1024);
            }
        }
    };
    // at most the pieces issued after the last piece of tile t + 1 stay in flight: B(0), A(0), B(1) of the NS - 1 tiles
    // after it and A(1) of NS - 2 of them
    auto wait_tile = [&](bool full) {
        if (!full) { pq_wait_vmcnt<0>(); return; }
        constexpr int LATE_A1 = (NS - 2) * G::NA1;
        if constexpr (G::PA0 % 8 != 0) {
            if (third) pq_wait_vmcnt<(NS - 1) * (2 * G::NB + G::NA0) + LATE_A1>();
            else pq_wait_vmcnt<(NS - 1) * (2 * G::NB + G::NA0 - 1) + LATE_A1>();
        } else {
            pq_wait_vmcnt<(NS - 1) * (2 * G::NB + G::NA0) + LATE_A1>();
        }
    };

    // ---- fragments ---------------------------------------------------------------------------------------------
    // K-contiguous: lane (row l16, 16-B chunk 4 ks + lg) of a 16-row block, one ds_read_b128; the swizzle depends
    // on l16 only. Reduction-major: 8 k-values (32 ks + 8 lg + j) of column l16, two ds_read_b64_tr_b16.
    const int frag_x0 = l16 * ROW_BYTES + (((0 + lg) ^ (l16 >> 1)) * 16);
    const int frag_x1 = l16 * ROW_BYTES + (((4 + lg) ^ (l16 >> 1)) * 16);
    // Transposing reads: lane (tq = l16 >> 2, tp = l16 & 3) of lane group lg addresses k-row 32 ks + 8 lg + tq (+4
    // for the second read) and the 8 bytes tp & 1 of chunk (2 blk + (tp >> 1)) ^ swizzle(row). Both swizzles depend
    // on tq and lg only (256-B rows: the second read's key is the first's ^ 1; 128-B rows: the same key), and
    // 2 blk only touches chunk bits 1-3, so one per-lane offset serves every block, k-step and read of a
    // half-tile: address = tile + (lane_off ^ 32 blk [^ 16]) + pitch * (32 ks [+ 4]), the last term an immediate.
    const int tq = l16 >> 2, tp = l16 & 3;
    const int rm_lane256 = 256 * (8 * lg + tq) + 16 * ((tp >> 1) ^ ((tq << 2) | ((2 * lg) & 3))) + 8 * (tp & 1);
    const int rm_lane128 = 128 * (8 * lg + tq) + 16 * ((tp >> 1) ^ pq_swz_rm128(8 * lg + tq)) + 8 * (tp & 1);
    auto frag_rm = [&](const char *t, int pitch, int blk, int ks) -> bf16x8 {   // 16-column block blk of the half
        const int base = (pitch == 256 ? rm_lane256 : rm_lane128) ^ (32 * blk);
        const int base_hi = pitch == 256 ? base ^ 16 : base;
        const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) v4s *)(t + base + pitch * 32 * ks));
        const v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) v4s *)(t + base_hi + pitch * (32 * ks + 4)));
        return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    };
    bf16x8 fa[QA0][2], fb0[QB][2], fb1[QB][2];
    bool abl_first = true;
    auto read_a = [&](const char *stage, int q, int ks) {
        if constexpr (ABL & 1) { if (!abl_first) return; }
        if constexpr (ARM) {
            const char *t = stage + (q ? G::OFF_A1 : G::OFF_A0);
#pragma unroll
            for (int i = 0; i < QA0; ++i) fa[i][ks] = frag_rm(t, ARP, wr * QA0 + i, ks);
            return;
        }
        const char *t = stage + (q ? G::OFF_A1 + wr * 16 * QA1 * ROW_BYTES : G::OFF_A0 + wr * 16 * QA0 * ROW_BYTES);
#pragma unroll
        for (int i = 0; i < QA0; ++i) {
            if (q && i >= QA1) break;
            fa[i][ks] = *reinterpret_cast<const bf16x8 *>(t + i * 16 * ROW_BYTES + (ks ? frag_x1 : frag_x0));
        }
    };
    auto read_b = [&](const char *stage, int q, bf16x8 (&fb)[QB][2]) {
        if constexpr (ABL & 1) { if (!abl_first) return; }
        if constexpr (BRM) {
            const char *t = stage + (q ? G::OFF_B1 : G::OFF_B0);
#pragma unroll
            for (int f = 0; f < QB; ++f) {
                fb[f][0] = frag_rm(t, BRP, wc * QB + f, 0);
                fb[f][1] = frag_rm(t, BRP, wc * QB + f, 1);
            }
            return;
        }
        const char *t = stage + (q ? G::OFF_B1 : G::OFF_B0) + wc * 16 * QB * ROW_BYTES;
#pragma unroll
        for (int f = 0; f < QB; ++f) {
            fb[f][0] = *reinterpret_cast<const bf16x8 *>(t + f * 16 * ROW_BYTES + frag_x0);
            fb[f][1] = *reinterpret_cast<const bf16x8 *>(t + f * 16 * ROW_BYTES + frag_x1);
        }
    };
    auto mfma_quadrant = [&](int qm, int qn, const bf16x8 (&fb)[QB][2]) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int i = 0; i < QA0; ++i) {
                if (qm && i >= QA1) break;
#pragma unroll
                for (int f = 0; f < QB; ++f) {
                    f32x4 &c = acc[qm ? QA0 + i : i][qn * QB + f];
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i][s], fb[f][s], c, 0, 0, 0);
                }
            }
        __builtin_amdgcn_s_setprio(0);
    };
    auto bar = [&]() { if constexpr (!(ABL & 4)) __builtin_amdgcn_s_barrier(); };
    auto lds_done = [&]() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); };

    if constexpr (NS == 3) {
        static_assert(!ARM, "the free-running schedule reads A K-contiguous");
        // ---- free-running schedule (header): fragment sets X / Y of half a k-tile each --------------------------
        bf16x8 ax[RF], bx[NF], ay[RF], by[NF];
        auto read_half = [&](const char *st, int ks, bf16x8 (&a)[RF], bf16x8 (&b)[NF]) {
            const int fx = ks ? frag_x1 : frag_x0;
            const char *a0 = st + G::OFF_A0 + wr * 16 * QA0 * ROW_BYTES, *a1 = st + G::OFF_A1 + wr * 16 * QA1 * ROW_BYTES;
#pragma unroll
            for (int i = 0; i < QA0; ++i) a[i] = *reinterpret_cast<const bf16x8 *>(a0 + i * 16 * ROW_BYTES + fx);
#pragma unroll
            for (int i = 0; i < QA1; ++i) a[QA0 + i] = *reinterpret_cast<const bf16x8 *>(a1 + i * 16 * ROW_BYTES + fx);
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int f = 0; f < QB; ++f) {
                    if constexpr (BRM) b[q * QB + f] = frag_rm(st + (q ? G::OFF_B1 : G::OFF_B0), BRP, wc * QB + f, ks);
                    else b[q * QB + f] = *reinterpret_cast<const bf16x8 *>(
                        st + (q ? G::OFF_B1 : G::OFF_B0) + (wc * QB + f) * 16 * ROW_BYTES + fx);
                }
        };
        auto mfma_half = [&](const bf16x8 (&a)[RF], const bf16x8 (&b)[NF]) {
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < RF; ++i)
#pragma unroll
                for (int f = 0; f < NF; ++f) acc[i][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[f], acc[i][f], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
        };
        auto issue_tile = [&](int u) {
            issue(u, 2);
            issue(u, 0);
            issue(u, 3);
            issue(u, 1);
        };
        // pieces this wave issues per tile: what may stay in flight when tile t must have landed (tile t + 1's)
        auto wait_landed = [&](bool next_in_flight) {
            if (!next_in_flight) { pq_wait_vmcnt<0>(); return; }
            if constexpr (G::PA0 % 8 != 0) {
                if (third) pq_wait_vmcnt<2 * G::NB + G::NA0 + G::NA1>();
                else pq_wait_vmcnt<2 * G::NB + G::NA0 - 1 + G::NA1>();
            } else {
                pq_wait_vmcnt<2 * G::NB + G::NA0 + G::NA1>();
            }
        };
        issue_tile(0);
        if (nt > 1) issue_tile(1);
        int ring = 0;
        for (int t = 0; t < nt; ++t) {
            const char *st = smem + ring * G::STAGE;
            ring = ring == 2 ? 0 : ring + 1;
            const bool more = t + 2 < nt;
            // own pieces of tile t have landed, own reads of tile t - 1 are in registers; then everyone's
            lds_done();
            wait_landed(t + 1 < nt);
            __builtin_amdgcn_sched_barrier(0);
            bar();
            __builtin_amdgcn_sched_barrier(0);
            if (wr == 0 && more) issue_tile(t + 2);            // into the stage tile t - 1 was read from
            __builtin_amdgcn_sched_barrier(0);
            read_half(st, 0, ax, bx);
            __builtin_amdgcn_sched_barrier(0);
            if (t > 0) mfma_half(ay, by);
            __builtin_amdgcn_sched_barrier(0);
            read_half(st, 1, ay, by);
            __builtin_amdgcn_sched_barrier(0);
            // (no control flow between the two read blocks: the compiler's wait for set X in front of its MFMAs is then
            // the counted lgkmcnt(11) that leaves the reads of set Y in flight, not lgkmcnt(0))
            if (wr == 1 && more) issue_tile(t + 2);            // the SIMD partner's pieces go under this wave's MFMAs
            __builtin_amdgcn_sched_barrier(0);
            mfma_half(ax, bx);
            __builtin_amdgcn_sched_barrier(0);
        }
        lds_done();
        bar();                                                 // every wave is done with the stages: the epilogue patches may go there
        mfma_half(ay, by);
        __builtin_amdgcn_sched_barrier(0);
    } else {
    // ---- prologue: k-tile 0 complete; the next NS - 2 whole tiles and B(0), A(0), B(1) of tile NS - 1 in flight ----
    issue(0, 0);
    issue(0, 2);
    issue(0, 3);
    issue(0, 1);
#pragma unroll
    for (int u = 1; u < NS; ++u) {
        if (u < nt) {
            issue(u, 2);
            issue(u, 0);
            issue(u, 3);
            if (u < NS - 1) issue(u, 1);
        }
    }
    wait_tile(nt >= NS);                              // (shorter reductions: everything has landed)
    bar();
    if (wr == 1) bar();                               // group 1 runs one barrier behind from here on

    int ring = 0;                                     // t % NS
    for (int t = 0; t < nt; ++t) {
        const char *stage = smem + ring * G::STAGE;
        ring = ring + 1 == NS ? 0 : ring + 1;
        const bool more1 = t + NS - 1 < nt, more2 = t + NS < nt;
        if constexpr (ABL & 1) abl_first = t == 0;
        // phase 1 (the B reads are issued first and retired before the barrier: B(0) is re-staged in phase 2)
        read_b(stage, 0, fb0);
        __builtin_amdgcn_sched_barrier(0);
        read_a(stage, 0, 0);
        if (more1) issue(t + NS - 1, 1);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!(ABL & 1)) pq_wait_lgkmcnt<(ARM ? 2 : 1) * QA0>();      // the B reads have returned
        __builtin_amdgcn_sched_barrier(0);
        read_a(stage, 0, 1);
        __builtin_amdgcn_sched_barrier(0);
        bar();
        lds_done();
        __builtin_amdgcn_sched_barrier(0);
        mfma_quadrant(0, 0, fb0);
        __builtin_amdgcn_sched_barrier(0);
        bar();
        // phase 2
        read_b(stage, 1, fb1);
        if (more2) issue(t + NS, 2);
        __builtin_amdgcn_sched_barrier(0);
        bar();
        lds_done();
        __builtin_amdgcn_sched_barrier(0);
        mfma_quadrant(0, 1, fb1);
        __builtin_amdgcn_sched_barrier(0);
        bar();
        // phase 3
        read_a(stage, 1, 0);
        read_a(stage, 1, 1);
        if (more2) issue(t + NS, 0);
        __builtin_amdgcn_sched_barrier(0);
        bar();
        lds_done();
        __builtin_amdgcn_sched_barrier(0);
        mfma_quadrant(1, 1, fb1);
        __builtin_amdgcn_sched_barrier(0);
        bar();
        // phase 4
        if (more2) issue(t + NS, 3);
        __builtin_amdgcn_sched_barrier(0);
        if (wr == 1) wait_tile(more2);
        bar();
        mfma_quadrant(1, 0, fb0);
        __builtin_amdgcn_sched_barrier(0);
        if (wr == 0) wait_tile(more2);
        bar();
    }
    if (wr == 0) bar();                               // rebalance the barrier count of the two groups
    }

    // ---- split K through slabs (NtArgs::ws_slab; cdna_hip_programming.md, "in-launch split-K reduction", the ticket form
    // with write-through stores): every slice stores its accumulators as they stand -- (i, f) fragment of thread tid at
    // float4 index (i NF + f) NT + tid of the slice's slab: 1 KiB of whole lines per wave-instruction, sc1 --, drains its
    // stores, and after a workgroup barrier ONE lane draws the tile's ticket (agent-scope add). The slice that draws the
    // last ticket puts the counter back to zero, acquires once, adds the other slices' slabs to its registers (sc1 loads)
    // and goes on as an UNSPLIT launch's workgroup would: every epilogue is available, nothing is zero-filled and no
    // float atomic is issued. No workgroup waits for another one, so there is nothing here that dispatch order, placement
    // or residency can stall. (All slices of a tile land on one XCD with this block order -- bid % 8 does not depend on
    // zs --, which the protocol does not rely on.)
    const bool slabs = g.splitk > 1 && g.ws_slab != nullptr;
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx942__) && !defined(__gfx950__)
#error "the slab ticket protocol's cache hints (aux = 16: sc1) are written for the gfx942 / gfx950 memory model"
#endif
    if (slabs) {
        constexpr unsigned SLAB_BYTES = (unsigned)G::BM * G::BN * 4u;
        typedef unsigned pq_u32x4 __attribute__((ext_vector_type(4)));
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            g.ws_slab + (size_t)ord * g.splitk * (G::BM * G::BN), 0, (int)(SLAB_BYTES * (unsigned)g.splitk), 0x00020000);
        const unsigned lane_off = (unsigned)threadIdx.x * 16u;
        const unsigned mine = (unsigned)zs * SLAB_BYTES;
#pragma unroll
        for (int i = 0; i < RF; ++i)
#pragma unroll
            for (int f = 0; f < NF; ++f)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(pq_u32x4, acc[i][f]), rs,
                                                       mine + (unsigned)(i * NF + f) * (NT * 16u) + lane_off, 0, 16);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // EVERY storing wave, before the barrier in front of the ticket
        __syncthreads();
        volatile unsigned *mail = reinterpret_cast<volatile unsigned *>(smem);     // (the stages are free: no second LDS object)
        if (threadIdx.x == 0)
            // ACQ_REL at agent scope: the release half orders this workgroup's (drained, barrier-joined) slab stores
            // before the ticket, the acquire half orders the last arriver's slab loads after it -- the memory model says so
            // by itself; the sc1 hints on the stores / loads and the explicit drain stay for speed, not for correctness
            mail[0] = __hip_atomic_fetch_add(g.ws_cnt + ord, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const unsigned ticket = mail[0];
        if (ticket != (unsigned)g.splitk - 1u) return;            // workgroup-uniform: the slab is this slice's whole result
        if (threadIdx.x == 0) {
            __hip_atomic_store(g.ws_cnt + ord, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // for the next launch
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();                                          // (also: every thread has read the mail word)
        constexpr int IB = RF % 3 == 0 ? 3 : (NF == 4 ? 2 : 4);   // fragment rows per batch: 6-12 loads of 16 B in flight per lane
        static_assert(RF % IB == 0, "whole batches");
#pragma unroll 1
        for (int sl = 0; sl < g.splitk; ++sl) {
            if (sl == zs) continue;
            const unsigned theirs = (unsigned)sl * SLAB_BYTES + lane_off;
#pragma unroll
            for (int i0 = 0; i0 < RF; i0 += IB) {
                pq_u32x4 t[IB][NF];
#pragma unroll
                for (int i = 0; i < IB; ++i)
#pragma unroll
                    for (int f = 0; f < NF; ++f)
                        t[i][f] = __builtin_amdgcn_raw_buffer_load_b128(rs, theirs + (unsigned)((i0 + i) * NF + f) * (NT * 16u), 0, 16);
#pragma unroll
                for (int i = 0; i < IB; ++i)
#pragma unroll
                    for (int f = 0; f < NF; ++f) acc[i0 + i][f] += __builtin_bit_cast(f32x4, t[i][f]);
            }
        }
    }

    // ---- epilogue through a wave-private LDS patch: 16 rows x 16 NF columns at a time ------------------------
    const int epi = g.epilogue;
    const bool lead = zs == 0;
    const bool split = g.splitk > 1 && !slabs;         // (float atomics; a slab launch's last arriver holds the complete sum)
    constexpr int LDW = G::EPI_LDW;
    float *patch = reinterpret_cast<float *>(smem) + wave * 16 * LDW;
    constexpr int LPR = 4 * NF;                        // lanes per patch row (one float4 each)
    constexpr int RPP = 64 / LPR;                      // rows per pass; NF passes cover the 16 rows
    const int prow = lane / LPR, pc4 = (lane % LPR) * 4;
    const int col = n0 + wc * 16 * NF + pc4;
    const bool col_ok = col < N;                       // N % 4 == 0 (host-checked): the whole float4 or nothing
    const bool with_bias = (!split || lead) && (epi == SEI_EPI_BIAS || epi == SEI_EPI_BIAS_GELU ||
                                                epi == SEI_EPI_BIAS_RES || epi == SEI_EPI_BIAS_ROWSCALE);
    f32x4 bias4 = f32x4{0.f, 0.f, 0.f, 0.f};
    if (with_bias && col_ok) bias4 = *reinterpret_cast<const f32x4 *>(g.bias + col);
    const bool dgelu_to_bf16 = g.D16 != nullptr && g.D32 == nullptr;     // the product is rounded to bf16 only
    const float *aux1 = nullptr, *aux2 = nullptr;
    if (!split) {
        if (epi == SEI_EPI_ACCUM) aux1 = g.D32;
        else if (epi == SEI_EPI_MUL_DGELU || epi == SEI_EPI_BIAS_RES) aux1 = g.R1;
        if (epi == SEI_EPI_BIAS_RES) aux2 = g.R2;
    } else if (lead && epi == SEI_EPI_BIAS_RES) {
        aux1 = g.R1;
        aux2 = g.R2;
    }
    float beta1 = 0.f, beta2 = 0.f, eps = 0.f, wd = 0.f, step_size = 0.f, inv_bc2_sqrt = 0.f;
    if constexpr (ADAM) {
        beta1 = g.adam_h[0]; beta2 = g.adam_h[1]; eps = g.adam_h[2]; wd = g.adam_h[3];
        step_size = g.adam_h[4]; inv_bc2_sqrt = g.adam_h[5];
    }
    f32x4 csum = f32x4{0.f, 0.f, 0.f, 0.f};            // NtArgs::colsum: this lane's four columns over its rows of the tile
#pragma unroll 1
    for (int i = 0; i < RF; ++i) {
        f32x4 c[NF];
        // static copies picked by a switch: a runtime-indexed acc[] would be demoted to scratch
#define PQ_PICK(I) case I: if constexpr (I < RF) { _Pragma("unroll") for (int f = 0; f < NF; ++f) c[f] = acc[I < RF ? I : 0][f]; } break;
        switch (i) {
            PQ_PICK(0) PQ_PICK(1) PQ_PICK(2) PQ_PICK(3) PQ_PICK(4) PQ_PICK(5) PQ_PICK(6) PQ_PICK(7)
            default: if constexpr (RF > 8) { _Pragma("unroll") for (int f = 0; f < NF; ++f) c[f] = acc[RF - 1][f]; } break;
        }
#undef PQ_PICK
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int j = 0; j < 4; ++j) patch[(4 * lg + j) * LDW + 16 * f + l16] = c[f][j];
        __builtin_amdgcn_sched_barrier(0);
        if (split) {
            // float atomics: one dword per lane on whole row segments (a float4-strided lane layout would make
            // every atomic instruction 16 quarter-filled 64-B requests: measured 2.4x the launch time)
            constexpr int CW = 16 * NF, RPI = 64 / CW;         // columns per row; rows per instruction
            const int sc = lane % CW, sr = lane / CW;
            const int scol = n0 + wc * CW + sc;
            const bool sok = scol < N;
            const float sbias = (with_bias && sok) ? g.bias[scol] : 0.f;
            float x[16 / RPI];
#pragma unroll
            for (int p = 0; p < 16 / RPI; ++p) x[p] = patch[(p * RPI + sr) * LDW + sc] + sbias;
            if (epi == SEI_EPI_BIAS_ROWSCALE && with_bias) {   // lead split: bias[n] * R1[m] instead of the plain bias
#pragma unroll
                for (int p = 0; p < 16 / RPI; ++p) {
                    const int row = m0 + wr * 16 * RF + 16 * i + p * RPI + sr;
                    x[p] += sbias * (g.R1[row < M ? row : 0] - 1.0f);
                }
            }
            if (aux1) {                                        // lead split of BIAS_RES: residuals, loads batched
#pragma unroll
                for (int p = 0; p < 16 / RPI; ++p) {
                    const int row = m0 + wr * 16 * RF + 16 * i + p * RPI + sr;
                    const bool ok = sok && row < M;
                    const size_t o = ok ? (size_t)row * N + scol : 0;
                    const float r1 = aux1[o], r2 = aux2 ? aux2[o] : 0.f;
                    x[p] += ok ? r1 + r2 : 0.f;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int p = 0; p < 16 / RPI; ++p) {
                const int row = m0 + wr * 16 * RF + 16 * i + p * RPI + sr;
                if (sok && row < M) atomicAdd(g.D32 + (size_t)row * N + scol, x[p]);
            }
            continue;
        }
        const int row_base = m0 + wr * 16 * RF + 16 * i + prow;
        if constexpr (ADAM) {
            // 3 NF independent 16-byte loads per lane in flight (parameter, both moments: streamed once per step, non-temporal
            // so that they do not push the GEMM operands out of L2 / MALL), then the element-wise step and four stores
            float4 gq[NF], pq[NF], mq[NF], vq[NF];
            size_t off[NF];
            bool ok[NF];
#pragma unroll
            for (int p = 0; p < NF; ++p) {
                const int row = row_base + p * RPP;
                ok[p] = col_ok && row < M;
                off[p] = ok[p] ? (size_t)row * N + col : 0;
                const f32x4 t = *reinterpret_cast<const f32x4 *>(patch + (p * RPP + prow) * LDW + pc4);
                gq[p] = make_float4(t[0], t[1], t[2], t[3]);
                pq[p] = nt_load4(g.adam_p + off[p]);
                mq[p] = nt_load4(g.adam_m + off[p]);
                vq[p] = nt_load4(g.adam_v + off[p]);
            }
#pragma unroll
            for (int p = 0; p < NF; ++p) {
                if (!ok[p]) continue;
                float4 o;
                o.x = sei_adam_element(pq[p].x, gq[p].x, mq[p].x, vq[p].x, beta1, beta2, eps, wd, step_size, inv_bc2_sqrt);
                o.y = sei_adam_element(pq[p].y, gq[p].y, mq[p].y, vq[p].y, beta1, beta2, eps, wd, step_size, inv_bc2_sqrt);
                o.z = sei_adam_element(pq[p].z, gq[p].z, mq[p].z, vq[p].z, beta1, beta2, eps, wd, step_size, inv_bc2_sqrt);
                o.w = sei_adam_element(pq[p].w, gq[p].w, mq[p].w, vq[p].w, beta1, beta2, eps, wd, step_size, inv_bc2_sqrt);
                nt_store4(g.adam_m + off[p], mq[p]);
                nt_store4(g.adam_v + off[p], vq[p]);
                nt_store4(g.adam_p + off[p], o);
                if (g.adam_p16) {
                    typedef unsigned pq_u32x2 __attribute__((ext_vector_type(2)));
                    pq_u32x2 t;
                    t.x = (unsigned)f2bf(o.x) | ((unsigned)f2bf(o.y) << 16);
                    t.y = (unsigned)f2bf(o.z) | ((unsigned)f2bf(o.w) << 16);
                    __builtin_nontemporal_store(t, reinterpret_cast<pq_u32x2 *>(g.adam_p16 + off[p]));
                }
            }
            continue;
        }
        f32x4 v[NF], a1[NF], a2[NF];
#pragma unroll
        for (int p = 0; p < NF; ++p) {
            v[p] = *reinterpret_cast<const f32x4 *>(patch + (p * RPP + prow) * LDW + pc4);
            a1[p] = a2[p] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if (aux1) {
#pragma unroll
            for (int p = 0; p < NF; ++p) {
                const int row = row_base + p * RPP;
                const bool ok = col_ok && row < M;
                const f32x4 x = *reinterpret_cast<const f32x4 *>(aux1 + (ok ? (size_t)row * N + col : 0));
                a1[p] = ok ? x : a1[p];
            }
        }
        if (aux2) {
#pragma unroll
            for (int p = 0; p < NF; ++p) {
                const int row = row_base + p * RPP;
                const bool ok = col_ok && row < M;
                const f32x4 x = *reinterpret_cast<const f32x4 *>(aux2 + (ok ? (size_t)row * N + col : 0));
                a2[p] = ok ? x : a2[p];
            }
        }
        if (epi == SEI_EPI_BIAS_ROWSCALE) {              // D = acc + bias[n] * R1[m]
#pragma unroll
            for (int p = 0; p < NF; ++p) {
                const int row = row_base + p * RPP;
                const float s = g.R1[row < M ? row : 0];
                v[p] += bias4 * s;
            }
        } else {
#pragma unroll
            for (int p = 0; p < NF; ++p) v[p] += bias4;
        }
        if (epi == SEI_EPI_MUL_DGELU && dgelu_to_bf16) {      // one uniform branch per 16-row chunk, not per value
#pragma unroll
            for (int p = 0; p < NF; ++p)
#pragma unroll
                for (int j = 0; j < 4; ++j) v[p][j] *= sei_dgelu_bf16out(a1[p][j]);
        } else if (epi == SEI_EPI_MUL_DGELU) {
#pragma unroll
            for (int p = 0; p < NF; ++p)
#pragma unroll
                for (int j = 0; j < 4; ++j) v[p][j] *= sei_dgelu(a1[p][j]);
        } else {
#pragma unroll
            for (int p = 0; p < NF; ++p) v[p] += a1[p] + a2[p];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int p = 0; p < NF; ++p) {
            const int row = row_base + p * RPP;
            if (!col_ok || row >= M) continue;
            const size_t o = (size_t)row * N + col;
            if (epi == SEI_EPI_BIAS_GELU) {
                ushort4 h;
                h.x = f2bf(sei_gelu_bf16out(v[p][0])); h.y = f2bf(sei_gelu_bf16out(v[p][1]));
                h.z = f2bf(sei_gelu_bf16out(v[p][2])); h.w = f2bf(sei_gelu_bf16out(v[p][3]));
                *reinterpret_cast<ushort4 *>(g.D2_16 + o) = h;
            }
            if (g.D32) *reinterpret_cast<f32x4 *>(g.D32 + o) = v[p];
            if (g.D16) {
                ushort4 h;
                h.x = f2bf(v[p][0]); h.y = f2bf(v[p][1]); h.z = f2bf(v[p][2]); h.w = f2bf(v[p][3]);
                *reinterpret_cast<ushort4 *>(g.D16 + o) = h;
                if (g.colsum) {                                   // (uniform) the ROUNDED values, as a pass over D16 would read them
                    csum[0] += __builtin_bit_cast(float, (unsigned)h.x << 16);
                    csum[1] += __builtin_bit_cast(float, (unsigned)h.y << 16);
                    csum[2] += __builtin_bit_cast(float, (unsigned)h.z << 16);
                    csum[3] += __builtin_bit_cast(float, (unsigned)h.w << 16);
                }
            }
        }
    }
    if (g.colsum && !split) {
        // lanes that share pc4 (lane % LPR) hold the same four columns for different patch rows: fold them, then one float
        // atomic per column and wave (two row halves x tiles_m workgroups add into each column)
#pragma unroll
        for (int off = LPR; off < 64; off <<= 1)
#pragma unroll
            for (int j = 0; j < 4; ++j) csum[j] += __shfl_xor(csum[j], off, 64);
        if (prow == 0 && col_ok) {
#pragma unroll
            for (int j = 0; j < 4; ++j) atomicAdd(g.colsum + col + j, csum[j]);
        }
    }
}

// Shapes the quadrant kernel takes: float4-able rows (K % 8 == 0 is checked by the entry points).
inline bool pq_eligible(const NtArgs &g, bool = false) {
    return g.N % 4 == 0 && g.K >= 32 &&
           (!g.D32 || (reinterpret_cast<uintptr_t>(g.D32) & 15) == 0) &&
           (!g.D16 || (reinterpret_cast<uintptr_t>(g.D16) & 7) == 0) &&
           (!g.D2_16 || (reinterpret_cast<uintptr_t>(g.D2_16) & 7) == 0) &&
           (!g.bias || (reinterpret_cast<uintptr_t>(g.bias) & 15) == 0) &&
           (!g.R1 || g.epilogue == SEI_EPI_BIAS_ROWSCALE || (reinterpret_cast<uintptr_t>(g.R1) & 15) == 0) &&
           (!g.R2 || (reinterpret_cast<uintptr_t>(g.R2) & 15) == 0);
}

// (three LDS stages wherever they fit and the operands are K-contiguous or B reduction-major: the 128-column tiles)
template <int RF, int NF, bool ARM = false, bool BRM = false, int ABL = 0, int NS = ((NF == 2 && !ARM) ? 3 : 2),
          bool ADAM = false>
int launch_pq(NtArgs &g, hipStream_t s) {
    using G = PqGeom<RF, NF>;
    g.tiles_m = (int)sei_ceil_div(g.M, G::BM);
    g.tiles_n = (int)sei_ceil_div(g.N, G::BN);
    const size_t tiles = (size_t)g.tiles_m * g.tiles_n;
    SEI_REQUIRE(tiles < ((size_t)1 << 27));
    int band = g.force_band > 0 ? g.force_band : 6;       // ~sqrt(32 tiles in flight per XCD)
    // (narrow outputs -- up to 16 tile columns -- take one band: with 6 the last band of an 8-column output is 2 columns wide;
    // tools/exp_band.py: 3456 x 2048 x 8192 132 -> 121 us, 2304 x 2048 x 8192 85.7 -> 83.2. Wide outputs keep 6: 576 x 8192 x
    // 32768 reads 266 us with 6, 290 with 16, 356 with 1 -- all row tiles of a tile column on one XCD is the WORST order)
    if (g.force_band <= 0 && g.tiles_n <= 16) band = g.tiles_n;
    if (band > g.tiles_n) band = g.tiles_n;
    g.band = band;
    g.tiles_per_xcd = (int)sei_ceil_div(tiles, 8);
    g.splitk = 1;
    g.k_per_split = g.K;
    g.ws_slab = nullptr;
    g.ws_cnt = nullptr;
    const size_t ktiles = sei_ceil_div(g.K, BK);
    const bool splittable = g.epilogue == SEI_EPI_NONE || g.epilogue == SEI_EPI_BIAS ||
                            g.epilogue == SEI_EPI_BIAS_RES || g.epilogue == SEI_EPI_ACCUM;
    const bool atomics_ok = splittable && g.D32 && !g.D16;               // partial sums added into a zero-filled float output
    // slabs (NtArgs::ws_slab): a 16-KiB block of tile counters, then tiles x splits slabs; any epilogue but ACCUM's running sum
    // (whose atomics need no zero fill) and Adam's; a workspace too small for the launch leaves it on the atomics / unsplit
    constexpr size_t WS_HEAD = 16384, SLAB = (size_t)G::BM * G::BN * 4;
    auto slabs_fit = [&](size_t sk) {
        return g.ws != nullptr && (reinterpret_cast<uintptr_t>(g.ws) & 255) == 0 && tiles <= WS_HEAD / 4 &&
               g.ws_bytes >= WS_HEAD + tiles * sk * SLAB && sk * SLAB < ((size_t)1 << 31);
    };
    const bool slabs_ok = !ADAM && g.epilogue != SEI_EPI_ACCUM && slabs_fit(2);
    if (!ADAM && (atomics_ok || slabs_ok) && g.K >= 8 * BK) {            // (the Adam epilogue needs the complete sum)
        const size_t slots = 256;
        const size_t max_sk = ktiles / 4 < 16 ? ktiles / 4 : 16;
        // in k-tile times: prologue + epilogue of every workgroup; split launches pay the slab store, the ticket and one
        // slab read per further slice (slabs) or the atomics (modelled at 2: measured 30-60 k-tile times per workgroup,
        // which is why the model kept most launches unsplit)
        const double overhead = 8.0 + (g.epilogue == SEI_EPI_ACCUM ? 0.0 : 2.0);
        double best = 1e30;
        size_t best_sk = 1;
        for (size_t sk = 1; sk <= max_sk; ++sk) {
            if (sk > 1 && slabs_ok && !slabs_fit(sk)) break;
            const double rounds = (double)sei_ceil_div(tiles * sk, slots);
            const double extra = sk == 1 ? 0.0 : (slabs_ok ? 6.0 + 3.0 * (double)(sk - 1) : 2.0);
            const double cost = rounds * ((double)sei_ceil_div(ktiles, sk) + overhead + extra);
            if (cost < best * 0.97) {
                best = cost;
                best_sk = sk;
            }
        }
#ifdef SEI_TUNING
        if (g_tuning_splitk > 0) best_sk = (size_t)g_tuning_splitk < ktiles ? (size_t)g_tuning_splitk : ktiles;
#endif
        if (g.force_splitk > 0 && (size_t)g.force_splitk <= ktiles && (atomics_ok || slabs_fit((size_t)g.force_splitk)))
            best_sk = (size_t)g.force_splitk;
        if (best_sk > 1) {
            g.k_per_split = (int)(sei_ceil_div(ktiles, best_sk) * BK);
            g.splitk = (int)sei_ceil_div(g.K, g.k_per_split);
            if (slabs_ok && g.splitk > 1 && slabs_fit((size_t)g.splitk)) {
                g.ws_cnt = reinterpret_cast<unsigned *>(g.ws);
                g.ws_slab = reinterpret_cast<float *>(reinterpret_cast<char *>(g.ws) + WS_HEAD);
            } else if (!atomics_ok) {
                g.splitk = 1;
                g.k_per_split = g.K;
            }
        }
    }
    if (g.plan) {
        // (bit 15 of the split count: the slices meet in slabs, not in float atomics)
        *g.plan = (2ull << 48) | ((unsigned long long)G::BM << 32) | ((unsigned long long)G::BN << 16) | (unsigned)g.splitk |
                  (g.ws_slab ? 0x8000u : 0u);
        return SEI_OK;
    }
    if (g.splitk > 1 && g.ws_slab == nullptr && g.epilogue != SEI_EPI_ACCUM) {
        const size_t n = (size_t)g.M * g.N;
        size_t zg = sei_ceil_div(n / 4 + 1, 256);
        if (zg > 2048) zg = 2048;
        hipLaunchKernelGGL(zero_fill_kernel, dim3((unsigned)zg), dim3(256), 0, s, g.D32, n);
    }
    hipLaunchKernelGGL((gemm_bf16pq_kernel<RF, NF, ARM, BRM, ABL, NS, ADAM>), dim3((unsigned)(8 * (size_t)g.tiles_per_xcd * g.splitk)),
                       dim3(NT), 0, s, g);
    return sei_launch_status();
}
