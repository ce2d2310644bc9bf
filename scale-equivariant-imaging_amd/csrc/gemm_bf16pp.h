// gemm_bf16pp_kernel: 256 x 256 x 64 "ping-pong" schedule of the bf16 GEMM (included by gemm_bf16nt.hip, which
// provides the LDS images, the DMA pieces and the fragment reads).
//
// Why a second schedule: the 128x128 / one-barrier-per-k-tile loop parks its waves ~50 % of the time at
// s_waitcnt + s_barrier (SQ_WAIT_ANY) and tops out near 0.8 PFLOP/s whatever is prefetched; that is the known
// ceiling of that structure (cdna_hip_programming.md, "the step-3 structure"). What lifts it is ONE workgroup
// per CU whose LDS-DMA stays in flight across raw barriers (counted vmcnt, never 0 in the loop) and whose two
// wave groups alternate: while one group's waves run MFMAs, the other group's waves on the same SIMDs issue
// their LDS reads and DMA pieces.
//
// Geometry: 8 waves = 2 (rows: wr) x 4 (cols: wc); a wave owns 128 x 64 of the tile = 4 x 2 accumulators of
// 32x32 (128 VGPRs). A k-tile (64) is four HALF-TILES of 16 KB: A rows 0-127 / 128-255 (one per wave group)
// and B cols 0-127 / 128-255; either a K-contiguous image (128 rows x 128 B, ds_read_b128) or a
// reduction-major one (64 k-rows x 256 B, ds_read_b64_tr_b16) -- the same 16 DMA pieces of 1 KiB either way,
// two per wave. LDS = 2 k-tiles x 4 half-tiles = 128 KB.
//
// A k-tile is computed in four PHASES, one 64 x 32 quadrant of the wave's block each (8 MFMAs = 256 cycles):
//     phase 1: read A(q0) [8 x b128] + B(q0) [4]    MFMA (q0,q0)      issue Bh1, Ah0 of tile t+1
//     phase 2: read B(q1) [4]                       MFMA (q0,q1)      issue Ah1 of tile t+1
//     phase 3: read A(q1) [8]                       MFMA (q1,q1)      --
//     phase 4: --                                   MFMA (q1,q0)      issue Bh0 of tile t+2
// each phase = [reads + DMA issue] s_barrier [lgkmcnt(0); MFMAs] s_barrier. Group 1 (wr = 1) executes one
// extra barrier up front, so its read segment coincides with group 0's MFMA segment and vice versa.
//
// Hazards (placed by count, never by "it passed"):
//   WAR  a half-tile is re-staged >= 2 phases after its last ds_read: B halves are last read in phase 2
//        (re-staged from phase 4), A halves in phase 3 (re-staged from the next tile's phase 1).
//   RAW  every wave retires its own pieces of tile t+1 with a counted vmcnt (2 = the Bh0(t+2) pieces just
//        issued may stay in flight) BEFORE the barrier event that ends phase 4 -- group 0 before its second
//        barrier of phase 4, group 1 (one barrier ahead) before its first -- and tile t+1 is first read after
//        that event.
//
// Status (round 1): correct on all three operand layouts and every epilogue (tests), opt-in only
// (sei_debug_set_nt_tile(20)). Measured on MI355X against the automatic choice: K-contiguous operands
// 848 vs 802 TFLOP/s on 4096^3 and 585 vs 509 on 2304x8192x2048; reduction-major operands lose (400 vs 659 on
// 2048x8192x3456: every fragment is two ds_read_b64_tr_b16, and the rr/kr variants sit at 247-256 VGPRs).
// The schedule is LDS-bandwidth-bound: a k-tile moves 192 KB of fragment reads + 64 KB of DMA writes through
// a 128 B/clk LDS = 2048 cycles, exactly its 2048 MFMA cycles, and phase 1 (12 reads + 4 pieces) is twice as
// long as its 8 MFMAs. Next step: balance the reads over the phases / feed one operand from registers.
#pragma once

constexpr int PP_BM = 256, PP_BN = 256;
constexpr int PP_HALF = 128 * ROW_BYTES;              // 16 KB
constexpr int PP_LDS = 2 * 4 * PP_HALF;               // 128 KB

// ABL (tuning aid, wrong results): 1 = no fragment reads in the loop, 2 = no DMA in the loop, 4 = no barriers.
template <bool ARM, bool BRM, int ABL = 0>
__global__ __launch_bounds__(NT) void gemm_bf16pp_kernel(NtArgs g) {
    __shared__ __attribute__((aligned(1024))) char smem[PP_LDS];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int li = lane & 31, lh = lane >> 5;

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
    const int m0 = tm_i * PP_BM, n0 = tn_i * PP_BN;
    const int M = g.M, N = g.N;
    const int k_begin = zs * g.k_per_split;
    const int k_end = min(g.K, k_begin + g.k_per_split);
    const int nt = (k_end - k_begin + BK - 1) / BK;

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- DMA: half-tile h (0,1 = A row halves; 2,3 = B column halves) of k-tile u, 2 pieces per wave ----------
    auto issue_half = [&](int u, int h) {
        if constexpr (ABL & 2) { if (u > 1) return; }
        char *dst = smem + ((u & 1) * 4 + h) * PP_HALF;
        const int k0 = k_begin + u * BK;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int q = 2 * wave + e;
            if (h < 2) {
                const int r0 = m0 + 128 * h;
                if constexpr (ARM) stage_piece_rmajor(g.A, g.A2, g.k_seg, g.lda, r0, M, k0, k_end, dst, q, lane);
                else stage_piece<128>(g.A, g.lda, r0, M, k0, k_end, dst, q, lane);
            } else {
                const int c0 = n0 + 128 * (h - 2);
                if constexpr (BRM) stage_piece_rmajor(g.B, g.B2, g.k_seg, g.ldb, c0, N, k0, k_end, dst, q, lane);
                else stage_piece<128>(g.B, g.ldb, c0, N, k0, k_end, dst, q, lane);
            }
        }
    };

    // ---- fragments ---------------------------------------------------------------------------------------------
    const int sw = (li >> 1) & 7;
    bf16x8 fa[2][4], fb0[4], fb1[4];                  // A: 2 row fragments x 4 k-steps; B: column fragment 0 / 1
    bool abl_first = true;
    auto read_a = [&](const char *buf, int qm) {      // rows 64*qm .. +64 of this wave group's A half
        if constexpr (ABL & 1) { if (!abl_first) return; }
        const char *t = buf + wr * PP_HALF;
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int r0 = 64 * qm + 32 * i;
                if constexpr (ARM) fa[i][s] = frag_rmajor(t, s, lane, r0);
                else fa[i][s] = *reinterpret_cast<const bf16x8 *>(t + (r0 + li) * ROW_BYTES + (((2 * s + lh) ^ sw) * 16));
            }
    };
    auto read_b = [&](const char *buf, int qn, bf16x8 (&fb)[4]) {   // cols 64*(wc&1) + 32*qn .. +32 of B half wc>>1
        if constexpr (ABL & 1) { if (!abl_first) return; }
        const char *t = buf + (2 + (wc >> 1)) * PP_HALF;
        const int c0 = 64 * (wc & 1) + 32 * qn;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if constexpr (BRM) fb[s] = frag_rmajor(t, s, lane, c0);
            else fb[s] = *reinterpret_cast<const bf16x8 *>(t + (c0 + li) * ROW_BYTES + (((2 * s + lh) ^ sw) * 16));
        }
    };
    auto mfma_quadrant = [&](int qm, int qn, const bf16x8 (&fb)[4]) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int i = 0; i < 2; ++i)
                acc[2 * qm + i][qn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][s], fb[s], acc[2 * qm + i][qn], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };
    auto bar = [&]() { if constexpr (!(ABL & 4)) __builtin_amdgcn_s_barrier(); };
    auto lds_done = [&]() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); };

    // ---- prologue: k-tile 0 complete, Bh0 of k-tile 1 in flight ---------------------------------------------
    issue_half(0, 0);
    issue_half(0, 1);
    issue_half(0, 2);
    issue_half(0, 3);
    if (nt > 1) {
        issue_half(1, 2);
        asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    bar();
    if (wr == 1) bar();                               // group 1 runs one barrier ahead from here on

    for (int t = 0; t < nt; ++t) {
        const char *buf = smem + (t & 1) * 4 * PP_HALF;
        const bool more1 = t + 1 < nt, more2 = t + 2 < nt;
        if constexpr (ABL & 1) abl_first = t == 0;
        // phase 1
        read_a(buf, 0);
        read_b(buf, 0, fb0);
        if (more1) {
            issue_half(t + 1, 3);
            issue_half(t + 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        bar();
        lds_done();
        mfma_quadrant(0, 0, fb0);
        __builtin_amdgcn_sched_barrier(0);
        bar();
        // phase 2
        read_b(buf, 1, fb1);
        if (more1) issue_half(t + 1, 1);
        __builtin_amdgcn_sched_barrier(0);
        bar();
        lds_done();
        mfma_quadrant(0, 1, fb1);
        __builtin_amdgcn_sched_barrier(0);
        bar();
        // phase 3
        read_a(buf, 1);
        __builtin_amdgcn_sched_barrier(0);
        bar();
        lds_done();
        mfma_quadrant(1, 1, fb1);
        __builtin_amdgcn_sched_barrier(0);
        bar();
        // phase 4: retire k-tile t+1 (only the Bh0(t+2) pieces just issued may stay in flight) before the
        // barrier event that ends the phase: group 1 is one barrier ahead, so it waits before its first.
        if (more2) issue_half(t + 2, 2);
        __builtin_amdgcn_sched_barrier(0);
        if (wr == 1) {
            if (more2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        bar();
        mfma_quadrant(1, 0, fb0);
        __builtin_amdgcn_sched_barrier(0);
        if (wr == 0) {
            if (more2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        bar();
    }
    if (wr == 0) bar();                               // rebalance the barrier count of the two groups

    // ---- epilogue (as gemm_bf16nt_kernel; auxiliary values gathered per 32x32 tile) ---------------------------
    const int epi = g.epilogue;
    const bool lead = zs == 0;
    const bool split = g.splitk > 1;
    // The eight 32x32 sub-tiles are walked by a ROLLED loop (its body inlines erf-based GELU / GELU' for 16
    // rows; fully unrolled it exceeds the unroller's budget, the loop stays, and a runtime-indexed acc[] is
    // demoted to scratch). The accumulator of the current sub-tile is picked by a switch of static copies.
#pragma unroll 1
    for (int idx = 0; idx < 8; ++idx) {
        f32x16 c;
        switch (idx) {
            case 0: c = acc[0][0]; break;
            case 1: c = acc[0][1]; break;
            case 2: c = acc[1][0]; break;
            case 3: c = acc[1][1]; break;
            case 4: c = acc[2][0]; break;
            case 5: c = acc[2][1]; break;
            case 6: c = acc[3][0]; break;
            default: c = acc[3][1]; break;
        }
        const int i = idx >> 1, j = idx & 1;
        const int col = n0 + wc * 64 + 32 * j + li;
        const bool col_ok = col < N;
        const int row_base = m0 + wr * 128 + 32 * i + 4 * lh;
        float bias = (col_ok && (!split || lead) &&
                      (epi == SEI_EPI_BIAS || epi == SEI_EPI_BIAS_GELU || epi == SEI_EPI_BIAS_RES ||
                       epi == SEI_EPI_BIAS_ROWSCALE))
                         ? g.bias[col] : 0.f;
        // Auxiliary inputs: ONE uniform decision per sub-tile, then 16 unconditional loads in flight together
        // (clamped address, masked afterwards). A per-row "load or not" branch makes hipcc wait vmcnt(0) per
        // row, and stores count on vmcnt too: every store of the previous rows was drained one by one
        // (measured: 38 us of a 47 us single-tile launch).
        float a1[16], a2[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) a1[r] = a2[r] = 0.f;
        const float *aux1 = nullptr, *aux2 = nullptr;
        if (!split) {
            if (epi == SEI_EPI_ACCUM) aux1 = g.D32;
            else if (epi == SEI_EPI_MUL_DGELU || epi == SEI_EPI_BIAS_RES) aux1 = g.R1;
            if (epi == SEI_EPI_BIAS_RES) aux2 = g.R2;
        } else if (lead && epi == SEI_EPI_BIAS_RES) {
            aux1 = g.R1;
            aux2 = g.R2;
        }
        if (aux1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row_base + (r & 3) + 8 * (r >> 2);
                const bool ok = col_ok && row < M;
                const float v = aux1[ok ? (size_t)row * N + col : 0];
                a1[r] = ok ? v : 0.f;
            }
        }
        if (aux2) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row_base + (r & 3) + 8 * (r >> 2);
                const bool ok = col_ok && row < M;
                const float v = aux2[ok ? (size_t)row * N + col : 0];
                a2[r] = ok ? v : 0.f;
            }
        }
        if (epi == SEI_EPI_BIAS_ROWSCALE) {              // D = acc + bias[n] * R1[m]
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row_base + (r & 3) + 8 * (r >> 2);
                a1[r] = bias * g.R1[row < M ? row : 0];
            }
            bias = 0.f;
        }
        __builtin_amdgcn_sched_barrier(0);
        // values first (this consumes every gathered input: one wait), stores afterwards with nothing to wait for
        float v[16];
        if (epi == SEI_EPI_MUL_DGELU && !split) {
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = (c[r] + bias) * sei_dgelu(a1[r]);
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = c[r] + bias + a1[r] + a2[r];
        }
        __builtin_amdgcn_sched_barrier(0);
        float *d32 = g.D32 ? g.D32 + (size_t)row_base * N + col : nullptr;
        unsigned short *d16 = g.D16 ? g.D16 + (size_t)row_base * N + col : nullptr;
        if (split) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int dr = (r & 3) + 8 * (r >> 2);
                if (col_ok && row_base + dr < M) atomicAdd(d32 + (size_t)dr * N, v[r]);
            }
        } else {
            if (epi == SEI_EPI_BIAS_GELU) {
                unsigned short *d2 = g.D2_16 + (size_t)row_base * N + col;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int dr = (r & 3) + 8 * (r >> 2);
                    if (col_ok && row_base + dr < M) d2[(size_t)dr * N] = f2bf(sei_gelu(v[r]));
                }
            }
            if (d32) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int dr = (r & 3) + 8 * (r >> 2);
                    if (col_ok && row_base + dr < M) d32[(size_t)dr * N] = v[r];
                }
            }
            if (d16) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int dr = (r & 3) + 8 * (r >> 2);
                    if (col_ok && row_base + dr < M) d16[(size_t)dr * N] = f2bf(v[r]);
                }
            }
        }
    }
}

template <bool ARM, bool BRM, int ABL = 0>
int launch_pp(NtArgs &g, hipStream_t s) {
    g.tiles_m = (int)sei_ceil_div(g.M, PP_BM);
    g.tiles_n = (int)sei_ceil_div(g.N, PP_BN);
    const size_t tiles = (size_t)g.tiles_m * g.tiles_n;
    SEI_REQUIRE(tiles < ((size_t)1 << 27));
    int band = g.force_band > 0 ? g.force_band : 6;       // ~sqrt(32 tiles in flight per XCD)
    if (band > g.tiles_n) band = g.tiles_n;
    g.band = band;
    g.tiles_per_xcd = (int)sei_ceil_div(tiles, 8);
    g.splitk = 1;
    g.k_per_split = g.K;
    const bool splittable = g.epilogue == SEI_EPI_NONE || g.epilogue == SEI_EPI_BIAS ||
                            g.epilogue == SEI_EPI_BIAS_RES || g.epilogue == SEI_EPI_ACCUM;
    if (splittable && g.D32 && !g.D16 && g.K >= 8 * BK) {
        const size_t slots = 256, ktiles = sei_ceil_div(g.K, BK);
        const size_t max_sk = ktiles / 4 < 16 ? ktiles / 4 : 16;
        const double overhead = 8.0 + (g.epilogue == SEI_EPI_ACCUM ? 0.0 : 2.0);
        double best = 1e30;
        size_t best_sk = 1;
        for (size_t sk = 1; sk <= max_sk; ++sk) {
            const double rounds = (double)sei_ceil_div(tiles * sk, slots);
            const double cost = rounds * ((double)sei_ceil_div(ktiles, sk) + overhead + (sk > 1 ? 2.0 : 0.0));
            if (cost < best * 0.97) {
                best = cost;
                best_sk = sk;
            }
        }
        if (best_sk > 1) {
            g.k_per_split = (int)(sei_ceil_div(ktiles, best_sk) * BK);
            g.splitk = (int)sei_ceil_div(g.K, g.k_per_split);
        }
    }
    if (g.splitk > 1 && g.epilogue != SEI_EPI_ACCUM) {
        const size_t n = (size_t)g.M * g.N;
        size_t zg = sei_ceil_div(n / 4 + 1, 256);
        if (zg > 2048) zg = 2048;
        if ((reinterpret_cast<uintptr_t>(g.D32) & 15) != 0) return SEI_ERR_BAD_ARG;
        hipLaunchKernelGGL(zero_fill_kernel, dim3((unsigned)zg), dim3(256), 0, s, g.D32, n);
    }
    hipLaunchKernelGGL((gemm_bf16pp_kernel<ARM, BRM, ABL>), dim3((unsigned)(8 * (size_t)g.tiles_per_xcd * g.splitk)),
                       dim3(NT), 0, s, g);
    return sei_launch_status();
}
