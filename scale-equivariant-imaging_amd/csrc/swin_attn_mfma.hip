// 8x8-window multi-head self-attention of SwinIR on the bf16 MFMA (v_mfma_f32_32x32x16_bf16), gfx950.
// (reference: WindowAttention.forward of the official SwinIR / deepinv.models.SwinIR, src/models/__init__.py:51-74.)
//
// Throughput-mode twin of swin_kernels.hip's exact-f32 kernels. Layout: the qkv projection of ALL tokens in natural
// (b, y, x) order, bf16, heads padded from 30 to HP = 32 dims (the two pad dims are exactly zero because the packed
// qkv weights have zero rows there): qkv16 (B*H*W, 3*heads*32), out16 / dout16 (B*H*W, heads*32). The cyclic shift,
// window partition / reverse, relative-position bias and shift mask are index arithmetic, as in the f32 kernels.
//
// One wave per (window, head); a workgroup of 4 waves walks a strided list of windows of ONE head, so that the
// bias-table gradient is summed in LDS across all of them and leaves as 225 float atomics per workgroup.
//
//   forward   S^T = K Q^T as 2x2 tiles of 32x32 (keys on the accumulator rows, the query on the lane: the softmax
//             over keys is lane-local plus one exchange with lane^32), scale + bias + mask + softmax in registers,
//             then O = P V with the P^T accumulators re-used directly as the A operand (cdna_hip_programming.md
//             section 3, "an accumulator tile as the next MFMA's operand": the k order inside a step is permuted, and
//             V is read from LDS with ds_read_b64_tr_b16 in that same order).
//   backward  recomputes P^T; dP^T = V dO^T; dS^T = P^T (dP^T - delta) -> dQ = scale dS K (reduction over keys =
//             accumulator rows of dS^T); then the same products in the other orientation (S = Q K^T, queries on the
//             rows) -> dV = P^T dO and dK = scale dS^T Q (reduction over queries = accumulator rows of P, dS).
//             56 MFMAs per (window, head), no transposes through LDS.
#include "sei_common.h"

namespace {

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f32x16 = __attribute__((ext_vector_type(16))) float;
typedef short v4s __attribute__((ext_vector_type(4)));
typedef short v8s __attribute__((ext_vector_type(8)));

constexpr int WS = 8, NTOK = 64, HP = 32, NB = (2 * WS - 1) * (2 * WS - 1), WAVES = 4;
constexpr int TILE_BYTES = NTOK * HP * 2;                  // one [token][32] bf16 matrix: 4 KiB, 64-byte rows

struct MGeom {
    int H, W, nwy, nwx, shift, heads, nwin;               // nwin = B * nwy * nwx
};

__device__ __forceinline__ void win_token(const MGeom &g, int win, int i, int &tok, int &region) {
    const int per = g.nwy * g.nwx;
    const int b = win / per, w = win - b * per;
    const int wy = w / g.nwx, wx = w - wy * g.nwx;
    const int sy = wy * WS + (i >> 3), sx = wx * WS + (i & 7);
    int oy = sy + g.shift, ox = sx + g.shift;
    if (oy >= g.H) oy -= g.H;
    if (ox >= g.W) ox -= g.W;
    tok = (b * g.H + oy) * g.W + ox;
    region = 0;
    if (g.shift > 0) {
        const int ry = sy < g.H - WS ? 0 : (sy < g.H - g.shift ? 1 : 2);
        const int rx = sx < g.W - WS ? 0 : (sx < g.W - g.shift ? 1 : 2);
        region = 3 * ry + rx;
    }
}

__device__ __forceinline__ int bias_bin(int i, int j) {
    return ((i >> 3) - (j >> 3) + WS - 1) * (2 * WS - 1) + ((i & 7) - (j & 7) + WS - 1);
}

__device__ __forceinline__ f32x16 mfma(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// rows r0 .. r0+31 of a [token][32] tile as the A operand (row = lane & 31) or, equally, as the B operand of the
// TRANSPOSED matrix (column = lane & 31): k = 16 s + 8 (lane >> 5) + 0..7 -> 16 contiguous bytes of the row.
__device__ __forceinline__ bf16x8 row_frag(const char *tile, int r0, int s, int lane) {
    return *reinterpret_cast<const bf16x8 *>(tile + (r0 + (lane & 31)) * (HP * 2) + (16 * s + 8 * (lane >> 5)) * 2);
}

// B operand [k = token][col = d] of a [token][32] tile for a k-step whose A operand is an ACCUMULATOR tile converted
// in place: element e of lane half h is token t0 + 8 (e >> 2) + 4 h + (e & 3). Two transposing reads.
__device__ __forceinline__ bf16x8 tr_frag_perm(const char *tile, int t0, int lane) {
    const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    const char *a = tile + (t0 + 4 * (g >> 1) + q) * (HP * 2) + (16 * (g & 1) + 4 * p) * 2;
    const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s *)a);
    const v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s *)(a + 8 * HP * 2));
    const v8s both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, both);
}

// registers 8 s .. 8 s + 7 of an accumulator tile as the A operand of k-step s (rows of the tile = reduction index)
__device__ __forceinline__ bf16x8 acc_frag(const f32x16 &x, int s) {
    bf16x8 a;
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] = (__bf16)x[8 * s + e];
    return a;
}

__device__ __forceinline__ int acc_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

// per-wave LDS: four [token][32] tiles + bias column of this head + token / region tables
struct __attribute__((aligned(16))) WaveLds {
    char q[TILE_BYTES], k[TILE_BYTES], v[TILE_BYTES], g[TILE_BYTES];
    float rowstat[NTOK][4];                               // per query: delta, max, 1 / sum (backward), pad
    int tok[NTOK], region[NTOK];
    char st[32 * HP * 2];                                 // one 32 x 32 bf16 result tile on its way out (16-byte stores)
};

// One 32-token x 32-dim result tile (accumulator layout: column = lane & 31, rows in the registers) to
// dst[tok[i0 + row]][col0 ..]: through LDS, so that it leaves as 16-byte stores of whole 64-byte rows (two store
// instructions per tile instead of sixteen 2-byte ones: the epilogue was store-issue-bound).
__device__ __forceinline__ void store_tile(WaveLds &L, const f32x16 &o, float mul, int lane, unsigned short *dst,
                                           size_t ld, int col0, int i0, bool live) {
    unsigned short *st = reinterpret_cast<unsigned short *>(L.st);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const __bf16 v = (__bf16)(o[r] * mul);
        st[acc_row(r, lane) * HP + (lane & 31)] = __builtin_bit_cast(unsigned short, v);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // same wave: LDS is in order; this pins the compiler too
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int row = 16 * pass + (lane >> 2), chunk = lane & 3;
        const uint4 v = *reinterpret_cast<const uint4 *>(L.st + row * (HP * 2) + chunk * 16);
        if (live) *reinterpret_cast<uint4 *>(dst + (size_t)L.tok[i0 + row] * ld + col0 + chunk * 8) = v;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

// The tiles of one (window, head), lane = token (4 x 16 bytes per matrix row), in two halves so that the global loads
// of item t+1 are in flight while item t computes: FETCH (global -> registers) and COMMIT (registers -> LDS). Plain
// named registers (a struct passed by reference ended up in scratch memory).
#define SWIN_FETCH(WIN, WITH_G)                                                                              \
    {                                                                                                        \
        win_token(g, (WIN), lane, s_tok, s_region);                                                          \
        const uint4 *src_ = reinterpret_cast<const uint4 *>(qkv + (size_t)s_tok * 3 * C + h * HP);            \
        s_q0 = src_[0]; s_q1 = src_[1]; s_q2 = src_[2]; s_q3 = src_[3];                                      \
        s_k0 = src_[C / 8]; s_k1 = src_[C / 8 + 1]; s_k2 = src_[C / 8 + 2]; s_k3 = src_[C / 8 + 3];          \
        s_v0 = src_[C / 4]; s_v1 = src_[C / 4 + 1]; s_v2 = src_[C / 4 + 2]; s_v3 = src_[C / 4 + 3];          \
        if (WITH_G) {                                                                                        \
            const uint4 *gs_ = reinterpret_cast<const uint4 *>(dout + (size_t)s_tok * C + h * HP);           \
            s_g0 = gs_[0]; s_g1 = gs_[1]; s_g2 = gs_[2]; s_g3 = gs_[3];                                      \
        }                                                                                                    \
    }
#define SWIN_COMMIT(WITH_G)                                                                                  \
    {                                                                                                        \
        L.tok[lane] = s_tok;                                                                                 \
        L.region[lane] = s_region;                                                                           \
        uint4 *d_ = reinterpret_cast<uint4 *>(L.q + lane * HP * 2);                                          \
        d_[0] = s_q0; d_[1] = s_q1; d_[2] = s_q2; d_[3] = s_q3;                                              \
        d_ = reinterpret_cast<uint4 *>(L.k + lane * HP * 2);                                                 \
        d_[0] = s_k0; d_[1] = s_k1; d_[2] = s_k2; d_[3] = s_k3;                                              \
        d_ = reinterpret_cast<uint4 *>(L.v + lane * HP * 2);                                                 \
        d_[0] = s_v0; d_[1] = s_v1; d_[2] = s_v2; d_[3] = s_v3;                                              \
        if (WITH_G) {                                                                                        \
            d_ = reinterpret_cast<uint4 *>(L.g + lane * HP * 2);                                             \
            d_[0] = s_g0; d_[1] = s_g1; d_[2] = s_g2; d_[3] = s_g3;                                          \
        }                                                                                                    \
    }
#define SWIN_STAGE_REGS                                                                                      \
    uint4 s_q0, s_q1, s_q2, s_q3, s_k0, s_k1, s_k2, s_k3, s_v0, s_v1, s_v2, s_v3, s_g0, s_g1, s_g2, s_g3;  \
    int s_tok, s_region;                                                                                     \
    (void)s_g0; (void)s_g1; (void)s_g2; (void)s_g3;

// ---- elementwise part: what dominated the first version --------------------------------------------------------
// Measured: 16 MFMAs per item are 0.2 us, the item took 4-5 us, almost all of it integer / LDS work per score element
// (bias bin, mask region, two table lookups: ~25 VALU instructions x 64 elements per lane). Now:
//  * bias: bin(i, j) = (iy*15 + ix + 112) - (jy*15 + jx). In the accumulator layout one of the two terms is a
//    compile-time constant per (tile, register) plus 4 * (lane >> 5), the other a per-lane constant for the whole
//    kernel: the lookup is ONE ds_read_b32 with an immediate offset from a per-lane base.
//  * mask (shift 4 only): with window 8 and shift 4 the 3 x 3 regions of calculate_mask reduce, inside a window of the
//    last window row / column, to "upper / lower half" x "left / right half"; key half and query half are constants
//    per (query tile, key tile, lane), never per element: the -100 is one per-lane value per tile pair, and windows
//    away from the last row and column (25 of 36) have none.
struct LaneGeom {
    const float *bias_o1[2];      // orientation 1 (keys on rows): base for query tile it; index 108 - c1(jt, r)
    const float *bias_o2[2];      // orientation 2 (queries on rows): base for key tile jt; index c2(it, r)
    bool q_low_x[2];              // orientation 1: query column in the right half (ix >= 4), per query tile
    bool q_low_y[2];              //                query row in the lower half (iy >= 4)
    bool k_low_x2[2];             // orientation 2: key column in the right half, per key tile
};
__device__ __forceinline__ LaneGeom lane_geom(const float *bias_col, int lane) {
    LaneGeom G;
    const int h = lane >> 5;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int i = 32 * t + (lane & 31), iy = i >> 3, ix = i & 7;
        G.bias_o1[t] = bias_col + (iy * 15 + ix + 112 - 4 * h) - 108;
        G.bias_o2[t] = bias_col + 4 * h + 112 - (iy * 15 + ix);          // (here i plays the key on the lane)
        G.q_low_x[t] = ix >= 4;
        G.q_low_y[t] = iy >= 4;
        G.k_low_x2[t] = ix >= 4;
    }
    return G;
}
__device__ __forceinline__ constexpr int c1(int jt, int r) { return (4 * jt + (r >> 2)) * 15 + (r & 3); }   // jy*15 + jx - 4h
__device__ __forceinline__ constexpr int c2(int it, int r) { return (4 * it + (r >> 2)) * 15 + (r & 3); }   // iy*15 + ix - 4h

// S^T tiles [jt][it] (rows = keys of tile jt, column = query 32 it + (lane & 31)): scale, bias, mask, softmax over
// the keys. On return p[jt][it] holds the probabilities.
template <bool SHIFTED, bool KEEP_STATS>
__device__ __forceinline__ void scores_T(WaveLds &L, const LaneGeom &G, float scale, int lane, bool last_row,
                                         bool last_col, f32x16 (&p)[2][2]) {
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            f32x16 acc = {0};
#pragma unroll
            for (int s = 0; s < 2; ++s) acc = mfma(row_frag(L.k, 32 * jt, s, lane), row_frag(L.q, 32 * it, s, lane), acc);
            p[jt][it] = acc;
        }
    const bool k_low_x = (lane >> 5) == 1;               // key column (r & 3) + 4h >= 4
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        float mx = -3.0e38f;
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) {
            float pen = 0.f;                             // key row 4 jt + (r >> 2) >= 4  <=>  jt == 1
            if (SHIFTED && ((last_row && G.q_low_y[it] != (jt == 1)) || (last_col && G.q_low_x[it] != k_low_x))) pen = -100.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = fmaf(p[jt][it][r], scale, G.bias_o1[it][108 - c1(jt, r)]) + pen;
                p[jt][it][r] = v;
                mx = fmaxf(mx, v);
            }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float e = __expf(p[jt][it][r] - mx);
                p[jt][it][r] = e;
                sum += e;
            }
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.0f / sum;
        if (KEEP_STATS && lane < 32) {
            L.rowstat[32 * it + lane][1] = mx;
            L.rowstat[32 * it + lane][2] = inv;
        }
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int r = 0; r < 16; ++r) p[jt][it][r] *= inv;
    }
}

template <bool SHIFTED>
__global__ __launch_bounds__(64 * WAVES) void swin_attn_fwd_mfma_kernel(const unsigned short *__restrict__ qkv,
                                                                          const float *__restrict__ table,
                                                                          unsigned short *__restrict__ out, MGeom g,
                                                                          float scale, int groups) {
    __shared__ WaveLds lds[WAVES];
    __shared__ float bias_col[NB];
    const int h = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int b = threadIdx.x; b < NB; b += 64 * WAVES) bias_col[b] = table[b * g.heads + h];
    WaveLds &L = lds[wave];
    const LaneGeom G = lane_geom(bias_col, lane);
    const int C = g.heads * HP;
    const int stride = groups * WAVES;
    const int rounds = (g.nwin + stride - 1) / stride;
    auto window_of = [&](int rd) { return min((rd * groups + (int)blockIdx.x) * WAVES + wave, g.nwin - 1); };
    const unsigned short *dout = nullptr;
    SWIN_STAGE_REGS
    SWIN_FETCH(window_of(0), false)
    for (int rd = 0; rd < rounds; ++rd) {
        const bool live = (rd * groups + (int)blockIdx.x) * WAVES + wave < g.nwin;
        const int win = window_of(rd);
        const int wloc = win % (g.nwy * g.nwx);
        const bool last_row = wloc / g.nwx == g.nwy - 1, last_col = wloc % g.nwx == g.nwx - 1;
        __syncthreads();                                   // previous round's LDS reads are done
        SWIN_COMMIT(false)
        __syncthreads();
        if (rd + 1 < rounds) SWIN_FETCH(window_of(rd + 1), false)      // in flight under the MFMAs
        f32x16 p[2][2];
        scores_T<SHIFTED, false>(L, G, scale, lane, last_row, last_col, p);
        // O[it] = sum over keys P[query][key] V[key][d]  =  (P^T tile)^T V
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            f32x16 o = {0};
#pragma unroll
            for (int jt = 0; jt < 2; ++jt)
#pragma unroll
                for (int s = 0; s < 2; ++s) o = mfma(acc_frag(p[jt][it], s), tr_frag_perm(L.v, 32 * jt + 16 * s, lane), o);
            store_tile(L, o, 1.0f, lane, out, C, h * HP, 32 * it, live);
        }
    }
}

template <bool SHIFTED>
__global__ __launch_bounds__(64 * WAVES) void swin_attn_bwd_mfma_kernel(const unsigned short *__restrict__ qkv,
                                                                          const float *__restrict__ table,
                                                                          const unsigned short *__restrict__ dout,
                                                                          unsigned short *__restrict__ dqkv,
                                                                          float *__restrict__ dtable, MGeom g,
                                                                          float scale, int groups) {
    __shared__ WaveLds lds[WAVES];
    __shared__ float bias_col[NB], bins[NB];
    const int h = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int b = threadIdx.x; b < NB; b += 64 * WAVES) {
        bias_col[b] = table[b * g.heads + h];
        bins[b] = 0.f;
    }
    WaveLds &L = lds[wave];
    const LaneGeom G = lane_geom(bias_col, lane);
    // dS summed over every window this wave walks: element (jt, it, r) of a lane always belongs to the same bias bin
    // (the bin depends on the positions inside the window only), so the table gradient needs no atomics in the loop
    f32x16 dsum[2][2] = {{{0}, {0}}, {{0}, {0}}};
    const int C = g.heads * HP;
    const int stride = groups * WAVES;
    const int rounds = (g.nwin + stride - 1) / stride;
    auto window_of = [&](int rd) { return min((rd * groups + (int)blockIdx.x) * WAVES + wave, g.nwin - 1); };
    SWIN_STAGE_REGS
    SWIN_FETCH(window_of(0), true)
    for (int rd = 0; rd < rounds; ++rd) {
        const bool live = (rd * groups + (int)blockIdx.x) * WAVES + wave < g.nwin;
        const int win = window_of(rd);
        const int wloc = win % (g.nwy * g.nwx);
        const bool last_row = wloc / g.nwx == g.nwy - 1, last_col = wloc % g.nwx == g.nwx - 1;
        __syncthreads();
        SWIN_COMMIT(true)
        __syncthreads();
        if (rd + 1 < rounds) SWIN_FETCH(window_of(rd + 1), true)
        // ---- orientation 1: keys on the accumulator rows ------------------------------------------------
        f32x16 p[2][2];
        scores_T<SHIFTED, true>(L, G, scale, lane, last_row, last_col, p);
        // dP^T[jt][it] = V dO^T ; delta[query] = sum_keys P dP ; dS^T = P (dP - delta), kept in dp
        f32x16 dp[2][2];
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            float delta = 0.f;
#pragma unroll
            for (int jt = 0; jt < 2; ++jt) {
                f32x16 acc = {0};
#pragma unroll
                for (int s = 0; s < 2; ++s)
                    acc = mfma(row_frag(L.v, 32 * jt, s, lane), row_frag(L.g, 32 * it, s, lane), acc);
                dp[jt][it] = acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) delta = fmaf(p[jt][it][r], acc[r], delta);
            }
            delta += __shfl_xor(delta, 32, 64);
            if (lane < 32) L.rowstat[32 * it + lane][0] = delta;             // needed again in orientation 2
            const float keep = live ? 1.0f : 0.0f;
#pragma unroll
            for (int jt = 0; jt < 2; ++jt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float ds = p[jt][it][r] * (dp[jt][it][r] - delta);
                    dp[jt][it][r] = ds;
                    dsum[jt][it][r] = fmaf(keep, ds, dsum[jt][it][r]);
                }
        }
        // dQ[it] = scale * (dS^T tile)^T K
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            f32x16 o = {0};
#pragma unroll
            for (int jt = 0; jt < 2; ++jt)
#pragma unroll
                for (int s = 0; s < 2; ++s) o = mfma(acc_frag(dp[jt][it], s), tr_frag_perm(L.k, 32 * jt + 16 * s, lane), o);
            store_tile(L, o, scale, lane, dqkv, 3 * (size_t)C, h * HP, 32 * it, live);
        }
        // ---- orientation 2: queries on the accumulator rows, the key on the lane ----------------------------
        // S = Q K^T and dP = dO V^T again with the operands swapped (16 MFMAs); the per-query max, 1 / sum and delta
        // come from orientation 1 through LDS, so the softmax here is elementwise.
        __syncthreads();                                   // rowstat written by lanes < 32 of this wave
        f32x16 dvacc[2] = {{0}, {0}}, dkacc[2] = {{0}, {0}};
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            f32x16 s2[2], d2[2];
#pragma unroll
            for (int jt = 0; jt < 2; ++jt) {
                f32x16 a = {0}, b2 = {0};
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    a = mfma(row_frag(L.q, 32 * it, s, lane), row_frag(L.k, 32 * jt, s, lane), a);
                    b2 = mfma(row_frag(L.g, 32 * it, s, lane), row_frag(L.v, 32 * jt, s, lane), b2);
                }
                s2[jt] = a;
                d2[jt] = b2;
            }
            // here the query is the accumulator row: its row / column half is (it == 1) / (lane >> 5)
            float pen[2];
#pragma unroll
            for (int jt = 0; jt < 2; ++jt)
                pen[jt] = (SHIFTED && ((last_row && (it == 1) != (jt == 1)) ||
                                       (last_col && ((lane >> 5) == 1) != G.k_low_x2[jt]))) ? -100.0f : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float4 st = *reinterpret_cast<const float4 *>(L.rowstat[32 * it + acc_row(r, lane)]);   // delta, max, 1/sum
#pragma unroll
                for (int jt = 0; jt < 2; ++jt) {
                    const float v = fmaf(s2[jt][r], scale, G.bias_o2[jt][c2(it, r)]) + pen[jt];
                    const float pr = __expf(v - st.y) * st.z;              // P[query][key]
                    s2[jt][r] = pr;
                    d2[jt][r] = pr * (d2[jt][r] - st.x);                   // dS[query][key]
                }
            }
            // dV[jt] += (P tile)^T dO ; dK[jt] += (dS tile)^T Q     (reduction over the queries of tile it)
#pragma unroll
            for (int jt = 0; jt < 2; ++jt)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    dvacc[jt] = mfma(acc_frag(s2[jt], s), tr_frag_perm(L.g, 32 * it + 16 * s, lane), dvacc[jt]);
                    dkacc[jt] = mfma(acc_frag(d2[jt], s), tr_frag_perm(L.q, 32 * it + 16 * s, lane), dkacc[jt]);
                }
        }
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) {
            store_tile(L, dkacc[jt], scale, lane, dqkv, 3 * (size_t)C, C + h * HP, 32 * jt, live);
            store_tile(L, dvacc[jt], 1.0f, lane, dqkv, 3 * (size_t)C, 2 * C + h * HP, 32 * jt, live);
        }
    }
    // table gradient: one LDS add per element and lane for the whole kernel, then 225 global atomics per workgroup
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int i = 32 * it + (lane & 31);
        const int base = (i >> 3) * 15 + (i & 7) + 112 - 4 * (lane >> 5);
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int r = 0; r < 16; ++r) atomicAdd(&bins[base - c1(jt, r)], dsum[jt][it][r]);
    }
    __syncthreads();
    for (int b = threadIdx.x; b < NB; b += 64 * WAVES) atomicAdd(dtable + b * g.heads + h, bins[b]);
}

inline int check(int B, int H, int W, int heads, int shift) {
    SEI_REQUIRE(B > 0 && H >= WS && W >= WS && H % WS == 0 && W % WS == 0 && heads > 0 && heads <= 64);
    SEI_REQUIRE(shift == 0 || shift == WS / 2);           // SwinIR: window 8, shift 0 / 4 (the mask logic relies on it)
    SEI_REQUIRE((size_t)B * H * W < ((size_t)1 << 31));
    return SEI_OK;
}

// groups x heads workgroups, each looping over its windows. Two workgroups fit a CU (80 KB of LDS each): 512 resident.
// 128 groups x 6 heads = 768 ran as one full round and a half-empty one; 85 x 6 = 510 are all resident and walk
// more windows each.
inline int group_count(int nwin, int heads, int resident = 512) {
    const int groups = (nwin + WAVES - 1) / WAVES;
    const int cap = resident / heads > 0 ? resident / heads : 1;
    return groups > cap ? cap : groups;
}

}  // namespace

extern "C" int sei_swin_attn_fwd_bf16(const uint16_t *qkv, const float *table, uint16_t *out, int B, int H, int W,
                                      int heads, int shift, float scale, void *stream) {
    SEI_REQUIRE(qkv && table && out);
    SEI_REQUIRE((((uintptr_t)qkv | (uintptr_t)out) & 15) == 0);
    if (int rc = check(B, H, W, heads, shift)) return rc;
    MGeom g{H, W, H / WS, W / WS, shift, heads, B * (H / WS) * (W / WS)};
    const int groups = group_count(g.nwin, heads);
    if (shift)
        hipLaunchKernelGGL(swin_attn_fwd_mfma_kernel<true>, dim3((unsigned)groups, (unsigned)heads), dim3(64 * WAVES), 0,
                           (hipStream_t)stream, qkv, table, out, g, scale, groups);
    else
        hipLaunchKernelGGL(swin_attn_fwd_mfma_kernel<false>, dim3((unsigned)groups, (unsigned)heads), dim3(64 * WAVES), 0,
                           (hipStream_t)stream, qkv, table, out, g, scale, groups);
    return sei_launch_status();
}

extern "C" int sei_swin_attn_bwd_bf16(const uint16_t *qkv, const float *table, const uint16_t *dout, uint16_t *dqkv,
                                      float *dtable, int B, int H, int W, int heads, int shift, float scale,
                                      void *stream) {
    SEI_REQUIRE(qkv && table && dout && dqkv && dtable);
    SEI_REQUIRE((((uintptr_t)qkv | (uintptr_t)dout | (uintptr_t)dqkv) & 15) == 0);
    if (int rc = check(B, H, W, heads, shift)) return rc;
    MGeom g{H, W, H / WS, W / WS, shift, heads, B * (H / WS) * (W / WS)};
    const int groups = group_count(g.nwin, heads, 256);        // 494 registers: ONE workgroup per CU
    if (shift)
        hipLaunchKernelGGL(swin_attn_bwd_mfma_kernel<true>, dim3((unsigned)groups, (unsigned)heads), dim3(64 * WAVES), 0,
                           (hipStream_t)stream, qkv, table, dout, dqkv, dtable, g, scale, groups);
    else
        hipLaunchKernelGGL(swin_attn_bwd_mfma_kernel<false>, dim3((unsigned)groups, (unsigned)heads), dim3(64 * WAVES), 0,
                           (hipStream_t)stream, qkv, table, dout, dqkv, dtable, g, scale, groups);
    return sei_launch_status();
}
