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
//   backward  ONE orientation (queries on the accumulator rows, the key on the lane), every wave on its own: no
//             workgroup barrier inside the loop, LDS tiles are wave-private. S = Q K^T is accumulated on top of
//             (bias + mask) / scale, so that P = exp2(c S - lse) is one FMA and one v_exp per element with the
//             per-query log-sum-exp the FORWARD pass stored (log2 units) -- no maxima, no sums, no second set of
//             score products; delta = rowsum(dO * O) comes from the forward output (v_dot2_f32_bf16, lane = token).
//             dS = P (dP - delta). All three results are produced TRANSPOSED (head dim on the accumulator rows, the
//             token on the lane): dV^T += dO^T P, dK^T += Q^T dS with the P / dS accumulators converted in place as
//             the B operand, dQ^T = K^T dS^T with dS^T through a 2-KiB LDS tile (ds_write_b64 / ds_read_b64_tr_b16).
//             A lane then holds 16 of its token's 32 output values after one v_permlane32_swap per register pair:
//             two 16-byte stores per tile, no LDS round trip. 40 MFMAs per (window, head), ~250 registers: two
//             workgroups per CU.
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

// per-wave LDS of the forward kernel: three [token][32] tiles + token / region tables
struct __attribute__((aligned(16))) WaveLds {
    char q[TILE_BYTES], k[TILE_BYTES], v[TILE_BYTES];
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
#define SWIN_FETCH(WIN)                                                                                      \
    {                                                                                                        \
        win_token(g, (WIN), lane, s_tok, s_region);                                                          \
        const uint4 *src_ = reinterpret_cast<const uint4 *>(qkv + (size_t)s_tok * 3 * C + h * HP);            \
        s_q0 = src_[0]; s_q1 = src_[1]; s_q2 = src_[2]; s_q3 = src_[3];                                      \
        s_k0 = src_[C / 8]; s_k1 = src_[C / 8 + 1]; s_k2 = src_[C / 8 + 2]; s_k3 = src_[C / 8 + 3];          \
        s_v0 = src_[C / 4]; s_v1 = src_[C / 4 + 1]; s_v2 = src_[C / 4 + 2]; s_v3 = src_[C / 4 + 3];          \
    }
#define SWIN_COMMIT()                                                                                        \
    {                                                                                                        \
        L.tok[lane] = s_tok;                                                                                 \
        L.region[lane] = s_region;                                                                           \
        uint4 *d_ = reinterpret_cast<uint4 *>(L.q + lane * HP * 2);                                          \
        d_[0] = s_q0; d_[1] = s_q1; d_[2] = s_q2; d_[3] = s_q3;                                              \
        d_ = reinterpret_cast<uint4 *>(L.k + lane * HP * 2);                                                 \
        d_[0] = s_k0; d_[1] = s_k1; d_[2] = s_k2; d_[3] = s_k3;                                              \
        d_ = reinterpret_cast<uint4 *>(L.v + lane * HP * 2);                                                 \
        d_[0] = s_v0; d_[1] = s_v1; d_[2] = s_v2; d_[3] = s_v3;                                              \
    }
#define SWIN_STAGE_REGS                                                                                      \
    uint4 s_q0, s_q1, s_q2, s_q3, s_k0, s_k1, s_k2, s_k3, s_v0, s_v1, s_v2, s_v3;                          \
    int s_tok, s_region;

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
// the keys. On return p[jt][it] holds the probabilities; lse_item (64 floats of this window and head, or null) gets
// the rows' log-sum-exp.
template <bool SHIFTED>
__device__ __forceinline__ void scores_T(WaveLds &L, const LaneGeom &G, float scale, int lane, bool last_row,
                                         bool last_col, f32x16 (&p)[2][2], float *lse_item) {
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
        // log2-domain log-sum-exp of the query's row: the backward pass rebuilds P = exp2(log2e v - lse) from it
        if (lse_item && lane < 32) lse_item[32 * it + lane] = fmaf(mx, 1.44269504088896341f, __log2f(sum));
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int r = 0; r < 16; ++r) p[jt][it][r] *= inv;
    }
}

template <bool SHIFTED>
__global__ __launch_bounds__(64 * WAVES) void swin_attn_fwd_mfma_kernel(const unsigned short *__restrict__ qkv,
                                                                          const float *__restrict__ table,
                                                                          unsigned short *__restrict__ out,
                                                                          float *__restrict__ lse, MGeom g, float scale,
                                                                          int groups) {
    __shared__ WaveLds lds[WAVES];
    __shared__ float bias_pair[2][NB];
    // A workgroup = two windows x a PAIR of neighbouring heads (as the backward kernel): the two heads' 64-byte row
    // segments are the halves of the same 128-byte lines. (One head per workgroup fetched every line twice: 256 MB
    // beyond L2 per launch for 124 MB of operands, profiles/r04_h_*.)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int hsel = wave & 1, hreal = 2 * (int)blockIdx.y + hsel;
    const bool head_ok = hreal < g.heads;                  // (odd head count: the last pair's second wave computes a copy)
    const int h = head_ok ? hreal : g.heads - 1;
    for (int b = threadIdx.x; b < 2 * NB; b += 64 * WAVES) {
        const int hh = min(2 * (int)blockIdx.y + b / NB, g.heads - 1);
        bias_pair[0][b] = table[(b % NB) * g.heads + hh];
    }
    WaveLds &L = lds[wave];
    const LaneGeom G = lane_geom(bias_pair[hsel], lane);
    const int C = g.heads * HP;
    const int stride = groups * (WAVES / 2);
    const int rounds = (g.nwin + stride - 1) / stride;
    auto window_of = [&](int rd) { return min((rd * groups + (int)blockIdx.x) * (WAVES / 2) + (wave >> 1), g.nwin - 1); };
    SWIN_STAGE_REGS
    SWIN_FETCH(window_of(0))
    for (int rd = 0; rd < rounds; ++rd) {
        const bool live = head_ok && (rd * groups + (int)blockIdx.x) * (WAVES / 2) + (wave >> 1) < g.nwin;
        const int win = window_of(rd);
        const int wloc = win % (g.nwy * g.nwx);
        const bool last_row = wloc / g.nwx == g.nwy - 1, last_col = wloc % g.nwx == g.nwx - 1;
        __syncthreads();                                   // previous round's LDS reads are done
        SWIN_COMMIT()
        __syncthreads();
        if (rd + 1 < rounds) SWIN_FETCH(window_of(rd + 1))      // in flight under the MFMAs
        f32x16 p[2][2];
        scores_T<SHIFTED>(L, G, scale, lane, last_row, last_col, p,
                          lse && live ? lse + ((size_t)h * g.nwin + win) * NTOK : nullptr);
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

// ---- backward ----------------------------------------------------------------------------------------------
// The LDS executes one wave's instructions in the order they were issued: a read queued behind a write of the same wave
// sees it, a write queued behind a read cannot overtake it. Between the wave-private tiles' writers and readers only
// the COMPILER has to be kept from reordering (differently typed accesses do not alias for it); no s_waitcnt: the
// waits the compiler places in front of the first use of each loaded value are the only ones.
#define SWIN_LDS_ORDER() asm volatile("" ::: "memory")
// Per-wave LDS of the backward kernel. The four [token][32] tiles are chunk-swizzled (16-byte chunk c of row r at slot
// c ^ ((r >> 2) & 3)): the operand reads of 32 rows x 16 bytes (64-byte pitch) then fall on all 64 banks, and the
// transposing reads (4 rows x 32 bytes per 16 lanes) stay conflict-free. The dS tile is [32 keys][32 queries] with
// the same swizzle.
struct __attribute__((aligned(16))) BwdLds {
    char q[TILE_BYTES], k[TILE_BYTES], v[TILE_BYTES], g[TILE_BYTES];
    char t[32 * HP * 2];
    float nlse[NTOK], delta[NTOK];                        // per query: -(log-sum-exp, log2 units), rowsum(dO * O)
    int tok[2][NTOK];                                     // token inside the image, of this item and of the one being staged
};

struct BwdLane {                                          // per-lane address pieces, fixed for the whole kernel
    int row[2];                                           // operand rows: + 64 * r0 ; k-step s
    int tr[2];                                            // transposing reads of a token tile: + 64 * t0 (t0 % 16 == 0); rows +0 / +8
    int tw[4];                                            // dS tile writes: key row, queries 8 a + 4 h .. + 3
    int hb[2];                                            // dS tile as the A operand of the bias-gradient product
    int quad;                                             // four lanes per row: row (lane >> 2) [+ 16 k], chunk lane & 3
};
__device__ __forceinline__ BwdLane bwd_lane(int lane) {
    BwdLane A;
    const int h = lane >> 5, r = lane & 31;
    for (int s = 0; s < 2; ++s) A.row[s] = r * 64 + (((2 * s + h) ^ ((r >> 2) & 3)) << 4);
    const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    const int trow = 4 * (g >> 1) + q, chunk = 2 * (g & 1) + (p >> 1);
    A.tr[0] = trow * 64 + ((chunk ^ (g >> 1)) << 4) + 8 * (p & 1);
    A.tr[1] = (trow + 8) * 64 + ((chunk ^ ((g >> 1) | 2)) << 4) + 8 * (p & 1);
    for (int a = 0; a < 4; ++a) A.tw[a] = r * 64 + ((a ^ ((r >> 2) & 3)) << 4) + 8 * h;
    // bias-gradient product (16x16x32): row m = lane & 15 = (query row iy' = m >> 2, key row jy' = m & 3) of the tile,
    // k = (key column jx = 4 jxh + g, query column ix = e): 16 bytes of dS-tile row 8 jy' + jx at queries 8 iy' ..
    for (int jxh = 0; jxh < 2; ++jxh) {
        const int m = lane & 15, j = 8 * (m & 3) + 4 * jxh + g;
        A.hb[jxh] = j * 64 + (((m >> 2) ^ ((j >> 2) & 3)) << 4);
    }
    A.quad = (lane >> 2) * 64 + (((lane & 3) ^ ((lane >> 4) & 3)) << 4);
    return A;
}
__device__ __forceinline__ bf16x8 lds_frag(const char *p) { return *reinterpret_cast<const bf16x8 *>(p); }
__device__ __forceinline__ bf16x8 lds_tr_frag(const char *a, const char *b) {
    const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s *)a);
    const v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s *)b);
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}
__device__ __forceinline__ unsigned pack2(float a, float b) {
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    const bf2 v = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, v);
}
// A transposed result tile (rows = head dim on the registers: dims 8 a + 4 h + 0..3 in register quad a; column = token
// (lane & 31) of the tile) on its way out: through the (then free) dS tile in the swizzled layout, 8 bytes per lane and
// quad in, whole 64-byte rows out (four lanes per row, 16 rows per store instruction). tok: the tile's 32 tokens.
__device__ __forceinline__ void store_tile_T(char *stage, const int (&tw)[4], int quad, const int *tok, const f32x16 &o,
                                             float mul, int lane, char *dst, unsigned row_bytes) {
#pragma unroll
    for (int a = 0; a < 4; ++a)
        *reinterpret_cast<uint2 *>(stage + tw[a]) = uint2{pack2(o[4 * a] * mul, o[4 * a + 1] * mul),
                                                          pack2(o[4 * a + 2] * mul, o[4 * a + 3] * mul)};
    SWIN_LDS_ORDER();
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const uint4 v = *reinterpret_cast<const uint4 *>(stage + 1024 * pass + quad);
        const unsigned off = (unsigned)tok[16 * pass + (lane >> 2)] * row_bytes + 16u * (lane & 3);
        *reinterpret_cast<uint4 *>(dst + off) = v;
    }
    SWIN_LDS_ORDER();
}
__device__ __forceinline__ float dot8(uint4 a, uint4 b, float acc) {
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, a.x), __builtin_bit_cast(bf2, b.x), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, a.y), __builtin_bit_cast(bf2, b.y), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, a.z), __builtin_bit_cast(bf2, b.z), acc, false);
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, a.w), __builtin_bit_cast(bf2, b.w), acc, false);
}

// Rows of a (window, head) as they travel: FOUR lanes per 64-byte token row (lane = 4 * row-in-group + chunk), 16 rows
// per instruction -- every request is a whole row segment (one lane per row = 64 quarter-filled requests per
// instruction: the loads and above all the stores were request-bound, not byte-bound, that way).
// Token of row 16 k + (lane >> 2) of window `win`: window row iy = 2 k + (lane >> 5), column ix = (lane >> 2) & 7.
// Addresses are a per-item SCALAR base (the window's image) plus a 32-bit lane offset (token inside the image: below
// 2^31 / row bytes by `check`): one address register per load instead of two, immediate offsets between q, k and v.
__device__ __forceinline__ void quad_tokens(const MGeom &g, int win, int lane, int &image, int (&loc)[4]) {
    const int per = g.nwy * g.nwx;
    image = win / per;
    const int w = win - image * per;
    const int wy = w / g.nwx, wx = w - wy * g.nwx;
    int ox = wx * WS + ((lane >> 2) & 7) + g.shift;
    if (ox >= g.W) ox -= g.W;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int oy = wy * WS + 2 * k + (lane >> 5) + g.shift;
        if (oy >= g.H) oy -= g.H;
        loc[k] = oy * g.W + ox;
    }
}
// (plain named registers: arrays of staged rows ended up in scratch memory)
#define SWIN_BWD_FETCH_ROW(K)                                                                                \
    {                                                                                                        \
        const unsigned oq_ = (unsigned)tok_[K] * (unsigned)(6 * C) + 16u * (lf_ & 3);                        \
        const unsigned og_ = (unsigned)tok_[K] * (unsigned)(2 * C) + 16u * (lf_ & 3);                        \
        s_q##K = *reinterpret_cast<const uint4 *>(qb_ + oq_);                                                \
        s_k##K = *reinterpret_cast<const uint4 *>(kb_ + oq_);                                                \
        s_v##K = *reinterpret_cast<const uint4 *>(vb_ + oq_);                                                \
        s_g##K = *reinterpret_cast<const uint4 *>(gb_ + og_);                                                \
        s_o##K = *reinterpret_cast<const uint4 *>(ob_ + og_);                                                \
    }
#define SWIN_BWD_FETCH(WIN)                                                                                  \
    {                                                                                                        \
        /* an opaque copy of the lane: what the addresses derive from it is recomputed per item (a dozen */  \
        /* instructions) instead of living in registers across the whole loop, where there are none to spare */ \
        int lf_ = lane;                                                                                      \
        asm volatile("" : "+v"(lf_));                                                                        \
        int tok_[4], image_;                                                                                 \
        quad_tokens(g, (WIN), lf_, image_, tok_);                                                            \
        image_ = __builtin_amdgcn_readfirstlane(image_);                                                     \
        /* lane (row-in-group r, chunk c) files what belongs to row 16 c + r; the table of the item in flight is */ \
        /* still read by its last stores: two tables */                                                      \
        {                                                                                                    \
            const int c_ = lf_ & 3;                                                                          \
            L.tok[tok_sel ^ 1][16 * c_ + (lf_ >> 2)] = c_ == 0 ? tok_[0] : c_ == 1 ? tok_[1] : c_ == 2 ? tok_[2] : tok_[3]; \
        }                                                                                                    \
        const size_t first_ = (size_t)image_ * g.H * g.W;                                                    \
        const char *qb_ = reinterpret_cast<const char *>(qkv + first_ * 3 * C + h * HP);                     \
        const char *kb_ = qb_ + 2 * C, *vb_ = qb_ + 4 * C;                                                   \
        const char *gb_ = reinterpret_cast<const char *>(dout + first_ * C + h * HP);                        \
        const char *ob_ = reinterpret_cast<const char *>(out + first_ * C + h * HP);                         \
        SWIN_BWD_FETCH_ROW(0) SWIN_BWD_FETCH_ROW(1) SWIN_BWD_FETCH_ROW(2) SWIN_BWD_FETCH_ROW(3)              \
        const char *lb_ = reinterpret_cast<const char *>(lse + ((size_t)h * g.nwin + (WIN)) * NTOK);         \
        s_lse = *reinterpret_cast<const float *>(lb_ + 4u * lf_);                                            \
    }
#define SWIN_BWD_COMMIT_ROW(K)                                                                               \
    {                                                                                                        \
        *reinterpret_cast<uint4 *>(L.q + 1024 * K + A.quad) = s_q##K;                                        \
        *reinterpret_cast<uint4 *>(L.k + 1024 * K + A.quad) = s_k##K;                                        \
        *reinterpret_cast<uint4 *>(L.v + 1024 * K + A.quad) = s_v##K;                                        \
        *reinterpret_cast<uint4 *>(L.g + 1024 * K + A.quad) = s_g##K;                                        \
        d##K = quad_sum(dot8(s_g##K, s_o##K, 0.f));                                                          \
    }

__device__ __forceinline__ float quad_sum(float x) {      // sum over the four lanes of a quad, in all four
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xf, 0xf, false));
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xf, 0xf, false));
    return x;
}

template <bool SHIFTED>
__global__ __launch_bounds__(64 * WAVES, 2) void swin_attn_bwd_mfma_kernel(
    const unsigned short *__restrict__ qkv, const float *__restrict__ table, const unsigned short *__restrict__ out,
    const float *__restrict__ lse, const unsigned short *__restrict__ dout, unsigned short *__restrict__ dqkv,
    float *__restrict__ dtable, MGeom g, float scale, int groups) {
    __shared__ BwdLds lds[WAVES];
    // A workgroup = two windows x a PAIR of neighbouring heads: the two heads' 64-byte row segments are the halves of
    // the same 128-byte lines of qkv / out / dout / dqkv and travel within microseconds of each other. (One head per
    // workgroup left half of every fetched line to be fetched again by another workgroup, usually after its eviction.)
    __shared__ float bias_pair[2][NB], bins_pair[2][NB];  // bias / scale: what the score accumulators start from
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int bx = (int)blockIdx.x, by = (int)blockIdx.y;
    const int hsel = wave & 1, h = 2 * by + hsel;
    const float inv_scale = 1.0f / scale;
    for (int b = threadIdx.x; b < 2 * NB; b += 64 * WAVES) {
        const int hh = 2 * by + b / NB;
        bias_pair[0][b] = hh < g.heads ? table[(b % NB) * g.heads + hh] * inv_scale : 0.f;
        bins_pair[0][b] = 0.f;
    }
    __syncthreads();
    float *bias_s = bias_pair[hsel], *bins = bins_pair[hsel];
    BwdLds &L = lds[wave];
    const BwdLane A = bwd_lane(lane);
    const int hl = lane >> 5;
    // bias bin of element (it, jt, r): c2(it, r) + 4 hl + 112 - (jy * 15 + jx), key j = 32 jt + (lane & 31)
    int bin_base[2];
#pragma unroll
    for (int jt = 0; jt < 2; ++jt) {
        const int j = 32 * jt + (lane & 31);
        bin_base[jt] = 4 * hl + 112 - ((j >> 3) * 15 + (j & 7));
    }
    const bool key_right = (lane & 7) >= 4;
    const float cs = scale * 1.44269504088896341f;        // scores -> log2 units
    const float pen_acc = -100.0f * inv_scale;
    // The bias-table gradient sums dS over every window the wave walks. Element-wise that is a 64 x 64 matrix = 64
    // registers per lane (what kept the first version of this kernel at one wave per SIMD). The bin of (query i, key j)
    // is ((iy - jy + 7), (ix - jx + 7)): the sum over the column pairs (ix, jx) with ix - jx = dx is a product of the
    // (bf16) dS tile, read back from LDS as [(iy', jy')][(jx, ix)], with a constant 0 / 1 matrix [(jx, ix)][dx] --
    // two 16x16x32 MFMAs per tile into 4 accumulator registers per (it, jt), 16 in all.
    using f32x4 = __attribute__((ext_vector_type(4))) float;
    f32x4 hsum[2][2] = {{{0}, {0}}, {{0}, {0}}};
    bf16x8 sel[2];
#pragma unroll
    for (int jxh = 0; jxh < 2; ++jxh)
#pragma unroll
        for (int e = 0; e < 8; ++e) sel[jxh][e] = (e - (4 * jxh + (lane >> 4)) + 7 == (lane & 15)) ? (__bf16)1.0f : (__bf16)0.0f;
    const int C = g.heads * HP;
    const int stride = groups * (WAVES / 2);
    uint4 s_q0, s_q1, s_q2, s_q3, s_k0, s_k1, s_k2, s_k3, s_v0, s_v1, s_v2, s_v3, s_g0, s_g1, s_g2, s_g3, s_o0, s_o1, s_o2,
        s_o3;
    float s_lse;
    int tok_sel = 1;                                      // the first fetch files its tokens in table 0
    int win = h < g.heads ? bx * (WAVES / 2) + (wave >> 1) : g.nwin;      // (odd head count: an idle wave)
    if (win < g.nwin) SWIN_BWD_FETCH(win)
    // (the first item's rows are waited for HERE: the waits the compiler places at the top of the loop are the stricter
    // of its two entries, and entering with loads as the youngest memory operations would make them vmcnt(0) -- which,
    // on the way round, also waits for the eight stores issued after the next item's loads)
    __builtin_amdgcn_s_waitcnt(0x0F70);                   // vmcnt(0)
    for (; win < g.nwin; win += stride) {
        const int wloc = win % (g.nwy * g.nwx);
        const bool last_row = wloc / g.nwx == g.nwy - 1, last_col = wloc % g.nwx == g.nwx - 1;
        const bool masked = SHIFTED && (last_row || last_col);
        char *const dq_image = reinterpret_cast<char *>(dqkv + (size_t)(win / (g.nwy * g.nwx)) * g.H * g.W * 3 * C + h * HP);
        // ---- commit: staged rows -> swizzled LDS tiles; delta, -lse, token table -------------------------------
        {
            asm volatile("" ::: "memory");
            float d0, d1, d2, d3;
            SWIN_BWD_COMMIT_ROW(0) SWIN_BWD_COMMIT_ROW(1) SWIN_BWD_COMMIT_ROW(2) SWIN_BWD_COMMIT_ROW(3)
            // lane (row-in-group r, chunk c) files what belongs to row 16 c + r
            const int c = lane & 3;
            L.delta[16 * c + (lane >> 2)] = c == 0 ? d0 : c == 1 ? d1 : c == 2 ? d2 : d3;
            tok_sel ^= 1;
            L.nlse[lane] = -s_lse;
            SWIN_LDS_ORDER();
        }
        f32x16 dvt[2] = {{0}, {0}}, dkt[2] = {{0}, {0}};
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            f32x16 dqt = {0};
#pragma unroll
            for (int jt = 0; jt < 2; ++jt) {
                // S (on top of bias / scale [+ mask]) and dP for queries of tile it, keys of tile jt
                f32x16 s, dp = {0};
                {
                    const float *bl = bias_s + bin_base[jt];
#pragma unroll
                    for (int r = 0; r < 16; ++r) s[r] = bl[c2(it, r)];
                    if (masked) {
                        const float pen = ((last_row && it != jt) || (last_col && (hl == 1) != key_right)) ? pen_acc : 0.f;
#pragma unroll
                        for (int r = 0; r < 16; ++r) s[r] += pen;
                    }
                }
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    s = mfma(lds_frag(L.q + 2048 * it + A.row[ks]), lds_frag(L.k + 2048 * jt + A.row[ks]), s);
                    dp = mfma(lds_frag(L.g + 2048 * it + A.row[ks]), lds_frag(L.v + 2048 * jt + A.row[ks]), dp);
                }
                // P = exp2(cs S - lse), dS = P (dP - delta); rows of register quad a: queries 32 it + 8 a + 4 hl + 0..3
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    const float4 n4 = *reinterpret_cast<const float4 *>(L.nlse + 32 * it + 8 * a + 4 * hl);
                    const float4 d4 = *reinterpret_cast<const float4 *>(L.delta + 32 * it + 8 * a + 4 * hl);
                    const float nl[4] = {n4.x, n4.y, n4.z, n4.w}, dl[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const int r = 4 * a + c;
                        const float p = __builtin_amdgcn_exp2f(fmaf(s[r], cs, nl[c]));
                        const float ds = p * (dp[r] - dl[c]);
                        s[r] = p;
                        dp[r] = ds;
                    }
                }
                const bf16x8 pf[2] = {acc_frag(s, 0), acc_frag(s, 1)}, df[2] = {acc_frag(dp, 0), acc_frag(dp, 1)};
                // dS^T on its way to the dQ product: lane = key, four consecutive queries per register quad
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const uint4 w = __builtin_bit_cast(uint4, df[ks]);
                    *reinterpret_cast<uint2 *>(L.t + A.tw[2 * ks]) = uint2{w.x, w.y};
                    *reinterpret_cast<uint2 *>(L.t + A.tw[2 * ks + 1]) = uint2{w.z, w.w};
                }
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const int t0 = 64 * (32 * it + 16 * ks);
                    dvt[jt] = mfma(lds_tr_frag(L.g + t0 + A.tr[0], L.g + t0 + A.tr[1]), pf[ks], dvt[jt]);
                    dkt[jt] = mfma(lds_tr_frag(L.q + t0 + A.tr[0], L.q + t0 + A.tr[1]), df[ks], dkt[jt]);
                }
                SWIN_LDS_ORDER();
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const int t0 = 64 * (32 * jt + 16 * ks);
                    dqt = mfma(lds_tr_frag(L.k + t0 + A.tr[0], L.k + t0 + A.tr[1]),
                               lds_tr_frag(L.t + 1024 * ks + A.tr[0], L.t + 1024 * ks + A.tr[1]), dqt);
                }
                // bias gradient: hsum[it][jt][(iy', jy')][dx] += sum over (ix, jx) of dS [ix - jx == dx]
#pragma unroll
                for (int jxh = 0; jxh < 2; ++jxh)
                    hsum[it][jt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_frag(L.t + A.hb[jxh]), sel[jxh], hsum[it][jt], 0, 0, 0);
                SWIN_LDS_ORDER();
            }
            store_tile_T(L.t, A.tw, A.quad, L.tok[tok_sel] + 32 * it, dqt, scale, lane, dq_image, 6u * C);
        }
        // The next item's rows start travelling BEFORE this item's last eight stores: the memory counter is in order,
        // so a wait for a load also waits for every store issued before it, not for the ones behind it.
        if (win + stride < g.nwin) SWIN_BWD_FETCH(win + stride)
        store_tile_T(L.t, A.tw, A.quad, L.tok[tok_sel], dkt[0], scale, lane, dq_image + 2 * C, 6u * C);
        store_tile_T(L.t, A.tw, A.quad, L.tok[tok_sel] + 32, dkt[1], scale, lane, dq_image + 2 * C, 6u * C);
        store_tile_T(L.t, A.tw, A.quad, L.tok[tok_sel], dvt[0], 1.0f, lane, dq_image + 4 * C, 6u * C);
        store_tile_T(L.t, A.tw, A.quad, L.tok[tok_sel] + 32, dvt[1], 1.0f, lane, dq_image + 4 * C, 6u * C);
    }
    // table gradient: row m = 4 (lane >> 4) + c of hsum[it][jt] is (iy' = lane >> 4, jy' = c), column lane & 15 = dx + 7
    if ((lane & 15) < 15) {
#pragma unroll
        for (int it = 0; it < 2; ++it)
#pragma unroll
            for (int jt = 0; jt < 2; ++jt)
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    atomicAdd(&bins[(4 * (it - jt) + (lane >> 4) - c + 7) * 15 + (lane & 15)], hsum[it][jt][c]);
    }
    __syncthreads();
    for (int b = threadIdx.x; b < 2 * NB; b += 64 * WAVES) {
        const int hh = 2 * by + b / NB;
        if (hh < g.heads) atomicAdd(dtable + (b % NB) * g.heads + hh, bins_pair[0][b]);
    }
}

inline int check(int B, int H, int W, int heads, int shift) {
    SEI_REQUIRE(B > 0 && H >= WS && W >= WS && H % WS == 0 && W % WS == 0 && heads > 0 && heads <= 64);
    SEI_REQUIRE(shift == 0 || shift == WS / 2);           // SwinIR: window 8, shift 0 / 4 (the mask logic relies on it)
    SEI_REQUIRE((size_t)B * H * W < ((size_t)1 << 31));
    return SEI_OK;
}

}  // namespace

extern "C" int sei_swin_attn_fwd_bf16(const uint16_t *qkv, const float *table, uint16_t *out, float *lse, int B, int H,
                                      int W, int heads, int shift, float scale, void *stream) {
    SEI_REQUIRE(qkv && table && out);
    SEI_REQUIRE((((uintptr_t)qkv | (uintptr_t)out) & 15) == 0);
    if (int rc = check(B, H, W, heads, shift)) return rc;
    MGeom g{H, W, H / WS, W / WS, shift, heads, B * (H / WS) * (W / WS)};
    // two workgroups per CU, each two windows x two heads at a time
    const int pairs = (heads + 1) / 2;
    const int want = (g.nwin + 1) / 2, cap = 512 / pairs > 0 ? 512 / pairs : 1;
    const int groups = want > cap ? cap : want;
    if (shift)
        hipLaunchKernelGGL(swin_attn_fwd_mfma_kernel<true>, dim3((unsigned)groups, (unsigned)pairs), dim3(64 * WAVES), 0,
                           (hipStream_t)stream, qkv, table, out, lse, g, scale, groups);
    else
        hipLaunchKernelGGL(swin_attn_fwd_mfma_kernel<false>, dim3((unsigned)groups, (unsigned)pairs), dim3(64 * WAVES), 0,
                           (hipStream_t)stream, qkv, table, out, lse, g, scale, groups);
    return sei_launch_status();
}

extern "C" int sei_swin_attn_bwd_bf16(const uint16_t *qkv, const float *table, const uint16_t *out, const float *lse,
                                      const uint16_t *dout, uint16_t *dqkv, float *dtable, int B, int H, int W, int heads,
                                      int shift, float scale, void *stream) {
    SEI_REQUIRE(qkv && table && out && lse && dout && dqkv && dtable && scale > 0.f);
    SEI_REQUIRE((((uintptr_t)qkv | (uintptr_t)out | (uintptr_t)dout | (uintptr_t)dqkv) & 15) == 0);
    if (int rc = check(B, H, W, heads, shift)) return rc;
    SEI_REQUIRE((size_t)H * W * heads * HP * 6 < ((size_t)1 << 32));      // 32-bit lane offsets inside one image
    MGeom g{H, W, H / WS, W / WS, shift, heads, B * (H / WS) * (W / WS)};
    // two workgroups per CU (80 KB of LDS, <= 256 registers), each two windows x two heads at a time
    const int pairs = (heads + 1) / 2;
    const int want = (g.nwin + 1) / 2, cap = 512 / pairs > 0 ? 512 / pairs : 1;
    const int groups = want > cap ? cap : want;
    const dim3 grid((unsigned)groups, (unsigned)pairs);
    if (shift)
        hipLaunchKernelGGL(swin_attn_bwd_mfma_kernel<true>, grid, dim3(64 * WAVES), 0, (hipStream_t)stream, qkv, table, out,
                           lse, dout, dqkv, dtable, g, scale, groups);
    else
        hipLaunchKernelGGL(swin_attn_bwd_mfma_kernel<false>, grid, dim3(64 * WAVES), 0, (hipStream_t)stream, qkv, table, out,
                           lse, dout, dqkv, dtable, g, scale, groups);
    return sei_launch_status();
}
