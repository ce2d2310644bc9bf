// Depthwise 7x7 convolution of the ConvBlocks for gfx950, NHWC float32
// (reference: src/models/convolutional.py:36-38,46 -- nn.Conv2d(C, C, 7, padding=3, groups=C)).
//
//   sei_dwconv7_fwd         forward; with flip=1 and a residual it is the data gradient of the block
//   sei_dwconv7_bwd_weight  weight + bias gradient, two deterministic stages (partials, then a fold)
//
// Every level of the default network moves the same 18.9 MB per tensor (C quadruples as H*W quarters), so
// all three regimes below matter equally:
//   * tiled    (H, W >= 8, C % 4 == 0): a workgroup stages a (th+6) x (tw+6) x 32-channel halo tile in LDS
//              with 16-byte coalesced loads (all in flight together), then each thread slides a 7x7 register
//              window along one tile row: 7 LDS reads + 49 FMA per output, lanes = channels (conflict-free).
//   * whole    (3x3 and 6x6 images, the two deepest levels): one thread holds the whole image of one channel
//              in registers; only the taps that can touch the image are evaluated.
//   * generic  any other shape: sliding window straight from global memory.
// The accumulation order per output (bias, then taps row-major) is the same in all three, so they agree
// bit for bit on finite data.
#include "sei_common.h"

namespace {

constexpr int DW_THREADS = 256;
constexpr int DW_SEG = 16;           // generic kernel: output columns per worker segment unless the caller says

inline unsigned capped_grid(size_t work_items, int per_block, unsigned cap) {
    size_t g = sei_ceil_div(work_items, (size_t)per_block);
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (unsigned)g;
}

// =================================================================================================
// generic: one thread = one channel; a "worker" (Cc consecutive threads) walks a segment of one output row
// with a 7x7 register window that slides by one column per step (7 new loads + 49 FMA per output).
// =================================================================================================
template <bool WEIGHT_GRAD>
__global__ __launch_bounds__(DW_THREADS) void dwconv7_kernel(
    const float *__restrict__ x, const float *__restrict__ w, const float *__restrict__ bias,
    const float *__restrict__ res, float res_scale, float *__restrict__ y, const float *__restrict__ gy,
    float *__restrict__ gw, int B, int H, int W, int C, int flip, int Cc, int nseg, int total_rowsegs,
    int seg_len) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int workers = DW_THREADS / Cc;
    const int cl = threadIdx.x % Cc, worker = threadIdx.x / Cc;
    const int c = blockIdx.y * Cc + cl;
    const bool c_ok = c < C;

    float wr[49];
    float acc_w[49];
    float acc_b = 0.f;
    if (!WEIGHT_GRAD) {
#pragma unroll
        for (int t = 0; t < 49; ++t) wr[t] = c_ok ? w[(size_t)c * 49 + (flip ? 48 - t : t)] : 0.f;
    } else {
#pragma unroll
        for (int t = 0; t < 49; ++t) acc_w[t] = 0.f;
    }
    const float bv = (!WEIGHT_GRAD && bias && c_ok) ? bias[c] : 0.f;

    for (int rs = blockIdx.x * workers + worker; rs < total_rowsegs; rs += gridDim.x * workers) {
        if (!c_ok) continue;
        const int seg = rs % nseg;
        const int bi = rs / nseg;
        const int i = bi % H, b = bi / H;
        const int j0 = seg * seg_len, j1 = min(W, j0 + seg_len);
        const float *xb = x + (size_t)b * H * W * C + c;
        float win[7][7];   // win[di][slot]
#pragma unroll
        for (int dj = 0; dj < 6; ++dj) {
            const int jj = j0 - 3 + dj;
#pragma unroll
            for (int di = 0; di < 7; ++di) {
                const int ii = i - 3 + di;
                win[di][dj] = (ii >= 0 && ii < H && jj >= 0 && jj < W) ? xb[((size_t)ii * W + jj) * C] : 0.f;
            }
        }
        for (int jb = j0; jb < j1; jb += 7) {
#pragma unroll
            for (int s = 0; s < 7; ++s) {
                const int j = jb + s;
                if (j < j1) {
                    const int jj = j + 3;
#pragma unroll
                    for (int di = 0; di < 7; ++di) {
                        const int ii = i - 3 + di;
                        win[di][(s + 6) % 7] =
                            (ii >= 0 && ii < H && jj < W) ? xb[((size_t)ii * W + jj) * C] : 0.f;
                    }
                    const size_t o = (((size_t)b * H + i) * W + j) * C + c;
                    if (!WEIGHT_GRAD) {
                        float a = bv;
#pragma unroll
                        for (int di = 0; di < 7; ++di)
#pragma unroll
                            for (int dj = 0; dj < 7; ++dj) a = fmaf(wr[di * 7 + dj], win[di][(s + dj) % 7], a);
                        if (res) a = fmaf(res_scale, res[o], a);
                        y[o] = a;
                    } else {
                        const float g = gy[o];
                        acc_b += g;
#pragma unroll
                        for (int di = 0; di < 7; ++di)
#pragma unroll
                            for (int dj = 0; dj < 7; ++dj)
                                acc_w[di * 7 + dj] = fmaf(g, win[di][(s + dj) % 7], acc_w[di * 7 + dj]);
                    }
                }
            }
        }
    }
    if (WEIGHT_GRAD) {
        float *red = smem;   // [worker][50][Cc]
#pragma unroll
        for (int t = 0; t < 49; ++t) red[(worker * 50 + t) * Cc + cl] = acc_w[t];
        red[(worker * 50 + 49) * Cc + cl] = acc_b;
        __syncthreads();
        float *part = gw + (size_t)blockIdx.x * 50 * C;      // partial sums of this workgroup: part[bx][t][c]
        for (int e = threadIdx.x; e < 50 * Cc; e += DW_THREADS) {
            const int t = e / Cc, c2 = e % Cc;
            const int cg = blockIdx.y * Cc + c2;
            if (cg >= C) continue;
            float s = 0.f;
            for (int wk = 0; wk < workers; ++wk) s += red[(wk * 50 + t) * Cc + c2];
            part[(size_t)t * C + cg] = s;
        }
    }
}

// =================================================================================================
// tiled: LDS halo tile, 32 channels x 8 row slots per workgroup
// =================================================================================================
constexpr int DT_CC = 32;                          // channels per workgroup = lanes of a half wave
constexpr int DT_ROWS = DW_THREADS / DT_CC;        // 8 row slots
constexpr int DT_MAXW = 16;                        // widest tile
constexpr int DT_LW = DT_MAXW + 6;                 // staged columns
constexpr int DT_RS = DT_LW * DT_CC + 32;          // LDS row stride (floats); +32 puts the two rows of a
                                                   // wave on different banks
constexpr int DT_TILE = (DT_ROWS + 6) * DT_RS;     // 10304 floats = 41 KB
constexpr int DT_RED = DT_ROWS * 50 * DT_CC;       // weight-gradient reduction scratch, 12800 floats
constexpr int DT_SWEEPS = ((DT_ROWS + 6) * DT_LW + 31) / 32;   // 10 sweeps of 32 pixels cover the largest tile
static_assert(DT_MAXW + 6 >= 11, "a sweep of 32 pixels wraps at most three staged rows");

struct DwTiling {
    int th, tw, tiles_i, tiles_j;
    size_t ntiles;
};
inline DwTiling dw_tiling(int B, int H, int W) {
    DwTiling t;
    t.th = (H % 8 == 0) ? 8 : ((H % 6 == 0) ? 6 : ((H % 7 == 0) ? 7 : 8));
    t.tw = (W % 16 == 0) ? 16 : ((W % 12 == 0) ? 12 : ((W % 14 == 0) ? 14 : 16));
    t.tiles_i = (int)sei_ceil_div(H, t.th);
    t.tiles_j = (int)sei_ceil_div(W, t.tw);
    t.ntiles = (size_t)B * t.tiles_i * t.tiles_j;
    return t;
}

#ifndef SEI_DW_WGRAD_OCC
#define SEI_DW_WGRAD_OCC 3          // workgroups per CU the weight-gradient instantiation is compiled for (A/B builds: 2)
#endif
template <bool WEIGHT_GRAD>
__global__ __launch_bounds__(DW_THREADS, WEIGHT_GRAD ? SEI_DW_WGRAD_OCC : 3) void dwconv7_tiled_kernel(
    const float *__restrict__ x, const float *__restrict__ w, const float *__restrict__ bias,
    const float *__restrict__ res, float res_scale, float *__restrict__ y, const float *__restrict__ gy,
    float *__restrict__ part, int H, int W, int C, int flip, int th, int tw, int tiles_i, int tiles_j,
    int ntiles, int tiles_per_block) {
    __shared__ __attribute__((aligned(16))) float lds[WEIGHT_GRAD ? (DT_RED > DT_TILE ? DT_RED : DT_TILE) : DT_TILE];
    const int cl = threadIdx.x & (DT_CC - 1), tr = threadIdx.x >> 5;
    const int c0 = blockIdx.y * DT_CC, c = c0 + cl;
    const bool c_ok = c < C;

    float wr[49];
    float acc_w[49];
    float acc_b = 0.f;
    if (!WEIGHT_GRAD) {
        // The 32 x 49 weights of the channel group are one contiguous run of w: fetched coalesced into LDS, then each
        // lane picks its channel's row (stride 49 floats: odd, conflict-free).  Read straight from global memory the 49
        // per-lane loads are 32-line gathers, ~3 us per workgroup that does ~5 us of arithmetic.
        const int nw = min(DT_CC, C - c0) * 49;
        for (int e = threadIdx.x; e < nw; e += DW_THREADS) lds[e] = w[(size_t)c0 * 49 + e];
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 49; ++t) wr[t] = c_ok ? lds[cl * 49 + (flip ? 48 - t : t)] : 0.f;
        __syncthreads();
    } else {
#pragma unroll
        for (int t = 0; t < 49; ++t) acc_w[t] = 0.f;
    }
    const float bv = (!WEIGHT_GRAD && bias && c_ok) ? bias[c] : 0.f;
    const int lw = tw + 6;

    const int t_begin = blockIdx.x * tiles_per_block, t_end = min(ntiles, t_begin + tiles_per_block);
    for (int tile = t_begin; tile < t_end; ++tile) {
        const int tj = tile % tiles_j, ti = (tile / tiles_j) % tiles_i, b = tile / (tiles_j * tiles_i);
        const int i0 = ti * th, j0 = tj * tw;
        const float *xb = x + (size_t)b * H * W * C;
        if (tile != t_begin) __syncthreads();                    // the previous tile's readers are done
        {   // Stage the halo tile (zero outside the image): thread = (channel quad q, pixel), 32 pixels a sweep.
            // All DT_SWEEPS loads are issued before the first LDS store (clamped addresses keep them
            // unconditional), so a workgroup has its whole tile in flight at once.
            const int q = threadIdx.x & 7;
            const bool q_ok = c0 + 4 * q < C;                    // C % 4 == 0: a quad is all in or all out
            const float *xq = xb + (q_ok ? c0 + 4 * q : 0);
            int lc = threadIdx.x >> 3, lr = 0;
            {
                const int k = (lc >= lw) + (lc >= 2 * lw);
                lc -= k * lw;
                lr += k;
            }
            float4 v[DT_SWEEPS];
            int off[DT_SWEEPS];
#pragma unroll
            for (int it = 0; it < DT_SWEEPS; ++it) {
                const int ii = i0 - 3 + lr, jj = j0 - 3 + lc;
                const bool live = lr < th + 6;
                const bool inside = live && q_ok && ii >= 0 && ii < H && jj >= 0 && jj < W;
                const int ic = min(max(ii, 0), H - 1), jc = min(max(jj, 0), W - 1);
                v[it] = *reinterpret_cast<const float4 *>(xq + ((size_t)ic * W + jc) * C);
                off[it] = live ? ((lr * DT_RS + lc * DT_CC + 4 * q) << 1) | (inside ? 1 : 0) : -1;
                lc += DW_THREADS / 8;
                const int k = (lc >= lw) + (lc >= 2 * lw) + (lc >= 3 * lw);
                lc -= k * lw;
                lr += k;
            }
#pragma unroll
            for (int it = 0; it < DT_SWEEPS; ++it)
                if (off[it] >= 0)
                    *reinterpret_cast<float4 *>(lds + (off[it] >> 1)) =
                        (off[it] & 1) ? v[it] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        __syncthreads();
        const int i = i0 + tr;
        if (tr < th && i < H && c_ok) {
            const float *trow = lds + tr * DT_RS + cl;           // window origin: image (i-3, j0-3)
            const int jn = min(tw, W - j0);                      // outputs of this row run
            const size_t obase = (((size_t)b * H + i) * W + j0) * C + c;
            float gv[DT_MAXW];                                   // the run's gy (weight grad) or residual values
            if (WEIGHT_GRAD) {
#pragma unroll
                for (int jl = 0; jl < DT_MAXW; ++jl) gv[jl] = (jl < jn) ? gy[obase + (size_t)jl * C] : 0.f;
            } else if (res) {
#pragma unroll
                for (int jl = 0; jl < DT_MAXW; ++jl) gv[jl] = (jl < jn) ? res[obase + (size_t)jl * C] : 0.f;
            }
            float win[7][7];
#pragma unroll
            for (int dj = 0; dj < 6; ++dj)
#pragma unroll
                for (int di = 0; di < 7; ++di) win[di][dj] = trow[di * DT_RS + dj * DT_CC];
#pragma unroll
            for (int jb = 0; jb < 21; jb += 7) {
#pragma unroll
                for (int s = 0; s < 7; ++s) {
                    const int jl = jb + s;
                    if (jl < DT_MAXW && jl < jn) {
#pragma unroll
                        for (int di = 0; di < 7; ++di) win[di][(s + 6) % 7] = trow[di * DT_RS + (jl + 6) * DT_CC];
                        if (!WEIGHT_GRAD) {
                            float a = bv;
#pragma unroll
                            for (int di = 0; di < 7; ++di)
#pragma unroll
                                for (int dj = 0; dj < 7; ++dj)
                                    a = fmaf(wr[di * 7 + dj], win[di][(s + dj) % 7], a);
                            if (res) a = fmaf(res_scale, gv[jl], a);
                            y[obase + (size_t)jl * C] = a;
                        } else {
                            const float g = gv[jl];
                            acc_b += g;
#pragma unroll
                            for (int di = 0; di < 7; ++di)
#pragma unroll
                                for (int dj = 0; dj < 7; ++dj)
                                    acc_w[di * 7 + dj] = fmaf(g, win[di][(s + dj) % 7], acc_w[di * 7 + dj]);
                        }
                    }
                }
            }
        }
    }
    if (WEIGHT_GRAD) {
        __syncthreads();
        float *red = lds;                                        // [row slot][50][32]
#pragma unroll
        for (int t = 0; t < 49; ++t) red[(tr * 50 + t) * DT_CC + cl] = acc_w[t];
        red[(tr * 50 + 49) * DT_CC + cl] = acc_b;
        __syncthreads();
        float *out = part + (size_t)blockIdx.x * 50 * C;
        for (int e = threadIdx.x; e < 50 * DT_CC; e += DW_THREADS) {
            const int t = e >> 5, c2 = e & 31;
            if (c0 + c2 >= C) continue;
            float s = 0.f;
#pragma unroll
            for (int r = 0; r < DT_ROWS; ++r) s += red[(r * 50 + t) * DT_CC + c2];
            out[(size_t)t * C + c0 + c2] = s;
        }
    }
}

// =================================================================================================
// whole image in registers (S x S images, S = 3 or 6): thread = (image group, channel)
// =================================================================================================
template <int S, bool WEIGHT_GRAD>
__global__ __launch_bounds__(DW_THREADS) void dwconv7_whole_kernel(
    const float *__restrict__ x, const float *__restrict__ w, const float *__restrict__ bias,
    const float *__restrict__ res, float res_scale, float *__restrict__ y, const float *__restrict__ gy,
    float *__restrict__ part, int B, int C, int flip, int imgs_per_thread) {
    __shared__ float wl[WEIGHT_GRAD ? 1 : DW_THREADS * 49];
    const int c0 = blockIdx.x * DW_THREADS, c = c0 + threadIdx.x;
    if (!WEIGHT_GRAD) {      // the workgroup's weights are one contiguous run of w: coalesced into LDS, rows picked below
        const int nw = min(DW_THREADS, C - c0) * 49;
        for (int e = threadIdx.x; e < nw; e += DW_THREADS) wl[e] = w[(size_t)c0 * 49 + e];
        __syncthreads();
    }
    if (c >= C) return;
    const int b0 = blockIdx.y * imgs_per_thread, b1 = min(B, b0 + imgs_per_thread);
    constexpr int LO = (S >= 4) ? 0 : 4 - S, HI = 6 - LO;       // taps that can reach the image: |d-3| <= S-1
    float wr[49];
    float acc_w[49];
    float acc_b = 0.f;
#pragma unroll
    for (int t = 0; t < 49; ++t) {
        const int di = t / 7, dj = t % 7;
        const bool used = di >= LO && di <= HI && dj >= LO && dj <= HI;
        if (!WEIGHT_GRAD) wr[t] = used ? wl[threadIdx.x * 49 + (flip ? 48 - t : t)] : 0.f;
        else acc_w[t] = 0.f;
    }
    const float bv = (!WEIGHT_GRAD && bias) ? bias[c] : 0.f;
    for (int b = b0; b < b1; ++b) {
        const size_t base = (size_t)b * S * S * C + c;
        float xi[S][S];
#pragma unroll
        for (int i = 0; i < S; ++i)
#pragma unroll
            for (int j = 0; j < S; ++j) xi[i][j] = x[base + (size_t)(i * S + j) * C];
        if (!WEIGHT_GRAD) {
            float rv[S][S];
            if (res) {
#pragma unroll
                for (int i = 0; i < S; ++i)
#pragma unroll
                    for (int j = 0; j < S; ++j) rv[i][j] = res[base + (size_t)(i * S + j) * C];
            }
#pragma unroll
            for (int i = 0; i < S; ++i)
#pragma unroll
                for (int j = 0; j < S; ++j) {
                    float a = bv;
#pragma unroll
                    for (int di = 0; di < 7; ++di)
#pragma unroll
                        for (int dj = 0; dj < 7; ++dj) {
                            const int ii = i + di - 3, jj = j + dj - 3;
                            if (ii >= 0 && ii < S && jj >= 0 && jj < S) a = fmaf(wr[di * 7 + dj], xi[ii][jj], a);
                        }
                    if (res) a = fmaf(res_scale, rv[i][j], a);
                    y[base + (size_t)(i * S + j) * C] = a;
                }
        } else {
            float g[S][S];
#pragma unroll
            for (int i = 0; i < S; ++i)
#pragma unroll
                for (int j = 0; j < S; ++j) {
                    g[i][j] = gy[base + (size_t)(i * S + j) * C];
                    acc_b += g[i][j];
                }
#pragma unroll
            for (int i = 0; i < S; ++i)
#pragma unroll
                for (int j = 0; j < S; ++j)
#pragma unroll
                    for (int di = 0; di < 7; ++di)
#pragma unroll
                        for (int dj = 0; dj < 7; ++dj) {
                            const int ii = i + di - 3, jj = j + dj - 3;
                            if (ii >= 0 && ii < S && jj >= 0 && jj < S)
                                acc_w[di * 7 + dj] = fmaf(g[i][j], xi[ii][jj], acc_w[di * 7 + dj]);
                        }
        }
    }
    if (WEIGHT_GRAD) {
        float *out = part + (size_t)blockIdx.y * 50 * C + c;
#pragma unroll
        for (int t = 0; t < 49; ++t) out[(size_t)t * C] = acc_w[t];
        out[(size_t)49 * C] = acc_b;
    }
}

// stage 2 of the weight gradient: gw[c][t] += sum_p part[p][t][c]; gbias[c] += sum_p part[p][49][c].
// A workgroup owns FIN_E consecutive (t, c) entries and splits the partials over FIN_S interleaved slices
// (many short chains of independent loads), then folds the slices through LDS in a fixed order
// (bitwise reproducible; no atomics).
constexpr int FIN_E = 16, FIN_S = 16;
__global__ __launch_bounds__(FIN_E * FIN_S) void dwconv7_wgrad_finish_kernel(const float *__restrict__ part,
                                                                             int nparts, int C,
                                                                             float *__restrict__ gw,
                                                                             float *__restrict__ gbias) {
    __shared__ float red[FIN_S][FIN_E];
    const int el = threadIdx.x % FIN_E, slice = threadIdx.x / FIN_E;
    const int e = blockIdx.x * FIN_E + el;               // e = t*C + c
    const size_t stride = (size_t)50 * C;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (e < 50 * C) {
        int p = slice;
        for (; p + 3 * FIN_S < nparts; p += 4 * FIN_S) {
            s0 += part[(size_t)p * stride + e];
            s1 += part[(size_t)(p + FIN_S) * stride + e];
            s2 += part[(size_t)(p + 2 * FIN_S) * stride + e];
            s3 += part[(size_t)(p + 3 * FIN_S) * stride + e];
        }
        for (; p < nparts; p += FIN_S) s0 += part[(size_t)p * stride + e];
    }
    red[slice][el] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (slice == 0 && e < 50 * C) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < FIN_S; ++k) s += red[k][el];
        const int t = e / C, c = e - t * C;
        if (t < 49) gw[(size_t)c * 49 + t] += s;
        else if (gbias) gbias[c] += s;
    }
}

enum DwPath { DW_GENERIC, DW_TILED, DW_WHOLE3, DW_WHOLE6 };
// seg: the `seg` argument of the _ex entry points: 0 = choose by shape, 1..64 = the generic kernels with that many
// output columns per worker segment (tests compare the paths; tools/exp_dwseg.py times them)
inline DwPath dw_path(int H, int W, int C, int seg) {
    if (seg > 0) return DW_GENERIC;
    if (H == 3 && W == 3) return DW_WHOLE3;
    if (H == 6 && W == 6) return DW_WHOLE6;
    if (H >= 8 && W >= 8 && C % 4 == 0) return DW_TILED;
    return DW_GENERIC;
}

inline int generic_cc(int C) {
    return C >= 64 ? 64 : (C >= 32 ? 32 : (C >= 16 ? 16 : (C >= 8 ? 8 : (C >= 4 ? 4 : (C >= 2 ? 2 : 1)))));
}

// launch geometry of the weight-gradient stage 1; nparts = rows of the [nparts][50][C] workspace
struct DwWgradPlan {
    DwPath path;
    DwTiling tiling;
    int tiles_per_block, imgs_per_thread, Cc, nseg, seg;
    size_t total;
    unsigned gx, gy;
    size_t nparts;
};
inline DwWgradPlan dw_wgrad_plan(int B, int H, int W, int C, int seg) {
    DwWgradPlan p{};
    p.path = dw_path(H, W, C, seg);
    p.seg = seg > 0 ? seg : DW_SEG;
    if (p.path == DW_TILED) {
        p.tiling = dw_tiling(B, H, W);
        p.gy = (unsigned)sei_ceil_div(C, DT_CC);
        size_t want = 768 / p.gy > 0 ? 768 / p.gy : 1;          // one resident round: 256 CUs x 3 workgroups
        if (want < 64) want = 64;
        p.tiles_per_block = (int)sei_ceil_div(p.tiling.ntiles, want);
        p.gx = (unsigned)sei_ceil_div(p.tiling.ntiles, (size_t)p.tiles_per_block);
        p.nparts = p.gx;
    } else if (p.path == DW_WHOLE3 || p.path == DW_WHOLE6) {
        p.gx = (unsigned)sei_ceil_div(C, DW_THREADS);
        // image groups = partial sets.  Each set is 50 x C floats written here and read back by the finish kernel, against
        // H x W x C floats of x and of gy per image: keep the partials below about half of the input bytes, but at least
        // 8 groups (and ~256 workgroups) so the chip still has waves to hide the strided image loads behind.
        size_t want = (size_t)B * H * W / 64;
        if (want < 8) want = 8;
        if (want * p.gx < 256) want = sei_ceil_div(256, p.gx);
        if (want > (size_t)B) want = B;
        p.imgs_per_thread = (int)sei_ceil_div(B, want);
        p.gy = (unsigned)sei_ceil_div(B, p.imgs_per_thread);
        p.nparts = p.gy;
    } else {
        p.Cc = generic_cc(C);
        const int workers = DW_THREADS / p.Cc;
        p.nseg = (int)sei_ceil_div(W, p.seg);
        p.total = (size_t)B * H * p.nseg;
        p.gy = (unsigned)sei_ceil_div(C, p.Cc);
        p.gx = capped_grid(p.total, workers * 2, 65535);
        const unsigned max_gx = 4096 / p.gy > 0 ? 4096 / p.gy : 1;
        if (p.gx > max_gx) p.gx = max_gx;
        p.nparts = p.gx;
    }
    return p;
}

}  // namespace

// -------------------------------------------------------------------------------------------------
extern "C" int sei_dwconv7_fwd_ex(const float *x, const float *w, const float *bias, const float *res,
                                  float res_scale, float *y, int B, int H, int W, int C, int flip, int seg,
                                  void *stream) {
    SEI_REQUIRE(x && w && y && x != y && B > 0 && H > 0 && W > 0 && C > 0 && seg >= 0 && seg <= 66);
    hipStream_t s = (hipStream_t)stream;
    // seg 65 / 66: the first tiled kernel / the pipelined kernel, explicitly (tests compare them bit for bit)
    if (seg == 66) return sei_dwconv7_pipe_launch(x, w, bias, res, res_scale, y, B, H, W, C, flip, s);
    const bool old_tiled = seg == 65;
    if (old_tiled) seg = 0;
    const int gseg = seg > 0 ? seg : DW_SEG;
    const float *nof = nullptr;
    float *nom = nullptr;
    if (!old_tiled && dw_path(H, W, C, seg) == DW_TILED && sei_dwconv7_pipe_eligible(x, w, B, H, W, C))
        return sei_dwconv7_pipe_launch(x, w, bias, res, res_scale, y, B, H, W, C, flip, s);
    switch (dw_path(H, W, C, seg)) {
        case DW_TILED: {
            const DwTiling t = dw_tiling(B, H, W);
            SEI_REQUIRE(t.ntiles < (size_t)1 << 31);
            hipLaunchKernelGGL(dwconv7_tiled_kernel<false>, dim3((unsigned)t.ntiles, (unsigned)sei_ceil_div(C, DT_CC)),
                               dim3(DW_THREADS), 0, s, x, w, bias, res, res_scale, y, nof, nom, H, W, C, flip ? 1 : 0,
                               t.th, t.tw, t.tiles_i, t.tiles_j, (int)t.ntiles, 1);
            break;
        }
        case DW_WHOLE3:
        case DW_WHOLE6: {
            const unsigned gx = (unsigned)sei_ceil_div(C, DW_THREADS);
            size_t groups = 2048 / gx > 0 ? 2048 / gx : 1;
            if (groups > (size_t)B) groups = B;
            const int ipt = (int)sei_ceil_div(B, groups);
            const dim3 grid(gx, (unsigned)sei_ceil_div(B, ipt));
            if (H == 3)
                hipLaunchKernelGGL((dwconv7_whole_kernel<3, false>), grid, dim3(DW_THREADS), 0, s, x, w, bias, res,
                                   res_scale, y, nof, nom, B, C, flip ? 1 : 0, ipt);
            else
                hipLaunchKernelGGL((dwconv7_whole_kernel<6, false>), grid, dim3(DW_THREADS), 0, s, x, w, bias, res,
                                   res_scale, y, nof, nom, B, C, flip ? 1 : 0, ipt);
            break;
        }
        default: {
            const int Cc = generic_cc(C);
            const int workers = DW_THREADS / Cc;
            const int nseg = (int)sei_ceil_div(W, gseg);
            const size_t total = (size_t)B * H * nseg;
            SEI_REQUIRE(total < (size_t)1 << 31);
            dim3 grid(capped_grid(total, workers, 65535), (unsigned)sei_ceil_div(C, Cc));
            hipLaunchKernelGGL(dwconv7_kernel<false>, grid, dim3(DW_THREADS), 0, s, x, w, bias, res, res_scale, y, nof,
                               nom, B, H, W, C, flip ? 1 : 0, Cc, nseg, (int)total, gseg);
        }
    }
    return sei_launch_status();
}

extern "C" int sei_dwconv7_fwd(const float *x, const float *w, const float *bias, const float *res,
                               float res_scale, float *y, int B, int H, int W, int C, int flip, void *stream) {
    return sei_dwconv7_fwd_ex(x, w, bias, res, res_scale, y, B, H, W, C, flip, 0, stream);
}

extern "C" size_t sei_dwconv7_bwd_weight_workspace_ex(int B, int H, int W, int C, int seg) {
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || seg < 0 || seg > 64) return 0;
    return dw_wgrad_plan(B, H, W, C, seg).nparts * 50 * (size_t)C;
}

extern "C" size_t sei_dwconv7_bwd_weight_workspace(int B, int H, int W, int C) {
    return sei_dwconv7_bwd_weight_workspace_ex(B, H, W, C, 0);
}

extern "C" int sei_dwconv7_bwd_weight_ex(const float *x, const float *gy, float *gw, float *gbias, int B, int H,
                                         int W, int C, float *work, size_t work_floats, int seg, void *stream) {
    // gw = NULL: the partial sums ([parts][50][C], parts = workspace / (50 C)) stay in `work` for sei_fold_many
    SEI_REQUIRE(x && gy && work && B > 0 && H > 0 && W > 0 && C > 0 && seg >= 0 && seg <= 64 && (gw || !gbias));
    const DwWgradPlan p = dw_wgrad_plan(B, H, W, C, seg);
    SEI_REQUIRE(work_floats >= p.nparts * 50 * (size_t)C);
    hipStream_t s = (hipStream_t)stream;
    const float *nof = nullptr;
    float *nom = nullptr;
    // stage 1: per-workgroup (or per image group) partial sums into the workspace
    if (p.path == DW_TILED) {
        SEI_REQUIRE(p.tiling.ntiles < (size_t)1 << 31);
        hipLaunchKernelGGL(dwconv7_tiled_kernel<true>, dim3(p.gx, p.gy), dim3(DW_THREADS), 0, s, x, nof, nof, nof, 0.f,
                           nom, gy, work, H, W, C, 0, p.tiling.th, p.tiling.tw, p.tiling.tiles_i, p.tiling.tiles_j,
                           (int)p.tiling.ntiles, p.tiles_per_block);
    } else if (p.path == DW_WHOLE3) {
        hipLaunchKernelGGL((dwconv7_whole_kernel<3, true>), dim3(p.gx, p.gy), dim3(DW_THREADS), 0, s, x, nof, nof, nof,
                           0.f, nom, gy, work, B, C, 0, p.imgs_per_thread);
    } else if (p.path == DW_WHOLE6) {
        hipLaunchKernelGGL((dwconv7_whole_kernel<6, true>), dim3(p.gx, p.gy), dim3(DW_THREADS), 0, s, x, nof, nof, nof,
                           0.f, nom, gy, work, B, C, 0, p.imgs_per_thread);
    } else {
        SEI_REQUIRE(p.total < (size_t)1 << 31);
        const size_t lds = sizeof(float) * (size_t)(DW_THREADS / p.Cc) * 50 * p.Cc;
        hipLaunchKernelGGL(dwconv7_kernel<true>, dim3(p.gx, p.gy), dim3(DW_THREADS), lds, s, x, nof, nof, nof, 0.f, nom,
                           gy, work, B, H, W, C, 0, p.Cc, p.nseg, (int)p.total, p.seg);
    }
    // stage 2: fold the partials into the running gradient
    if (gw)
        hipLaunchKernelGGL(dwconv7_wgrad_finish_kernel, dim3((unsigned)sei_ceil_div((size_t)50 * C, FIN_E)),
                       dim3(FIN_E * FIN_S), 0, s,
                       (const float *)work, (int)p.nparts, C, gw, gbias);
    return sei_launch_status();
}

extern "C" int sei_dwconv7_bwd_weight(const float *x, const float *gy, float *gw, float *gbias, int B, int H,
                                      int W, int C, float *work, size_t work_floats, void *stream) {
    return sei_dwconv7_bwd_weight_ex(x, gy, gw, gbias, B, H, W, C, work, work_floats, 0, stream);
}
