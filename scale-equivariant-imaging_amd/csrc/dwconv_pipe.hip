// Pipelined depthwise 7x7 convolution for gfx950, NHWC float32, with an optional fused channel LayerNorm
// (reference: src/models/convolutional.py:36-39 -- conv1 = nn.Conv2d(C, C, 7, padding=3, groups=C) followed by
//  LayerNorm over the channels, :21-30; SURVEY 8(b) sei_dwconv7_ln_fwd).
//
// What the first tiled kernel (dwconv_kernels.hip) left on the table: a workgroup loaded its halo tile, waited for
// it, computed, stored and exited -- load latency, weight fetch and arithmetic in series, three workgroups per CU to
// overlap them. Here a workgroup walks a run of (tile, 32-channel group) stages with TWO halo tiles in LDS:
//   * stage s+1's tile (and its 32 x 49 weights when the group changes) is fetched by LDS-DMA
//     (global_load_lds_dwordx4: 1 KiB per wave-instruction, no VGPR staging; pixels outside the image come from a
//     16-byte zero chunk through the per-lane source address) while stage s is computed: one barrier per stage;
//   * lanes = 32 channels of a half wave, 8 row slots per workgroup, a 7x7 register window sliding along a row of
//     the tile: 7 conflict-free LDS reads + 49 FMA per output, the accumulation order (bias, then taps row-major)
//     of the other depthwise kernels, so the results agree with them bit for bit.
// Fused LayerNorm (C = 32 * LNG, LNG in {1, 4}: the two shallow levels of the default network): a workgroup owns
// ALL channels of its pixels; the LNG convolution results of a pixel stay in registers, the mean and the variance
// (two passes, as the stand-alone kernel) are summed over the half wave's 32 lanes with a transposing butterfly
// (16 exchanges for the 16 pixels of a row run instead of 16 x 5), and the normalised rows leave as bf16 (or f32)
// together with mean / rstd and the f32 convolution result that the backward pass re-reads. h1 is not re-read,
// the LayerNorm launch is gone.
#include "sei_common.h"

namespace {

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void glb_void;

__device__ __attribute__((aligned(16))) float g_dp_zero_chunk[4] = {0.f, 0.f, 0.f, 0.f};

constexpr int DP_THREADS = 256;
constexpr int DP_CC = 32;                       // channels per stage = lanes of a half wave
constexpr int DP_MAXW = 16;                     // widest tile
constexpr int DP_MAXK = 10;                     // tile pieces per wave: ceil(ceil(14 * 22 / 8) / 4)
constexpr int DP_WBYTES = DP_CC * 49 * 4;       // 6272 B of weights per group
constexpr int DP_WPIECES = 7;                   // ... in 7 pieces of 1 KiB (the tail fed from the zero chunk)
constexpr int DP_WBUF = DP_WPIECES * 1024;

__device__ __forceinline__ unsigned short dp_f2bf(float v) {
    const __bf16 b = (__bf16)v;
    return __builtin_bit_cast(unsigned short, b);
}

// Sum of v[p] over the 32 lanes of a half wave, for 16 values at once: after the five exchange steps lane cl holds
// the total of value (cl >> 1). 16 exchanges + 16 adds (a butterfly per value would be 80 + 80).
__device__ __forceinline__ float dp_transpose_sum16(const float (&v)[16], int cl) {
    float u[8], t[4], p[2];
    const bool b4 = cl & 16, b3 = cl & 8, b2 = cl & 4, b1 = cl & 2;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float keep = b4 ? v[k + 8] : v[k], send = b4 ? v[k] : v[k + 8];
        u[k] = keep + __shfl_xor(send, 16, 64);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float keep = b3 ? u[k + 4] : u[k], send = b3 ? u[k] : u[k + 4];
        t[k] = keep + __shfl_xor(send, 8, 64);
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const float keep = b2 ? t[k + 2] : t[k], send = b2 ? t[k] : t[k + 2];
        p[k] = keep + __shfl_xor(send, 4, 64);
    }
    const float keep = b1 ? p[1] : p[0], send = b1 ? p[0] : p[1];
    float r = keep + __shfl_xor(send, 2, 64);
    r += __shfl_xor(r, 1, 64);
    return r;
}

// s_waitcnt vmcnt(n) for a wave-uniform n in 0..16: wait until all but the wave's n youngest vector-memory operations
// are done (the counter runs in issue order over loads, stores and LDS-DMA alike).
__device__ __forceinline__ void dp_wait_vmcnt(int n) {
    switch (n) {
#define SEI_DP_W(N_) case N_: asm volatile("s_waitcnt vmcnt(" #N_ ")" ::: "memory"); break;
        SEI_DP_W(1) SEI_DP_W(2) SEI_DP_W(3) SEI_DP_W(4) SEI_DP_W(5) SEI_DP_W(6) SEI_DP_W(7) SEI_DP_W(8)
        SEI_DP_W(9) SEI_DP_W(10) SEI_DP_W(11) SEI_DP_W(12) SEI_DP_W(13) SEI_DP_W(14) SEI_DP_W(15) SEI_DP_W(16)
#undef SEI_DP_W
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

struct DpTile {
    int b, i0, j0, g;
};

// LN: fused LayerNorm over C = 32 * LNG channels (stages run tile-major, the LNG groups of a tile back to back);
// otherwise stages run group-major (weights change at most once or twice per workgroup). OUT16: bf16 LayerNorm output.
// TW: tile width (12 or 16 output columns), WB: double-buffered weight images in LDS (more than one channel group);
// together they fix the static LDS size: <= 80 KB (two workgroups per CU) except for 16-wide tiles with weight buffers.
constexpr int dp_pieces(int tw) { return ((8 + 6) * (tw + 6) + 7) / 8; }
constexpr int dp_lds_bytes(int tw, bool wb) { return 2 * dp_pieces(tw) * 1024 + (wb ? 2 * DP_WBUF : 0); }

template <bool LN, int LNG, bool OUT16, int TW, bool WB>
__global__ __launch_bounds__(DP_THREADS, 2) void dwconv7_pipe_kernel(
    const float *__restrict__ x, const float *__restrict__ w, const float *__restrict__ bias,
    const float *__restrict__ res, float res_scale, float *__restrict__ y, const float *__restrict__ gamma,
    const float *__restrict__ beta, void *__restrict__ h2, float *__restrict__ mean, float *__restrict__ rstd,
    float eps, int H, int W, int C, int flip, int th, int tiles_i, int tiles_j, int ntiles, int stages,
    int stages_per_block, int npix, int npieces) {
    constexpr int tw = TW, tile_bytes = dp_pieces(TW) * 1024;
    constexpr int wbuf_off = WB ? 2 * tile_bytes : tile_bytes;
    __shared__ __attribute__((aligned(16))) char smem[dp_lds_bytes(TW, WB)];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int cl = threadIdx.x & (DP_CC - 1), tr = threadIdx.x >> 5;
    const int lw = tw + 6;
    const int s0 = blockIdx.x * stages_per_block, s1 = min(stages, s0 + stages_per_block);
    if (s0 >= s1) return;

    // (row, column) of the tile pixel this lane fetches in each of its pieces: fixed for the launch
    int pp[DP_MAXK];
#pragma unroll
    for (int k = 0; k < DP_MAXK; ++k) {
        const int pixel = 8 * (wave + 4 * k) + (lane >> 3);
        const int lr = pixel / lw;
        pp[k] = pixel < npix ? (lr << 8) | (pixel - lr * lw) : -1;
    }

    auto decode = [&](int s) {
        DpTile t;
        int tile;
        if (LN) {
            tile = s / LNG;
            t.g = s - tile * LNG;
        } else {
            t.g = s / ntiles;
            tile = s - t.g * ntiles;
        }
        const int tj = tile % tiles_j, rest = tile / tiles_j;
        const int ti = rest % tiles_i;
        t.b = rest / tiles_i;
        t.i0 = ti * th;
        t.j0 = tj * tw;
        return t;
    };
    auto issue_weights = [&](int g, char *dst) {
        const char *wsrc = reinterpret_cast<const char *>(w + (size_t)g * DP_CC * 49);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int p = wave + 4 * k;
            if (p < DP_WPIECES) {                                   // wave-uniform
                const int off = 1024 * p + 16 * lane;
                const char *src = off < DP_WBYTES ? wsrc + off : reinterpret_cast<const char *>(g_dp_zero_chunk);
                __builtin_amdgcn_global_load_lds((glb_void *)src, (lds_void *)(dst + p * 1024), 16, 0, 0);
            }
        }
    };
    auto issue_tile = [&](const DpTile &t, char *dst) {
        const float *xb = x + (size_t)t.b * H * W * C + t.g * DP_CC + 4 * (lane & 7);
#pragma unroll
        for (int k = 0; k < DP_MAXK; ++k) {
            const int q = wave + 4 * k;
            if (q < npieces) {                                      // wave-uniform
                const int lr = pp[k] >> 8, lc = pp[k] & 255;
                const int ii = t.i0 - 3 + lr, jj = t.j0 - 3 + lc;
                const bool ok = pp[k] >= 0 && ii >= 0 && ii < H && jj >= 0 && jj < W;
                const float *src = ok ? xb + ((size_t)ii * W + jj) * C : g_dp_zero_chunk;
                __builtin_amdgcn_global_load_lds((glb_void *)src, (lds_void *)(dst + q * 1024), 16, 0, 0);
            }
        }
    };

    float wr[49];
    float bv = 0.f;
    int wg_loaded = -1;
    auto read_weights = [&](int g, const char *src) {
        const float *wl = reinterpret_cast<const float *>(src);
#pragma unroll
        for (int t = 0; t < 49; ++t) wr[t] = wl[cl * 49 + (flip ? 48 - t : t)];     // stride 49 floats: conflict-free
        bv = bias ? bias[g * DP_CC + cl] : 0.f;
        wg_loaded = g;
    };
    char *const tile0 = smem;
    char *const wbuf0 = smem + wbuf_off;            // double-buffered weights, or (one group only) tile buffer 1

    // prologue: first tile + first weights in flight
    DpTile cur = decode(s0);
    issue_tile(cur, tile0);
    issue_weights(cur.g, wbuf0);
    constexpr bool w_in_tile1 = !WB;                                // no weight buffers: they pass through tile 1 once
    if (w_in_tile1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        read_weights(cur.g, wbuf0);
        __syncthreads();                                            // tile buffer 1 is free for stage s0 + 1
    }

    float av[LN ? LNG : 1][DP_MAXW];                                // LN: the convolution results of the tile's groups
    float gam[LN ? LNG : 1], bet[LN ? LNG : 1];
    if (LN) {
#pragma unroll
        for (int g = 0; g < LNG; ++g) {
            gam[g] = gamma[g * DP_CC + cl];
            bet[g] = beta[g * DP_CC + cl];
        }
    }

    // The stores of a stage are issued AFTER the DMA of the next tile, so "the next tile has landed" is
    // vmcnt(stores issued since) -- not vmcnt(0), which would drain the stores (1-2 us under load) at every stage.
    // `later` is a lower bound of that number for this wave (exactly the row run's jn stores of a wave with a live row
    // slot; the LayerNorm stores of a tile's last group only add to it), so the wait is never too weak.
    int later = 0;
    for (int sb = s0; sb < s1; sb += (LN ? LNG : 1)) {
#pragma unroll
        for (int gi = 0; gi < (LN ? LNG : 1); ++gi) {
            const int s = sb + gi;                                  // (LN: stages_per_block is a multiple of LNG)
            const int par = (s - s0) & 1;
            const char *tile = tile0 + par * tile_bytes;
            dp_wait_vmcnt(later);
            __builtin_amdgcn_s_barrier();                           // stage s has landed; stage s-1's readers are done
            asm volatile("" ::: "memory");
            if (!w_in_tile1 && cur.g != wg_loaded) read_weights(cur.g, wbuf0 + par * DP_WBUF);
            const int i = cur.i0 + tr;
            const bool row_ok = tr < th && i < H;
            const int jn = min(tw, W - cur.j0);
            const size_t obase = (((size_t)cur.b * H + i) * W + cur.j0) * C + cur.g * DP_CC + cl;
            float gv[DP_MAXW];
            if (!LN && res && row_ok) {                             // (before the DMA issue: their wait must not cover it)
#pragma unroll
                for (int jl = 0; jl < DP_MAXW; ++jl) gv[jl] = (jl < jn) ? res[obase + (size_t)jl * C] : 0.f;
            }
            asm volatile("" ::: "memory");
            DpTile nxt = cur;
            if (s + 1 < s1) {
                nxt = decode(s + 1);
                issue_tile(nxt, tile0 + (par ^ 1) * tile_bytes);
                if (!w_in_tile1 && nxt.g != cur.g) issue_weights(nxt.g, wbuf0 + (par ^ 1) * DP_WBUF);
            }
            asm volatile("" ::: "memory");
            {   // a wave holds row slots 2 * wave and 2 * wave + 1
                const int r0 = 2 * wave;
                later = (r0 < th && cur.i0 + r0 < H) ? jn : 0;
            }
            if (LN) {
#pragma unroll
                for (int jl = 0; jl < DP_MAXW; ++jl) av[gi][jl] = 0.f;
            }
            if (row_ok) {
                // Four outputs at a time: their 49-term chains (bias, then taps row-major -- the order of every other
                // depthwise kernel, bit for bit) are independent, so the FMAs of one cover the latency of the others;
                // a single chain issued one dependent FMA per ~8 cycles and left the SIMD three quarters idle. All TW
                // outputs of the run are computed (the staged tile is always TW + 6 columns wide), only the stores are
                // predicated. The window is a ring of RW columns: 10 live ones, the next block's four loaded under them.
                const float *trow = reinterpret_cast<const float *>(tile) + (tr * lw) * DP_CC + cl;
                constexpr int rs_ = (TW + 6) * DP_CC;                // floats per staged row
                constexpr int RW = LN ? (LNG > 1 ? 10 : 14) : 12;
                float ring[7][RW];
#pragma unroll
                for (int c = 0; c < 6; ++c)
#pragma unroll
                    for (int di = 0; di < 7; ++di) ring[di][c % RW] = trow[di * rs_ + c * DP_CC];
#pragma unroll
                for (int blk = 0; blk < TW / 4; ++blk) {
#pragma unroll
                    for (int c = 4 * blk + 6; c < 4 * blk + 10; ++c)
#pragma unroll
                        for (int di = 0; di < 7; ++di) ring[di][c % RW] = trow[di * rs_ + c * DP_CC];
                    float a[4] = {bv, bv, bv, bv};
#pragma unroll
                    for (int di = 0; di < 7; ++di)
#pragma unroll
                        for (int dj = 0; dj < 7; ++dj)
#pragma unroll
                            for (int o = 0; o < 4; ++o)     // (volatile: keeps the four chains interleaved as written)
                                asm volatile("v_fmac_f32 %0, %1, %2"
                                             : "+v"(a[o])
                                             : "v"(wr[di * 7 + dj]), "v"(ring[di][(4 * blk + o + dj) % RW]));
#pragma unroll
                    for (int o = 0; o < 4; ++o) {
                        const int jl = 4 * blk + o;
                        if (LN) av[gi][jl] = jl < jn ? a[o] : 0.f;
                        else if (res) a[o] = fmaf(res_scale, gv[jl], a[o]);
                        if (jl < jn) y[obase + (size_t)jl * C] = a[o];
                    }
                }
            }
            if (LN && gi == LNG - 1 && row_ok) {
                // ---- LayerNorm over the C = 32 * LNG channels of each of the run's pixels (half-wave uniform branch)
                const float invC = 1.0f / (float)(DP_CC * LNG);
                const int hw = lane & 32;
                float acc[DP_MAXW];
#pragma unroll
                for (int jl = 0; jl < DP_MAXW; ++jl) {
                    float t = av[0][jl];
#pragma unroll
                    for (int g = 1; g < LNG; ++g) t += av[g][jl];
                    acc[jl] = t;
                }
                const float mu_own = dp_transpose_sum16(acc, cl) * invC;           // pixel cl >> 1
                float mu[DP_MAXW];
#pragma unroll
                for (int jl = 0; jl < DP_MAXW; ++jl) mu[jl] = __shfl(mu_own, hw + 2 * jl, 64);
#pragma unroll
                for (int jl = 0; jl < DP_MAXW; ++jl) {
                    float q = 0.f;
#pragma unroll
                    for (int g = 0; g < LNG; ++g) {
                        const float d = av[g][jl] - mu[jl];
                        q = fmaf(d, d, q);
                    }
                    acc[jl] = q;
                }
                const float rs_own = 1.0f / sqrtf(dp_transpose_sum16(acc, cl) * invC + eps);
                const size_t pix0 = ((size_t)cur.b * H + i) * W + cur.j0;
                if ((cl & 1) == 0 && (cl >> 1) < jn) {
                    mean[pix0 + (cl >> 1)] = mu_own;
                    rstd[pix0 + (cl >> 1)] = rs_own;
                }
                const size_t hbase = pix0 * C + cl;
#pragma unroll
                for (int jl = 0; jl < DP_MAXW; ++jl) {
                    const float rs = __shfl(rs_own, hw + 2 * jl, 64);
                    if (jl < jn) {
#pragma unroll
                        for (int g = 0; g < LNG; ++g) {
                            const float o = fmaf((av[g][jl] - mu[jl]) * rs, gam[g], bet[g]);
                            const size_t e = hbase + (size_t)jl * C + g * DP_CC;
                            if (OUT16) reinterpret_cast<unsigned short *>(h2)[e] = dp_f2bf(o);
                            else reinterpret_cast<float *>(h2)[e] = o;
                        }
                    }
                }
            }
            cur = nxt;
        }
    }
}

struct DpPlan {
    bool ok;
    int th, tw, tiles_i, tiles_j, ntiles, G, stages, spb, npix, npieces;
    bool wb;
    unsigned grid;
};

// ln_groups: 0 = plain convolution, otherwise C / 32 (1 or 4)
inline DpPlan dp_plan(int B, int H, int W, int C, int ln_groups) {
    DpPlan p{};
    p.ok = false;
    if (H < 8 || W < 8 || C % DP_CC != 0) return p;
    p.G = C / DP_CC;
    if (ln_groups && (ln_groups != p.G || (p.G != 1 && p.G != 4))) return p;     // (2 groups: 89 spilled VGPRs)
    p.th = (H % 8 == 0) ? 8 : ((H % 6 == 0) ? 6 : ((H % 7 == 0) ? 7 : 8));
    // with weight buffers in LDS (more than one channel group) a 16-wide tile leaves room for one workgroup per CU
    // only: prefer 12 columns where they divide the row
    p.wb = p.G > 1;
    if (p.wb) p.tw = (W % 12 == 0) ? 12 : 16;
    else p.tw = (W % 16 != 0 && W % 12 == 0) ? 12 : 16;
    p.tiles_i = (int)sei_ceil_div(H, p.th);
    p.tiles_j = (int)sei_ceil_div(W, p.tw);
    const size_t nt = (size_t)B * p.tiles_i * p.tiles_j;
    if (nt * p.G >= ((size_t)1 << 30)) return p;
    p.ntiles = (int)nt;
    p.stages = p.ntiles * p.G;
    p.npix = (p.th + 6) * (p.tw + 6);
    p.npieces = (p.npix + 7) / 8;
    // one resident round of two workgroups per CU (one when the tiles do not leave room for two)
    const int slots = dp_lds_bytes(p.tw, p.wb) <= 80 * 1024 ? 512 : 256;
    const int unit = ln_groups ? ln_groups : 1;
    int spb = (int)sei_ceil_div((size_t)p.stages, (size_t)slots);
    spb = (int)sei_ceil_div((size_t)spb, (size_t)unit) * unit;
    if (spb < unit) spb = unit;
    p.spb = spb;
    p.grid = (unsigned)sei_ceil_div((size_t)p.stages, (size_t)spb);
    p.ok = true;
    return p;
}

inline bool dp_aligned(const void *a, const void *b) {
    return ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) & 15) == 0;
}

}  // namespace

template <bool LN, int LNG, bool OUT16, int TW, bool WB>
static int dp_launch_one(const DpPlan &p, const float *x, const float *w, const float *bias, const float *res,
                         float res_scale, float *y, const float *gamma, const float *beta, void *h2, float *mean,
                         float *rstd, float eps, int H, int W, int C, int flip, hipStream_t s) {
    hipLaunchKernelGGL((dwconv7_pipe_kernel<LN, LNG, OUT16, TW, WB>), dim3(p.grid), dim3(DP_THREADS), 0, s, x, w, bias,
                       res, res_scale, y, gamma, beta, h2, mean, rstd, eps, H, W, C, flip ? 1 : 0, p.th, p.tiles_i,
                       p.tiles_j, p.ntiles, p.stages, p.spb, p.npix, p.npieces);
    return sei_launch_status();
}

// Plain depthwise convolution on the pipelined kernel; SEI_ERR_BAD_ARG when the shape is not eligible (the caller
// falls back to the kernels of dwconv_kernels.hip).
int sei_dwconv7_pipe_launch(const float *x, const float *w, const float *bias, const float *res, float res_scale,
                            float *y, int B, int H, int W, int C, int flip, hipStream_t s) {
    const DpPlan p = dp_plan(B, H, W, C, 0);
    if (!p.ok || !dp_aligned(x, w)) return SEI_ERR_BAD_ARG;
#define SEI_DP_PLAIN(TW_, WB_)                                                                                    \
    return dp_launch_one<false, 1, false, TW_, WB_>(p, x, w, bias, res, res_scale, y, nullptr, nullptr, nullptr,  \
                                                    nullptr, nullptr, 0.f, H, W, C, flip, s);
    if (p.tw == 12) {
        if (p.wb) { SEI_DP_PLAIN(12, true) }
        SEI_DP_PLAIN(12, false)
    }
    if (p.wb) { SEI_DP_PLAIN(16, true) }
    SEI_DP_PLAIN(16, false)
#undef SEI_DP_PLAIN
}

// Where the pipelined kernel is the default choice (measured on MI355X against the first tiled kernel, round 3:
// 64 x 24 x 24 x 128 forward / data gradient 18.2 / 18.5 us against 20.4 / 23.1, 64 x 12 x 12 x 512 20.0 / 20.5 against
// 24.6 / 25.9, 64 x 192 x 192 x 32 223 / 240 against 258 / 271; with fewer than two stages per resident workgroup -- the
// 32-crop passes -- its prologue is not amortised: 13.2 against 12.2 us).
bool sei_dwconv7_pipe_eligible(const float *x, const float *w, int B, int H, int W, int C) {
    const DpPlan p = dp_plan(B, H, W, C, 0);
    return p.ok && dp_aligned(x, w) && p.stages >= 1024;
}

// One fused launch for conv1 -> LayerNorm where that measured faster than the two launches: one channel group
// (C = 32: 18.3 against 21.5 us at 32 crops, 285 against 321 us at 64 x 192 x 192). With four groups (C = 128) the
// LayerNorm arithmetic lands in a kernel that is already bound by the f32 FMA issue rate (SQ_ACTIVE_INST_VALU = 0.69
// of the SIMD cycles) and the fusion only ties (27.6 against 27.1 us): it stays available (sei_dwconv7_ln_fwd_ex).
bool sei_dwconv7_ln_fused_eligible(const float *x, const float *w, int B, int H, int W, int C) {
    return C == DP_CC && dp_plan(B, H, W, C, 1).ok && dp_aligned(x, w);
}

int sei_dwconv7_ln_fused_launch(const float *x, const float *w, const float *bias, const float *gamma,
                                const float *beta, float *h1, void *h2, int out16, float *mean, float *rstd, int B, int H,
                                int W, int C, float eps, hipStream_t s) {
    const DpPlan p = dp_plan(B, H, W, C, C / DP_CC);
    if (!p.ok || !dp_aligned(x, w)) return SEI_ERR_BAD_ARG;
#define SEI_DP_LN(G_, O_, TW_, WB_)                                                                                  \
    return dp_launch_one<true, G_, O_, TW_, WB_>(p, x, w, bias, nullptr, 0.f, h1, gamma, beta, h2, mean, rstd, eps, \
                                                 H, W, C, 0, s);
    if (p.G == 1) {
        if (p.tw == 12) { if (out16) { SEI_DP_LN(1, true, 12, false) } SEI_DP_LN(1, false, 12, false) }
        if (out16) { SEI_DP_LN(1, true, 16, false) }
        SEI_DP_LN(1, false, 16, false)
    }
    if (p.G == 4) {
        if (p.tw == 12) { if (out16) { SEI_DP_LN(4, true, 12, true) } SEI_DP_LN(4, false, 12, true) }
        if (out16) { SEI_DP_LN(4, true, 16, true) }
        SEI_DP_LN(4, false, 16, true)
    }
    return SEI_ERR_BAD_ARG;
#undef SEI_DP_LN
}

// -------------------------------------------------------------------------------------------------
// ConvBlock.conv1 -> LayerNorm in one call (convolutional.py:36-39). h1 = dwconv7(x) + bias (f32, kept for the
// backward pass), h2 = LayerNorm_C(h1) * gamma + beta as bf16 (out16 = 1) or f32, mean / rstd per pixel.
// C = 32 on images of at least 8 x 8: ONE launch of the pipelined kernel (C = 128 on request: fuse = 1); any other
// shape: the depthwise kernel followed by the stand-alone LayerNorm (same results, two launches).
extern "C" int sei_dwconv7_ln_fwd_ex(const float *x, const float *w, const float *bias, const float *gamma,
                                     const float *beta, float *h1, void *h2, int out16, float *mean, float *rstd, int B,
                                     int H, int W, int C, float eps, int fuse, void *stream) {
    SEI_REQUIRE(x && w && gamma && beta && h1 && h2 && mean && rstd && x != h1 && B > 0 && H > 0 && W > 0 && C > 0);
    SEI_REQUIRE(fuse >= 0 && fuse <= 2);
    hipStream_t s = (hipStream_t)stream;
    // fuse: 0 = by measurement (above), 1 = the fused launch or SEI_ERR_BAD_ARG, 2 = the two launches
    if (fuse == 1)
        return sei_dwconv7_ln_fused_launch(x, w, bias, gamma, beta, h1, h2, out16, mean, rstd, B, H, W, C, eps, s);
    if (fuse == 0 && sei_dwconv7_ln_fused_eligible(x, w, B, H, W, C))
        return sei_dwconv7_ln_fused_launch(x, w, bias, gamma, beta, h1, h2, out16, mean, rstd, B, H, W, C, eps, s);
    int rc = sei_dwconv7_fwd(x, w, bias, nullptr, 1.0f, h1, B, H, W, C, 0, stream);
    if (rc != SEI_OK) return rc;
    const size_t rows = (size_t)B * H * W;
    return out16 ? sei_ln_fwd_bf16(h1, gamma, beta, (uint16_t *)h2, mean, rstd, rows, C, eps, stream)
                 : sei_ln_fwd(h1, gamma, beta, (float *)h2, mean, rstd, rows, C, eps, stream);
}

extern "C" int sei_dwconv7_ln_fwd(const float *x, const float *w, const float *bias, const float *gamma,
                                  const float *beta, float *h1, void *h2, int out16, float *mean, float *rstd, int B,
                                  int H, int W, int C, float eps, void *stream) {
    return sei_dwconv7_ln_fwd_ex(x, w, bias, gamma, beta, h1, h2, out16, mean, rstd, B, H, W, C, eps, 0, stream);
}

// Kernel launches sei_dwconv7_ln_fwd issues for this shape (x and w 16-byte aligned): 1 (fused) or 2.
extern "C" size_t sei_dwconv7_ln_fwd_launches(int B, int H, int W, int C) {
    return (B > 0 && H > 0 && W > 0 && C == DP_CC && dp_plan(B, H, W, C, 1).ok) ? 1 : 2;
}
