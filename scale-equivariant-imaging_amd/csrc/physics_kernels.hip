// Physics operators and the EI scale transform for gfx950 (MI355X).
//
// All three are HBM-streaming stencils on NCHW planar float32 images: each input tile (plus halo) is
// read once from HBM into LDS, both separable passes run out of LDS, the output is written once.
// Algorithmic bytes per plane: 4*H*W in + 4*Ho*Wo out.
//
//   sei_blur_sep_circ / sei_blur_dense_circ : BlurV2.A and its adjoint  (blur/__init__.py:205-227)
//   sei_resample_sepband                    : Downsampling.A, adjoints, AA prefilter
//   sei_scale_resample_fwd/bwd              : padded_downsampling_transform (transforms.py:27-83)
#include "sei_common.h"

namespace {

constexpr int BLUR_THREADS = 256;
constexpr int MAX_TAPS = 63;

// ------------------------------------------------------------------------------------------------
// circular separable blur. One workgroup = one (plane, tile). LDS: taps | input tile + halo | row pass.
// y[i,j] = sum_a tv[a] sum_b th[b] x[(i - a + sv) mod H, (j - b + sh) mod W]
// (the adjoint is the same kernel with flipped taps and sv' = kv-1-sv, done on the host side of the ABI)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLUR_THREADS) void blur_sep_circ_kernel(
    const float *__restrict__ x, float *__restrict__ y, const float *__restrict__ tv,
    const float *__restrict__ th, int kv, int kh, int sv, int sh, int flip, int H, int W, int tile_h,
    int tile_w, int tiles_y, int tiles_x) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int tiles = tiles_y * tiles_x;
    const int plane = blockIdx.x / tiles;
    const int t = blockIdx.x - plane * tiles;
    const int i0 = (t / tiles_x) * tile_h, j0 = (t % tiles_x) * tile_w;
    const int ih = tile_h + kv - 1, iw = tile_w + kh - 1;

    float *s_tv = smem;              // 64 floats reserved each, keeps the tiles 16-byte aligned
    float *s_th = smem + 64;
    float *s_in = smem + 128;
    float *s_row = s_in + ((ih * iw + 3) & ~3);

    if (tid < kv) s_tv[tid] = tv[flip ? kv - 1 - tid : tid];
    if (tid < kh) s_th[tid] = th[flip ? kh - 1 - tid : tid];

    const float *xp = x + (size_t)plane * H * W;
    const int r0 = i0 - kv + 1 + sv, c0 = j0 - kh + 1 + sh;
    for (int idx = tid; idx < ih * iw; idx += BLUR_THREADS) {
        const int r = idx / iw, l = idx - r * iw;
        s_in[idx] = xp[(size_t)sei_mod(r0 + r, H) * W + sei_mod(c0 + l, W)];
    }
    __syncthreads();
    for (int idx = tid; idx < ih * tile_w; idx += BLUR_THREADS) {
        const int r = idx / tile_w, c = idx - r * tile_w;
        const float *src = s_in + r * iw + c + kh - 1;
        float acc = 0.f;
        for (int b = 0; b < kh; ++b) acc = fmaf(s_th[b], src[-b], acc);
        s_row[idx] = acc;
    }
    __syncthreads();
    float *yp = y + (size_t)plane * H * W;
    for (int idx = tid; idx < tile_h * tile_w; idx += BLUR_THREADS) {
        const int r = idx / tile_w, c = idx - r * tile_w;
        if (i0 + r < H && j0 + c < W) {
            const float *src = s_row + (r + kv - 1) * tile_w + c;
            float acc = 0.f;
            for (int a = 0; a < kv; ++a) acc = fmaf(s_tv[a], src[-a * tile_w], acc);
            yp[(size_t)(i0 + r) * W + j0 + c] = acc;
        }
    }
}

// dense (non-separable) circular blur: same tiling, one pass of kv*kh taps out of LDS.
__global__ __launch_bounds__(BLUR_THREADS) void blur_dense_circ_kernel(
    const float *__restrict__ x, float *__restrict__ y, const float *__restrict__ k, int kv, int kh,
    int sv, int sh, int flip, int H, int W, int tile_h, int tile_w, int tiles_y, int tiles_x) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int tiles = tiles_y * tiles_x;
    const int plane = blockIdx.x / tiles;
    const int t = blockIdx.x - plane * tiles;
    const int i0 = (t / tiles_x) * tile_h, j0 = (t % tiles_x) * tile_w;
    const int ih = tile_h + kv - 1, iw = tile_w + kh - 1;
    const int nk = kv * kh;
    float *s_k = smem;
    float *s_in = smem + ((nk + 3) & ~3);
    for (int idx = tid; idx < nk; idx += BLUR_THREADS) s_k[idx] = k[flip ? nk - 1 - idx : idx];
    const float *xp = x + (size_t)plane * H * W;
    const int r0 = i0 - kv + 1 + sv, c0 = j0 - kh + 1 + sh;
    for (int idx = tid; idx < ih * iw; idx += BLUR_THREADS) {
        const int r = idx / iw, l = idx - r * iw;
        s_in[idx] = xp[(size_t)sei_mod(r0 + r, H) * W + sei_mod(c0 + l, W)];
    }
    __syncthreads();
    float *yp = y + (size_t)plane * H * W;
    for (int idx = tid; idx < tile_h * tile_w; idx += BLUR_THREADS) {
        const int r = idx / tile_w, c = idx - r * tile_w;
        if (i0 + r < H && j0 + c < W) {
            float acc = 0.f;
            for (int a = 0; a < kv; ++a) {
                const float *src = s_in + (r + kv - 1 - a) * iw + c + kh - 1;
                for (int b = 0; b < kh; ++b) acc = fmaf(s_k[a * kh + b], src[-b], acc);
            }
            yp[(size_t)(i0 + r) * W + j0 + c] = acc;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// separable banded resampling: y = Wv x Wh^T with band-stored weights.
// One workgroup = one (plane, output tile): stage the input footprint, horizontal pass, vertical pass.
// ------------------------------------------------------------------------------------------------
constexpr int RS_THREADS = 256;

__global__ __launch_bounds__(RS_THREADS) void resample_sepband_kernel(
    const float *__restrict__ x, float *__restrict__ y, int Hi, int Wi, int Ho, int Wo,
    const float *__restrict__ wv, const int *__restrict__ lov, int nbv, const float *__restrict__ wh,
    const int *__restrict__ loh, int nbh, int tile_h, int tile_w, int tiles_y, int tiles_x, int fh_max,
    int fw_max) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int tiles = tiles_y * tiles_x;
    const int plane = blockIdx.x / tiles;
    const int t = blockIdx.x - plane * tiles;
    const int oi0 = (t / tiles_x) * tile_h, oj0 = (t % tiles_x) * tile_w;
    const int th = min(tile_h, Ho - oi0), tw = min(tile_w, Wo - oj0);
    const int r0 = lov[oi0], c0 = loh[oj0];
    // input footprint of this tile (clipped to the LDS the host sized from the band-step bound)
    const int fh = min(min(lov[oi0 + th - 1] + nbv, Hi) - r0, fh_max);
    const int fw = min(min(loh[oj0 + tw - 1] + nbh, Wi) - c0, fw_max);
    float *s_in = smem;                              // fh_max * fw_max
    float *s_row = smem + ((fh_max * fw_max + 3) & ~3);   // fh_max * tile_w
    const float *xp = x + (size_t)plane * Hi * Wi;
    for (int idx = tid; idx < fh * fw; idx += RS_THREADS) {
        const int r = idx / fw, c = idx - r * fw;
        s_in[r * fw_max + c] = xp[(size_t)(r0 + r) * Wi + c0 + c];
    }
    __syncthreads();
    for (int idx = tid; idx < fh * tw; idx += RS_THREADS) {
        const int r = idx / tw, c = idx - r * tw;
        const int oj = oj0 + c;
        const int base = loh[oj] - c0;
        const int n = min(nbh, fw - base);
        const float *wrow = wh + (size_t)oj * nbh;
        const float *src = s_in + r * fw_max + base;
        float acc = 0.f;
        for (int b = 0; b < n; ++b) acc = fmaf(wrow[b], src[b], acc);
        s_row[r * tile_w + c] = acc;
    }
    __syncthreads();
    float *yp = y + (size_t)plane * Ho * Wo;
    for (int idx = tid; idx < th * tw; idx += RS_THREADS) {
        const int r = idx / tw, c = idx - r * tw;
        const int oi = oi0 + r;
        const int base = lov[oi] - r0;
        const int n = min(nbv, fh - base);
        const float *wcol = wv + (size_t)oi * nbv;
        float acc = 0.f;
        for (int a = 0; a < n; ++a) acc = fmaf(wcol[a], s_row[(base + a) * tile_w + c], acc);
        yp[(size_t)oi * Wo + oj0 + c] = acc;
    }
}

// ------------------------------------------------------------------------------------------------
// EI scale transform: fused grid generation + bicubic (A=-0.75) gather with reflection.
// The coordinate arithmetic follows the reference's float32 op order (transforms.py:32-41, then
// ATen's grid_sampler unnormalize with align_corners=True), with contraction to FMA disabled
// (__f*_rn intrinsics) so that the sampling positions agree with torch to the last bit wherever
// the division and reciprocal are correctly rounded.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float cubic1(float x) {   // |x| <= 1, A = -0.75
    const float A = -0.75f;
    return ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f;
}
__device__ __forceinline__ float cubic2(float x) {   // 1 < |x| < 2
    const float A = -0.75f;
    return ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A;
}
__device__ __forceinline__ int reflect_clip(int v, int n) {
    // reflect_coordinates(v, 0, 2(n-1)) then clip, on an integer tap position
    if (n == 1) return 0;
    const int span = n - 1;
    v = v < 0 ? -v : v;
    const int flips = v / span;
    const int extra = v - flips * span;
    const int r = (flips & 1) ? span - extra : extra;
    return min(max(r, 0), n - 1);
}

struct ScaleTaps {
    int xs[4], ys[4];
    float wx[4], wy[4];
};

// (H, W): output / grid size; (Hi, Wi): size of the sampled image (they differ only for the
// antialiased variant, where the reference samples the pre-shrunk image on the original-size grid)
__device__ __forceinline__ void scale_taps(int p, int H, int W, int Hi, int Wi, float two_over_h,
                                           float two_over_w, float inv_rate, float cx, float cy,
                                           ScaleTaps &tp) {
    // position of output pixel p = i*W + j inside the reference's (w,h) meshgrid viewed as (h,w)
    const int aa = p / H, bb = p - aa * H;
    const float v = __fsub_rn(__fmul_rn(two_over_h, (float)bb), 1.f);
    const float u = __fsub_rn(__fmul_rn(two_over_w, (float)aa), 1.f);
    const float gx = __fadd_rn(__fmul_rn(inv_rate, __fsub_rn(v, cx)), cx);
    const float gy = __fadd_rn(__fmul_rn(inv_rate, __fsub_rn(u, cy)), cy);
    const float ix = __fmul_rn(__fdiv_rn(__fadd_rn(gx, 1.f), 2.f), (float)(Wi - 1));
    const float iy = __fmul_rn(__fdiv_rn(__fadd_rn(gy, 1.f), 2.f), (float)(Hi - 1));
    const float fx = floorf(ix), fy = floorf(iy);
    const float tx = ix - fx, ty = iy - fy;
    tp.wx[0] = cubic2(tx + 1.f); tp.wx[1] = cubic1(tx); tp.wx[2] = cubic1(1.f - tx); tp.wx[3] = cubic2(2.f - tx);
    tp.wy[0] = cubic2(ty + 1.f); tp.wy[1] = cubic1(ty); tp.wy[2] = cubic1(1.f - ty); tp.wy[3] = cubic2(2.f - ty);
    // clamp before the int conversion: far-out-of-range coordinates only arise from absurd rates
    const int bx = (int)fminf(fmaxf(fx, -1.0e8f), 1.0e8f), by = (int)fminf(fmaxf(fy, -1.0e8f), 1.0e8f);
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        tp.xs[d] = reflect_clip(bx - 1 + d, Wi);
        tp.ys[d] = reflect_clip(by - 1 + d, Hi);
    }
}

__global__ __launch_bounds__(256) void scale_resample_fwd_kernel(
    const float *__restrict__ x, float *__restrict__ y, const float *__restrict__ rate,
    const float *__restrict__ center, int B, int C, int Hi, int Wi, int H, int W, float two_over_h,
    float two_over_w) {
    const size_t total = (size_t)B * H * W;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int b = (int)(idx / ((size_t)H * W));
    const int p = (int)(idx - (size_t)b * H * W);
    ScaleTaps tp;
    scale_taps(p, H, W, Hi, Wi, two_over_h, two_over_w, __fdiv_rn(1.f, rate[b]), center[2 * b],
               center[2 * b + 1], tp);
    for (int c = 0; c < C; ++c) {
        const float *xp = x + ((size_t)b * C + c) * Hi * Wi;
        float acc = 0.f;
#pragma unroll
        for (int dy = 0; dy < 4; ++dy) {
            const float *row = xp + (size_t)tp.ys[dy] * Wi;
            float r = tp.wx[0] * row[tp.xs[0]];
            r = fmaf(tp.wx[1], row[tp.xs[1]], r);
            r = fmaf(tp.wx[2], row[tp.xs[2]], r);
            r = fmaf(tp.wx[3], row[tp.xs[3]], r);
            acc = fmaf(tp.wy[dy], r, acc);
        }
        y[((size_t)b * C + c) * H * W + p] = acc;
    }
}

__global__ __launch_bounds__(256) void scale_resample_bwd_kernel(
    const float *__restrict__ gy, float *__restrict__ gx, const float *__restrict__ rate,
    const float *__restrict__ center, int B, int C, int Hi, int Wi, int H, int W, float two_over_h,
    float two_over_w) {
    const size_t total = (size_t)B * H * W;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int b = (int)(idx / ((size_t)H * W));
    const int p = (int)(idx - (size_t)b * H * W);
    ScaleTaps tp;
    scale_taps(p, H, W, Hi, Wi, two_over_h, two_over_w, __fdiv_rn(1.f, rate[b]), center[2 * b],
               center[2 * b + 1], tp);
    for (int c = 0; c < C; ++c) {
        const float g = gy[((size_t)b * C + c) * H * W + p];
        float *gp = gx + ((size_t)b * C + c) * Hi * Wi;
#pragma unroll
        for (int dy = 0; dy < 4; ++dy)
#pragma unroll
            for (int dx = 0; dx < 4; ++dx)
                atomicAdd(gp + (size_t)tp.ys[dy] * Wi + tp.xs[dx], g * tp.wy[dy] * tp.wx[dx]);
    }
}

// ---- nearest-neighbour rotation about the image centre (deepinv.transform.Rotate -> torchvision rotate) ------------
// Source pixel of output (i, j), in the arithmetic of torchvision's tensor path: base grid xg = j + 0.5 - W/2,
// yg = i + 0.5 - H/2 (its linspace is exact: half-integers), theta rows divided by (W/2, H/2), the three-term product of
// the bmm, then grid_sample's un-normalisation ((g + 1) * size - 1) / 2 and nearbyint (ties to even); out-of-range
// sources read as zero.  All float32, unfused (the file is built with -ffp-contract=off).
struct RotateMap {
    float r00, r10, r01, r11;   // theta[0][0] / (W/2), theta[0][1] / (W/2), theta[1][0] / (H/2), theta[1][1] / (H/2)
};

__device__ __forceinline__ int rotate_source(int p, int H, int W, const RotateMap &m) {
    const int i = p / W, j = p - i * W;
    const float xg = ((float)j + 0.5f) - 0.5f * (float)W, yg = ((float)i + 0.5f) - 0.5f * (float)H;
    const float gx = xg * m.r00 + yg * m.r10, gy = xg * m.r01 + yg * m.r11;
    const float ix = ((gx + 1.f) * (float)W - 1.f) * 0.5f, iy = ((gy + 1.f) * (float)H - 1.f) * 0.5f;
    const float nx = nearbyintf(ix), ny = nearbyintf(iy);
    if (!(nx >= 0.f && nx <= (float)(W - 1) && ny >= 0.f && ny <= (float)(H - 1))) return -1;
    return (int)ny * W + (int)nx;
}

__global__ __launch_bounds__(256) void rotate_nearest_fwd_kernel(const float *__restrict__ x, float *__restrict__ y,
                                                                 int planes, int H, int W, RotateMap m) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= H * W) return;
    const int src = rotate_source(p, H, W, m);
    for (int c = blockIdx.y; c < planes; c += gridDim.y)
        y[(size_t)c * H * W + p] = src < 0 ? 0.f : x[(size_t)c * H * W + src];
}

__global__ __launch_bounds__(256) void rotate_nearest_bwd_kernel(const float *__restrict__ gy, float *__restrict__ gx,
                                                                 int planes, int H, int W, RotateMap m) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= H * W) return;
    const int src = rotate_source(p, H, W, m);
    if (src < 0) return;
    for (int c = blockIdx.y; c < planes; c += gridDim.y)      // several outputs may share a source pixel
        atomicAdd(gx + (size_t)c * H * W + src, gy[(size_t)c * H * W + p]);
}

// pick a tile so that tile + halo fits comfortably in LDS and small images are one tile
inline void blur_tiles(int H, int W, int kv, int kh, int &th, int &tw) {
    th = H < 64 ? H : 64;
    tw = W < 64 ? W : 64;
    (void)kv; (void)kh;
}

}  // namespace

extern "C" int sei_blur_sep_circ(const float *x, float *y, const float *tv, const float *th, int kv,
                                 int kh, int planes, int H, int W, int transpose, void *stream) {
    SEI_REQUIRE(x && y && tv && th && x != y);
    SEI_REQUIRE(planes > 0 && H > 0 && W > 0 && kv > 0 && kh > 0);
    if (kv > MAX_TAPS || kh > MAX_TAPS) return SEI_ERR_TOO_LARGE;
    int tile_h, tile_w;
    blur_tiles(H, W, kv, kh, tile_h, tile_w);
    const int tiles_y = (int)sei_ceil_div(H, tile_h), tiles_x = (int)sei_ceil_div(W, tile_w);
    const int ih = tile_h + kv - 1, iw = tile_w + kh - 1;
    const size_t lds = sizeof(float) * (128 + ((ih * iw + 3) & ~3) + (size_t)ih * tile_w);
    if (lds > 160 * 1024) return SEI_ERR_TOO_LARGE;
    // forward: shift kv/2, taps as given; adjoint: flipped taps, shift kv-1-kv/2
    const int sv = transpose ? kv - 1 - kv / 2 : kv / 2, sh = transpose ? kh - 1 - kh / 2 : kh / 2;
    const size_t grid = (size_t)planes * tiles_y * tiles_x;
    hipLaunchKernelGGL(blur_sep_circ_kernel, dim3((unsigned)grid), dim3(BLUR_THREADS), lds, (hipStream_t)stream,
                       x, y, tv, th, kv, kh, sv, sh, transpose ? 1 : 0, H, W, tile_h, tile_w, tiles_y, tiles_x);
    return sei_launch_status();
}

extern "C" int sei_blur_dense_circ(const float *x, float *y, const float *k, int kv, int kh, int planes,
                                   int H, int W, int transpose, void *stream) {
    SEI_REQUIRE(x && y && k && x != y);
    SEI_REQUIRE(planes > 0 && H > 0 && W > 0 && kv > 0 && kh > 0);
    if (kv > MAX_TAPS || kh > MAX_TAPS) return SEI_ERR_TOO_LARGE;
    int tile_h = H < 32 ? H : 32, tile_w = W < 64 ? W : 64;
    const int tiles_y = (int)sei_ceil_div(H, tile_h), tiles_x = (int)sei_ceil_div(W, tile_w);
    const int ih = tile_h + kv - 1, iw = tile_w + kh - 1;
    const size_t lds = sizeof(float) * (((kv * kh + 3) & ~3) + (size_t)ih * iw);
    if (lds > 160 * 1024) return SEI_ERR_TOO_LARGE;
    const int sv = transpose ? kv - 1 - kv / 2 : kv / 2, sh = transpose ? kh - 1 - kh / 2 : kh / 2;
    const size_t grid = (size_t)planes * tiles_y * tiles_x;
    hipLaunchKernelGGL(blur_dense_circ_kernel, dim3((unsigned)grid), dim3(BLUR_THREADS), lds, (hipStream_t)stream,
                       x, y, k, kv, kh, sv, sh, transpose ? 1 : 0, H, W, tile_h, tile_w, tiles_y, tiles_x);
    return sei_launch_status();
}

extern "C" int sei_resample_sepband(const float *x, float *y, int planes, int Hi, int Wi, int Ho, int Wo,
                                    const float *wv, const int *lov, int nbv, int stepv,
                                    const float *wh, const int *loh, int nbh, int steph, void *stream) {
    SEI_REQUIRE(x && y && wv && lov && wh && loh && x != y);
    SEI_REQUIRE(planes > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0 && nbv > 0 && nbh > 0);
    SEI_REQUIRE(stepv >= 0 && steph >= 0);
    // Output tile 16 x 32; its input footprint is at most (tile-1)*step + band per axis.
    const int tile_h = Ho < 16 ? Ho : 16, tile_w = Wo < 32 ? Wo : 32;
    int fh_max = (tile_h - 1) * stepv + nbv, fw_max = (tile_w - 1) * steph + nbh;
    if (fh_max > Hi) fh_max = Hi;
    if (fw_max > Wi) fw_max = Wi;
    const size_t lds = sizeof(float) * (((size_t)(fh_max * fw_max + 3) & ~(size_t)3) + (size_t)fh_max * tile_w);
    if (lds > 160 * 1024) return SEI_ERR_TOO_LARGE;
    const int tiles_y = (int)sei_ceil_div(Ho, tile_h), tiles_x = (int)sei_ceil_div(Wo, tile_w);
    const size_t grid = (size_t)planes * tiles_y * tiles_x;
    hipLaunchKernelGGL(resample_sepband_kernel, dim3((unsigned)grid), dim3(RS_THREADS), lds, (hipStream_t)stream,
                       x, y, Hi, Wi, Ho, Wo, wv, lov, nbv, wh, loh, nbh, tile_h, tile_w, tiles_y, tiles_x,
                       fh_max, fw_max);
    return sei_launch_status();
}

extern "C" int sei_scale_resample_fwd(const float *x, float *y, const float *rate, const float *center,
                                      int B, int C, int Hi, int Wi, int H, int W, void *stream) {
    SEI_REQUIRE(x && y && rate && center && x != y);
    SEI_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && Hi > 0 && Wi > 0);
    const size_t total = (size_t)B * H * W;
    // python: 2 / w evaluated in double, then rounded to float32 when it multiplies a float tensor
    const float two_over_h = (float)(2.0 / (double)H), two_over_w = (float)(2.0 / (double)W);
    hipLaunchKernelGGL(scale_resample_fwd_kernel, dim3((unsigned)sei_ceil_div(total, 256)), dim3(256), 0,
                       (hipStream_t)stream, x, y, rate, center, B, C, Hi, Wi, H, W, two_over_h, two_over_w);
    return sei_launch_status();
}

extern "C" int sei_scale_resample_bwd(const float *gy, float *gx, const float *rate, const float *center,
                                      int B, int C, int Hi, int Wi, int H, int W, void *stream) {
    SEI_REQUIRE(gy && gx && rate && center && gx != gy);
    SEI_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && Hi > 0 && Wi > 0);
    const size_t total = (size_t)B * H * W;
    const float two_over_h = (float)(2.0 / (double)H), two_over_w = (float)(2.0 / (double)W);
    hipLaunchKernelGGL(scale_resample_bwd_kernel, dim3((unsigned)sei_ceil_div(total, 256)), dim3(256), 0,
                       (hipStream_t)stream, gy, gx, rate, center, B, C, Hi, Wi, H, W, two_over_h, two_over_w);
    return sei_launch_status();
}

static RotateMap rotate_map(const float theta[4], int H, int W) {
    const float hw = 0.5f * (float)W, hh = 0.5f * (float)H;
    return RotateMap{theta[0] / hw, theta[1] / hw, theta[2] / hh, theta[3] / hh};
}

extern "C" int sei_rotate_nearest_fwd(const float *x, float *y, int planes, int H, int W, float t00, float t01,
                                      float t10, float t11, void *stream) {
    SEI_REQUIRE(x && y && x != y && planes > 0 && H > 0 && W > 0 && (size_t)H * W < (1u << 30));
    const float theta[4] = {t00, t01, t10, t11};
    const dim3 grid((unsigned)sei_ceil_div((size_t)H * W, 256), (unsigned)(planes < 1024 ? planes : 1024));
    hipLaunchKernelGGL(rotate_nearest_fwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, y, planes, H, W,
                       rotate_map(theta, H, W));
    return sei_launch_status();
}

extern "C" int sei_rotate_nearest_bwd(const float *gy, float *gx, int planes, int H, int W, float t00, float t01,
                                      float t10, float t11, void *stream) {
    SEI_REQUIRE(gy && gx && gx != gy && planes > 0 && H > 0 && W > 0 && (size_t)H * W < (1u << 30));
    const float theta[4] = {t00, t01, t10, t11};
    const dim3 grid((unsigned)sei_ceil_div((size_t)H * W, 256), (unsigned)(planes < 1024 ? planes : 1024));
    hipLaunchKernelGGL(rotate_nearest_bwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, gy, gx, planes, H, W,
                       rotate_map(theta, H, W));
    return sei_launch_status();
}
