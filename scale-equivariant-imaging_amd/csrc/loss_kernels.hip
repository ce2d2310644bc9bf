// Loss-side streaming kernels for gfx950: axpy (noise / SURE probe), fused SURE terms + gradients,
// fused MSE + gradient. HBM-bound; reductions are two-stage (per-block partials in a caller-provided
// workspace, then one finishing block) so results are bitwise reproducible run to run (no float atomics).
//
//   sei_axpy       : y + sigma*n (deepinv GaussianNoise), y + tau*b (losses/sure.py:24)
//   sei_sure_terms : losses/sure.py:24-31 (mc_div) and :57-62 (cropped mse) + d/dy1, d/dy2
//   sei_mse_terms  : deepinv `mse` metric inside EILoss (losses/__init__.py:117-122) + d/dx3
#include "sei_common.h"

namespace {

constexpr int RED_THREADS = 256;

__global__ __launch_bounds__(256) void axpy_kernel(const float *__restrict__ a, const float *__restrict__ b,
                                                   float alpha, float *__restrict__ out, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t n4 = n / 4;
    const float4 *a4 = reinterpret_cast<const float4 *>(a);
    const float4 *b4 = reinterpret_cast<const float4 *>(b);
    float4 *o4 = reinterpret_cast<float4 *>(out);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const float4 va = a4[i], vb = b4[i];
        o4[i] = make_float4(fmaf(alpha, vb.x, va.x), fmaf(alpha, vb.y, va.y), fmaf(alpha, vb.z, va.z),
                            fmaf(alpha, vb.w, va.w));
    }
    for (size_t i = n4 * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        out[i] = fmaf(alpha, b[i], a[i]);
}

// stage 1: per-block partial sums of the two SURE terms + elementwise gradients
__global__ __launch_bounds__(RED_THREADS) void sure_terms_kernel(
    const float *__restrict__ y, const float *__restrict__ y1, const float *__restrict__ y2,
    const float *__restrict__ b, size_t total, int H, int W, int md, int mm, float inv_tau, float c_mse,
    float c_div, float *__restrict__ g1, float *__restrict__ g2, float *__restrict__ work) {
    __shared__ float scratch[RED_THREADS / 64];
    float s_div = 0.f, s_mse = 0.f;
    const size_t stride = (size_t)gridDim.x * RED_THREADS;
    for (size_t i = (size_t)blockIdx.x * RED_THREADS + threadIdx.x; i < total; i += stride) {
        const int p = (int)(i % ((size_t)H * W));
        const int r = p / W, c = p - r * W;
        const bool in_d = r >= md && r < H - md && c >= md && c < W - md;
        const bool in_m = r >= mm && r < H - mm && c >= mm && c < W - mm;
        const float v1 = y1[i];
        float ga = 0.f, gb = 0.f;
        if (in_d) {
            const float bb = b[i];
            s_div += bb * (y2[i] - v1) * inv_tau;
            gb = c_div * bb * inv_tau;
            ga = -gb;
        }
        if (in_m) {
            const float d = v1 - y[i];
            s_mse = fmaf(d, d, s_mse);
            ga = fmaf(2.f * c_mse, d, ga);
        }
        g1[i] = ga;
        g2[i] = gb;
    }
    const float bd = sei_block_sum<RED_THREADS>(s_div, scratch);
    const float bm = sei_block_sum<RED_THREADS>(s_mse, scratch);
    if (threadIdx.x == 0) {
        work[blockIdx.x] = bd;
        work[SEI_REDUCE_BLOCKS + blockIdx.x] = bm;
    }
}

__global__ __launch_bounds__(RED_THREADS) void mse_terms_kernel(const float *__restrict__ a,
                                                                const float *__restrict__ b, size_t n,
                                                                float scale, float *__restrict__ ga,
                                                                float *__restrict__ work) {
    __shared__ float scratch[RED_THREADS / 64];
    float s = 0.f;
    const size_t stride = (size_t)gridDim.x * RED_THREADS;
    for (size_t i = (size_t)blockIdx.x * RED_THREADS + threadIdx.x; i < n; i += stride) {
        const float d = a[i] - b[i];
        s = fmaf(d, d, s);
        ga[i] = scale * d;
    }
    const float bs = sei_block_sum<RED_THREADS>(s, scratch);
    if (threadIdx.x == 0) work[blockIdx.x] = bs;
}

// sum over pixels of (Y(a) - Y(b))^2 for planar RGB images (3, npix): the numerator of the luma PSNR
__global__ __launch_bounds__(RED_THREADS) void luma_sqerr_kernel(const float *__restrict__ a,
                                                                 const float *__restrict__ b, size_t npix,
                                                                 float *__restrict__ work) {
    __shared__ float scratch[RED_THREADS / 64];
    float s = 0.f;
    const size_t stride = (size_t)gridDim.x * RED_THREADS;
    for (size_t i = (size_t)blockIdx.x * RED_THREADS + threadIdx.x; i < npix; i += stride) {
        const float ya = 0.299f * a[i] + 0.587f * a[npix + i] + 0.114f * a[2 * npix + i];
        const float yb = 0.299f * b[i] + 0.587f * b[npix + i] + 0.114f * b[2 * npix + i];
        const float d = ya - yb;
        s = fmaf(d, d, s);
    }
    const float bs = sei_block_sum<RED_THREADS>(s, scratch);
    if (threadIdx.x == 0) work[blockIdx.x] = bs;
}

// stage 2: one block sums `nparts` partials of each of `nout` quantities (stored SEI_REDUCE_BLOCKS apart)
__global__ __launch_bounds__(RED_THREADS) void finish_sums_kernel(const float *__restrict__ work, int nparts,
                                                                  int nout, float *__restrict__ out) {
    __shared__ float scratch[RED_THREADS / 64];
    for (int q = 0; q < nout; ++q) {
        float v = 0.f;
        for (int i = threadIdx.x; i < nparts; i += RED_THREADS) v += work[q * SEI_REDUCE_BLOCKS + i];
        const float s = sei_block_sum<RED_THREADS>(v, scratch);
        if (threadIdx.x == 0) out[q] = s;
    }
}

// stage 2 with the loss value formed on the device: out[nout] = sum_q coef[q] * out[q] + bias (the SURE loss = c_mse * mse sum
// + c_div * div sum - constant; the weighted mean squared error): the step's scalar arithmetic is not a chain of 0-dim
// torch kernels (round 5: five fewer launches in the forward pass of a proposed-loss step)
struct FinishCoef {
    float coef[2], bias;
};
__global__ __launch_bounds__(RED_THREADS) void finish_loss_kernel(const float *__restrict__ work, int nparts, int nout,
                                                                  FinishCoef fc, float *__restrict__ out) {
    __shared__ float scratch[RED_THREADS / 64];
    float total = fc.bias;
    for (int q = 0; q < nout; ++q) {
        float v = 0.f;
        for (int i = threadIdx.x; i < nparts; i += RED_THREADS) v += work[q * SEI_REDUCE_BLOCKS + i];
        const float s = sei_block_sum<RED_THREADS>(v, scratch);
        if (threadIdx.x == 0) out[q] = s;
        total = fmaf(fc.coef[q], s, total);             // (only thread 0's `s` is the block sum; only it writes below)
    }
    if (threadIdx.x == 0) out[nout] = total;
}

inline int reduce_blocks(size_t n) {
    size_t g = sei_ceil_div(n, (size_t)RED_THREADS * 4);
    if (g < 1) g = 1;
    if (g > SEI_REDUCE_BLOCKS) g = SEI_REDUCE_BLOCKS;
    return (int)g;
}

// zero up to 8 slices of one buffer in one launch (the gaps of the gradient bucket between the stored weight gradients)
struct ZeroRanges {
    unsigned long long off[8], len[8], start[9];     // start: prefix sums of the lengths (in elements)
    int count;
};
__global__ __launch_bounds__(256) void zero_ranges_kernel(float *__restrict__ base, ZeroRanges r) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < r.start[r.count]; i += stride) {
        int k = 0;
        while (k + 1 < r.count && i >= r.start[k + 1]) ++k;
        base[r.off[k] + (i - r.start[k])] = 0.f;
    }
}
}  // namespace

extern "C" int sei_zero_ranges(float *base, const unsigned long long *off_len_pairs, int count, void *stream) {
    SEI_REQUIRE(base && off_len_pairs && count > 0 && count <= 8);
    ZeroRanges r;
    r.count = count;
    r.start[0] = 0;
    for (int k = 0; k < count; ++k) {
        r.off[k] = off_len_pairs[2 * k];
        r.len[k] = off_len_pairs[2 * k + 1];
        r.start[k + 1] = r.start[k] + r.len[k];
    }
    if (r.start[count] == 0) return 0;
    size_t grid = sei_ceil_div((size_t)r.start[count], 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(zero_ranges_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, base, r);
    return sei_launch_status();
}

extern "C" int sei_axpy(const float *a, const float *b, float alpha, float *out, size_t n, void *stream) {
    SEI_REQUIRE(a && b && out && n > 0);
    SEI_REQUIRE((((uintptr_t)a | (uintptr_t)b | (uintptr_t)out) & 15) == 0);
    size_t grid = sei_ceil_div(n / 4 + 1, 256);
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(axpy_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, a, b, alpha, out, n);
    return sei_launch_status();
}

// The arithmetic behind the scale transform's two uniform draws (reference src/transforms.py:5-24): rate = table[floor(n u)],
// centre = 2 u' - 1 -- the float32 operations torch performs one elementwise kernel at a time, in one launch.
__global__ __launch_bounds__(256) void scale_params_kernel(const float *__restrict__ u, const float *__restrict__ v,
                                                           const float *__restrict__ table, int ntable, int B,
                                                           float *__restrict__ rate, float *__restrict__ center) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B) {
        int k = (int)floorf((float)ntable * u[i]);
        k = k < 0 ? 0 : (k >= ntable ? ntable - 1 : k);            // (rand() < 1: never clamps)
        rate[i] = table[k];
    }
    if (i < 2 * B) center[i] = 2.0f * v[i] - 1.0f;
}

extern "C" int sei_scale_params(const float *u, const float *v, const float *table, int ntable, int B, float *rate,
                                float *center, void *stream) {
    SEI_REQUIRE(u && v && table && rate && center && ntable > 0 && B > 0);
    hipLaunchKernelGGL(scale_params_kernel, dim3((unsigned)sei_ceil_div((size_t)2 * B, 256)), dim3(256), 0,
                       (hipStream_t)stream, u, v, table, ntable, B, rate, center);
    return sei_launch_status();
}

extern "C" int sei_sure_terms(const float *y, const float *y1, const float *y2, const float *b, int planes,
                              int H, int W, int margin_div, int margin_mse, float tau, float c_mse,
                              float c_div, float *out2, float *g1, float *g2, float *work, void *stream) {
    SEI_REQUIRE(y && y1 && y2 && b && out2 && g1 && g2 && work);
    SEI_REQUIRE(planes > 0 && H > 0 && W > 0 && margin_div >= 0 && margin_mse >= 0 && tau != 0.f);
    SEI_REQUIRE(2 * margin_div < H && 2 * margin_div < W && 2 * margin_mse < H && 2 * margin_mse < W);
    const size_t total = (size_t)planes * H * W;
    const int grid = reduce_blocks(total);
    hipLaunchKernelGGL(sure_terms_kernel, dim3(grid), dim3(RED_THREADS), 0, (hipStream_t)stream, y, y1, y2, b,
                       total, H, W, margin_div, margin_mse, 1.0f / tau, c_mse, c_div, g1, g2, work);
    hipLaunchKernelGGL(finish_sums_kernel, dim3(1), dim3(RED_THREADS), 0, (hipStream_t)stream,
                       (const float *)work, grid, 2, out2);
    return sei_launch_status();
}

// sei_sure_terms + the loss value: out3[0] = div sum, out3[1] = mse sum, out3[2] = c_mse out3[1] + c_div out3[0] - cst.
extern "C" int sei_sure_loss(const float *y, const float *y1, const float *y2, const float *b, int planes, int H, int W,
                             int margin_div, int margin_mse, float tau, float c_mse, float c_div, float cst, float *out3,
                             float *g1, float *g2, float *work, void *stream) {
    SEI_REQUIRE(y && y1 && y2 && b && out3 && g1 && g2 && work);
    SEI_REQUIRE(planes > 0 && H > 0 && W > 0 && margin_div >= 0 && margin_mse >= 0 && tau != 0.f);
    SEI_REQUIRE(2 * margin_div < H && 2 * margin_div < W && 2 * margin_mse < H && 2 * margin_mse < W);
    const size_t total = (size_t)planes * H * W;
    const int grid = reduce_blocks(total);
    hipLaunchKernelGGL(sure_terms_kernel, dim3(grid), dim3(RED_THREADS), 0, (hipStream_t)stream, y, y1, y2, b,
                       total, H, W, margin_div, margin_mse, 1.0f / tau, c_mse, c_div, g1, g2, work);
    FinishCoef fc;
    fc.coef[0] = c_div; fc.coef[1] = c_mse; fc.bias = -cst;
    hipLaunchKernelGGL(finish_loss_kernel, dim3(1), dim3(RED_THREADS), 0, (hipStream_t)stream, (const float *)work, grid, 2,
                       fc, out3);
    return sei_launch_status();
}

// sei_mse_terms + the loss value: out2[0] = sum (a - b)^2, out2[1] = value_scale * out2[0].
extern "C" int sei_mse_loss(const float *a, const float *b, size_t n, float grad_scale, float value_scale, float *out2,
                            float *ga, float *work, void *stream) {
    SEI_REQUIRE(a && b && out2 && ga && work && n > 0);
    const int grid = reduce_blocks(n);
    hipLaunchKernelGGL(mse_terms_kernel, dim3(grid), dim3(RED_THREADS), 0, (hipStream_t)stream, a, b, n, grad_scale, ga,
                       work);
    FinishCoef fc;
    fc.coef[0] = value_scale; fc.coef[1] = 0.f; fc.bias = 0.f;
    hipLaunchKernelGGL(finish_loss_kernel, dim3(1), dim3(RED_THREADS), 0, (hipStream_t)stream, (const float *)work, grid, 1,
                       fc, out2);
    return sei_launch_status();
}

extern "C" int sei_mse_terms(const float *a, const float *b, size_t n, float scale, float *out1, float *ga,
                             float *work, void *stream) {
    SEI_REQUIRE(a && b && out1 && ga && work && n > 0);
    const int grid = reduce_blocks(n);
    hipLaunchKernelGGL(mse_terms_kernel, dim3(grid), dim3(RED_THREADS), 0, (hipStream_t)stream, a, b, n, scale,
                       ga, work);
    hipLaunchKernelGGL(finish_sums_kernel, dim3(1), dim3(RED_THREADS), 0, (hipStream_t)stream,
                       (const float *)work, grid, 1, out1);
    return sei_launch_status();
}


extern "C" int sei_luma_sqerr(const float *a, const float *b, size_t npix, float *out1, float *work, void *stream) {
    SEI_REQUIRE(a && b && out1 && work && npix > 0);
    const int grid = reduce_blocks(npix);
    hipLaunchKernelGGL(luma_sqerr_kernel, dim3(grid), dim3(RED_THREADS), 0, (hipStream_t)stream, a, b, npix, work);
    hipLaunchKernelGGL(finish_sums_kernel, dim3(1), dim3(RED_THREADS), 0, (hipStream_t)stream,
                       (const float *)work, grid, 1, out1);
    return sei_launch_status();
}
