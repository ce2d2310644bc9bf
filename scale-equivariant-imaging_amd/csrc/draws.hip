// The eager prologue of a training step in two launches (gfx950): the step's device-side random numbers, and the crop of
// the measurement batch -- plus the stacked SURE input inside the step.
// (reference: src/losses/sure.py:13-22 (probe b on the interior), src/transforms.py:5-24 (rate = values[floor(2 U)],
//  centre = 2 U - 1), deepinv GaussianNoise (y + sigma randn) inside EILoss, src/crop.py:26-38 (one window per batch),
//  src/losses/sure.py:24 (y + tau b).)
//
// sei_proposed_draws replaces the five launches an eager step spends on its draws -- torch.randn (interior) + the strided
// copy into b, torch.rand(B), torch.rand(B, 2) + sei_scale_params, torch.randn (noise) -- and walks THE SAME RANDOM STREAM
// as those torch calls for the same generator state: torch's CUDA / HIP generator is Philox4x32-10, and for a tensor of
// numel <= 256 * 2048 elements its kernel (ATen distribution_elementwise_grid_stride_kernel with calc_execution_policy:
// 256-thread blocks, one element per thread when the tensor fits 2048 blocks) gives element i the FIRST component of
// rocrand_normal4 / rocrand_uniform4 of the engine (seed, subsequence i, offset); every call then advances the
// generator's offset by 4. The caller passes (seed, offset) of torch's generator and advances it by 16 afterwards
// (losses.ProposedLoss.draw_into), so a run that mixes these launches with torch's own draws consumes one stream.
// Measured against torch 2.10.0+rocm7.0 on MI355X (tools/probe_torch_philox.py): the mapping and the offsets are torch's;
// the uniform draws (rates, centres) are bit-identical; of the normal draws 99.3 % are bit-identical and the rest differ
// by 1-3 ulp (<= 1e-6 relative) -- Box-Muller's logf / __sincosf come from this build's device library (ROCm 7.2), ATen's
// from the one it was built with. rocRAND's device functions are called as they are (the headers hipRAND / ATen compile
// against), with floating-point contraction off like the rest of the library (with contraction on only 85 % agree).
#include "sei_common.h"
#include <rocrand/rocrand_philox4x32_10.h>
#include <rocrand/rocrand_normal.h>
#include <rocrand/rocrand_uniform.h>

namespace {

constexpr size_t DRAW_MAX_NUMEL = (size_t)256 * 2048;     // beyond it torch's threads take several elements each

__device__ __forceinline__ float draw_normal(unsigned long long seed, unsigned long long sub, unsigned long long off) {
    rocrand_state_philox4x32_10 st;
    rocrand_init(seed, sub, off, &st);
    return rocrand_normal4(&st).x;                      // (normal_: rand * std + mean with std 1, mean 0: exact)
}
__device__ __forceinline__ float draw_uniform(unsigned long long seed, unsigned long long sub, unsigned long long off) {
    rocrand_state_philox4x32_10 st;
    rocrand_init(seed, sub, off, &st);
    const float r = rocrand_uniform4(&st).x;            // in (0, 1]; ATen's uniform_(0, 1) maps 1 to 0
    return r == 1.0f ? 0.0f : r;
}

struct DrawArgs {
    unsigned long long seed, offset;
    float *b;
    int B, C, H, W, margin;
    const float *table;
    int ntable;
    float *rate, *center, *noise;
    unsigned n_int, n_full;                              // interior / full element counts
};

// one thread per element of: [interior of b | rate (B) | centre (2 B) | noise], in torch's call order
__global__ __launch_bounds__(256) void proposed_draws_kernel(DrawArgs g) {
    unsigned i = blockIdx.x * 256u + threadIdx.x;
    if (i < g.n_int) {
        const int hi = g.H - 2 * g.margin, wi = g.W - 2 * g.margin;
        const int x = i % wi, t = i / wi, y = t % hi, pc = t / hi;          // pc = image * C + channel
        g.b[((size_t)pc * g.H + y + g.margin) * g.W + x + g.margin] = draw_normal(g.seed, i, g.offset);
        return;
    }
    i -= g.n_int;
    if (i < (unsigned)g.B) {
        const float u = draw_uniform(g.seed, i, g.offset + 4);
        int k = (int)floorf((float)g.ntable * u);                           // (sei_scale_params' arithmetic)
        k = k < 0 ? 0 : (k >= g.ntable ? g.ntable - 1 : k);
        g.rate[i] = g.table[k];
        return;
    }
    i -= g.B;
    if (i < 2u * g.B) {
        g.center[i] = 2.0f * draw_uniform(g.seed, i, g.offset + 8) - 1.0f;
        return;
    }
    i -= 2u * g.B;
    if (i < g.n_full) g.noise[i] = draw_normal(g.seed, i, g.offset + 12);
}

// out[p][r][s] = y[p][i0 + r][j0 + s] inside y, 0 in the zero padding MinSizePadding appends (src/crop.py:50-57 on a
// 4-D batch: the batched-crop quirk); p = image * C + channel
__global__ __launch_bounds__(256) void crop_window_kernel(const float *__restrict__ y, float *__restrict__ out, int planes,
                                                          int H, int W, int i0, int j0, int S) {
    const size_t n = (size_t)planes * S * S;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) {
        const int s = (int)(e % S), r = (int)((e / S) % S);
        const size_t p = e / ((size_t)S * S);
        const int yy = i0 + r, xx = j0 + s;
        out[e] = (yy < H && xx < W) ? y[(p * H + yy) * W + xx] : 0.f;
    }
}

// out[0:n] = a, out[n:2n] = a + alpha b (axpy_kernel's fmaf): the 2B-image input of the fused SURE pass in one launch
__global__ __launch_bounds__(256) void stack_axpy_kernel(const float *__restrict__ a, const float *__restrict__ b, float alpha,
                                                         float *__restrict__ out, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x, n4 = n / 4;
    const float4 *a4 = reinterpret_cast<const float4 *>(a), *b4 = reinterpret_cast<const float4 *>(b);
    float4 *o0 = reinterpret_cast<float4 *>(out), *o1 = reinterpret_cast<float4 *>(out + n);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const float4 va = a4[i], vb = b4[i];
        o0[i] = va;
        o1[i] = make_float4(fmaf(alpha, vb.x, va.x), fmaf(alpha, vb.y, va.y), fmaf(alpha, vb.z, va.z), fmaf(alpha, vb.w, va.w));
    }
}

// out = [a (na floats) | b (nb floats)]: the joint backward's 3B-row gradient from the two model calls' gradients, a model
// input into its arena buffer (nb = 0) -- one launch where autograd-side torch code issued one memcpy node per piece
__global__ __launch_bounds__(256) void concat2_kernel(const float *__restrict__ a, size_t na, const float *__restrict__ b,
                                                      size_t nb, float *__restrict__ out) {
    const size_t stride = (size_t)gridDim.x * blockDim.x, n4 = (na + nb) / 4, a4 = na / 4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride)
        reinterpret_cast<float4 *>(out)[i] = i < a4 ? reinterpret_cast<const float4 *>(a)[i]
                                                    : reinterpret_cast<const float4 *>(b)[i - a4];
}

// out = x * (*s): a stored gradient times the scalar gradient of the loss value (the backward of the loss-term functions)
__global__ __launch_bounds__(256) void scale_dev_kernel(const float *__restrict__ x, const float *__restrict__ s,
                                                        float *__restrict__ out, size_t n) {
    const float k = *s;
    const size_t stride = (size_t)gridDim.x * blockDim.x, n4 = n / 4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 v = reinterpret_cast<const float4 *>(x)[i];
        v.x *= k; v.y *= k; v.z *= k; v.w *= k;
        reinterpret_cast<float4 *>(out)[i] = v;
    }
}

__global__ void add_scalars_kernel(const float *a, const float *b, float *out) { *out = *a + *b; }

}  // namespace

extern "C" int sei_concat2_f32(const float *a, size_t na, const float *b, size_t nb, float *out, void *stream) {
    SEI_REQUIRE(a && out && na > 0 && na % 4 == 0 && nb % 4 == 0 && (nb == 0 || b));
    SEI_REQUIRE((((uintptr_t)a | (uintptr_t)b | (uintptr_t)out) & 15) == 0);
    size_t grid = sei_ceil_div((na + nb) / 4, 256);
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(concat2_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, a, na, b, nb, out);
    return sei_launch_status();
}

extern "C" int sei_scale_dev_f32(const float *x, const float *scalar, float *out, size_t n, void *stream) {
    SEI_REQUIRE(x && scalar && out && n > 0 && n % 4 == 0 && (((uintptr_t)x | (uintptr_t)out) & 15) == 0);
    size_t grid = sei_ceil_div(n / 4, 256);
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(scale_dev_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, x, scalar, out, n);
    return sei_launch_status();
}

extern "C" int sei_add_scalars(const float *a, const float *b, float *out, void *stream) {
    SEI_REQUIRE(a && b && out);
    hipLaunchKernelGGL(add_scalars_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, a, b, out);
    return sei_launch_status();
}

extern "C" int sei_proposed_draws(unsigned long long seed, unsigned long long offset, float *b, int B, int C, int H, int W,
                                  int margin, const float *table, int ntable, float *rate, float *center, float *noise,
                                  void *stream) {
    SEI_REQUIRE(b && table && rate && center && noise && B > 0 && C > 0 && H > 0 && W > 0 && ntable > 0 && margin >= 0);
    SEI_REQUIRE(H > 2 * margin && W > 2 * margin && offset % 4 == 0);
    const size_t n_full = (size_t)B * C * H * W, n_int = (size_t)B * C * (H - 2 * margin) * (W - 2 * margin);
    SEI_REQUIRE(n_full <= DRAW_MAX_NUMEL);               // (larger tensors: torch's threads take several elements each)
    DrawArgs g;
    g.seed = seed; g.offset = offset; g.b = b; g.B = B; g.C = C; g.H = H; g.W = W; g.margin = margin;
    g.table = table; g.ntable = ntable; g.rate = rate; g.center = center; g.noise = noise;
    g.n_int = (unsigned)n_int; g.n_full = (unsigned)n_full;
    const size_t total = n_int + 3 * (size_t)B + n_full;
    hipLaunchKernelGGL(proposed_draws_kernel, dim3((unsigned)sei_ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, g);
    return sei_launch_status();
}

extern "C" size_t sei_proposed_draws_max_numel(void) { return DRAW_MAX_NUMEL; }

extern "C" int sei_crop_window(const float *y, float *out, int planes, int H, int W, int i0, int j0, int S, void *stream) {
    SEI_REQUIRE(y && out && planes > 0 && H > 0 && W > 0 && S > 0 && i0 >= 0 && j0 >= 0);
    const size_t n = (size_t)planes * S * S;
    size_t grid = sei_ceil_div(n, 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(crop_window_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, y, out, planes, H, W, i0,
                       j0, S);
    return sei_launch_status();
}

extern "C" int sei_stack_axpy(const float *a, const float *b, float alpha, float *out, size_t n, void *stream) {
    SEI_REQUIRE(a && b && out && n > 0 && n % 4 == 0);
    SEI_REQUIRE((((uintptr_t)a | (uintptr_t)b | (uintptr_t)out) & 15) == 0);
    size_t grid = sei_ceil_div(n / 4, 256);
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(stack_axpy_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, a, b, alpha, out, n);
    return sei_launch_status();
}
