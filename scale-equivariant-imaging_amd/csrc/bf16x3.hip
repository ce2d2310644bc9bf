// Elementwise kernels of the split-bf16 GEMM mode (--compute_dtype bf16x3; models/_ops.py gemm_x3), gfx950.
// (reference: src/models/convolutional.py:33-51,95-110,136-150 -- the 1x1 convolutions, float32 throughout. This mode
//  evaluates each of their GEMMs as three bf16 MFMA products with float32 accumulation:
//      a = a_hi + a_lo + r_a,  a_hi = bf16(a),  a_lo = bf16(a - a_hi)     (|r_a| <= 2^-17 |a|)
//      a b  ~  a_lo b_hi + a_hi b_lo + a_hi b_hi                          (a_lo b_lo ~ 2^-16 |a b| is dropped)
//  -- 16 mantissa bits per operand at a third of the bf16 MFMA rate instead of the 157 TFLOP/s float32 MFMA.)
// All three kernels stream: 16 bytes per lane, grid-stride.
#include "sei_common.h"

namespace {

__device__ __forceinline__ unsigned short x3_f2bf(float v) {
    const __bf16 b = (__bf16)v;
    return __builtin_bit_cast(unsigned short, b);
}
__device__ __forceinline__ float x3_bf2f(unsigned short u) { return __builtin_bit_cast(float, (unsigned)u << 16); }

// planes[i] = bf16(x[i]); planes[n + i] = bf16(x[i] - float(planes[i]))
__global__ __launch_bounds__(256) void split_bf16x2_kernel(const float *__restrict__ x, unsigned short *__restrict__ hi,
                                                           unsigned short *__restrict__ lo, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x, n4 = n / 4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const float4 v = reinterpret_cast<const float4 *>(x)[i];
        ushort4 h, l;
        h.x = x3_f2bf(v.x); h.y = x3_f2bf(v.y); h.z = x3_f2bf(v.z); h.w = x3_f2bf(v.w);
        l.x = x3_f2bf(v.x - x3_bf2f(h.x)); l.y = x3_f2bf(v.y - x3_bf2f(h.y));
        l.z = x3_f2bf(v.z - x3_bf2f(h.z)); l.w = x3_f2bf(v.w - x3_bf2f(h.w));
        reinterpret_cast<ushort4 *>(hi)[i] = h;
        reinterpret_cast<ushort4 *>(lo)[i] = l;
    }
    for (size_t i = n4 * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const unsigned short h = x3_f2bf(x[i]);
        hi[i] = h;
        lo[i] = x3_f2bf(x[i] - x3_bf2f(h));
    }
}

// three planes for a reduction that is concatenated over the three products (a weight gradient: the planes of a
// reduction-major operand stack along its reduction index): pattern 0 = [head, head, remainder] (the A side),
// pattern 1 = [head, remainder, head] (the B side) -- sum_p A_p^T B_p = a_hi b_hi + a_hi b_lo + a_lo b_hi
__global__ __launch_bounds__(256) void split_bf16x3_kernel(const float *__restrict__ x, unsigned short *__restrict__ p0,
                                                           unsigned short *__restrict__ p1, unsigned short *__restrict__ p2,
                                                           size_t n, int pattern) {
    const size_t stride = (size_t)gridDim.x * blockDim.x, n4 = n / 4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const float4 v = reinterpret_cast<const float4 *>(x)[i];
        ushort4 h, l;
        h.x = x3_f2bf(v.x); h.y = x3_f2bf(v.y); h.z = x3_f2bf(v.z); h.w = x3_f2bf(v.w);
        l.x = x3_f2bf(v.x - x3_bf2f(h.x)); l.y = x3_f2bf(v.y - x3_bf2f(h.y));
        l.z = x3_f2bf(v.z - x3_bf2f(h.z)); l.w = x3_f2bf(v.w - x3_bf2f(h.w));
        reinterpret_cast<ushort4 *>(p0)[i] = h;
        reinterpret_cast<ushort4 *>(p1)[i] = pattern ? l : h;
        reinterpret_cast<ushort4 *>(p2)[i] = pattern ? h : l;
    }
}

// y = gelu(x) (exact erf form, as the float32 GEMM epilogue SEI_EPI_BIAS_GELU applies it)
__global__ __launch_bounds__(256) void gelu_f32_kernel(const float *__restrict__ x, float *__restrict__ y, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x, n4 = n / 4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 v = reinterpret_cast<const float4 *>(x)[i];
        v.x = sei_gelu(v.x); v.y = sei_gelu(v.y); v.z = sei_gelu(v.z); v.w = sei_gelu(v.w);
        reinterpret_cast<float4 *>(y)[i] = v;
    }
    for (size_t i = n4 * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) y[i] = sei_gelu(x[i]);
}

// d *= gelu'(h) (SEI_EPI_MUL_DGELU's factor)
__global__ __launch_bounds__(256) void mul_dgelu_f32_kernel(float *__restrict__ d, const float *__restrict__ h, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x, n4 = n / 4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 v = reinterpret_cast<const float4 *>(d)[i];
        const float4 t = reinterpret_cast<const float4 *>(h)[i];
        v.x *= sei_dgelu(t.x); v.y *= sei_dgelu(t.y); v.z *= sei_dgelu(t.z); v.w *= sei_dgelu(t.w);
        reinterpret_cast<float4 *>(d)[i] = v;
    }
    for (size_t i = n4 * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) d[i] *= sei_dgelu(h[i]);
}

inline unsigned x3_grid(size_t n) {
    size_t grid = sei_ceil_div(n / 4 + 1, 256);
    return (unsigned)(grid > 4096 ? 4096 : grid);
}

}  // namespace

extern "C" int sei_split_bf16x2(const float *x, uint16_t *planes, size_t n, void *stream) {
    SEI_REQUIRE(x && planes && n > 0);
    SEI_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)planes & 7) == 0 && n % 4 == 0);     // (both planes 8-byte aligned)
    hipLaunchKernelGGL(split_bf16x2_kernel, dim3(x3_grid(n)), dim3(256), 0, (hipStream_t)stream, x, planes, planes + n, n);
    return sei_launch_status();
}

extern "C" int sei_split_bf16x3(const float *x, uint16_t *planes, size_t n, int pattern, void *stream) {
    SEI_REQUIRE(x && planes && n > 0 && (pattern == 0 || pattern == 1));
    SEI_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)planes & 7) == 0 && n % 4 == 0);
    hipLaunchKernelGGL(split_bf16x3_kernel, dim3(x3_grid(n)), dim3(256), 0, (hipStream_t)stream, x, planes, planes + n,
                       planes + 2 * n, n, pattern);
    return sei_launch_status();
}

extern "C" int sei_gelu_f32(const float *x, float *y, size_t n, void *stream) {
    SEI_REQUIRE(x && y && n > 0 && (((uintptr_t)x | (uintptr_t)y) & 15) == 0);
    hipLaunchKernelGGL(gelu_f32_kernel, dim3(x3_grid(n)), dim3(256), 0, (hipStream_t)stream, x, y, n);
    return sei_launch_status();
}

extern "C" int sei_mul_dgelu_f32(float *d, const float *h, size_t n, void *stream) {
    SEI_REQUIRE(d && h && n > 0 && (((uintptr_t)d | (uintptr_t)h) & 15) == 0);
    hipLaunchKernelGGL(mul_dgelu_f32_kernel, dim3(x3_grid(n)), dim3(256), 0, (hipStream_t)stream, d, h, n);
    return sei_launch_status();
}
