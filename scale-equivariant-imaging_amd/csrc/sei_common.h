// Shared device/host helpers for libsei_hip.so (gfx950 only: wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/sei_hip.h"

#define SEI_WAVE 64

#define SEI_REQUIRE(cond) \
    do {                  \
        if (!(cond)) return SEI_ERR_BAD_ARG; \
    } while (0)

static inline int sei_launch_status() { return (int)hipGetLastError(); }

static inline size_t sei_ceil_div(size_t a, size_t b) { return (a + b - 1) / b; }

__device__ __forceinline__ int sei_mod(int v, int n) {
    int r = v % n;
    return r < 0 ? r + n : r;
}

__device__ __forceinline__ float sei_wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// Sum over the whole block (THREADS a multiple of 64, <= 1024); result valid in thread 0.
// `scratch` must hold THREADS/64 floats of LDS.
template <int THREADS>
__device__ __forceinline__ float sei_block_sum(float v, float *scratch) {
    v = sei_wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) scratch[wave] = v;
    __syncthreads();
    float r = 0.f;
    if (wave == 0) {
        r = (lane < THREADS / 64) ? scratch[lane] : 0.f;
        r = sei_wave_sum(r);
    }
    __syncthreads();
    return r;
}

// LDS-DMA (one 1-KiB piece: 16 bytes per lane to M0 + 16 lane) issued by INLINE ASM, not by
// __builtin_amdgcn_global_load_lds (round 5). With the builtin the compiler knows an LDS store is in flight and puts
// `s_waitcnt vmcnt(0)` in front of every ds_read_b64_tr_b16 that follows (it has no alias information for the
// transposing read's intrinsic; plain ds_read_b128 loads are not affected): in every kernel with a reduction-major
// operand -- all weight gradients, every data gradient dX = dY W -- the DMA of the next k-tile was drained before the
// current tile's first fragment read, i.e. nothing was prefetched (found in the ISA of the loops: "global_load_lds x10;
// s_waitcnt vmcnt(0); ds_read_b64_tr_b16 x15"). The kernels that use these helpers wait for their DMA explicitly anyway (counted
// s_waitcnt vmcnt + s_barrier), so the compiler's tracking buys nothing. M0 is written inside the asm; a kernel
// that uses these helpers issues NO LDS-DMA through the builtin (a hoisted M0 of the compiler's own would be clobbered).
// dma16_base: uniform 64-bit base + per-lane 32-bit byte offset (the lean path); dma16_lane: per-lane 64-bit address.
// (The "m0" clobber makes the asm a definition of M0 for the backend's M0-initialisation hoisting; clang warns that a
// reserved register on a clobber list is not saved around the asm, which is exactly what is wanted here.)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void dma16_base(const char *base, unsigned off, char *lds_dst) {
    const unsigned m = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void *)lds_dst;
    asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(off), "s"(base), "s"(m) : "memory", "m0");
}
__device__ __forceinline__ void dma16_lane(const void *src, char *lds_dst) {
    const unsigned m = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void *)lds_dst;
    asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(m) : "memory", "m0");
}
#pragma clang diagnostic pop

// exact (erf) GELU and its derivative, as torch.nn.GELU() (approximate='none')
__device__ __forceinline__ float sei_gelu(float x) {
    return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}
__device__ __forceinline__ float sei_dgelu(float x) {
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}

// The same two functions for results that are ROUNDED TO bf16 right away (bf16 mode only: gelu(h3) -> h4 and
// the GELU' factor of the data gradient): Phi(x) from Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7 on erf, four
// orders below the bf16 rounding of the result; one exponential shared by the cdf and the pdf, ~16 instructions
// against ~50 for erff). A 288 x 256 tile evaluates 73,728 of them in its epilogue with nothing to overlap.
__device__ __forceinline__ void sei_phi_pdf_bf16out(float x, float &cdf, float &pdf) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    const float e = __expf(-z * z);                       // exp(-x^2 / 2)
    float poly = fmaf(1.061405429f, t, -1.453152027f);
    poly = fmaf(poly, t, 1.421413741f);
    poly = fmaf(poly, t, -0.284496736f);
    poly = fmaf(poly, t, 0.254829592f);
    const float q = 0.5f * poly * t * e;                  // Phi(-|x|), no cancellation in the tail
    cdf = x >= 0.f ? 1.0f - q : q;
    pdf = 0.39894228040143267794f * e;
}
__device__ __forceinline__ float sei_gelu_bf16out(float x) {
    float cdf, pdf;
    sei_phi_pdf_bf16out(x, cdf, pdf);
    return x * cdf;
}
__device__ __forceinline__ float sei_dgelu_bf16out(float x) {
    float cdf, pdf;
    sei_phi_pdf_bf16out(x, cdf, pdf);
    return fmaf(x, pdf, cdf);
}

// One element of torch.optim.Adam (single-tensor form): lerp for exp_avg, addcmul for exp_avg_sq.
__device__ __forceinline__ float sei_adam_element(float pi, float gi, float &mi, float &vi, float beta1, float beta2,
                                              float eps, float wd, float step_size, float inv_bc2_sqrt) {
    if (wd != 0.f) gi = fmaf(wd, pi, gi);
    mi = mi + (gi - mi) * (1.f - beta1);
    vi = fmaf(beta2, vi, (1.f - beta2) * gi * gi);
    const float denom = sqrtf(vi) * inv_bc2_sqrt + eps;
    return pi - step_size * (mi / denom);
}


// mlp128.hip: the fused pointwise MLP of the 128-channel level (internal linkage between translation units; mlp_fused.hip
// dispatches to it from the sei_mlp_fused_* entry points)
bool sei_mlp128_eligible(int M, int C);
int sei_mlp128_fwd_launch(const uint16_t *h2, const uint16_t *W2, const float *b2, const uint16_t *W3, const float *b3,
                          const float *x, float res_scale, float *out, int M, int C, hipStream_t s);
int sei_mlp128_bwd_launch(const float *go, const uint16_t *h2, const uint16_t *W2, const float *b2, const uint16_t *W3T,
                          const uint16_t *W2T, float *gh2, uint16_t *go16, uint16_t *h4, uint16_t *gh3, int M, int C,
                          hipStream_t s);

// dwconv_pipe.hip: the pipelined depthwise 7x7 kernel (internal linkage between translation units, not part of the ABI)
int sei_dwconv7_pipe_launch(const float *x, const float *w, const float *bias, const float *res, float res_scale,
                            float *y, int B, int H, int W, int C, int flip, hipStream_t s);
bool sei_dwconv7_pipe_eligible(const float *x, const float *w, int B, int H, int W, int C);
bool sei_dwconv7_ln_fused_eligible(const float *x, const float *w, int B, int H, int W, int C);
int sei_dwconv7_ln_fused_launch(const float *x, const float *w, const float *bias, const float *gamma,
                                const float *beta, float *h1, void *h2, int out16, float *mean, float *rstd, int B, int H,
                                int W, int C, float eps, hipStream_t s);

// swin_bf16_kernels.hip: out[c] += sum over groups of part[group][..][c] (internal linkage between translation units)
int sei_fold_partials3(const float *part, int groups, int C, float *a, float *b, float *c3, hipStream_t s);
