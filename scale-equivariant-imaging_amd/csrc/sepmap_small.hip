// sepmap_small_kernel: the ideal resamplers of the DEEP levels in one pass (round 5).
//
//   y[b, :, :, c] = L1 X R1^T + L2 X R2^T,   X = x[b, :, :, c]  (Hi x Wi),  L: (Ho, Hi), R: (Wo, Wi)
//
// (reference: IdealDownsample / IdealUpsample, src/models/convolutional.py:54-92,113-133 -- rfft2, fftshift, crop or
// embed, the discarded ifftshift, irfft2 -- which models/_mats.py states as this separable rank-2 map.)
//
// The 48-pixel crop reaches the deep levels as 12 x 12, 6 x 6 and 3 x 3 images with 512 - 8192 channels. sepmap_mfma_kernel
// takes extents of 24 - 64; below that the maps ran on the two-launch float32 kernels (W pass into a two-plane HBM
// intermediate, H pass out of it): 15 of a step's 24 resampler launches, ~28 us each for 24 - 94 MB of traffic.
// Here a workgroup item is one image x 64 channels: its Hi x Wi x 64 input tile goes through LDS once (<= 48 KB: three or
// more workgroups per CU overlap each other's loads), a thread owns (channel, output column): the W pass of both terms
// stays in registers (2 Hi values), the H pass reads them back from there, whole 256-byte channel rows are stored; the
// matrix entries are wave-uniform and come through the scalar cache. Nothing but x in and y out touches HBM; the
// arithmetic is float32 FMAs (so the bf16 mode loses no precision here).
//
// Measured (tools/exp_sepmap_small.py, 64 / 96 images, us per launch, this kernel vs the two launches): 6 -> 3 at 2048
// channels 12.4 vs 15.5 / 12.3 vs 19.3; 3 -> 6 at 8192 33.9 vs 39.8 / 47.2 vs 60.6; 6 -> 12 at 2048 38.9 vs 41.5 / 49.5 vs 58.7
// (2.4 - 3.0 TB/s of x + y). With 12 x 12 inputs it LOSES -- 12 -> 6 at 512 channels 23.8 vs 17.2, 12 -> 24 112 vs 49: 144
// ds_read_b32 and 288 + 576 FMAs per (lane, output column) and 24 dependent scalar-load rounds make it VALU- and latency-
// bound -- so input extents stop at 8 and those maps stay on the two-launch kernels.
#include "sei_common.h"

namespace {

constexpr int SS_THREADS = 256;
constexpr int SS_CH = 64;              // channels per item (one per lane)
constexpr int SS_MAX_IN = 8;           // largest input extent: 12 x 12 inputs LOSE to the two-launch kernels (below)
constexpr int SS_MAX_OUT = 24;         // largest output extent
constexpr int SS_MAX_PIX = 64;         // Hi * Wi: the tile is Hi * Wi * 256 bytes of LDS (<= 16 KB)

struct SsGeom {
    int B, Hi, Wi, Ho, Wo, C;
    int out16;             // y is bf16 (round to nearest even of the float32 result): the GEMM operand the Downsample's
};                         // 1x1 convolution reads -- no float32 copy, no cast pass

template <int MAXE>
__global__ __launch_bounds__(SS_THREADS) void sepmap_small_kernel(const float *__restrict__ x, float *__restrict__ y,
                                                                  const float *__restrict__ L1, const float *__restrict__ R1,
                                                                  const float *__restrict__ L2, const float *__restrict__ R2,
                                                                  SsGeom g, int items) {
    extern __shared__ __attribute__((aligned(16))) float sX[];  // [Hi * Wi][64]
    const int tid = threadIdx.x, Hi = g.Hi, Wi = g.Wi, Ho = g.Ho, Wo = g.Wo, C = g.C;
    const int groups = C / SS_CH, npix = Hi * Wi;
    const int q = tid & 15, p0 = tid >> 4;                     // tile loads: 16 lanes x float4 per 64-channel pixel row
    const int c = tid & 63;                                    // arithmetic: lane = channel, wave = output column (+ 4 k)
    const int w0 = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int item = blockIdx.x; item < items; item += gridDim.x) {
        const int b = item / groups, c0 = (item - b * groups) * SS_CH;
        __syncthreads();                                       // the previous item's reads of sX
        const float *xb = x + (size_t)b * npix * C + c0 + 4 * q;
        for (int p = p0; p < npix; p += SS_THREADS / 16)
            *reinterpret_cast<float4 *>(sX + p * SS_CH + 4 * q) = *reinterpret_cast<const float4 *>(xb + (size_t)p * C);
        __syncthreads();
        for (int wo = w0; wo < Wo; wo += SS_THREADS / 64) {    // wave-uniform: the matrix entries below are SCALAR loads
            // (every coefficient is the same for the 64 lanes of a wave: read through the scalar cache into SGPRs, an FMA
            // takes it as its scalar operand. As LDS broadcast reads they were one LDS instruction per FMA: 12 -> 24 at
            // 512 channels spent ~750 LDS instructions per wave and output column, 576 of them coefficients -- 48 us per
            // launch against 28 on the two-launch kernels this one replaces.)
            const float *r1p = R1 + wo * Wi, *r2p = R2 + wo * Wi;
            float t1[MAXE], t2[MAXE];
#pragma unroll
            for (int hi = 0; hi < MAXE; ++hi) {
                float a1 = 0.f, a2 = 0.f;
                if (hi < Hi) {
#pragma unroll
                    for (int wi = 0; wi < MAXE; ++wi) {
                        if (wi < Wi) {
                            const float v = sX[(hi * Wi + wi) * SS_CH + c];
                            a1 = fmaf(r1p[wi], v, a1);
                            a2 = fmaf(r2p[wi], v, a2);
                        }
                    }
                }
                t1[hi] = a1;
                t2[hi] = a2;
            }
            const size_t ybase = ((size_t)b * Ho * Wo + wo) * C + c0 + c;
            float *yb = y + ybase;
            unsigned short *yb16 = reinterpret_cast<unsigned short *>(y) + ybase;
            for (int ho = 0; ho < Ho; ++ho) {
                const float *l1p = L1 + ho * Hi, *l2p = L2 + ho * Hi;
                float acc = 0.f;
#pragma unroll
                for (int hi = 0; hi < MAXE; ++hi) {
                    if (hi < Hi) {
                        acc = fmaf(l1p[hi], t1[hi], acc);
                        acc = fmaf(l2p[hi], t2[hi], acc);
                    }
                }
                if (g.out16) {
                    const __bf16 h = (__bf16)acc;
                    yb16[(size_t)ho * Wo * C] = __builtin_bit_cast(unsigned short, h);
                } else {
                    yb[(size_t)ho * Wo * C] = acc;
                }
            }
        }
    }
}

// The same map with NO LDS and no barrier (3 x 3 and 6 x 6 inputs: the deep levels of a 48-pixel crop): a WAVE item is one
// image x 64 channels, lane = channel; the HI x WI inputs of the lane's channel sit in registers, both passes run there,
// the coefficients come through the scalar cache. The staged kernel above spends two workgroup barriers per 2 - 9 KB tile
// and, with 6 output columns on 4 waves, leaves two waves idle in every second round; here waves never wait for one
// another and 32 of them fit a CU. Same FMA order: bit-identical results. tools/exp_sepmap_small.py, us per launch, staged
// -> this kernel (64 / 96 images): 3 -> 6 at 8192 channels 34.7 -> 19.8 / 48.3 -> 25.0 (4.8 / 5.7 TB/s), 6 -> 12 at 2048
// 40.4 -> 27.9 / 51.0 -> 33.6, 6 -> 3 at 2048 11.5 -> 11.4 / 12.5 -> 11.0. (Extents as template arguments: with run-time
// extents the 64 load offsets took 128 SGPRs and the kernel spilled 255 of them.)
template <int HI, int WI>
__global__ __launch_bounds__(SS_THREADS) void sepmap_small_wave_kernel(const float *__restrict__ x, float *__restrict__ y,
                                                                       const float *__restrict__ L1, const float *__restrict__ R1,
                                                                       const float *__restrict__ L2, const float *__restrict__ R2,
                                                                       SsGeom g, int items, int parts) {
    const int Ho = g.Ho, Wo = g.Wo, C = g.C;
    const int groups = C / SS_CH;
    const int c = threadIdx.x & 63;
    const int w0 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // parts > 1: an item's output columns are dealt to `parts` waves (each reads the item's inputs: L1 / L2 hits) -- with few
    // items (32 images x 2048 channels = 1024) one wave per SIMD walked a whole item on its own: load, 2592 FMAs, 144 stores
    const int cols = Wo / parts;
    for (int wi_ = blockIdx.x * (SS_THREADS / 64) + w0; wi_ < items * parts; wi_ += gridDim.x * (SS_THREADS / 64)) {
        const int item = wi_ / parts, part = wi_ - item * parts;
        const int b = item / groups, c0 = (item - b * groups) * SS_CH;
        const float *xb = x + (size_t)b * HI * WI * C + c0 + c;
        float xin[HI][WI];
#pragma unroll
        for (int hi = 0; hi < HI; ++hi)
#pragma unroll
            for (int wi = 0; wi < WI; ++wi) xin[hi][wi] = xb[(size_t)(hi * WI + wi) * C];
        for (int wo = part * cols; wo < (part + 1) * cols; ++wo) {
            const float *r1p = R1 + wo * WI, *r2p = R2 + wo * WI;
            float t1[HI], t2[HI];
#pragma unroll
            for (int hi = 0; hi < HI; ++hi) {
                float a1 = 0.f, a2 = 0.f;
#pragma unroll
                for (int wi = 0; wi < WI; ++wi) {
                    a1 = fmaf(r1p[wi], xin[hi][wi], a1);
                    a2 = fmaf(r2p[wi], xin[hi][wi], a2);
                }
                t1[hi] = a1;
                t2[hi] = a2;
            }
            const size_t ybase = ((size_t)b * Ho * Wo + wo) * C + c0 + c;
            float *yb = y + ybase;
            unsigned short *yb16 = reinterpret_cast<unsigned short *>(y) + ybase;
            for (int ho = 0; ho < Ho; ++ho) {
                const float *l1p = L1 + ho * HI, *l2p = L2 + ho * HI;
                float acc = 0.f;
#pragma unroll
                for (int hi = 0; hi < HI; ++hi) {
                    acc = fmaf(l1p[hi], t1[hi], acc);
                    acc = fmaf(l2p[hi], t2[hi], acc);
                }
                if (g.out16) {
                    const __bf16 h = (__bf16)acc;
                    yb16[(size_t)ho * Wo * C] = __builtin_bit_cast(unsigned short, h);
                } else {
                    yb[(size_t)ho * Wo * C] = acc;
                }
            }
        }
    }
}

inline bool ss_plan(int B, int Hi, int Wi, int Ho, int Wo, int C) {
    if (B <= 0 || C <= 0 || C % SS_CH != 0) return false;
    // 12 x 12 inputs REDUCED to at most 6 x 6 (the transposed 6 -> 12 upsampler of the backward pass, 2048 channels): the
    // barrier-free wave kernel with the 144 inputs in registers -- 36 outputs per lane, unlike the 12 -> 24 case above
    // (tools/exp_sepmap_reduce12.py, us, this kernel / the matrix-core kernel: 96 x 2048 channels 40 / 78, 96 x 512 15.7 / 22.6,
    // 64 x 512 14.2 / 16.0, 32 x 512 15.3 / 9.8 -- from 512 wave items on)
    const bool reduce12 = Hi == 12 && Wi == 12 && Ho <= 6 && Wo <= 6 && (size_t)B * (C / SS_CH) >= 512;
    if (Hi < 1 || Wi < 1 || Ho < 1 || Wo < 1 || ((Hi > SS_MAX_IN || Wi > SS_MAX_IN) && !reduce12) || Ho > SS_MAX_OUT ||
        Wo > SS_MAX_OUT)
        return false;
    if (Hi * Wi > SS_MAX_PIX && !reduce12) return false;
    return (size_t)B * (C / SS_CH) < ((size_t)1 << 31) && (size_t)B * Ho * Wo * C < ((size_t)1 << 40);
}

template <int MAXE>
int ss_launch(const float *x, float *y, const float *L1, const float *R1, const float *L2, const float *R2, const SsGeom &g,
              hipStream_t s) {
    const int items = g.B * (g.C / SS_CH);
    const size_t lds = (size_t)(g.Hi * g.Wi * SS_CH) * sizeof(float);
    // resident workgroups per CU by LDS (160 KB), at most 8; a persistent grid of that many walks the items
    size_t per_cu = (160 * 1024) / lds;
    if (per_cu > 8) per_cu = 8;
    if (per_cu < 1) return SEI_ERR_TOO_LARGE;
    const size_t grid = (size_t)items < 256 * per_cu ? (size_t)items : 256 * per_cu;
    hipLaunchKernelGGL(sepmap_small_kernel<MAXE>, dim3((unsigned)grid), dim3(SS_THREADS), lds, s, x, y, L1, R1, L2, R2, g, items);
    return sei_launch_status();
}

template <int HI, int WI>
int ss_launch_wave(const float *x, float *y, const float *L1, const float *R1, const float *L2, const float *R2,
                   const SsGeom &g, hipStream_t s) {
    const int items = g.B * (g.C / SS_CH);                      // wave items: four per workgroup
    // output columns of an item over 2, 3, 4 or 6 waves while that keeps the launch under ~8 waves per SIMD
    int parts = 1;
    for (int cand : {2, 3, 4, 6})
        if (g.Wo % cand == 0 && (size_t)items * cand <= 8192) parts = cand;
    const int wgs = (items * parts + 3) / 4;
    const int grid = wgs < 256 * 8 ? wgs : 256 * 8;            // 29 - 62 VGPRs: eight workgroups (32 waves) per CU
    hipLaunchKernelGGL((sepmap_small_wave_kernel<HI, WI>), dim3((unsigned)grid), dim3(SS_THREADS), 0, s, x, y, L1, R1, L2, R2, g,
                       items, parts);
    return sei_launch_status();
}

}  // namespace

// 1 when sei_sepmap2_small serves this shape (input extents <= 8, output extents <= 24, C % 64 == 0), else 0.
extern "C" size_t sei_sepmap2_small_eligible(int B, int Hi, int Wi, int Ho, int Wo, int C) {
    return ss_plan(B, Hi, Wi, Ho, Wo, C) ? 1 : 0;
}

// x: (B, Hi, Wi, C) -> y: (B, Ho, Wo, C), NHWC; x float32, y float32 or (out_bf16) bf16; L1, L2: (Ho, Hi), R1, R2: (Wo, Wi)
// float32 row-major (device).
extern "C" int sei_sepmap2_small(const float *x, void *y, int out_bf16, int B, int Hi, int Wi, int Ho, int Wo, int C,
                                 const float *L1, const float *R1, const float *L2, const float *R2, void *stream) {
    SEI_REQUIRE(x && y && L1 && R1 && L2 && R2 && (const void *)x != (const void *)y);
    SEI_REQUIRE((((uintptr_t)x | (uintptr_t)y) & 15) == 0 && (out_bf16 == 0 || out_bf16 == 1));
    if (!ss_plan(B, Hi, Wi, Ho, Wo, C)) return SEI_ERR_BAD_ARG;
    SsGeom g{B, Hi, Wi, Ho, Wo, C, out_bf16};
    hipStream_t s = (hipStream_t)stream;
    // the two input shapes of a 48-pixel crop's deep levels: the barrier-free kernel, extents as template arguments
    if (Hi == 3 && Wi == 3) return ss_launch_wave<3, 3>(x, reinterpret_cast<float *>(y), L1, R1, L2, R2, g, s);
    if (Hi == 6 && Wi == 6) return ss_launch_wave<6, 6>(x, reinterpret_cast<float *>(y), L1, R1, L2, R2, g, s);
    if (Hi == 12 && Wi == 12) return ss_launch_wave<12, 12>(x, reinterpret_cast<float *>(y), L1, R1, L2, R2, g, s);
    const int e = Hi > Wi ? Hi : Wi;
    if (e <= 4) return ss_launch<4>(x, reinterpret_cast<float *>(y), L1, R1, L2, R2, g, s);
    return ss_launch<8>(x, reinterpret_cast<float *>(y), L1, R1, L2, R2, g, s);
}
