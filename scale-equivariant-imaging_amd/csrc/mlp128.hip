// Fused pointwise MLP of a ConvBlock at the 128-channel level of the U-Net, throughput (bf16) mode:
//     out = res_scale * x + conv3(gelu(conv2(h2)))        (reference: src/models/convolutional.py:40-51)
// conv2 (128 -> 512) and conv3 (512 -> 128) are 1x1 convolutions = GEMMs over the (pixels, channels) view.
//
// The first fused kernel (mlp_fused.hip, built for C = 32) lost at this level: 36,864 pixels were 288 four-wave
// workgroups of 128 pixels -- 1.125 rounds on 256 CUs -- and each walked the 16 weight slices behind a register-staged
// load and a barrier per slice with nothing in flight (63-67 us forward against 51 for the two GEMMs, SQ_WAIT_ANY 0.68).
// This one is cut to the level's own numbers:
//   * every pixel count of the level is a multiple of 144 = 9 x 16 (24 x 24 pixels x the batch): a workgroup is NINE waves,
//     each owning 16 pixels (one 16x16x32 MFMA column block) for all 512 hidden units, 144 pixels per workgroup, one
//     workgroup per CU for the 2B pass of batch 32 -- no ragged second round;
//   * the weights (2 x 128 KB per pass, the same for every workgroup: L2 hits) stream through a four-stage LDS ring of
//     32-hidden-unit slices filled by LDS-DMA three slices ahead (global_load_lds, counted vmcnt, one barrier per slice);
//     the per-lane SOURCE addresses swizzle the images so that the A fragments (ds_read_b128 rows) and the "permuted k"
//     B fragments (two ds_read_b64 per fragment) are bank-conflict-free;
//   * H^T = W2 h2^T is computed with the hidden unit on the accumulator rows and the pixel on the lane, bias + GELU run on
//     the accumulators, and two such tiles are the A operand of the second product as they stand (cdna_hip_programming.md
//     section 3: an accumulator tile as the next MFMA's operand); the 512-wide hidden activation never exists in memory;
//   * results leave through a wave-private LDS patch as whole rows: 16-byte residual loads and stores.
// The backward kernel recomputes h3 the same way (the same float32 bits as the forward), forms G^T = W3^T go^T,
// gh3 = G gelu'(h3) and gh2 = gh3 W2 through the same step, and writes what the weight gradients need (bf16 go, gelu(h3),
// gh3; csrc/dw_stream.hip reads them).
//
// Same values as the unfused path: h3 in f32, gelu / gelu' by sei_phi_pdf_bf16out, h4 and gh3 rounded to bf16 once.
#include "sei_common.h"

namespace {

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x4 = __attribute__((ext_vector_type(4))) __bf16;
using f32x4 = __attribute__((ext_vector_type(4))) float;
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void glb_void;

constexpr int WAVES = 9, NT = 64 * WAVES, PX = 16 * WAVES;   // 144 pixels per workgroup
constexpr int NSTAGE = 4;
// Per width C (32 or 128): 4 C hidden units in slices of 32; a slice image is 32 x C or C x 32 bf16 = C / 16 pieces of 1 KiB.
// At C = 32 all four slices (16 KB) fit the ring at once: the same loop, whose late refills then rewrite consumed stages.

__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// ---- LDS-DMA of one 1-KiB piece (p = 0..7) of a slice image; the destination is lane-linear, the source per lane ----
// "rows" image: 32 rows (hidden units 32 ht ..) x 128 channels of a (512, 128) matrix; 16-byte chunk c of row r sits at
// chunk position 16 r + (c ^ (r & 15)): the 16 lanes one ds_read_b128 serves together then hit 16 different bank groups.
// (C = 32: 64-byte rows, chunk c of row r at 4 r + (c ^ ((-(r >> 2)) & 3)) -- the XOR pattern that spreads the four rows which
// share r % 4 over the four bank groups for every 16-lane group of a ds_read_b128.)
template <int C>
__device__ __forceinline__ void issue_rows(char *dst, const unsigned short *m, int ht, int p, int lane) {
    int r, c;
    if constexpr (C == 128) { r = 4 * p + (lane >> 4); c = (lane & 15) ^ (r & 15); }
    else { r = 16 * p + (lane >> 2); c = (lane & 3) ^ ((-(r >> 2)) & 3); }
    __builtin_amdgcn_global_load_lds((glb_void *)(m + (size_t)(32 * ht + r) * C + 8 * c), (lds_void *)(dst + p * 1024), 16, 0, 0);
}
// "cols" image: 128 rows (channels) x 32 hidden units (32 ht ..) of a (128, 512) matrix; chunk c (0..3) of row r sits at
// chunk position 4 r + (c ^ ((r >> 2) & 3)): the 32 lanes of one ds_read_b64 group hit 32 different bank pairs.
template <int C>
__device__ __forceinline__ void issue_cols(char *dst, const unsigned short *m, int ht, int p, int lane) {
    const int r = 16 * p + (lane >> 2), c = (lane & 3) ^ ((r >> 2) & 3);
    __builtin_amdgcn_global_load_lds((glb_void *)(m + (size_t)r * (4 * C) + 32 * ht + 8 * c), (lds_void *)(dst + p * 1024), 16, 0, 0);
}
// A operand: hidden unit 16 t + l16 (row), channels 32 ks + 8 lg .. + 7
template <int C>
__device__ __forceinline__ bf16x8 rows_frag(const char *img, int t, int ks, int l16, int lg) {
    const int r = 16 * t + l16, c = 4 * ks + lg;
    if constexpr (C == 128) return *reinterpret_cast<const bf16x8 *>(img + (16 * r + (c ^ (r & 15))) * 16);
    else return *reinterpret_cast<const bf16x8 *>(img + (4 * r + (c ^ ((-(r >> 2)) & 3))) * 16);
}
// B operand in the permuted k order of two stacked accumulator tiles: column (channel) 16 cb + l16; k slots 0-3 = hidden
// units 4 lg .. + 3, slots 4-7 = hidden units 16 + 4 lg .. + 3 of the slice
__device__ __forceinline__ bf16x8 cols_frag(const char *img, int cb, int l16, int lg) {
    const int r = 16 * cb + l16, f = 2 * ((r >> 2) & 3);
    const bf16x4 lo = *reinterpret_cast<const bf16x4 *>(img + (8 * r + (lg ^ f)) * 8);
    const bf16x4 hi = *reinterpret_cast<const bf16x4 *>(img + (8 * r + ((lg + 4) ^ f)) * 8);
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}
__device__ __forceinline__ bf16x8 pack_tiles(const f32x4 &t0, const f32x4 &t1) {
    bf16x8 a;
#pragma unroll
    for (int e = 0; e < 4; ++e) { a[e] = (__bf16)t0[e]; a[4 + e] = (__bf16)t1[e]; }
    return a;
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}

// ---------------------------------------------------------------------------------------------------------------
// Forward. Grid = M / 144 workgroups of nine waves.
template <int C>
__global__ __launch_bounds__(NT) void mlp128_fwd_kernel(const unsigned short *__restrict__ h2,
                                                        const unsigned short *__restrict__ W2,
                                                        const float *__restrict__ b2,
                                                        const unsigned short *__restrict__ W3,
                                                        const float *__restrict__ b3, const float *__restrict__ x,
                                                        float res_scale, float *__restrict__ out) {
    constexpr int H4 = 4 * C, NSLICE = H4 / 32, IMG = 64 * C, LDP = C + 4, KS2 = C / 32, CB = C / 16, NPC = C / 16;
    constexpr int STAGE = 2 * IMG;                                  // W2 rows slice + W3 cols slice
    constexpr int RING = NSTAGE * STAGE, PATCH = WAVES * 16 * LDP * 4;
    __shared__ __attribute__((aligned(1024))) char smem[(RING > PATCH ? RING : PATCH) + H4 * 4];
    float *lb2 = reinterpret_cast<float *>(smem + (RING > PATCH ? RING : PATCH));
    const int tid = threadIdx.x, lane = tid & 63, l16 = lane & 15, lg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const size_t px0 = (size_t)blockIdx.x * PX + 16 * wave;

    // this wave's pixels: the B operand of the first product (pixel l16, channels 32 ks + 8 lg ..), kept for the whole pass
    bf16x8 f[KS2];
#pragma unroll
    for (int ks = 0; ks < KS2; ++ks)
        f[ks] = *reinterpret_cast<const bf16x8 *>(h2 + (px0 + l16) * C + 32 * ks + 8 * lg);
    if (tid < H4) lb2[tid] = b2[tid];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // before the ring: its counted waits see DMA pieces only
    __builtin_amdgcn_sched_barrier(0);

    auto issue = [&](int ht) {                                      // slice ht -> stage ht % NSTAGE: two pieces per wave < NPC
        if (wave < NPC) {                                           // (past the end: the last slice again: uniform DMA count)
            const int hs = ht < NSLICE ? ht : NSLICE - 1;
            char *dst = smem + (ht % NSTAGE) * STAGE;
            issue_rows<C>(dst, W2, hs, wave, lane);
            issue_cols<C>(dst + IMG, W3, hs, wave, lane);
        }
    };
    f32x4 acc[CB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) acc[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
    issue(0);
    issue(1);
    issue(2);
#pragma unroll 1
    for (int ht = 0; ht < NSLICE; ++ht) {
        if (wave < NPC) wait_vmcnt<4>();                            // this wave's pieces of slice ht have landed
        lds_barrier();                                              // ... everyone's; slice ht - 1 is read out (b2 on the first)
        issue(ht + 3);                                              // (past the end: into the stage of slice ht - 1, unread)
        const char *st = smem + (ht % NSTAGE) * STAGE;
        // the slice's A fragments in one burst, then the first product; the B fragments of the second product are requested
        // BEFORE the GELU arithmetic, which covers their latency
        bf16x8 wa[2][KS2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int ks = 0; ks < KS2; ++ks) wa[t][ks] = rows_frag<C>(st, t, ks, l16, lg);
        __builtin_amdgcn_sched_barrier(0);
        f32x4 hT[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            hT[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS2; ++ks) hT[t] = mfma16(wa[t][ks], f[ks], hT[t]);
        }
        bf16x8 wb[CB];
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) wb[cb] = cols_frag(st + IMG, cb, l16, lg);
        f32x4 bias[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) bias[t] = *reinterpret_cast<const f32x4 *>(lb2 + 32 * ht + 16 * t + 4 * lg);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) hT[t][r] = sei_gelu_bf16out(hT[t][r] + bias[t][r]);
        const bf16x8 a = pack_tiles(hT[0], hT[1]);
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) acc[cb] = mfma16(a, wb[cb], acc[cb]);
    }
    wait_vmcnt<0>();                                                // the clamped slices still in flight
    __syncthreads();                                                // every wave is done with the ring: it becomes the patches

    // ---- epilogue through a wave-private patch: accumulators in, whole rows out
    constexpr int QR = C / 4, ITS = 16 * QR / 64;                   // float4 per row; sweeps of the wave over its 16 rows
    float *patch = reinterpret_cast<float *>(smem) + wave * 16 * LDP;
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int r = 0; r < 4; ++r) patch[(4 * lg + r) * LDP + 16 * cb + l16] = acc[cb][r];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");              // (wave-private: no barrier)
    f32x4 xr[ITS];
#pragma unroll
    for (int it = 0; it < ITS; ++it) {
        const int idx = it * 64 + lane, p = idx / QR, q = idx % QR;
        xr[it] = *reinterpret_cast<const f32x4 *>(x + (px0 + p) * C + 4 * q);
    }
#pragma unroll
    for (int it = 0; it < ITS; ++it) {
        const int idx = it * 64 + lane, p = idx / QR, q = idx % QR;
        const f32x4 v = *reinterpret_cast<const f32x4 *>(patch + p * LDP + 4 * q);
        const f32x4 bc = *reinterpret_cast<const f32x4 *>(b3 + 4 * q);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = fmaf(res_scale, xr[it][e], v[e] + bc[e]);
        *reinterpret_cast<f32x4 *>(out + (px0 + p) * C + 4 * q) = o;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Backward. W3T (512, 128) and W2T (128, 512): transposed bf16 copies of the two weights.
template <int C>
__global__ __launch_bounds__(NT) void mlp128_bwd_kernel(const float *__restrict__ go, const unsigned short *__restrict__ h2,
                                                        const unsigned short *__restrict__ W2,
                                                        const float *__restrict__ b2,
                                                        const unsigned short *__restrict__ W3T,
                                                        const unsigned short *__restrict__ W2T,
                                                        float *__restrict__ gh2, unsigned short *__restrict__ go16,
                                                        unsigned short *__restrict__ h4,
                                                        unsigned short *__restrict__ gh3) {
    constexpr int H4 = 4 * C, NSLICE = H4 / 32, IMG = 64 * C, LDP = C + 4, KS2 = C / 32, CB = C / 16, NPC = C / 16;
    constexpr int STAGE = 3 * IMG;                                  // W2 rows, W3T rows, W2T cols
    constexpr int RING = NSTAGE * STAGE, PATCH = WAVES * 16 * LDP * 4;
    __shared__ __attribute__((aligned(1024))) char smem[(RING > PATCH ? RING : PATCH) + H4 * 4];
    float *lb2 = reinterpret_cast<float *>(smem + (RING > PATCH ? RING : PATCH));
    const int tid = threadIdx.x, lane = tid & 63, l16 = lane & 15, lg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const size_t px0 = (size_t)blockIdx.x * PX + 16 * wave;

    bf16x8 f[KS2], g[KS2];
#pragma unroll
    for (int ks = 0; ks < KS2; ++ks) {
        const size_t o = (px0 + l16) * C + 32 * ks + 8 * lg;
        f[ks] = *reinterpret_cast<const bf16x8 *>(h2 + o);
        const f32x4 a = *reinterpret_cast<const f32x4 *>(go + o), b = *reinterpret_cast<const f32x4 *>(go + o + 4);
        g[ks] = pack_tiles(a, b);
        *reinterpret_cast<bf16x8 *>(go16 + o) = g[ks];
    }
    if (tid < H4) lb2[tid] = b2[tid];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);

    auto issue = [&](int ht) {                                      // three pieces per slice and wave < NPC
        if (wave < NPC) {
            const int hs = ht < NSLICE ? ht : NSLICE - 1;
            char *dst = smem + (ht % NSTAGE) * STAGE;
            issue_rows<C>(dst, W2, hs, wave, lane);
            issue_rows<C>(dst + IMG, W3T, hs, wave, lane);
            issue_cols<C>(dst + 2 * IMG, W2T, hs, wave, lane);
        }
    };
    f32x4 acc[CB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) acc[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
    issue(0);
    issue(1);
    issue(2);
    unsigned short *h4p = h4 + (px0 + l16) * H4 + 4 * lg;
    unsigned short *g3p = gh3 + (px0 + l16) * H4 + 4 * lg;
#pragma unroll 1
    for (int ht = 0; ht < NSLICE; ++ht) {
        // this wave's pieces of slice ht have landed. vmcnt retires in order and counts the stores too: behind slice ht's
        // three pieces sit, in issue order, [stores ht-3] slice ht+1 [stores ht-2] slice ht+2 [stores ht-1] = 4 + 3 + 4 + 3 + 4
        // operations in the steady state, fewer on the first three slices (nothing was stored before slice 0)
        if (wave < NPC) {
            if (ht >= 3) wait_vmcnt<18>();
            else if (ht == 2) wait_vmcnt<14>();
            else if (ht == 1) wait_vmcnt<10>();
            else wait_vmcnt<6>();
        }
        lds_barrier();
        issue(ht + 3);
        const char *st = smem + (ht % NSTAGE) * STAGE;
        f32x4 hT[2], gT[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {                               // (tile by tile: both tiles' 16 fragments at once spill)
            bf16x8 wa[KS2], wg[KS2];
#pragma unroll
            for (int ks = 0; ks < KS2; ++ks) {
                wa[ks] = rows_frag<C>(st, t, ks, l16, lg);
                wg[ks] = rows_frag<C>(st + IMG, t, ks, l16, lg);
            }
            __builtin_amdgcn_sched_barrier(0);
            hT[t] = f32x4{0.f, 0.f, 0.f, 0.f};
            gT[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS2; ++ks) {
                hT[t] = mfma16(wa[ks], f[ks], hT[t]);
                gT[t] = mfma16(wg[ks], g[ks], gT[t]);
            }
        }
        bf16x8 wb[CB];                                              // requested before the GELU' arithmetic, which covers them
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) wb[cb] = cols_frag(st + 2 * IMG, cb, l16, lg);
        f32x4 bias[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) bias[t] = *reinterpret_cast<const f32x4 *>(lb2 + 32 * ht + 16 * t + 4 * lg);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            bf16x4 a4, g4;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float h3 = hT[t][r] + bias[t][r];
                float cdf, pdf;
                sei_phi_pdf_bf16out(h3, cdf, pdf);
                const float gv = gT[t][r] * fmaf(h3, pdf, cdf);
                gT[t][r] = gv;
                a4[r] = (__bf16)(h3 * cdf);
                g4[r] = (__bf16)gv;
            }
            // lane = pixel, registers = four consecutive hidden units: 8-byte stores
            *reinterpret_cast<bf16x4 *>(h4p + 32 * ht + 16 * t) = a4;
            *reinterpret_cast<bf16x4 *>(g3p + 32 * ht + 16 * t) = g4;
        }
        const bf16x8 a = pack_tiles(gT[0], gT[1]);
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) acc[cb] = mfma16(a, wb[cb], acc[cb]);
    }
    wait_vmcnt<0>();
    __syncthreads();

    constexpr int QR = C / 4, ITS = 16 * QR / 64;
    float *patch = reinterpret_cast<float *>(smem) + wave * 16 * LDP;
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int r = 0; r < 4; ++r) patch[(4 * lg + r) * LDP + 16 * cb + l16] = acc[cb][r];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int it = 0; it < ITS; ++it) {
        const int idx = it * 64 + lane, p = idx / QR, q = idx % QR;
        *reinterpret_cast<f32x4 *>(gh2 + (px0 + p) * C + 4 * q) = *reinterpret_cast<const f32x4 *>(patch + p * LDP + 4 * q);
    }
}

}  // namespace

// Pixel counts these kernels take (internal linkage between translation units: mlp_fused.hip dispatches to them).
bool sei_mlp128_eligible(int M, int Cc) { return (Cc == 128 || Cc == 32) && M > 0 && M % 144 == 0; }

int sei_mlp128_fwd_launch(const uint16_t *h2, const uint16_t *W2, const float *b2, const uint16_t *W3, const float *b3,
                          const float *x, float res_scale, float *out, int M, int Cc, hipStream_t s) {
    if (Cc == 128)
        hipLaunchKernelGGL(mlp128_fwd_kernel<128>, dim3((unsigned)(M / 144)), dim3(NT), 0, s, h2, W2, b2, W3, b3, x, res_scale, out);
    else
        hipLaunchKernelGGL(mlp128_fwd_kernel<32>, dim3((unsigned)(M / 144)), dim3(NT), 0, s, h2, W2, b2, W3, b3, x, res_scale, out);
    return sei_launch_status();
}

int sei_mlp128_bwd_launch(const float *go, const uint16_t *h2, const uint16_t *W2, const float *b2, const uint16_t *W3T,
                          const uint16_t *W2T, float *gh2, uint16_t *go16, uint16_t *h4, uint16_t *gh3, int M, int Cc,
                          hipStream_t s) {
    if (Cc == 128)
        hipLaunchKernelGGL(mlp128_bwd_kernel<128>, dim3((unsigned)(M / 144)), dim3(NT), 0, s, go, h2, W2, b2, W3T, W2T, gh2,
                           go16, h4, gh3);
    else
        hipLaunchKernelGGL(mlp128_bwd_kernel<32>, dim3((unsigned)(M / 144)), dim3(NT), 0, s, go, h2, W2, b2, W3T, W2T, gh2,
                           go16, h4, gh3);
    return sei_launch_status();
}
