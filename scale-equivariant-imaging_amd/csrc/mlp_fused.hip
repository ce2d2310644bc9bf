// Fused pointwise MLP of a ConvBlock for the SHALLOW levels of the U-Net (C = 32, 128) in throughput (bf16) mode:
//     out = res_scale * x + conv3(gelu(conv2(h2)))        (reference: src/models/convolutional.py:40-51)
// where conv2 (C -> 4C) and conv3 (4C -> C) are 1x1 convolutions = GEMMs over the (pixels, channels) view.
//
// Why: at these levels the two GEMMs hold 0.2 % of the step's FLOPs but were 1.8 ms of it, because the 4C-wide hidden
// activation crossed HBM five times per pass (f32 pre-GELU h3 and bf16 gelu(h3) written by conv2, h4 read by conv3,
// h3 and h4 read again by the backward pass): 1344 B per pixel forward against 320 B of true inputs and outputs.
// Here the hidden activation never leaves the registers:
//
//   forward    per wave and 32-pixel tile: H^T = W2 h2^T as 32x32 MFMA tiles with the hidden unit on the accumulator
//              ROWS and the pixel on the lane, bias + GELU in registers, and the tile is re-used directly as the A
//              operand of the second product (cdna_hip_programming.md section 3, "an accumulator tile as the next
//              MFMA's operand"): out += (H^T tile)^T W3^T sums over the tile's rows = hidden units.
//   backward   recomputes h3 = conv2(h2) the same way (bit-identical to the forward), G^T = W3^T go^T,
//              gh3 = G * gelu'(h3), gh2 = gh3 W2 through the same accumulator-as-operand step, and writes what
//              the two weight-gradient GEMMs (sei_gemm_bf16nt_dw2, merged across the step's model calls) need:
//              bf16 go, gelu(h3) and gh3. In the accumulator layout a lane holds FOUR consecutive hidden units of
//              its pixel per register quad, so those two tensors leave as 8-byte stores.
//
// The weights are walked in slices of 32 hidden units; a workgroup (4 waves = 128 or 256 pixels) stages each slice
// in LDS once (padded rows: conflict-free fragment reads) while the previous slice is being consumed (the global
// loads of slice t+1 are issued before the MFMAs of slice t and written to the other LDS buffer after them).
//
// The same values as the unfused path: h3 in f32, gelu / gelu' by sei_gelu_bf16out / sei_dgelu_bf16out, h4 and gh3
// rounded to bf16 once.
#include "sei_common.h"

namespace {

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x4 = __attribute__((ext_vector_type(4))) __bf16;
using f32x16 = __attribute__((ext_vector_type(16))) float;

__device__ __forceinline__ f32x16 mfma(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ int acc_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }
__device__ __forceinline__ bf16x8 acc_frag(const f32x16 &x, int s) {
    bf16x8 a;
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] = (__bf16)x[8 * s + e];
    return a;
}

constexpr int WAVES = 4, THREADS = 64 * WAVES;

// LDS images of one 32-hidden-unit slice.
//   RowSlice: 32 rows (hidden units) x C channels, row pitch C*2 + 16 bytes  -> A operand rows via ds_read_b128
//   ColSlice: C rows (channels) x 32 hidden units, row pitch 72 bytes         -> "permuted k" B operand via ds_read_b64
template <int C>
struct Slices {
    static constexpr int ROW_PITCH = C * 2 + 16, COL_PITCH = 72;
    static constexpr int ROW_BYTES = 32 * ROW_PITCH, COL_BYTES = C * COL_PITCH;
    static constexpr int ROW_PIECES = 32 * C / 8, COL_PIECES = C * 4;          // 16-byte pieces
    static constexpr int ROW_PER_THREAD = (ROW_PIECES + THREADS - 1) / THREADS;
    static constexpr int COL_PER_THREAD = (COL_PIECES + THREADS - 1) / THREADS;
};

// global -> registers (issue early) and registers -> LDS (late) of one RowSlice: rows 32 ht .. of a (4C, C) matrix
template <int C>
__device__ __forceinline__ void row_slice_load(const unsigned short *m, int ht, uint4 (&r)[Slices<C>::ROW_PER_THREAD]) {
#pragma unroll
    for (int i = 0; i < Slices<C>::ROW_PER_THREAD; ++i) {
        const int p = threadIdx.x + i * THREADS;
        if (p < Slices<C>::ROW_PIECES)
            r[i] = *reinterpret_cast<const uint4 *>(m + (size_t)(32 * ht + p / (C / 8)) * C + 8 * (p % (C / 8)));
    }
}
template <int C>
__device__ __forceinline__ void row_slice_store(char *lds, const uint4 (&r)[Slices<C>::ROW_PER_THREAD]) {
#pragma unroll
    for (int i = 0; i < Slices<C>::ROW_PER_THREAD; ++i) {
        const int p = threadIdx.x + i * THREADS;
        if (p < Slices<C>::ROW_PIECES)
            *reinterpret_cast<uint4 *>(lds + (p / (C / 8)) * Slices<C>::ROW_PITCH + 16 * (p % (C / 8))) = r[i];
    }
}
// ColSlice: columns 32 ht .. of a (C, 4C) matrix
template <int C>
__device__ __forceinline__ void col_slice_load(const unsigned short *m, int ht, uint4 (&r)[Slices<C>::COL_PER_THREAD]) {
#pragma unroll
    for (int i = 0; i < Slices<C>::COL_PER_THREAD; ++i) {
        const int p = threadIdx.x + i * THREADS;
        if (p < Slices<C>::COL_PIECES)
            r[i] = *reinterpret_cast<const uint4 *>(m + (size_t)(p >> 2) * (4 * C) + 32 * ht + 8 * (p & 3));
    }
}
template <int C>
__device__ __forceinline__ void col_slice_store(char *lds, const uint4 (&r)[Slices<C>::COL_PER_THREAD]) {
#pragma unroll
    for (int i = 0; i < Slices<C>::COL_PER_THREAD; ++i) {
        const int p = threadIdx.x + i * THREADS;
        if (p < Slices<C>::COL_PIECES) {
            uint2 *d = reinterpret_cast<uint2 *>(lds + (p >> 2) * Slices<C>::COL_PITCH + 16 * (p & 3));   // 8-byte aligned
            d[0] = make_uint2(r[i].x, r[i].y);
            d[1] = make_uint2(r[i].z, r[i].w);
        }
    }
}
template <int C>
__device__ __forceinline__ bf16x8 row_frag(const char *slice, int s, int lane) {                       // A operand
    return *reinterpret_cast<const bf16x8 *>(slice + (lane & 31) * Slices<C>::ROW_PITCH + (16 * s + 8 * (lane >> 5)) * 2);
}
template <int C>
__device__ __forceinline__ bf16x8 perm_frag(const char *slice, int ct, int s, int lane) {              // B operand
    const char *p = slice + (32 * ct + (lane & 31)) * Slices<C>::COL_PITCH + (16 * s + 4 * (lane >> 5)) * 2;
    const bf16x4 lo = *reinterpret_cast<const bf16x4 *>(p), hi = *reinterpret_cast<const bf16x4 *>(p + 16);
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// ---------------------------------------------------------------------------------------------------------------
template <int C, int RT>
__global__ __launch_bounds__(THREADS) void mlp_fwd_kernel(const unsigned short *__restrict__ h2,
                                                           const unsigned short *__restrict__ W2,
                                                           const float *__restrict__ b2,
                                                           const unsigned short *__restrict__ W3,
                                                           const float *__restrict__ b3, const float *__restrict__ x,
                                                           float res_scale, float *__restrict__ out, int M) {
    using S = Slices<C>;
    constexpr int KS = C / 16, CT = C / 32, HT = 4 * C / 32;
    __shared__ __attribute__((aligned(16))) char lds[2 * (S::ROW_BYTES + S::COL_BYTES)];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ngroup = (M + 32 * RT * WAVES - 1) / (32 * RT * WAVES);
    for (int group = blockIdx.x; group < ngroup; group += gridDim.x) {
        const int row0 = (group * WAVES + wave) * 32 * RT;
        bf16x8 f[RT][KS];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const int r = max(0, min(row0 + 32 * rt + (lane & 31), M - 1));
#pragma unroll
            for (int s = 0; s < KS; ++s)
                f[rt][s] = *reinterpret_cast<const bf16x8 *>(h2 + (size_t)r * C + 16 * s + 8 * (lane >> 5));
        }
        f32x16 acc[RT][CT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) acc[rt][ct] = f32x16{0};
        uint4 pr[S::ROW_PER_THREAD], pc[S::COL_PER_THREAD];
        row_slice_load<C>(W2, 0, pr);
        col_slice_load<C>(W3, 0, pc);
        __syncthreads();                                   // the previous group's readers are done with buffer 0
        row_slice_store<C>(lds, pr);
        col_slice_store<C>(lds + S::ROW_BYTES, pc);
        __syncthreads();
#pragma unroll 1
        for (int ht = 0; ht < HT; ++ht) {
            const char *cur = lds + (ht & 1) * (S::ROW_BYTES + S::COL_BYTES);
            char *nxt = lds + ((ht + 1) & 1) * (S::ROW_BYTES + S::COL_BYTES);
            if (ht + 1 < HT) {
                row_slice_load<C>(W2, ht + 1, pr);
                col_slice_load<C>(W3, ht + 1, pc);
            }
            f32x16 hT[RT];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) hT[rt] = f32x16{0};
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const bf16x8 w = row_frag<C>(cur, s, lane);
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) hT[rt] = mfma(w, f[rt][s], hT[rt]);
            }
            float bias[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) bias[r] = b2[32 * ht + acc_row(r, lane)];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int r = 0; r < 16; ++r) hT[rt][r] = sei_gelu_bf16out(hT[rt][r] + bias[r]);
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    const bf16x8 w = perm_frag<C>(cur + S::ROW_BYTES, ct, s, lane);
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) acc[rt][ct] = mfma(acc_frag(hT[rt], s), w, acc[rt][ct]);
                }
            if (ht + 1 < HT) {                             // the other buffer was last read in iteration ht - 1
                row_slice_store<C>(nxt, pr);
                col_slice_store<C>(nxt + S::ROW_BYTES, pc);
            }
            __syncthreads();
        }
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const int c = 32 * ct + (lane & 31);
                const float bc = b3[c];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = row0 + 32 * rt + acc_row(r, lane);
                    if (row < M) {
                        const size_t o = (size_t)row * C + c;
                        out[o] = fmaf(res_scale, x[o], acc[rt][ct][r] + bc);
                    }
                }
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------
template <int C, int RT>
__global__ __launch_bounds__(THREADS) void mlp_bwd_kernel(const float *__restrict__ go,
                                                           const unsigned short *__restrict__ h2,
                                                           const unsigned short *__restrict__ W2,
                                                           const float *__restrict__ b2,
                                                           const unsigned short *__restrict__ W3T,
                                                           const unsigned short *__restrict__ W2T,
                                                           float *__restrict__ gh2, unsigned short *__restrict__ go16,
                                                           unsigned short *__restrict__ h4,
                                                           unsigned short *__restrict__ gh3, int M) {
    using S = Slices<C>;
    constexpr int KS = C / 16, CT = C / 32, HT = 4 * C / 32;
    constexpr int STAGE = 2 * S::ROW_BYTES + S::COL_BYTES;          // W2 slice, W3T slice, W2T slice
    __shared__ __attribute__((aligned(16))) char lds[2 * STAGE];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ngroup = (M + 32 * RT * WAVES - 1) / (32 * RT * WAVES);
    for (int group = blockIdx.x; group < ngroup; group += gridDim.x) {
        const int row0 = (group * WAVES + wave) * 32 * RT;
        bf16x8 f[RT][KS], g[RT][KS];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const int rr = row0 + 32 * rt + (lane & 31);
            const bool live = rr < M;
            const int r = max(0, min(rr, M - 1));
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const size_t o = (size_t)r * C + 16 * s + 8 * (lane >> 5);
                f[rt][s] = *reinterpret_cast<const bf16x8 *>(h2 + o);
                const float4 a = *reinterpret_cast<const float4 *>(go + o), b = *reinterpret_cast<const float4 *>(go + o + 4);
                bf16x8 t;
                t[0] = (__bf16)a.x; t[1] = (__bf16)a.y; t[2] = (__bf16)a.z; t[3] = (__bf16)a.w;
                t[4] = (__bf16)b.x; t[5] = (__bf16)b.y; t[6] = (__bf16)b.z; t[7] = (__bf16)b.w;
                g[rt][s] = t;
                if (live) *reinterpret_cast<bf16x8 *>(go16 + o) = t;
            }
        }
        f32x16 acc[RT][CT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) acc[rt][ct] = f32x16{0};
        uint4 p2[S::ROW_PER_THREAD], p3[S::ROW_PER_THREAD], pt[S::COL_PER_THREAD];
        row_slice_load<C>(W2, 0, p2);
        row_slice_load<C>(W3T, 0, p3);
        col_slice_load<C>(W2T, 0, pt);
        __syncthreads();
        row_slice_store<C>(lds, p2);
        row_slice_store<C>(lds + S::ROW_BYTES, p3);
        col_slice_store<C>(lds + 2 * S::ROW_BYTES, pt);
        __syncthreads();
#pragma unroll 1
        for (int ht = 0; ht < HT; ++ht) {
            const char *cur = lds + (ht & 1) * STAGE;
            char *nxt = lds + ((ht + 1) & 1) * STAGE;
            if (ht + 1 < HT) {
                row_slice_load<C>(W2, ht + 1, p2);
                row_slice_load<C>(W3T, ht + 1, p3);
                col_slice_load<C>(W2T, ht + 1, pt);
            }
            float bias[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) bias[r] = b2[32 * ht + acc_row(r, lane)];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                f32x16 hT = {0}, gT = {0};
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    hT = mfma(row_frag<C>(cur, s, lane), f[rt][s], hT);
                    gT = mfma(row_frag<C>(cur + S::ROW_BYTES, s, lane), g[rt][s], gT);
                }
                // lane = pixel, registers = hidden units: quads of 4 consecutive units -> 8-byte stores
                const int pix = row0 + 32 * rt + (lane & 31);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    bf16x4 a4, g4;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int r = 4 * q + e;
                        const float h3 = hT[r] + bias[r];
                        float cdf, pdf;
                        sei_phi_pdf_bf16out(h3, cdf, pdf);
                        const float gv = gT[r] * fmaf(h3, pdf, cdf);
                        gT[r] = gv;
                        a4[e] = (__bf16)(h3 * cdf);
                        g4[e] = (__bf16)gv;
                    }
                    if (pix < M) {
                        const size_t o = (size_t)pix * (4 * C) + 32 * ht + 8 * q + 4 * (lane >> 5);
                        *reinterpret_cast<bf16x4 *>(h4 + o) = a4;
                        *reinterpret_cast<bf16x4 *>(gh3 + o) = g4;
                    }
                }
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct)
                        acc[rt][ct] = mfma(acc_frag(gT, s), perm_frag<C>(cur + 2 * S::ROW_BYTES, ct, s, lane), acc[rt][ct]);
            }
            if (ht + 1 < HT) {
                row_slice_store<C>(nxt, p2);
                row_slice_store<C>(nxt + S::ROW_BYTES, p3);
                col_slice_store<C>(nxt + 2 * S::ROW_BYTES, pt);
            }
            __syncthreads();
        }
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const int c = 32 * ct + (lane & 31);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = row0 + 32 * rt + acc_row(r, lane);
                    if (row < M) gh2[(size_t)row * C + c] = acc[rt][ct][r];
                }
            }
    }
}

inline unsigned mlp_grid(int M, int rows_per_group) {
    size_t g = sei_ceil_div((size_t)M, (size_t)rows_per_group);
    return (unsigned)(g < 1 ? 1 : (g > 2048 ? 2048 : g));
}

}  // namespace

// 1 where the fused form is the faster one: the 32-channel level always, the 128-channel level for the pixel counts
// csrc/mlp128.hip is cut for (multiples of 144 = a quarter of a 24 x 24 level image); the caller takes the GEMMs otherwise.
extern "C" size_t sei_mlp_fused_eligible(long long M, int C) {
    if (M <= 0 || M >= (1ll << 31)) return 0;
    if (C == 32) return 1;
    // C = 128: the nine-wave kernels on whole groups of 144 pixels (+ the first fused kernel on a tail of < 144: the
    // 256-pixel grids of the un-cropped series are powers of two)
    return (C == 128 && M >= 144 * 64) || sei_mlp128_eligible((int)M, C) ? 1 : 0;
}

namespace {
// pixels the nine-wave kernels take when M is not a multiple of their 144-pixel groups (0: none, the old kernel takes all)
inline int mlp128_main(int M, int C) {
    if ((C != 32 && C != 128) || M < 144 * 64) return 0;
    return M - M % 144;
}
}  // namespace

extern "C" int sei_mlp_fused_fwd(const uint16_t *h2, const uint16_t *W2, const float *b2, const uint16_t *W3,
                                 const float *b3, const float *x, float res_scale, float *out, int M, int C,
                                 void *stream) {
    SEI_REQUIRE(h2 && W2 && b2 && W3 && b3 && x && out && M > 0 && (C == 32 || C == 128));
    SEI_REQUIRE((((uintptr_t)h2 | (uintptr_t)W2 | (uintptr_t)W3) & 15) == 0);
    hipStream_t s = (hipStream_t)stream;
    if (sei_mlp128_eligible(M, C)) {
        SEI_REQUIRE((((uintptr_t)x | (uintptr_t)out) & 15) == 0);
        return sei_mlp128_fwd_launch(h2, W2, b2, W3, b3, x, res_scale, out, M, C, s);
    }
    if (const int main = mlp128_main(M, C)) {                        // whole 144-pixel groups there, the tail below
        SEI_REQUIRE((((uintptr_t)x | (uintptr_t)out) & 15) == 0);
        if (int rc = sei_mlp128_fwd_launch(h2, W2, b2, W3, b3, x, res_scale, out, main, C, s)) return rc;
        const size_t o = (size_t)main * C;
        h2 += o; x += o; out += o;
        M -= main;
    }
    if (C == 32)
        hipLaunchKernelGGL((mlp_fwd_kernel<32, 2>), dim3(mlp_grid(M, 64 * WAVES)), dim3(THREADS), 0, s, h2, W2, b2, W3, b3,
                           x, res_scale, out, M);
    else
        hipLaunchKernelGGL((mlp_fwd_kernel<128, 1>), dim3(mlp_grid(M, 32 * WAVES)), dim3(THREADS), 0, s, h2, W2, b2, W3, b3,
                           x, res_scale, out, M);
    return sei_launch_status();
}

extern "C" int sei_mlp_fused_bwd(const float *go, const uint16_t *h2, const uint16_t *W2, const float *b2,
                                 const uint16_t *W3T, const uint16_t *W2T, float *gh2, uint16_t *go16, uint16_t *h4,
                                 uint16_t *gh3, int M, int C, void *stream) {
    SEI_REQUIRE(go && h2 && W2 && b2 && W3T && W2T && gh2 && go16 && h4 && gh3 && M > 0 && (C == 32 || C == 128));
    SEI_REQUIRE((((uintptr_t)go | (uintptr_t)h2 | (uintptr_t)W2 | (uintptr_t)W3T | (uintptr_t)W2T | (uintptr_t)go16) & 15) == 0 &&
                (((uintptr_t)h4 | (uintptr_t)gh3) & 7) == 0);
    hipStream_t s = (hipStream_t)stream;
    if (sei_mlp128_eligible(M, C)) {
        SEI_REQUIRE((((uintptr_t)gh2 | (uintptr_t)go) & 15) == 0);
        return sei_mlp128_bwd_launch(go, h2, W2, b2, W3T, W2T, gh2, go16, h4, gh3, M, C, s);
    }
    if (const int main = mlp128_main(M, C)) {
        SEI_REQUIRE((((uintptr_t)gh2 | (uintptr_t)go) & 15) == 0);
        if (int rc = sei_mlp128_bwd_launch(go, h2, W2, b2, W3T, W2T, gh2, go16, h4, gh3, main, C, s)) return rc;
        const size_t o = (size_t)main * C;
        go += o; h2 += o; gh2 += o; go16 += o; h4 += 4 * o; gh3 += 4 * o;
        M -= main;
    }
    if (C == 32)
        hipLaunchKernelGGL((mlp_bwd_kernel<32, 2>), dim3(mlp_grid(M, 64 * WAVES)), dim3(THREADS), 0, s, go, h2, W2, b2, W3T,
                           W2T, gh2, go16, h4, gh3, M);
    else
        hipLaunchKernelGGL((mlp_bwd_kernel<128, 1>), dim3(mlp_grid(M, 32 * WAVES)), dim3(THREADS), 0, s, go, h2, W2, b2,
                           W3T, W2T, gh2, go16, h4, gh3, M);
    return sei_launch_status();
}
