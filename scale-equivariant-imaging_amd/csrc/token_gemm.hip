// Token-streaming GEMMs for layer-sized weights (SwinIR's linear layers: 180 -> 540 / 180 / 360 -> 180 channels
// over 10^5 tokens; reference: deepinv's SwinIR built by /root/reference/src/models/__init__.py:51-74).
//
// The matrices of such a layer are a few hundred KB; the token operands are 50-170 MB. The tiled GEMMs of
// gemm_bf16nt.hip treat them like any other product -- every 128 x 128 tile re-stages its weight panel, the 192- and
// 576-wide outputs are cut into 1.5 and 4.5 tiles, and nothing overlaps a tile's epilogue -- and run at 0.3-0.45 of
// what the token bytes alone would cost. The kernels here are built the other way round: ONE workgroup per CU,
// persistent, streaming its share of the tokens exactly once through an LDS ring of LDS-DMA stages, with the whole
// weight (or the whole weight gradient) resident in registers.
//
//   sei_tokgrad_bf16   dW (Mo x Ni) += dY^T X over the tokens (both operands token-major as stored, two token
//                      segments = the step's two model calls). A workgroup owns a 192 x 192 block of dW in the
//                      accumulators of its four waves (96 x 96 each: 36 16x16x32 MFMAs per 12 transposing fragment
//                      reads, 83 B of LDS per clock against 188 for the 64 x 32 wave tiles of the 128 x 128 loop,
//                      which that loop's 520 TFLOP/s were bound by) and a contiguous range of tokens; blocks of the
//                      same token range sit on the same XCD, so X crosses the fabric once. Float atomics at the end.
//   sei_rowgemm_bf16   D = epilogue(A W^T), the layer's matrix in registers, row tiles streamed (described further down)
//     .._ln_bf16       + the LayerNorm BEHIND a residual layer in that layer's epilogue (proj -> norm2, fc2 -> next norm1)
//     .._lnbwd_bf16    + LayerNorm BACKWARD (+ residual gradient, bf16 copy, column sums) behind a data gradient
//     .._dgelu_bf16    fc2's data gradient with the GELU' input recomputed by a second product in registers
//     .._gelu_bf16     fc1's forward, bf16 output only, with a ones column that carries fc2's bias gradient
#include "sei_common.h"

namespace {

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f32x4 = __attribute__((ext_vector_type(4))) float;
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void glb_void;
typedef short v4s __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// 16-B chunk swizzle of a [64 tokens][64 columns] image (128-B rows, two per 256-B bank row): as gemm_bf16pq.h
__device__ __forceinline__ int tg_swz(int row) { return (((row >> 1) & 1) << 1) | (((row >> 3) & 1) << 2); }

constexpr int TG_NT = 256;                 // four waves, one per SIMD: 144 accumulator registers each (NW = 4)
constexpr int TG_IMG = 64 * 128;           // one [64 tokens][64 columns] bf16 image
constexpr int TG_STAGE = 6 * TG_IMG;       // 192 columns of dY + 192 columns of X
constexpr int TG_NSTAGE = 3;
constexpr int TG_MAX_BLOCKS = SEI_TOKGRAD_MAX_BLOCKS;

struct TokGradArgs {
    SeiTokGradBlock blk[TG_MAX_BLOCKS];
    int nblk;
    int kt_seg, kt_total;     // 64-token k-tiles in the first segment / in both
    int workers_per_xcd;      // token ranges per XCD (each served by `nblk` workgroups of that XCD)
};

// NW = 4: four waves of 96 x 96; NW = 8: eight waves of 96 x 48, two per SIMD -- one wave's fragment reads under the
// other's MFMAs (each stage is read 1.5 x as often from LDS, which has the room: the loop was bound by the read -> wait ->
// MFMA sequence of its single wave per SIMD, not by LDS bandwidth)
template <int NW>
__global__ __launch_bounds__(64 * NW) void tokgrad_kernel(TokGradArgs g) {
    constexpr int WRN = NW == 16 ? 4 : 2, WCN = NW / WRN;            // row groups x column groups of waves
    constexpr int RBk = 12 / WRN, CB = 12 / WCN, NPIECE = 48 / NW;   // 16-row / 16-column blocks per wave, DMA pieces per wave
    __shared__ __attribute__((aligned(1024))) char smem[TG_NSTAGE * TG_STAGE];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave / WCN, wc = wave % WCN;
    const int l16 = lane & 15, lg = lane >> 4;

    // workgroups are dealt round-robin to the XCDs: slot s of XCD x = token range (s / nblk) * 8 + x, block s % nblk --
    // the blocks of a token range (the three 192-row blocks of a qkv gradient share X) go through the same L2
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int wslot = slot / g.nblk, bsel = slot - wslot * g.nblk;
    if (wslot >= g.workers_per_xcd) return;
    const int worker = wslot * 8 + xcd, workers = 8 * g.workers_per_xcd;
    const int kt0 = (int)((long long)g.kt_total * worker / workers);
    const int kt1 = (int)((long long)g.kt_total * (worker + 1) / workers);
    const int nt = kt1 - kt0;
    if (nt <= 0) return;                                           // block-uniform, before any barrier
    const SeiTokGradBlock &b = g.blk[bsel];                         // uniform index: scalar loads from the argument block
    const unsigned short *Y1 = b.Y1, *Y2 = b.Y2, *X1 = b.X1, *X2 = b.X2;
    const int ldy = b.ldy, ldx = b.ldx;

    // ---- DMA: 48 1-KiB pieces per stage, 48 / NW per wave; piece q = image q / 8 (0-2: dY, 3-5: X), rows 8 (q % 8) ..
    unsigned off[NPIECE];
#pragma unroll
    for (int e = 0; e < NPIECE; ++e) {
        const int q = wave + NW * e, im = q >> 3, p = q & 7;
        const int krow = 8 * p + (lane >> 3);
        const int ch = (lane & 7) ^ tg_swz(krow);
        off[e] = im < 3 ? ((unsigned)krow * (unsigned)ldy + (unsigned)(b.y0 + im * 64 + 8 * ch)) * 2u
                        : ((unsigned)krow * (unsigned)ldx + (unsigned)(b.x0 + (im - 3) * 64 + 8 * ch)) * 2u;
    }
    auto issue = [&](int u) {                                       // k-tile u of this range (clamped: see the loop)
        const int kt = kt0 + min(u, nt - 1);
        const char *yb, *xb;
        if (kt < g.kt_seg) {
            yb = reinterpret_cast<const char *>(Y1 + (size_t)kt * 64 * ldy);
            xb = reinterpret_cast<const char *>(X1 + (size_t)kt * 64 * ldx);
        } else {
            yb = reinterpret_cast<const char *>(Y2 + (size_t)(kt - g.kt_seg) * 64 * ldy);
            xb = reinterpret_cast<const char *>(X2 + (size_t)(kt - g.kt_seg) * 64 * ldx);
        }
        char *dst = smem + (u % TG_NSTAGE) * TG_STAGE;
#pragma unroll
        for (int e = 0; e < NPIECE; ++e) {
            const int q = wave + NW * e;                            // wave-uniform
            dma16_base((q >> 3) < 3 ? yb : xb, off[e], dst + q * 1024);      // (inline asm: sei_common.h)
        }
    };

    // ---- fragments: 8 tokens (32 ks + 8 lg + j) of column l16 of a 16-column block, two transposing reads
    const int tq = l16 >> 2, tp = l16 & 3;
    const int rm_lane = 128 * (8 * lg + tq) + 16 * ((tp >> 1) ^ tg_swz(8 * lg + tq)) + 8 * (tp & 1);
    auto frag = [&](const char *img, int blk, int ks) -> bf16x8 {
        const int base = rm_lane ^ (32 * blk);
        const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) v4s *)(img + base + 128 * 32 * ks));
        const v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) v4s *)(img + base + 128 * (32 * ks + 4)));
        return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    };

    f32x4 acc[RBk][CB];
#pragma unroll
    for (int i = 0; i < RBk; ++i)
#pragma unroll
        for (int j = 0; j < CB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // Every iteration issues one stage (past the end: the last k-tile again, into a stage nobody reads), so the
    // counted wait below always leaves exactly the pieces of the next stage (12 or 6 per wave) in flight.
    issue(0);
    issue(1);
    for (int u = 0; u < nt; ++u) {
        if constexpr (NW == 4) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");   // this wave's pieces of stage u have landed
        else if constexpr (NW == 8) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        __builtin_amdgcn_s_barrier();                               // ... everyone's; and stage u - 1 is read out
        issue(u + 2);
        const char *st = smem + (u % TG_NSTAGE) * TG_STAGE;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 fa[RBk], fb[CB];
#pragma unroll
            for (int i = 0; i < RBk; ++i) {
                const int bi = RBk * wr + i;
                fa[i] = frag(st + (bi >> 2) * TG_IMG, bi & 3, ks);
            }
#pragma unroll
            for (int j = 0; j < CB; ++j) {
                const int bj = CB * wc + j;
                fb[j] = frag(st + (3 + (bj >> 2)) * TG_IMG, bj & 3, ks);
            }
#pragma unroll
            for (int i = 0; i < RBk; ++i)
#pragma unroll
                for (int j = 0; j < CB; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // the two clamped stages still in flight

    // ---- float atomics: accumulator element (row 4 lg + r, column l16) of block (i, j). One atomic per clock and L2
    // channel is what they cost (measured: 27 us for the 256 x 147 KB of a launch, whatever the scope or the number of
    // adders per address), which is why a launch should carry as many 192 x 192 blocks as it can: the token range of a
    // workgroup grows and the number of partial blocks per output shrinks with the block count.
    float *d = b.D + (size_t)(16 * RBk * wr + 4 * lg) * b.ldd + 16 * CB * wc + l16;
#pragma unroll
    for (int i = 0; i < RBk; ++i)
#pragma unroll
        for (int j = 0; j < CB; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) atomicAdd(d + (size_t)(16 * i + r) * b.ldd + 16 * j, acc[i][j][r]);
}

// The same product with the token stages travelling through REGISTERS: eight waves of 96 x 48, every wave loads its six
// 1-KiB pieces of a stage with plain 16-byte loads THREE stages ahead (3 x 24 registers per lane), writes them into one of
// two LDS stages one iteration before they are read, and computes as above. What the LDS-DMA ring is short of is bytes
// in flight: three 48-KB stages fill the LDS, one of them is being read, so 96 KB per CU are on their way -- 4.2 TB/s at the
// ~6 us a loaded request takes, where the same ring with nothing to compute (every slot reissued at once) streams 6.0.
// Registers have the room the LDS lacks: 144 KB per CU in flight.
__global__ __launch_bounds__(512, 2) void tokgrad_regs_kernel(TokGradArgs g) {
    constexpr int NW = 8, WCN = 4, RBk = 6, CB = 3, NPIECE = 6;
    __shared__ __attribute__((aligned(1024))) char smem[2 * TG_STAGE];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave / WCN, wc = wave % WCN;
    const int l16 = lane & 15, lg = lane >> 4;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int wslot = slot / g.nblk, bsel = slot - wslot * g.nblk;
    if (wslot >= g.workers_per_xcd) return;
    const int worker = wslot * 8 + xcd, workers = 8 * g.workers_per_xcd;
    const int kt0 = (int)((long long)g.kt_total * worker / workers);
    const int kt1 = (int)((long long)g.kt_total * (worker + 1) / workers);
    const int nt = kt1 - kt0;
    if (nt <= 0) return;
    const SeiTokGradBlock &b = g.blk[bsel];
    const unsigned short *Y1 = b.Y1, *Y2 = b.Y2, *X1 = b.X1, *X2 = b.X2;
    const int ldy = b.ldy, ldx = b.ldx;
    unsigned off[NPIECE];
#pragma unroll
    for (int e = 0; e < NPIECE; ++e) {
        const int q = wave + NW * e, im = q >> 3, p = q & 7;
        const int krow = 8 * p + (lane >> 3);
        const int ch = (lane & 7) ^ tg_swz(krow);
        off[e] = im < 3 ? ((unsigned)krow * (unsigned)ldy + (unsigned)(b.y0 + im * 64 + 8 * ch)) * 2u
                        : ((unsigned)krow * (unsigned)ldx + (unsigned)(b.x0 + (im - 3) * 64 + 8 * ch)) * 2u;
    }
    // loads the compiler does not track (it would wait for all of them at the first use of any): counted waits below
    auto load = [&](int u, u32x4 (&r)[NPIECE]) {
        const int kt = kt0 + min(u, nt - 1);
        const char *yb, *xb;
        if (kt < g.kt_seg) {
            yb = reinterpret_cast<const char *>(Y1 + (size_t)kt * 64 * ldy);
            xb = reinterpret_cast<const char *>(X1 + (size_t)kt * 64 * ldx);
        } else {
            yb = reinterpret_cast<const char *>(Y2 + (size_t)(kt - g.kt_seg) * 64 * ldy);
            xb = reinterpret_cast<const char *>(X2 + (size_t)(kt - g.kt_seg) * 64 * ldx);
        }
#pragma unroll
        for (int e = 0; e < NPIECE; ++e) {
            const int q = wave + NW * e;
            const char *src = ((q >> 3) < 3 ? yb : xb) + off[e];
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r[e]) : "v"(src) : "memory");
        }
    };
    auto commit = [&](int u, u32x4 (&r)[NPIECE]) {
        char *dst = smem + (u & 1) * TG_STAGE + lane * 16;
#pragma unroll
        for (int e = 0; e < NPIECE; ++e) {
            asm volatile("" : "+v"(r[e]));
            *reinterpret_cast<u32x4 *>(dst + (wave + NW * e) * 1024) = r[e];
        }
    };
    const int tq = l16 >> 2, tp = l16 & 3;
    const int rm_lane = 128 * (8 * lg + tq) + 16 * ((tp >> 1) ^ tg_swz(8 * lg + tq)) + 8 * (tp & 1);
    auto frag = [&](const char *img, int blk, int ks) -> bf16x8 {
        const int base = rm_lane ^ (32 * blk);
        const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) v4s *)(img + base + 128 * 32 * ks));
        const v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) v4s *)(img + base + 128 * (32 * ks + 4)));
        return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    };
    f32x4 acc[RBk][CB];
#pragma unroll
    for (int i = 0; i < RBk; ++i)
#pragma unroll
        for (int j = 0; j < CB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto compute = [&](int u) {
        const char *st = smem + (u & 1) * TG_STAGE;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 fa[RBk], fb[CB];
#pragma unroll
            for (int i = 0; i < RBk; ++i) {
                const int bi = RBk * wr + i;
                fa[i] = frag(st + (bi >> 2) * TG_IMG, bi & 3, ks);
            }
#pragma unroll
            for (int j = 0; j < CB; ++j) {
                const int bj = CB * wc + j;
                fb[j] = frag(st + (3 + (bj >> 2)) * TG_IMG, bj & 3, ks);
            }
#pragma unroll
            for (int i = 0; i < RBk; ++i)
#pragma unroll
                for (int j = 0; j < CB; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
    };
    // Stage u lives in register set u % 3 from three iterations before its turn until one before, then in LDS stage
    // u % 2. Iteration u: [stage u + 1 has landed: counted wait] [barrier: everyone is done reading LDS stage (u + 1) % 2
    // (iteration u - 1) and stage u's writes (iteration u - 1) are visible] stage u + 1 into LDS, loads of stage u + 4
    // into the freed set, MFMAs on stage u. No load is issued whose data is not used: an untracked load into a register
    // the compiler believes free would land on whatever it has put there since. (nt >= 8: the host sends shorter ranges
    // to the LDS-DMA kernel.)
    u32x4 r0[NPIECE], r1[NPIECE], r2[NPIECE];
    load(0, r0);
    load(1, r1);
    load(2, r2);
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    commit(0, r0);
    load(3, r0);
#define TG_STEP(U, RN)                                                                                          \
    {                                                                                                           \
        asm volatile("s_waitcnt vmcnt(12)" ::: "memory");                                                       \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                      \
        __builtin_amdgcn_s_barrier();                                                                           \
        commit((U) + 1, RN);                                                                                    \
        load((U) + 4, RN);                                                                                      \
        compute(U);                                                                                             \
    }
#define TG_TAIL(U, RN)                                                                                          \
    if ((U) < nt) {                                                                                             \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                        \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                      \
        __builtin_amdgcn_s_barrier();                                                                           \
        if ((U) + 1 < nt) commit((U) + 1, RN);                                                                  \
        if ((U) + 4 < nt) load((U) + 4, RN);                                                                    \
        compute(U);                                                                                             \
    }
    int u = 0;
    for (; u + 6 < nt; u += 3) {
        TG_STEP(u, r1)
        TG_STEP(u + 1, r2)
        TG_STEP(u + 2, r0)
    }
    TG_TAIL(u, r1)
    TG_TAIL(u + 1, r2)
    TG_TAIL(u + 2, r0)
    TG_TAIL(u + 3, r1)
    TG_TAIL(u + 4, r2)
    TG_TAIL(u + 5, r0)
#undef TG_STEP
#undef TG_TAIL
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float *d = b.D + (size_t)(16 * RBk * wr + 4 * lg) * b.ldd + 16 * CB * wc + l16;
#pragma unroll
    for (int i = 0; i < RBk; ++i)
#pragma unroll
        for (int j = 0; j < CB; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) atomicAdd(d + (size_t)(16 * i + r) * b.ldd + 16 * j, acc[i][j][r]);
}

// ---------------------------------------------------------------------------------------------------------------
// sei_rowgemm_bf16: D (M x N) = epilogue(A (M x K) W^T), W (N x K) the layer's matrix, K = 192 / 384 / 576 and
// N = 192 / 384 / 576 (zero-padded): nn.Linear forward and data gradient of the Swin blocks.
//
// One workgroup per CU, eight waves = column groups x row groups (x halves of K for K = 576). A wave keeps its slice of W
// (its columns x K) in registers for the whole launch (72-144 VGPRs) and streams row tiles (TR = 32 or 64 rows, every tile this workgroup
// owns: b, b + G, ...) through a three-stage LDS ring filled by LDS-DMA two tiles ahead -- A is read from HBM exactly once,
// W once per workgroup. Per tile: [the tile's residual / GELU' rows into registers, DMA of tile t + 2] MFMAs [one counted
// vmcnt: tile t + 1 has landed] accumulators into an LDS patch, and all 512 threads walk the patch as row
// quads: 16-byte auxiliary values (loaded before the MFMAs) and 16- / 8-byte stores on whole rows. Two barriers
// per tile. Against the tiled kernels (K = 192: 3 k-tiles of loop, then an epilogue nothing overlaps; 1.125-2.25 rounds
// of 128-row tiles on the chip) the loads of the next tiles and the stores of the last one are always in flight.
// ---------------------------------------------------------------------------------------------------------------
constexpr int RG_NT = 512;

struct RowGemmArgs {
    const unsigned short *A, *W;
    int lda, ldw;
    int tiles;                // M / TR
    const float *bias;        // nv entries or nullptr
    const float *R1, *R2;     // BIAS_RES: R1 (+ R2) residual rows; BIAS_SCALE_RES: R1 = one factor per row, R2 = residual rows;
    int ldr;                  // MUL_DGELU: R1 = the GELU' input rows.  Row stride of the row-shaped one(s)
    float *D32;
    unsigned short *D16;      // the result in bf16 (BIAS, NONE, MUL_DGELU) / gelu(result) in bf16 (BIAS_GELU)
    int ld32, ld16;
    int nv;                   // valid output columns of D32 / R (a multiple of 4); D16 always gets all 64 NB columns
    // RG_EPI_LNBWD (the product is the gradient with respect to a LayerNorm's output): the norm's input rows (R1, ldr),
    // its statistics and weight, the residual gradient rows (R2, ldr); D32 = the gradient with respect to the input;
    // optionally D16 = bf16(D32 * scale[row]) in 192-column rows; part: [workgroup][3][nv] column sums
    const float *mean, *rstd, *gamma, *scale;
    float *part;
    // RG_EPI_DGELU2 (D16 = bf16((A W^T) gelu'(A2 W2^T + bias))): the second product's operands -- the GELU' input is
    // RECOMPUTED from the layer's input rows instead of being stored by the forward pass and read back
    const unsigned short *A2, *W2;
    int lda2, ldw2;
    // RG_EPI_*_LN: the LayerNorm that follows the residual sum (weight gamma, bias beta2, eps): D16 = its output in bf16
    // rows (ones_col: 1.0 in padding column nv), mean / rstd written through mean_out / rstd_out
    const float *beta2;
    float *mean_out, *rstd_out;
    float eps;
    int ones_col;
    int gelu_one_at;          // SEI_EPI_BIAS_GELU: this column of D16 is written as 1.0 (-1: none)
};

constexpr int RG_EPI_LNBWD = 100;     // internal epilogue codes (sei_rowgemm_lnbwd_bf16, sei_rowgemm_dgelu_bf16,
constexpr int RG_EPI_DGELU2 = 101;    // sei_rowgemm_ln_bf16: SEI_EPI_BIAS_RES / _BIAS_SCALE_RES followed by a LayerNorm)
constexpr int RG_EPI_RES_LN = 102;
constexpr int RG_EPI_SCALE_RES_LN = 103;
constexpr int RG_EPI_GELU16 = 104;    // SEI_EPI_BIAS_GELU without the float32 pre-activation: bias + GELU in the accumulator
                                      // layout, bf16 patch, 16-byte stores; small enough for TWO workgroups per CU

template <int N>
__device__ __forceinline__ void rg_wait_vmcnt() {
    static_assert(N >= 0 && N < 64, "vmcnt is six bits");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// LDS writes of this wave done, then the barrier (a bare s_barrier does not wait for them; __syncthreads would also
// wait for every global load and LDS-DMA piece in flight)
__device__ __forceinline__ void rg_lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}

__device__ __forceinline__ unsigned rg_pack2(float a, float b) {
    const __bf16 x = (__bf16)a, y = (__bf16)b;
    return (unsigned)__builtin_bit_cast(unsigned short, x) | ((unsigned)__builtin_bit_cast(unsigned short, y) << 16);
}

// EPI: the SEI_EPI_* code; OUT16: SEI_EPI_NONE / _BIAS write bf16 (else float32)
// Wave layout: 8 = WN (column groups) x WR (row groups) x WK (halves of K). WK = 2 only where a wave's slice of W would
// not fit otherwise (K = 576): the two partial sums then meet in the patch. NBT 16-column blocks are dealt to the WN
// groups as evenly as they go (36 blocks on 8 groups: five each for the first four, four for the rest).
template <int KT, int NBT, int TR, int WN, int WR, int EPI, bool OUT16>
__global__ __launch_bounds__(RG_NT, (EPI == RG_EPI_GELU16 ? 4 : 1)) void rowgemm_kernel(RowGemmArgs g) {
    constexpr int WK = 8 / (WN * WR);
    static_assert(WN * WR * WK == 8 && (WK == 1 || WK == 2), "eight waves");
    constexpr int NB = (NBT + WN - 1) / WN;           // blocks of the widest column group
    constexpr int NBX = NBT % WN;                     // groups below NBX hold NB blocks, the others NB - 1 (0: all NB)
    constexpr int RB = TR / WR / 16;                  // 16-row blocks per wave
    constexpr int KS = 2 * KT / WK;                   // 32-wide k-steps per wave
    static_assert(TR % (16 * WR) == 0 && (2 * KT) % WK == 0, "whole blocks per wave");
    constexpr int NP = 16 * NBT;                      // padded output width
    constexpr int LDP = NP + 4;                       // patch row stride (floats): 4 LDP = 16 mod 64 banks
    constexpr bool TWO = EPI == RG_EPI_DGELU2;        // two products over the same rows: A W^T and A2 W2^T
    constexpr int OPS = TWO ? 2 : 1;
    constexpr int STAGE = OPS * KT * TR * 128;        // KT images of [TR rows][64 k] bf16 per operand
    constexpr int P = KT * TR / 8;                    // 1-KiB DMA pieces per stage
    constexpr int EMAX = (P + 7) / 8;
    constexpr int QR = NP / 4;                        // quads per row
    // what a thread finishes at a time: a quad (float32 outputs: 16-byte stores) or two neighbouring quads (bf16-only
    // outputs: 16-byte stores of eight values)
    constexpr bool LNB = EPI == RG_EPI_LNBWD;         // OUT16 then says: also write the scaled bf16 copy
    constexpr bool LNF = EPI == RG_EPI_RES_LN || EPI == RG_EPI_SCALE_RES_LN;
    constexpr bool LNX = LNB || LNF;                  // epilogues that reduce over the channels of a row
    static_assert(!LNX || (NBT == 12 && TR == 32), "the LayerNorm epilogues own 32 rows x 192 columns: 16 lanes per row");
    constexpr int GQ = (!LNX && (OUT16 || EPI == SEI_EPI_MUL_DGELU || TWO)) ? 2 : 1;
    // bf16-only outputs without auxiliary rows: bias and rounding happen in the accumulator layout and the patch holds bf16
    // (half the LDS bytes; the last pass is one 16-byte LDS read and one 16-byte store per eight values, no arithmetic)
    constexpr bool GELU16 = EPI == RG_EPI_GELU16;
    static_assert(!GELU16 || (OUT16 && WK == 1 && NBX == 0), "bf16 output only; whole column blocks per wave, all of K");
    constexpr bool P16 = ((EPI == SEI_EPI_BIAS || EPI == SEI_EPI_NONE) && OUT16) || TWO || GELU16;
    constexpr int LDP16 = NP + 8;                     // bf16 patch row stride: 4 rows = 64 B apart mod 256
    static_assert(!P16 || WK == 1, "the bf16 patch has nothing to add up");
    constexpr int IR = QR / GQ;                       // items per row
    constexpr int IPT = (TR * IR + RG_NT - 1) / RG_NT;   // items per thread (the last pass may be partly empty)
    constexpr bool RAGGED = TR * IR % RG_NT != 0;
    constexpr int QPT = IPT * GQ;                     // quads per thread
    static_assert(TR % 16 == 0 && QR % GQ == 0, "whole items");
    constexpr bool HAS_ROWS = EPI == SEI_EPI_BIAS_RES || EPI == SEI_EPI_BIAS_SCALE_RES || EPI == SEI_EPI_MUL_DGELU;
    constexpr bool HAS_BIAS = EPI == SEI_EPI_BIAS || EPI == SEI_EPI_BIAS_GELU || EPI == SEI_EPI_BIAS_RES ||
                              EPI == SEI_EPI_BIAS_SCALE_RES || LNF;   // (added to rows in the last pass; TWO: see below)
    constexpr int UNR = IPT > 3 ? (GQ == 2 ? 2 : 3) : IPT;   // items in flight per thread in the last pass
    constexpr int NAUX = LNF ? 3 + (EPI == RG_EPI_SCALE_RES_LN ? 1 : 0) : LNB ? 8 + (OUT16 ? 1 : 0)
                             : (HAS_ROWS ? QPT : 0) + (EPI == SEI_EPI_BIAS_SCALE_RES ? QPT : 0);   // loads per thread and tile
    // (the two-workgroups-per-CU variant sizes its patch for what it holds: bf16 rows, no bias row behind them)
    // TWO PATCHES where the LDS has the room (everything but GELU16, which spends it on a second workgroup, and K = 576):
    // tile t goes through patch t % 2, and the barrier that ended every tile ("the patch is free again") is gone -- a
    // patch is rewritten two tiles later, behind the mid-tile barrier of the tile in between, which every wave reaches
    // only after its last read of it; the ring stage of tile t is re-filled by the issue of tile t + 3, behind the
    // mid-tile barrier of tile t as before. One barrier per tile (two with the K halves) instead of two (three): waves
    // that are through with their rows start the next tile's MFMAs while the others still store.
    constexpr int PATCH_ONE = (P16 ? TR * LDP16 * 2 : TR * LDP * 4 + 15) / 16 * 16;
    constexpr int AUX_BYTES = GELU16 ? 0 : (LNF ? 3 : 1) * NP * 4;
    constexpr bool PATCH2 = !GELU16 && 3 * STAGE + 2 * PATCH_ONE + AUX_BYTES <= 160 * 1024;
    __shared__ __attribute__((aligned(1024))) char smem[3 * STAGE + (PATCH2 ? 2 : 1) * PATCH_ONE + AUX_BYTES];
    float *const patch0 = reinterpret_cast<float *>(smem + 3 * STAGE);
    float *lbias = reinterpret_cast<float *>(smem + 3 * STAGE + (PATCH2 ? 2 : 1) * PATCH_ONE);   // (not touched by the variant without it)
    float *lgam = lbias + NP, *lbet = lgam + NP;       // (LNF only)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave % WN, wr = (wave / WN) % WR, wk = wave / (WN * WR);
    const int nbw = (NBX == 0 || wn < NBX) ? NB : NB - 1;                          // this wave's blocks ...
    const int nb0 = (NBX == 0 || wn < NBX) ? NB * wn : NB * NBX + (NB - 1) * (wn - NBX);   // ... from block nb0 on
    const int l16 = lane & 15, lg = lane >> 4;
    const int G = gridDim.x, b = blockIdx.x;
    if (b >= g.tiles) return;
    const int nt = (g.tiles - b + G - 1) / G;          // tiles b, b + G, ...

    // ---- W slice of this wave: columns 16 NB wn .., k-steps wk KT .. (32 k each), MFMA B-operand layout
    bf16x8 wf[NB][KS];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int s = 0; s < KS; ++s)
            wf[nb][s] = *reinterpret_cast<const bf16x8 *>(g.W + (size_t)(16 * (nb0 + min(nb, nbw - 1)) + l16) * g.ldw +
                                                          32 * (wk * KS + s) + 8 * lg);
    bf16x8 wf2[TWO ? NB : 1][TWO ? KS : 1];
    float bias2[TWO ? NB : 1];
    if constexpr (TWO) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
            for (int s = 0; s < KS; ++s)
                wf2[nb][s] = *reinterpret_cast<const bf16x8 *>(g.W2 + (size_t)(16 * (nb0 + min(nb, nbw - 1)) + l16) * g.ldw2 +
                                                               32 * (wk * KS + s) + 8 * lg);
            const int c = 16 * (nb0 + min(nb, nbw - 1)) + l16;
            bias2[nb] = c < g.nv ? g.bias[c] : 0.f;        // the accumulator's column is the lane's for every row
        }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            asm volatile("" : "+v"(bias2[nb]));
#pragma unroll
            for (int s = 0; s < KS; ++s) asm volatile("" : "+v"(wf2[nb][s]));
        }
    }
    if (HAS_BIAS)
        for (int c = tid; c < NP; c += RG_NT) lbias[c] = c < g.nv ? g.bias[c] : 0.f;
    // W is in the registers before the ring starts: a use the compiler can see here, so that it does not wait for these
    // loads -- and with them for every LDS-DMA piece issued since -- in front of the first MFMA of every tile
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int s = 0; s < KS; ++s) asm volatile("" : "+v"(wf[nb][s]));

    float biasr[(P16 && (EPI == SEI_EPI_BIAS || GELU16)) ? NB : 1];
    bool is_one[GELU16 ? NB : 1];                     // this lane's column of block nb is the ones column
    if constexpr (GELU16) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) is_one[nb] = 16 * (nb0 + min(nb, nbw - 1)) + l16 == g.gelu_one_at;
    }
    if constexpr (P16 && (EPI == SEI_EPI_BIAS || GELU16)) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const int c = 16 * (nb0 + min(nb, nbw - 1)) + l16;
            biasr[nb] = c < g.nv ? g.bias[c] : 0.f;        // the accumulator's column is the lane's for every row
        }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) asm volatile("" : "+v"(biasr[nb]));
    }
    // ---- DMA pieces of a stage: piece q = image q / (TR / 8), rows 8 (q % (TR / 8)) ..; chunk swizzle (r >> 1) & 7
    unsigned offa[EMAX];
#pragma unroll
    for (int e = 0; e < EMAX; ++e) {
        const int q = min(wave + 8 * e, P - 1), im = q / (TR / 8), p = q % (TR / 8);
        const int r = 8 * p + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        offa[e] = (unsigned)r * (unsigned)g.lda * 2u + 128u * im + 16u * c;
    }
    unsigned offa2[TWO ? EMAX : 1];
    if constexpr (TWO) {
#pragma unroll
        for (int e = 0; e < EMAX; ++e) {
            const int q = min(wave + 8 * e, P - 1), im = q / (TR / 8), p = q % (TR / 8);
            const int r = 8 * p + (lane >> 3);
            const int c = (lane & 7) ^ ((r >> 1) & 7);
            offa2[e] = (unsigned)r * (unsigned)g.lda2 * 2u + 128u * im + 16u * c;
        }
    }
    auto issue = [&](int t) {                            // tile t of this workgroup (clamped: see the loop)
        const int tile = b + min(t, nt - 1) * G;
        const char *ab = reinterpret_cast<const char *>(g.A + (size_t)tile * TR * g.lda);
        char *dst = smem + (t % 3) * STAGE;
#pragma unroll
        for (int e = 0; e < EMAX; ++e) {
            if (P % 8 != 0 && e == EMAX - 1 && wave >= P % 8) break;           // wave-uniform
            dma16_base(ab, offa[e], dst + (wave + 8 * e) * 1024);
        }
        if constexpr (TWO) {
            const char *a2 = reinterpret_cast<const char *>(g.A2 + (size_t)tile * TR * g.lda2);
#pragma unroll
            for (int e = 0; e < EMAX; ++e) {
                if (P % 8 != 0 && e == EMAX - 1 && wave >= P % 8) break;
                dma16_base(a2, offa2[e], dst + KT * TR * 128 + (wave + 8 * e) * 1024);
            }
        }
    };
    constexpr int NDMA_LO = OPS * (P / 8);               // pieces per wave and stage: waves below P % 8 issue OPS more
    constexpr int NDMA_HI = OPS * (P / 8 + 1);

    // ---- the quads of a tile this thread finishes: quad i = (row, 4 columns)
    // Float32 outputs narrower than the tile (180 of 192 columns): the lanes past the edge redo the row's LAST valid quad --
    // same inputs, same result, same address -- instead of masking their stores: a branch around a store makes the store
    // count of the in-order vmcnt unknowable, and the compiler then waits for everything (the ring included) before it
    constexpr bool F32_ONLY = !(OUT16 || EPI == SEI_EPI_MUL_DGELU || EPI == SEI_EPI_BIAS_GELU || TWO);
    auto quad_row = [&](int q) { return (tid + RG_NT * (q / GQ)) / IR; };        // quad q = quad q % GQ of item q / GQ
    auto quad_col = [&](int q) {
        const int c = 4 * (GQ * ((tid + RG_NT * (q / GQ)) % IR) + q % GQ);
        return F32_ONLY ? min(c, g.nv - 4) : c;
    };
    static_assert(!(HAS_ROWS && RAGGED), "auxiliary rows are loaded for whole passes");
    f32x4 cur[HAS_ROWS ? QPT : 1];
    float curs[EPI == SEI_EPI_BIAS_SCALE_RES ? QPT : 1];
    auto load_aux = [&](int t, f32x4 (&a)[HAS_ROWS ? QPT : 1], float (&sc)[EPI == SEI_EPI_BIAS_SCALE_RES ? QPT : 1]) {
        if constexpr (HAS_ROWS) {
            const int tile = b + min(t, nt - 1) * G;
            const float *rows = EPI == SEI_EPI_BIAS_SCALE_RES ? g.R2 : g.R1;
#pragma unroll
            for (int i = 0; i < QPT; ++i) {
                const size_t row = (size_t)tile * TR + quad_row(i);
                const int qc = quad_col(i);
                // Loads the compiler does not track (it would wait for them, and with them for every LDS-DMA piece in
                // flight, with vmcnt(0)): the values are used only behind rg_wait_rows below
                const float *src = rows + row * g.ldr + (qc < g.nv ? qc : 0);
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(a[i]) : "v"(src) : "memory");
                if constexpr (EPI == SEI_EPI_BIAS_SCALE_RES) {
                    const float *ssrc = g.R1 + row;
                    asm volatile("global_load_dword %0, %1, off" : "=v"(sc[i]) : "v"(ssrc) : "memory");
                }
            }
        }
    };

    // ---- LayerNorm backward in the epilogue: thread = (row tid / 16, lane el of 16), quads el, el + 16, el + 32 of the row
    const int er = tid >> 4, el = tid & 15;
    const int nvq = g.nv >> 2;
    f32x4 lx[LNX ? 3 : 1], lr[LNB ? 3 : 1], ag[LNB ? 3 : 1], ab[LNB ? 3 : 1], ac[LNB ? 3 : 1];
    float lmu = 0.f, lrs = 0.f, lsc = 1.f;
    if constexpr (LNF) {
        for (int c = tid; c < NP; c += RG_NT) {
            lgam[c] = c < g.nv ? g.gamma[c] : 0.f;
            lbet[c] = c < g.nv ? g.beta2[c] : 0.f;
        }
    }
    if constexpr (LNB) {
        for (int c = tid; c < NP; c += RG_NT) lbias[c] = c < g.nv ? g.gamma[c] : 0.f;     // the norm's weight, read per tile
#pragma unroll
        for (int k = 0; k < 3; ++k) ag[k] = ab[k] = ac[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    auto load_ln = [&](int t) {
        if constexpr (LNF) {                             // the residual rows (and the row's stochastic-depth factor)
            const int tile = b + min(t, nt - 1) * G;
            const size_t row = (size_t)tile * TR + er;
            const float *rows = EPI == RG_EPI_SCALE_RES_LN ? g.R2 : g.R1;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float *xs = rows + row * g.ldr + 4 * min(el + 16 * k, nvq - 1);
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(lx[k]) : "v"(xs) : "memory");
            }
            if constexpr (EPI == RG_EPI_SCALE_RES_LN) {
                const float *cs = g.R1 + row;
                asm volatile("global_load_dword %0, %1, off" : "=v"(lsc) : "v"(cs) : "memory");
            }
        }
        if constexpr (LNB) {
            const int tile = b + min(t, nt - 1) * G;
            const size_t row = (size_t)tile * TR + er;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int c = 4 * min(el + 16 * k, nvq - 1);
                const float *xs = g.R1 + row * g.ldr + c, *rs = g.R2 + row * g.ldr + c;
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(lx[k]) : "v"(xs) : "memory");
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(lr[k]) : "v"(rs) : "memory");
            }
            const float *ms = g.mean + row, *ss = g.rstd + row;
            asm volatile("global_load_dword %0, %1, off" : "=v"(lmu) : "v"(ms) : "memory");
            asm volatile("global_load_dword %0, %1, off" : "=v"(lrs) : "v"(ss) : "memory");
            if constexpr (OUT16) {
                const float *cs = g.scale + row;
                asm volatile("global_load_dword %0, %1, off" : "=v"(lsc) : "v"(cs) : "memory");
            }
        }
    };

    // ---- prologue: tile 0 and its auxiliary rows landed, tile 1 in flight
    issue(0);
    issue(1);
    if (P % 8 != 0 && wave < P % 8) rg_wait_vmcnt<NDMA_HI>();
    else rg_wait_vmcnt<NDMA_LO>();
    rg_lds_barrier();

    for (int t = 0; t < nt; ++t) {
        float *const patch = PATCH2 ? reinterpret_cast<float *>(reinterpret_cast<char *>(patch0) + (t & 1) * PATCH_ONE) : patch0;
        load_aux(t, cur, curs);                           // this tile's residual / GELU' rows: used after the MFMAs
        load_ln(t);
        __builtin_amdgcn_sched_barrier(0);                // (the order the counted wait below assumes)
        issue(t + 2);
        __builtin_amdgcn_sched_barrier(0);
        // ---- MFMAs: this wave's rows x columns of the tile (its half of K when WK = 2)
        const char *st = smem + (t % 3) * STAGE;
        f32x4 acc[(GELU16 || TWO) ? 1 : RB][NB];
        if constexpr (GELU16 || TWO) {
            // One 16-row block at a time: MFMAs, the element-wise function in the accumulator layout, bf16 into the patch.
            // Half the accumulators (GELU16: what lets two workgroups share a CU's registers), and the VALU work of one
            // block stands next to the other block's MFMAs in the instruction stream.
            // TWO: the GELU' input of these very elements, recomputed -- acc2 = A2 W2^T in the same accumulator layout and
            // the same MFMA order as the forward kernel (the same float32 values it fed to GELU) --, then
            // acc * gelu'(acc2 + bias).
            static_assert(WK == 1 && NBX == 0, "whole column blocks per wave, all of K");
            unsigned short *pw16 = reinterpret_cast<unsigned short *>(patch) + (wr * RB * 16 + 4 * lg) * LDP16 + 16 * nb0 + l16;
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                f32x4 acc2[TWO ? NB : 1];
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    acc[0][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if constexpr (TWO) acc2[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    const char *img = st + (s >> 1) * (TR * 128) + (wr * RB * 16 + l16) * 128 + (((4 * (s & 1) + lg) ^ (l16 >> 1)) * 16);
                    const bf16x8 fa = *reinterpret_cast<const bf16x8 *>(img + rb * 16 * 128);
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb)
                        acc[0][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, wf[nb][s], acc[0][nb], 0, 0, 0);
                    if constexpr (TWO) {
                        const bf16x8 fa2 = *reinterpret_cast<const bf16x8 *>(img + KT * TR * 128 + rb * 16 * 128);
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb)
                            acc2[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa2, wf2[nb][s], acc2[nb], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float v;
                        if constexpr (TWO) v = acc[0][nb][j] * sei_dgelu_bf16out(acc2[nb][j] + bias2[nb]);
                        else v = is_one[nb] ? 1.0f : sei_gelu_bf16out(acc[0][nb][j] + biasr[nb]);
                        const __bf16 h = (__bf16)v;
                        pw16[(16 * rb + j) * LDP16 + 16 * nb] = __builtin_bit_cast(unsigned short, h);
                    }
            }
        } else {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[rb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int sg = wk * KS + s;                   // wave-uniform
            const char *img = st + (sg >> 1) * (TR * 128) + (wr * RB * 16 + l16) * 128 + (((4 * (sg & 1) + lg) ^ (l16 >> 1)) * 16);
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const bf16x8 fa = *reinterpret_cast<const bf16x8 *>(img + rb * 16 * 128);
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    if (NBX != 0 && nb == NB - 1 && nbw < NB) break;         // wave-uniform: this group has NB - 1 blocks
                    acc[rb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, wf[nb][s], acc[rb][nb], 0, 0, 0);
                }
            }
        }
        }
        // tile t + 1 (issued one iteration ago) and everything older have landed; what this iteration issued may fly
        if (P % 8 != 0 && wave < P % 8) rg_wait_vmcnt<NAUX + NDMA_HI>();
        else rg_wait_vmcnt<NAUX + NDMA_LO>();
        // ---- accumulators into the patch (WK = 2: the second half of K adds to the first)
        if constexpr (P16 && !GELU16 && !TWO) {
            unsigned short *pw16 = reinterpret_cast<unsigned short *>(patch) + (wr * RB * 16 + 4 * lg) * LDP16 + 16 * nb0 + l16;
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    if (NBX != 0 && nb == NB - 1 && nbw < NB) break;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float v = acc[rb][nb][j];
                        if constexpr (EPI == SEI_EPI_BIAS) v += biasr[nb];
                        const __bf16 h = (__bf16)v;
                        pw16[(16 * rb + j) * LDP16 + 16 * nb] = __builtin_bit_cast(unsigned short, h);
                    }
                }
        }
        float *pw = patch + (wr * RB * 16 + 4 * lg) * LDP + 16 * nb0 + l16;
        if (!P16 && wk == 0) {
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    if (NBX != 0 && nb == NB - 1 && nbw < NB) break;
#pragma unroll
                    for (int j = 0; j < 4; ++j) pw[(16 * rb + j) * LDP + 16 * nb] = acc[rb][nb][j];
                }
        }
        rg_lds_barrier();
        if constexpr (WK == 2) {
            if (wk == 1) {
#pragma unroll
                for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                        for (int j = 0; j < 4; ++j) pw[(16 * rb + j) * LDP + 16 * nb] += acc[rb][nb][j];
            }
            rg_lds_barrier();
        }
        // ---- rows out: the tile's auxiliary rows have landed when only this iteration's DMA pieces are in flight. The wait
        // first, between scheduling fences, and only then a use of the registers that the compiler can see: a copy it
        // makes for that use then reads landed data (tied to the wait itself, the copy was placed in front of it)
        if constexpr (HAS_ROWS) {
            __builtin_amdgcn_sched_barrier(0);
            if (P % 8 != 0 && wave < P % 8) rg_wait_vmcnt<NDMA_HI>();
            else rg_wait_vmcnt<NDMA_LO>();
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < QPT; ++i) {
                asm volatile("" : "+v"(cur[i]));
                if constexpr (EPI == SEI_EPI_BIAS_SCALE_RES) asm volatile("" : "+v"(curs[i]));
            }
        }
        const size_t row0 = (size_t)(b + t * G) * TR;
        if constexpr (LNF) {
            __builtin_amdgcn_sched_barrier(0);
            if (P % 8 != 0 && wave < P % 8) rg_wait_vmcnt<NDMA_HI>();
            else rg_wait_vmcnt<NDMA_LO>();
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < 3; ++k) asm volatile("" : "+v"(lx[k]));
            if constexpr (EPI == RG_EPI_SCALE_RES_LN) asm volatile("" : "+v"(lsc));
            // v = residual + [factor] (acc + bias): the block's new token row; then nn.LayerNorm over its nv channels
            // (mean, then the variance of the centred values, as sei_ln_fwd_bf16_pad)
            const size_t row = row0 + er;
            f32x4 v[3];
            float s1 = 0.f;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int qk = el + 16 * k;
                const bool valid = qk < nvq;
                v[k] = *reinterpret_cast<const f32x4 *>(patch + er * LDP + 4 * qk) + *reinterpret_cast<const f32x4 *>(lbias + 4 * qk);
                if constexpr (EPI == RG_EPI_SCALE_RES_LN) v[k] = lx[k] + lsc * v[k];
                else v[k] += lx[k];
                if (valid) {
                    *reinterpret_cast<f32x4 *>(g.D32 + row * g.ld32 + 4 * qk) = v[k];
                    s1 += (v[k][0] + v[k][1]) + (v[k][2] + v[k][3]);
                }
            }
#pragma unroll
            for (int off = 8; off > 0; off >>= 1) s1 += __shfl_xor(s1, off, 16);
            const float invC = 1.0f / (float)g.nv, mu = s1 * invC;
            float s2 = 0.f;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[k][j] -= mu;
                if (el + 16 * k < nvq) s2 += (v[k][0] * v[k][0] + v[k][1] * v[k][1]) + (v[k][2] * v[k][2] + v[k][3] * v[k][3]);
            }
#pragma unroll
            for (int off = 8; off > 0; off >>= 1) s2 += __shfl_xor(s2, off, 16);
            const float rs = rsqrtf(s2 * invC + g.eps);
            if (el == 0) {
                g.mean_out[row] = mu;
                g.rstd_out[row] = rs;
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int qk = el + 16 * k;
                const f32x4 gm = *reinterpret_cast<const f32x4 *>(lgam + 4 * qk), bt = *reinterpret_cast<const f32x4 *>(lbet + 4 * qk);
                f32x4 y = f32x4{0.f, 0.f, 0.f, 0.f};
                if (qk < nvq) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) y[j] = v[k][j] * rs * gm[j] + bt[j];
                } else if (qk == nvq && g.ones_col) {
                    y[0] = 1.0f;
                }
                uint2 h;
                h.x = rg_pack2(y[0], y[1]);
                h.y = rg_pack2(y[2], y[3]);
                *reinterpret_cast<uint2 *>(g.D16 + row * g.ld16 + 4 * qk) = h;
            }
        }
        if constexpr (LNB) {
            __builtin_amdgcn_sched_barrier(0);
            if (P % 8 != 0 && wave < P % 8) rg_wait_vmcnt<NDMA_HI>();
            else rg_wait_vmcnt<NDMA_LO>();
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                asm volatile("" : "+v"(lx[k]));
                asm volatile("" : "+v"(lr[k]));
            }
            asm volatile("" : "+v"(lmu));
            asm volatile("" : "+v"(lrs));
            if constexpr (OUT16) asm volatile("" : "+v"(lsc));
            // gx = rstd (t - mean_c(t) - xhat mean_c(t xhat)) + residual, t = g gamma (torch's native_layer_norm_backward)
            f32x4 gh[3], xh[3], tq[3];
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const bool valid = el + 16 * k < nvq;
                gh[k] = *reinterpret_cast<const f32x4 *>(patch + er * LDP + 4 * (el + 16 * k));
                const f32x4 gm = *reinterpret_cast<const f32x4 *>(lbias + 4 * (el + 16 * k));
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    xh[k][j] = (lx[k][j] - lmu) * lrs;
                    tq[k][j] = gh[k][j] * gm[j];
                }
                if (valid) {
                    s1 += (tq[k][0] + tq[k][1]) + (tq[k][2] + tq[k][3]);
                    s2 += (tq[k][0] * xh[k][0] + tq[k][1] * xh[k][1]) + (tq[k][2] * xh[k][2] + tq[k][3] * xh[k][3]);
                    ag[k] += gh[k] * xh[k];
                    ab[k] += gh[k];
                }
            }
#pragma unroll
            for (int off = 8; off > 0; off >>= 1) {
                s1 += __shfl_xor(s1, off, 16);
                s2 += __shfl_xor(s2, off, 16);
            }
            const float invC = 1.0f / (float)g.nv;
            s1 *= invC;
            s2 *= invC;
            const size_t row = row0 + er;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int qk = el + 16 * k;
                const bool valid = qk < nvq;
                f32x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = lrs * (tq[k][j] - s1 - xh[k][j] * s2) + lr[k][j];
                if (valid) *reinterpret_cast<f32x4 *>(g.D32 + row * g.ld32 + 4 * qk) = o;
                if constexpr (OUT16) {
                    f32x4 ys;
#pragma unroll
                    for (int j = 0; j < 4; ++j) ys[j] = valid ? o[j] * lsc : 0.f;
                    if constexpr (KT != 9) ac[k] += ys;              // (K = 576: no registers left for the third column sum)
                    uint2 h;
                    h.x = rg_pack2(ys[0], ys[1]);
                    h.y = rg_pack2(ys[2], ys[3]);
                    *reinterpret_cast<uint2 *>(g.D16 + row * g.ld16 + 4 * qk) = h;
                }
            }
        }
        if constexpr (P16) {
#pragma unroll
            for (int it = 0; it < IPT; ++it) {
                if (RAGGED && it == IPT - 1 && tid + RG_NT * it >= TR * IR) break;      // wave-uniform
                const int idx = tid + RG_NT * it, qr = idx / IR, qc = 8 * (idx % IR);
                const uint4 h = *reinterpret_cast<const uint4 *>(reinterpret_cast<const unsigned short *>(patch) + qr * LDP16 + qc);
                *reinterpret_cast<uint4 *>(g.D16 + (row0 + qr) * g.ld16 + qc) = h;
            }
        }
#pragma unroll UNR
        for (int it = 0; it < ((LNX || P16) ? 0 : IPT); ++it) {
            if (RAGGED && it == IPT - 1 && tid + RG_NT * it >= TR * IR) break;      // wave-uniform (whole waves past the end)
            f32x4 v[GQ];
#pragma unroll
            for (int gq = 0; gq < GQ; ++gq) {
                const int i = it * GQ + gq;
                const int qr = quad_row(i), qc = quad_col(i);
                v[gq] = *reinterpret_cast<const f32x4 *>(patch + qr * LDP + qc);
                if constexpr (HAS_BIAS) v[gq] += *reinterpret_cast<const f32x4 *>(lbias + qc);
                if constexpr (EPI == SEI_EPI_BIAS_RES) v[gq] += cur[i];
                if constexpr (EPI == SEI_EPI_BIAS_SCALE_RES) v[gq] = cur[i] + curs[i] * v[gq];
                if constexpr (EPI == SEI_EPI_MUL_DGELU) {
                    const bool ok = qc < g.nv;
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[gq][j] = ok ? v[gq][j] * sei_dgelu_bf16out(cur[i][j]) : 0.f;
                }
            }
            const int qr = quad_row(it * GQ), qc = quad_col(it * GQ);
            const size_t row = row0 + qr;
            if constexpr (F32_ONLY) *reinterpret_cast<f32x4 *>(g.D32 + row * g.ld32 + qc) = v[0];
            if constexpr (EPI == SEI_EPI_BIAS_GELU) {                // (nv = all the columns, host-checked; the float32
                if (g.D32) *reinterpret_cast<f32x4 *>(g.D32 + row * g.ld32 + qc) = v[0];   // pre-activation is optional:
            }                                                        // uniform branch, no aux loads in this variant)
            if constexpr (EPI == SEI_EPI_BIAS_GELU) {
                float ge[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) ge[j] = qc + j == g.gelu_one_at ? 1.0f : sei_gelu_bf16out(v[0][j]);
                uint2 h;
                h.x = rg_pack2(ge[0], ge[1]);
                h.y = rg_pack2(ge[2], ge[3]);
                *reinterpret_cast<uint2 *>(g.D16 + row * g.ld16 + qc) = h;
            } else if constexpr (GQ == 2) {
                uint4 h;
                h.x = rg_pack2(v[0][0], v[0][1]);
                h.y = rg_pack2(v[0][2], v[0][3]);
                h.z = rg_pack2(v[GQ - 1][0], v[GQ - 1][1]);
                h.w = rg_pack2(v[GQ - 1][2], v[GQ - 1][3]);
                *reinterpret_cast<uint4 *>(g.D16 + row * g.ld16 + qc) = h;
            }
        }
        if constexpr (!PATCH2) rg_lds_barrier();          // the patch and stage t % 3 are free again
    }
    rg_wait_vmcnt<0>();                                   // the clamped stages still in flight
    if constexpr (LNB) {
        // column sums of this workgroup: [3][32 rows][48 quads] through the (now idle) ring, then 3 x 48 threads add up
        // the 32 rows in a fixed order; [workgroup][3][nv] partials, folded by sei_fold_partials3
        static_assert(3 * STAGE >= 3 * 32 * 48 * 16, "the ring holds the partial sums");
        __builtin_amdgcn_s_barrier();
        f32x4 *red = reinterpret_cast<f32x4 *>(smem);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            red[(0 * 32 + er) * 48 + el + 16 * k] = ag[k];
            red[(1 * 32 + er) * 48 + el + 16 * k] = ab[k];
            red[(2 * 32 + er) * 48 + el + 16 * k] = ac[k];
        }
        __syncthreads();
        if (tid < 144) {
            const int sel = tid / 48, qk = tid - sel * 48;
            f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
            for (int r = 0; r < 32; ++r) a += red[(sel * 32 + r) * 48 + qk];
            if (qk < nvq) *reinterpret_cast<f32x4 *>(g.part + ((size_t)b * 3 + sel) * g.nv + 4 * qk) = a;
        }
    }
}

template <int KT, int NBT, int TR, int WN, int WR, int EPI, bool OUT16>
int rg_launch(const RowGemmArgs &g, int M, hipStream_t s, int per_cu = 1) {
    RowGemmArgs a = g;
    a.tiles = M / TR;
    const int grid = a.tiles < 256 * per_cu ? a.tiles : 256 * per_cu;
    hipLaunchKernelGGL((rowgemm_kernel<KT, NBT, TR, WN, WR, EPI, OUT16>), dim3((unsigned)grid), dim3(RG_NT), 0, s, a);
    return sei_launch_status();
}

bool tg_block_ok(const SeiTokGradBlock &b, long long K2) {
    if (!b.Y1 || !b.X1 || !b.D || (K2 && (!b.Y2 || !b.X2))) return false;
    if (b.ldy % 8 || b.ldx % 8 || b.y0 % 8 || b.x0 % 8 || b.y0 < 0 || b.x0 < 0) return false;
    if (b.ldy < b.y0 + 192 || b.ldx < b.x0 + 192 || b.ldd < 192) return false;
    if ((((uintptr_t)b.Y1 | (uintptr_t)b.Y2 | (uintptr_t)b.X1 | (uintptr_t)b.X2) & 15) != 0) return false;
    return (unsigned long long)(64 * (size_t)b.ldy) * 2 < (1ull << 32) && (unsigned long long)(64 * (size_t)b.ldx) * 2 < (1ull << 32);
}

}  // namespace

extern "C" int sei_tokgrad_bf16_blocks(const SeiTokGradBlock *blocks, int nblocks, long long K1, long long K2,
                                       void *stream) {
    SEI_REQUIRE(blocks && nblocks >= 1 && nblocks <= TG_MAX_BLOCKS);
    SEI_REQUIRE(K1 > 0 && K2 >= 0 && K1 % 64 == 0 && K2 % 64 == 0 && (K1 + K2) / 64 < (1ll << 24));
    TokGradArgs g;
    for (int i = 0; i < nblocks; ++i) {
        SEI_REQUIRE(tg_block_ok(blocks[i], K2));
        g.blk[i] = blocks[i];
        if (!K2) { g.blk[i].Y2 = blocks[i].Y1; g.blk[i].X2 = blocks[i].X1; }
    }
    for (int i = nblocks; i < TG_MAX_BLOCKS; ++i) g.blk[i] = g.blk[0];
    g.nblk = nblocks;
    g.kt_seg = (int)(K1 / 64);
    g.kt_total = (int)((K1 + K2) / 64);
    g.workers_per_xcd = 32 / nblocks;                               // 32 CUs per XCD, one workgroup each
    // Ranges of at least eight 64-token stages per workgroup: the stages travel through registers (three in flight);
    // shorter ones: the LDS-DMA ring with eight waves.
    if (g.kt_total / (8 * g.workers_per_xcd) >= 8)
        hipLaunchKernelGGL(tokgrad_regs_kernel, dim3(8 * (unsigned)(g.workers_per_xcd * nblocks)), dim3(512), 0,
                           (hipStream_t)stream, g);
    else
        hipLaunchKernelGGL(tokgrad_kernel<8>, dim3(8 * (unsigned)(g.workers_per_xcd * nblocks)), dim3(2 * TG_NT), 0,
                           (hipStream_t)stream, g);
    return sei_launch_status();
}

extern "C" size_t sei_tokgrad_bf16_eligible(int Mo, int Ni, int ldy, int ldx, long long K1, long long K2) {
    if (Mo <= 0 || Ni <= 0 || Mo % 192 || Ni % 192) return 0;
    const int ng = (Mo / 192) * (Ni / 192);
    if (ng > 8 || ldy % 8 || ldx % 8 || ldy < Mo || ldx < Ni) return 0;      // (one weight: at most eight blocks)
    if (K1 <= 0 || K2 < 0 || K1 % 64 || K2 % 64) return 0;
    if ((K1 + K2) / 64 >= (1ll << 24)) return 0;
    return (size_t)ng;
}

extern "C" int sei_tokgrad_bf16(const uint16_t *Y1, const uint16_t *Y2, int ldy, const uint16_t *X1, const uint16_t *X2,
                                int ldx, float *D, int ldd, int Mo, int Ni, long long K1, long long K2, void *stream) {
    SEI_REQUIRE(Y1 && X1 && D && sei_tokgrad_bf16_eligible(Mo, Ni, ldy, ldx, K1, K2));
    SEI_REQUIRE(ldd >= Ni);
    SeiTokGradBlock blocks[TG_MAX_BLOCKS];
    int n = 0;
    for (int gy = 0; gy < Mo / 192; ++gy)
        for (int gx = 0; gx < Ni / 192; ++gx) {
            SeiTokGradBlock &b = blocks[n++];
            b.Y1 = Y1; b.Y2 = Y2; b.X1 = X1; b.X2 = X2;
            b.ldy = ldy; b.ldx = ldx; b.y0 = 192 * gy; b.x0 = 192 * gx;
            b.D = D + (size_t)192 * gy * ldd + 192 * gx; b.ldd = ldd;
        }
    return sei_tokgrad_bf16_blocks(blocks, n, K1, K2, stream);
}

extern "C" size_t sei_rowgemm_bf16_eligible(long long M, int N, int K, int epilogue, int out16) {
    if (M <= 0 || M % 64 != 0 || M >= (1ll << 31)) return 0;
    const bool f32out = !out16;
    switch (epilogue) {
        case SEI_EPI_BIAS: return (N == 576 && K == 192 && out16) ? 1 : 0;
        case SEI_EPI_BIAS_RES:
        case SEI_EPI_BIAS_SCALE_RES: return (N == 192 && (K == 192 || K == 384) && f32out) ? 1 : 0;
        case SEI_EPI_BIAS_GELU: return (N == 384 && K == 192) ? 1 : 0;      // float32 pre-activation and / or bf16 gelu
        case SEI_EPI_MUL_DGELU: return (N == 384 && K == 192 && out16) ? 1 : 0;
        case SEI_EPI_NONE: return (N == 192 && ((K == 192 && out16) || ((K == 384 || K == 576) && f32out))) ? 1 : 0;
        default: return 0;
    }
}

extern "C" int sei_rowgemm_bf16(const uint16_t *A, int lda, const uint16_t *W, int ldw, float *D32, int ld32,
                                uint16_t *D16, int ld16, long long M, int N, int K, int nv, int epilogue,
                                const float *bias, const float *R1, const float *R2, int ldr, void *stream) {
    const int out16 = (D16 != nullptr && D32 == nullptr) ? 1 : 0;
    SEI_REQUIRE(A && W && sei_rowgemm_bf16_eligible(M, N, K, epilogue, out16));
    SEI_REQUIRE(lda >= K && ldw >= K && lda % 8 == 0 && ldw % 8 == 0);
    SEI_REQUIRE((((uintptr_t)A | (uintptr_t)W) & 15) == 0);
    SEI_REQUIRE(nv > 0 && nv <= N && nv % 4 == 0);
    SEI_REQUIRE((unsigned long long)64 * (size_t)lda * 2 < (1ull << 32));
    const bool has_bias = epilogue == SEI_EPI_BIAS || epilogue == SEI_EPI_BIAS_GELU || epilogue == SEI_EPI_BIAS_RES ||
                          epilogue == SEI_EPI_BIAS_SCALE_RES;
    SEI_REQUIRE(!has_bias || bias);
    if (D32) SEI_REQUIRE(ld32 >= nv && ld32 % 4 == 0 && ((uintptr_t)D32 & 15) == 0);
    if (D16) SEI_REQUIRE(ld16 >= N && ld16 % 8 == 0 && ((uintptr_t)D16 & 15) == 0);
    if (epilogue == SEI_EPI_BIAS_GELU) SEI_REQUIRE(D16 && nv == N);     // D32 (the float32 pre-activation) is optional
    if (epilogue == SEI_EPI_BIAS_RES || epilogue == SEI_EPI_MUL_DGELU)
        SEI_REQUIRE(R1 && !R2 && ldr >= nv && ldr % 4 == 0 && ((uintptr_t)R1 & 15) == 0);
    if (epilogue == SEI_EPI_BIAS_SCALE_RES) SEI_REQUIRE(R1 && R2 && ldr >= nv && ldr % 4 == 0 && ((uintptr_t)R2 & 15) == 0);
    RowGemmArgs g;
    g.A = A; g.W = W; g.lda = lda; g.ldw = ldw; g.tiles = 0; g.bias = bias; g.R1 = R1; g.R2 = R2; g.ldr = ldr;
    g.D32 = D32; g.D16 = D16; g.ld32 = ld32; g.ld16 = ld16; g.nv = nv; g.gelu_one_at = -1;
    hipStream_t s = (hipStream_t)stream;
    const int m = (int)M;
    // <k-tiles, 16-column blocks, rows per tile, column groups, row groups>
    switch (epilogue) {
        case SEI_EPI_BIAS: return rg_launch<3, 36, 32, 8, 1, SEI_EPI_BIAS, true>(g, m, s);
        case SEI_EPI_BIAS_RES:
            return K == 192 ? rg_launch<3, 12, 64, 4, 2, SEI_EPI_BIAS_RES, false>(g, m, s)
                            : rg_launch<6, 12, 32, 4, 2, SEI_EPI_BIAS_RES, false>(g, m, s);
        case SEI_EPI_BIAS_SCALE_RES:
            return K == 192 ? rg_launch<3, 12, 64, 4, 2, SEI_EPI_BIAS_SCALE_RES, false>(g, m, s)
                            : rg_launch<6, 12, 32, 4, 2, SEI_EPI_BIAS_SCALE_RES, false>(g, m, s);
        case SEI_EPI_BIAS_GELU:
            if (!D32) return rg_launch<3, 24, 32, 8, 1, RG_EPI_GELU16, true>(g, m, s, 2);
            return rg_launch<3, 24, 32, 8, 1, SEI_EPI_BIAS_GELU, false>(g, m, s);
        case SEI_EPI_MUL_DGELU: return rg_launch<3, 24, 32, 8, 1, SEI_EPI_MUL_DGELU, true>(g, m, s);
        default:
            if (K == 192) return rg_launch<3, 12, 64, 4, 2, SEI_EPI_NONE, true>(g, m, s);
            if (K == 384) return rg_launch<6, 12, 32, 4, 2, SEI_EPI_NONE, false>(g, m, s);
            return rg_launch<9, 12, 32, 4, 1, SEI_EPI_NONE, false>(g, m, s);
    }
}

extern "C" size_t sei_rowgemm_lnbwd_bf16_eligible(long long M, int K, int C) {
    return (M > 0 && M % 64 == 0 && M < (1ll << 31) && (K == 384 || K == 576) && C > 0 && C % 4 == 0 && C <= 192) ? 1 : 0;
}

extern "C" size_t sei_rowgemm_lnbwd_work_floats(int C) { return C > 0 ? (size_t)256 * 3 * (size_t)C : 0; }

extern "C" int sei_rowgemm_lnbwd_bf16(const uint16_t *A, int lda, const uint16_t *W, int ldw, long long M, int K, const float *x,
                                      const float *gamma, const float *mean, const float *rstd, const float *res, float *gx,
                                      int C, float *ggamma, float *gbeta, const float *row_scale, uint16_t *y16, int ldy,
                                      float *colsum, float *work, size_t work_floats, void *stream) {
    // ggamma = gbeta = NULL: the partial sums ([min(M / 32, 256)][3][C]) stay in `work` for sei_fold_many
    SEI_REQUIRE(A && W && x && gamma && mean && rstd && res && gx && work && (ggamma != nullptr) == (gbeta != nullptr));
    SEI_REQUIRE(sei_rowgemm_lnbwd_bf16_eligible(M, K, C) && work_floats >= sei_rowgemm_lnbwd_work_floats(C));
    SEI_REQUIRE(lda >= K && ldw >= K && lda % 8 == 0 && ldw % 8 == 0 && (((uintptr_t)A | (uintptr_t)W) & 15) == 0);
    SEI_REQUIRE((unsigned long long)64 * (size_t)lda * 2 < (1ull << 32));
    SEI_REQUIRE((((uintptr_t)x | (uintptr_t)gamma | (uintptr_t)res | (uintptr_t)gx | (uintptr_t)work) & 15) == 0);
    SEI_REQUIRE((y16 != nullptr) == (row_scale != nullptr) && (!colsum || y16));
    SEI_REQUIRE(!colsum || K == 384);         // (K = 576: the bf16 copy without its column sums -- no registers left for them)
    if (y16) SEI_REQUIRE(ldy >= 192 && ldy % 4 == 0 && ((uintptr_t)y16 & 7) == 0);
    RowGemmArgs g;
    g.A = A; g.W = W; g.lda = lda; g.ldw = ldw; g.tiles = 0; g.bias = nullptr; g.R1 = x; g.R2 = res; g.ldr = C;
    g.D32 = gx; g.D16 = y16; g.ld32 = C; g.ld16 = ldy; g.nv = C;
    g.mean = mean; g.rstd = rstd; g.gamma = gamma; g.scale = row_scale; g.part = work;
    hipStream_t s = (hipStream_t)stream;
    const int m = (int)M, groups = m / 32 < 256 ? m / 32 : 256;
    int rc;
    if (K == 384)
        rc = y16 ? rg_launch<6, 12, 32, 4, 1, RG_EPI_LNBWD, true>(g, m, s) : rg_launch<6, 12, 32, 4, 1, RG_EPI_LNBWD, false>(g, m, s);
    else
        rc = y16 ? rg_launch<9, 12, 32, 4, 1, RG_EPI_LNBWD, true>(g, m, s) : rg_launch<9, 12, 32, 4, 1, RG_EPI_LNBWD, false>(g, m, s);
    if (rc != 0 || !ggamma) return rc;
    return sei_fold_partials3(work, groups, C, ggamma, gbeta, colsum, s);
}

extern "C" size_t sei_rowgemm_dgelu_bf16_eligible(long long M, int N, int K) {
    return (M > 0 && M % 64 == 0 && M < (1ll << 31) && N == 384 && K == 192) ? 1 : 0;
}

extern "C" int sei_rowgemm_dgelu_bf16(const uint16_t *A, int lda, const uint16_t *W, int ldw, const uint16_t *A2, int lda2,
                                      const uint16_t *W2, int ldw2, const float *bias2, int nv, uint16_t *D16, int ld16,
                                      long long M, int N, int K, void *stream) {
    SEI_REQUIRE(A && W && A2 && W2 && bias2 && D16 && sei_rowgemm_dgelu_bf16_eligible(M, N, K));
    SEI_REQUIRE(lda >= K && ldw >= K && lda2 >= K && ldw2 >= K && lda % 8 == 0 && ldw % 8 == 0 && lda2 % 8 == 0 && ldw2 % 8 == 0);
    SEI_REQUIRE((((uintptr_t)A | (uintptr_t)W | (uintptr_t)A2 | (uintptr_t)W2 | (uintptr_t)D16) & 15) == 0);
    SEI_REQUIRE(nv > 0 && nv <= N && ld16 >= N && ld16 % 8 == 0);
    SEI_REQUIRE((unsigned long long)64 * (size_t)lda * 2 < (1ull << 32) && (unsigned long long)64 * (size_t)lda2 * 2 < (1ull << 32));
    RowGemmArgs g = {};
    g.A = A; g.W = W; g.lda = lda; g.ldw = ldw; g.A2 = A2; g.W2 = W2; g.lda2 = lda2; g.ldw2 = ldw2;
    g.bias = bias2; g.D16 = D16; g.ld16 = ld16; g.nv = nv;
    return rg_launch<3, 24, 32, 8, 1, RG_EPI_DGELU2, true>(g, (int)M, (hipStream_t)stream);
}

extern "C" size_t sei_rowgemm_ln_bf16_eligible(long long M, int K, int C) {
    return (M > 0 && M % 64 == 0 && M < (1ll << 31) && (K == 192 || K == 384) && C > 0 && C % 4 == 0 && C < 192) ? 1 : 0;
}

extern "C" int sei_rowgemm_ln_bf16(const uint16_t *A, int lda, const uint16_t *W, int ldw, long long M, int K, int C,
                                   const float *bias, const float *row_scale, const float *res, float *out, const float *gamma,
                                   const float *beta, float eps, int ones_col, uint16_t *h16, int ldh, float *mean,
                                   float *rstd, void *stream) {
    SEI_REQUIRE(A && W && bias && res && out && gamma && beta && h16 && mean && rstd && sei_rowgemm_ln_bf16_eligible(M, K, C));
    SEI_REQUIRE(lda >= K && ldw >= K && lda % 8 == 0 && ldw % 8 == 0 && (((uintptr_t)A | (uintptr_t)W) & 15) == 0);
    SEI_REQUIRE((unsigned long long)64 * (size_t)lda * 2 < (1ull << 32));
    SEI_REQUIRE((((uintptr_t)res | (uintptr_t)out) & 15) == 0 && ((uintptr_t)h16 & 7) == 0 && ldh >= 192 && ldh % 4 == 0);
    RowGemmArgs g = {};
    g.A = A; g.W = W; g.lda = lda; g.ldw = ldw; g.bias = bias; g.ldr = C; g.D32 = out; g.ld32 = C; g.D16 = h16; g.ld16 = ldh;
    g.nv = C; g.gamma = gamma; g.beta2 = beta; g.mean_out = mean; g.rstd_out = rstd; g.eps = eps; g.ones_col = ones_col;
    hipStream_t s = (hipStream_t)stream;
    const int m = (int)M;
    if (row_scale) {
        g.R1 = row_scale; g.R2 = res;
        return K == 192 ? rg_launch<3, 12, 32, 4, 2, RG_EPI_SCALE_RES_LN, false>(g, m, s)
                        : rg_launch<6, 12, 32, 4, 2, RG_EPI_SCALE_RES_LN, false>(g, m, s);
    }
    g.R1 = res; g.R2 = nullptr;
    return K == 192 ? rg_launch<3, 12, 32, 4, 2, RG_EPI_RES_LN, false>(g, m, s)
                    : rg_launch<6, 12, 32, 4, 2, RG_EPI_RES_LN, false>(g, m, s);
}

extern "C" int sei_rowgemm_gelu_bf16(const uint16_t *A, int lda, const uint16_t *W, int ldw, const float *bias, int nv,
                                     uint16_t *D16, int ld16, long long M, int N, int K, int one_at, void *stream) {
    SEI_REQUIRE(A && W && bias && D16 && sei_rowgemm_bf16_eligible(M, N, K, SEI_EPI_BIAS_GELU, 0));
    SEI_REQUIRE(lda >= K && ldw >= K && lda % 8 == 0 && ldw % 8 == 0 && (((uintptr_t)A | (uintptr_t)W) & 15) == 0);
    SEI_REQUIRE(nv == N && ld16 >= N && ld16 % 8 == 0 && ((uintptr_t)D16 & 15) == 0);       // bias: all N entries (zeros in the pad)
    SEI_REQUIRE((unsigned long long)64 * (size_t)lda * 2 < (1ull << 32) && one_at >= -1 && one_at < N);
    RowGemmArgs g = {};
    g.A = A; g.W = W; g.lda = lda; g.ldw = ldw; g.bias = bias; g.D16 = D16; g.ld16 = ld16; g.nv = nv; g.gelu_one_at = one_at;
    return rg_launch<3, 24, 32, 8, 1, RG_EPI_GELU16, true>(g, (int)M, (hipStream_t)stream, 2);
}
