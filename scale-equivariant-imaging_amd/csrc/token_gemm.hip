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
#include "sei_common.h"

namespace {

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f32x4 = __attribute__((ext_vector_type(4))) float;
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void glb_void;
typedef short v4s __attribute__((ext_vector_type(4)));

// 16-B chunk swizzle of a [64 tokens][64 columns] image (128-B rows, two per 256-B bank row): as gemm_bf16pq.h
__device__ __forceinline__ int tg_swz(int row) { return (((row >> 1) & 1) << 1) | (((row >> 3) & 1) << 2); }

constexpr int TG_NT = 256;                 // four waves, one per SIMD: 144 accumulator registers each
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

__global__ __launch_bounds__(TG_NT) void tokgrad_kernel(TokGradArgs g) {
    __shared__ __attribute__((aligned(1024))) char smem[TG_NSTAGE * TG_STAGE];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave >> 1, wc = wave & 1;
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

    // ---- DMA: 48 1-KiB pieces per stage, 12 per wave; piece q = image q / 8 (0-2: dY, 3-5: X), rows 8 (q % 8) ..
    unsigned off[12];
#pragma unroll
    for (int e = 0; e < 12; ++e) {
        const int q = wave + 4 * e, im = q >> 3, p = q & 7;
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
        for (int e = 0; e < 12; ++e) {
            const int q = wave + 4 * e;                             // wave-uniform
            __builtin_amdgcn_global_load_lds((glb_void *)(((q >> 3) < 3 ? yb : xb) + off[e]),
                                             (lds_void *)(dst + q * 1024), 16, 0, 0);
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

    f32x4 acc[6][6];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // Every iteration issues one stage (past the end: the last k-tile again, into a stage nobody reads), so the
    // counted wait below always leaves exactly the 12 pieces of the next stage in flight.
    issue(0);
    issue(1);
    for (int u = 0; u < nt; ++u) {
        asm volatile("s_waitcnt vmcnt(12)" ::: "memory");           // this wave's pieces of stage u have landed
        __builtin_amdgcn_s_barrier();                               // ... everyone's; and stage u - 1 is read out
        issue(u + 2);
        const char *st = smem + (u % TG_NSTAGE) * TG_STAGE;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 fa[6], fb[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const int bi = 6 * wr + i, bj = 6 * wc + i;
                fa[i] = frag(st + (bi >> 2) * TG_IMG, bi & 3, ks);
                fb[i] = frag(st + (3 + (bj >> 2)) * TG_IMG, bj & 3, ks);
            }
#pragma unroll
            for (int i = 0; i < 6; ++i)
#pragma unroll
                for (int j = 0; j < 6; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // the two clamped stages still in flight

    // ---- float atomics: accumulator element (row 4 lg + r, column l16) of block (i, j). One atomic per clock and L2
    // channel is what they cost (measured: 27 us for the 256 x 147 KB of a launch, whatever the scope or the number of
    // adders per address), which is why a launch should carry as many 192 x 192 blocks as it can: the token range of a
    // workgroup grows and the number of partial blocks per output shrinks with the block count.
    float *d = b.D + (size_t)(96 * wr + 4 * lg) * b.ldd + 96 * wc + l16;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) atomicAdd(d + (size_t)(16 * i + r) * b.ldd + 16 * j, acc[i][j][r]);
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
    hipLaunchKernelGGL(tokgrad_kernel, dim3(8 * (unsigned)(g.workers_per_xcd * nblocks)), dim3(TG_NT), 0,
                       (hipStream_t)stream, g);
    return sei_launch_status();
}

extern "C" size_t sei_tokgrad_bf16_eligible(int Mo, int Ni, int ldy, int ldx, long long K1, long long K2) {
    if (Mo <= 0 || Ni <= 0 || Mo % 192 || Ni % 192) return 0;
    const int ng = (Mo / 192) * (Ni / 192);
    if (ng > TG_MAX_BLOCKS || ldy % 8 || ldx % 8 || ldy < Mo || ldx < Ni) return 0;
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
