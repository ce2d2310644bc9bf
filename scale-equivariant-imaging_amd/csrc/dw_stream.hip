// Streamed weight gradients of the U-Net's SHALLOW levels (C = 32, 128): dW (Mo x Ni) += gy^T x over 10^4 - 10^5 pixels
// for 1x1 convolutions whose matrices are 4 K - 64 K elements (reference: src/models/convolutional.py:40-42 conv2 / conv3,
// :106 Upsample's and :143 Downsample's 1x1 convolution; autograd's conv2d weight gradient).
//
// On the tiled GEMM (sei_gemm_bf16nt_dw2) such a product is ONE 128 x 128 output tile per 16 K elements with a reduction
// 55,296 - 221,184 rows long: up to 256 K-splits of a tile, every split re-staging both operand panels through the
// 128-column reduction-major images, float atomics from all of them -- 30-90 us for 70 MB of operands that HBM delivers
// in 13 us, MFMA busy 0.09 (profiles/r03_d). Here the product is turned round, as csrc/token_gemm.hip does for SwinIR's
// layer-sized weights: a persistent workgroup owns a WHOLE block of the gradient in its accumulators and a contiguous
// range of pixels, which it streams exactly once through a three-stage LDS ring filled by LDS-DMA two stages ahead; and
// ALL such gradients of a backward pass share ONE launch (a job table, as sei_fold_many does for the
// folds), the workgroups dealt to the jobs in proportion to their bytes.
//
// Block shapes (P = the operand with the narrower rows, Q = the wider one; both token-major bf16 as stored):
//   WIDE  P = 128 columns (two 64-column LDS images), Q = a 256-column slice (four images): conv2 / conv3 of the C = 128
//         level (512 x 128, 128 x 512 -> two blocks each) and the 1x1 convolutions between the 128- and 512-channel
//         levels. Four waves as 2 x 2, a wave holds 64 x 128 of the block (32 16x16x32 MFMAs per 12 transposing reads).
//   PAIR  rows of 32 and 128 columns (C = 32): two consecutive pixels are read as ONE row of 64 / 256 columns, the block
//         is 64 x 256, and what is wanted sits in its two diagonal quadrants: dW[c][j] = blk[c][j] + blk[32 + c][128 + j]
//         (the off-diagonal quadrants pair pixel 2r with pixel 2r + 1 and are not computed). Waves (0,0) and (1,1) work.
// SWAP (the narrow operand is x, i.e. Ni < Mo): the MFMA takes Q as its A operand, so that the accumulator's lane
// dimension -- what one atomic instruction covers -- still runs along the gradient's contiguous (input-channel) rows.
//
// The end is float atomics into the gradient (which therefore must be zeroed, not stored, at the start of a step):
// every workgroup adds its partial block; the job table keeps their number per gradient near 256 / jobs.
#include "sei_common.h"

namespace {

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f32x4 = __attribute__((ext_vector_type(4))) float;
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void glb_void;
typedef short v4s __attribute__((ext_vector_type(4)));

// 16-B chunk swizzle of a [64 rows][64 columns] bf16 image (128-B rows): as token_gemm.hip / gemm_bf16pq.h
__device__ __forceinline__ int ds_swz(int row) { return (((row >> 1) & 1) << 1) | (((row >> 3) & 1) << 2); }

constexpr int DS_NT = 256;
constexpr int DS_IMG = 64 * 128;
constexpr int DS_NSTAGE = 3;
constexpr int DS_MAX_UNITS = 40;

struct DwUnit {                    // one block of one gradient
    const unsigned short *P1, *P2, *Q1, *Q2;   // the two row segments of each operand (rows = pixels, or pixel pairs)
    float *D;                      // the block's first element of the gradient
    float *bias;                   // NULL, or where this unit ADDS the column sums of its gy columns (the bias gradient)
    int ldp, ldq;                  // row strides in elements (of the rows as this kernel reads them)
    int q0;                        // first Q column of the block
    int ldd;                       // row stride of the gradient
    int kt_seg, kt_total;          // 64-row k-tiles in the first segment / in both
    int first;                     // first workgroup of this unit; the next unit's `first` ends it
    int kind;                      // 1 = WIDE, 2 = WIDE + SWAP, 3 = PAIR, 4 = PAIR + SWAP
};

struct DwArgs {
    DwUnit u[DS_MAX_UNITS];
    int first_end;                 // total workgroups
    int nunits;
};

// One workgroup's share of one unit: k-tiles [kt_total worker / workers, kt_total (worker + 1) / workers).
template <int PI, int QI, bool SWAP, bool PAIR>
__device__ __forceinline__ void dw_stream_body(const DwUnit &u, int worker, int workers, char *smem) {
    constexpr int NIMG = PI + QI;
    constexpr int STAGE = NIMG * DS_IMG;
    constexpr int NPW = 2 * NIMG;                  // 1-KiB DMA pieces per wave and stage
    constexpr int PB = 2 * PI, QB = 2 * QI;        // 16-column blocks of a wave's share (the block is cut 2 x 2)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int l16 = lane & 15, lg = lane >> 4;
    const int kt0 = (int)((long long)u.kt_total * worker / workers);
    const int kt1 = (int)((long long)u.kt_total * (worker + 1) / workers);
    const int nt = kt1 - kt0;
    if (nt <= 0) return;                                           // block-uniform, before any barrier
    const unsigned short *P1 = u.P1, *P2 = u.P2, *Q1 = u.Q1, *Q2 = u.Q2;
    const int ldp = u.ldp, ldq = u.ldq, kt_seg = u.kt_seg;

    // ---- DMA: piece q = wave + 4 e of a stage: image q / 8 (the first PI: P), rows 8 (q % 8) .. + 7
    unsigned off[NPW];
#pragma unroll
    for (int e = 0; e < NPW; ++e) {
        const int q = wave + 4 * e, im = q >> 3, p = q & 7;
        const int krow = 8 * p + (lane >> 3);
        const int ch = (lane & 7) ^ ds_swz(krow);
        off[e] = im < PI ? ((unsigned)krow * (unsigned)ldp + (unsigned)(im * 64 + 8 * ch)) * 2u
                         : ((unsigned)krow * (unsigned)ldq + (unsigned)(u.q0 + (im - PI) * 64 + 8 * ch)) * 2u;
    }
    auto issue = [&](int t) {                                       // k-tile t of this range (clamped: see the loop)
        const int kt = kt0 + min(t, nt - 1);
        const char *pb, *qb;
        if (kt < kt_seg) {
            pb = reinterpret_cast<const char *>(P1 + (size_t)kt * 64 * ldp);
            qb = reinterpret_cast<const char *>(Q1 + (size_t)kt * 64 * ldq);
        } else {
            pb = reinterpret_cast<const char *>(P2 + (size_t)(kt - kt_seg) * 64 * ldp);
            qb = reinterpret_cast<const char *>(Q2 + (size_t)(kt - kt_seg) * 64 * ldq);
        }
        char *dst = smem + (t % DS_NSTAGE) * STAGE;
#pragma unroll
        for (int e = 0; e < NPW; ++e) {
            const int q = wave + 4 * e;                             // wave-uniform
            // (inline-asm DMA: with the builtin the compiler drains vmcnt before every transposing read below -- the ring
            // never had more than the current stage in flight; sei_common.h)
            dma16_base((q >> 3) < PI ? pb : qb, off[e], dst + q * 1024);
        }
    };

    // ---- fragments: 8 rows (32 ks + 8 lg + j) of column l16 of a 16-column block, two transposing reads
    const int tq = l16 >> 2, tp = l16 & 3;
    const int rm_lane = 128 * (8 * lg + tq) + 16 * ((tp >> 1) ^ ds_swz(8 * lg + tq)) + 8 * (tp & 1);
    auto frag = [&](const char *img, int blk, int ks) -> bf16x8 {
        const int base = rm_lane ^ (32 * blk);
        const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) v4s *)(img + base + 128 * 32 * ks));
        const v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) v4s *)(img + base + 128 * (32 * ks + 4)));
        return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    };

    // acc[a][b]: a walks the MFMA's A operand (P, or Q when SWAP), b its B operand
    constexpr int AB = SWAP ? QB : PB, BB = SWAP ? PB : QB;
    f32x4 acc[AB][BB];
#pragma unroll
    for (int a = 0; a < AB; ++a)
#pragma unroll
        for (int b = 0; b < BB; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool works = !PAIR || wr == wc;                           // PAIR: the diagonal quadrants only (wave-uniform)
    // Bias gradient = the column sums of gy = gy^T 1: one more MFMA per A fragment against a fragment of ones (gy is the
    // MFMA's A operand in both orientations). Of the waves that hold the same gy tiles only one keeps the sums.
    const bool sums = u.bias != nullptr && works && (PAIR || (SWAP ? wr == 0 : wc == 0));
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;
    f32x4 accb[AB];
#pragma unroll
    for (int a = 0; a < AB; ++a) accb[a] = f32x4{0.f, 0.f, 0.f, 0.f};

    // Every iteration issues one stage (past the end: the last k-tile again, into a stage nobody reads), so the counted
    // wait below always leaves exactly the NPW pieces of the next stage in flight.
    issue(0);
    issue(1);
    for (int t = 0; t < nt; ++t) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPW) : "memory");  // this wave's pieces of stage t have landed
        __builtin_amdgcn_s_barrier();                               // ... everyone's; and stage t - 1 is read out
        issue(t + 2);
        if (!works) continue;
        const char *st = smem + (t % DS_NSTAGE) * STAGE;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 fp[PB], fq[QB];
#pragma unroll
            for (int i = 0; i < PB; ++i) {
                const int bi = PB * wr + i;
                fp[i] = frag(st + (bi >> 2) * DS_IMG, bi & 3, ks);
            }
#pragma unroll
            for (int j = 0; j < QB; ++j) {
                const int bj = QB * wc + j;
                fq[j] = frag(st + (PI + (bj >> 2)) * DS_IMG, bj & 3, ks);
            }
#pragma unroll
            for (int a = 0; a < AB; ++a)
#pragma unroll
                for (int b = 0; b < BB; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(SWAP ? fq[a] : fp[a], SWAP ? fp[b] : fq[b],
                                                                        acc[a][b], 0, 0, 0);
            if (sums) {                                             // wave-uniform
#pragma unroll
                for (int a = 0; a < AB; ++a)
                    accb[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(SWAP ? fq[a] : fp[a], ones, accb[a], 0, 0, 0);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // the two clamped stages still in flight
    if (!works) return;

    // ---- float atomics: accumulator element (row 16 a + 4 lg + r, column 16 b + l16). Index of the wave's share inside
    // the block: P rows from 32 PI wr, Q columns from 32 QI wc -- PAIR: both diagonal quadrants land on the SAME gradient
    // elements (pixel 2r's and pixel 2r + 1's contributions), hence offset 0.
    const int p_base = PAIR ? 0 : 32 * PI * wr, q_base = PAIR ? 0 : 32 * QI * wc;
    // gradient element of (p, q): not SWAP: D[p][q] (P = gy: rows); SWAP: D[q][p] (Q = gy)
    float *d = SWAP ? u.D + (size_t)(q_base + 4 * lg) * u.ldd + p_base + l16
                    : u.D + (size_t)(p_base + 4 * lg) * u.ldd + q_base + l16;
#pragma unroll
    for (int a = 0; a < AB; ++a)
#pragma unroll
        for (int b = 0; b < BB; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) atomicAdd(d + (size_t)(16 * a + r) * u.ldd + 16 * b, acc[a][b][r]);
    if (sums && l16 == 0) {                                         // every column of accb holds the same sums
        float *bp = u.bias + (SWAP ? q_base : p_base) + 4 * lg;
#pragma unroll
        for (int a = 0; a < AB; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r) atomicAdd(bp + 16 * a + r, accb[a][r]);
    }
}

// Every unit of a job table in ONE launch, whatever its block shape: the workgroups of a launch are dealt to ALL units, so
// a unit's pixel range per workgroup grows -- and its partial blocks (the float atomics at the end) shrink -- with the
// number of units, and the atomics of one workgroup run under the streaming of the others.
__global__ __launch_bounds__(DS_NT) void dw_stream_kernel(DwArgs g) {
    __shared__ __attribute__((aligned(1024))) char smem[DS_NSTAGE * 6 * DS_IMG];
    int ui = 0;
    for (int k = 1; k < g.nunits; ++k)
        if ((int)blockIdx.x >= g.u[k].first) ui = k;               // uniform: scalar compares on the argument block
    const DwUnit &u = g.u[ui];
    const int uend = ui + 1 < g.nunits ? g.u[ui + 1].first : g.first_end;
    const int worker = (int)blockIdx.x - u.first, workers = uend - u.first;
    switch (u.kind) {                                              // block-uniform
        case 1: dw_stream_body<2, 4, false, false>(u, worker, workers, smem); break;
        case 2: dw_stream_body<2, 4, true, false>(u, worker, workers, smem); break;
        case 3: dw_stream_body<1, 4, false, true>(u, worker, workers, smem); break;
        default: dw_stream_body<1, 4, true, true>(u, worker, workers, smem); break;
    }
}

struct DwPlan {                    // host-side: the units of one launch
    DwArgs args;
    double bytes[DS_MAX_UNITS];
    int n;
};

int dw_launch(DwPlan &pl, hipStream_t s) {
    if (pl.n == 0) return SEI_OK;
    // workgroups (one per CU: the ring is 120-144 KB) dealt to the units in proportion to the bytes they stream; a unit
    // gets at least one and never more than it has k-tiles
    double total = 0;
    for (int i = 0; i < pl.n; ++i) total += pl.bytes[i];
    const int budget = pl.n > 256 ? pl.n : 256;
    int first = 0;
    for (int i = 0; i < pl.n; ++i) {
        int w = (int)(budget * pl.bytes[i] / total + 0.5);
        if (w < 1) w = 1;
        if (w > pl.args.u[i].kt_total) w = pl.args.u[i].kt_total;
        pl.args.u[i].first = first;
        first += w;
    }
    pl.args.first_end = first;
    pl.args.nunits = pl.n;
    for (int i = pl.n; i < DS_MAX_UNITS; ++i) pl.args.u[i] = pl.args.u[0];
    hipLaunchKernelGGL(dw_stream_kernel, dim3((unsigned)first), dim3(DS_NT), 0, s, pl.args);
    pl.n = 0;
    return sei_launch_status();
}

// 0 = not built; 1 = WIDE, 2 = WIDE + SWAP, 3 = PAIR, 4 = PAIR + SWAP
int dw_kind(int Mo, int Ni, int ldy, int ldx, long long K1, long long K2) {
    if (Mo <= 0 || Ni <= 0 || ldy != Mo || ldx != Ni || K1 <= 0 || K2 < 0) return 0;
    const int narrow = Mo < Ni ? Mo : Ni, wide = Mo < Ni ? Ni : Mo;
    const bool swap = Ni < Mo;
    if (narrow == 128 && wide % 256 == 0 && wide <= 1024) {
        if (K1 % 64 || K2 % 64 || (K1 + K2) / 64 >= (1ll << 24)) return 0;
        return swap ? 2 : 1;
    }
    if (narrow == 32 && wide == 128) {
        if (K1 % 128 || K2 % 128 || (K1 + K2) / 128 >= (1ll << 24)) return 0;
        return swap ? 4 : 3;
    }
    return 0;
}

}  // namespace

extern "C" size_t sei_dwstream_bf16_eligible(int Mo, int Ni, int ldy, int ldx, long long K1, long long K2) {
    return (size_t)dw_kind(Mo, Ni, ldy, ldx, K1, K2);
}

extern "C" int sei_dwstream_bf16_jobs(const SeiDwStreamJob *jobs, int njobs, void *stream) {
    SEI_REQUIRE(jobs && njobs >= 1 && njobs <= SEI_DWSTREAM_MAX_JOBS);
    hipStream_t s = (hipStream_t)stream;
    DwPlan pl;
    pl.n = 0;
    for (int i = 0; i < njobs; ++i) {
        const SeiDwStreamJob &j = jobs[i];
        SEI_REQUIRE(j.Y1 && j.X1 && j.D && (j.K2 == 0 || (j.Y2 && j.X2)) && j.ldd >= j.Ni);
        SEI_REQUIRE((((uintptr_t)j.Y1 | (uintptr_t)j.X1 | (uintptr_t)j.Y2 | (uintptr_t)j.X2) & 15) == 0);
        const int kind = dw_kind(j.Mo, j.Ni, j.ldy, j.ldx, j.K1, j.K2);
        SEI_REQUIRE(kind != 0);
        const bool swap = kind == 2 || kind == 4, pair = kind >= 3;
        const int rows = pair ? 128 : 64;                            // pixels per k-tile
        const int nblk = pair ? 1 : (swap ? j.Mo : j.Ni) / 256;
        for (int b = 0; b < nblk; ++b) {
            if (pl.n == DS_MAX_UNITS) {                              // table full: launch what is there
                const int rc = dw_launch(pl, s);
                if (rc != SEI_OK) return rc;
            }
            DwUnit &u = pl.args.u[pl.n];
            u.kind = kind;
            // P = the operand with the narrower rows: gy unless SWAP
            u.P1 = swap ? j.X1 : j.Y1;
            u.P2 = j.K2 ? (swap ? j.X2 : j.Y2) : u.P1;
            u.Q1 = swap ? j.Y1 : j.X1;
            u.Q2 = j.K2 ? (swap ? j.Y2 : j.X2) : u.Q1;
            const int np = swap ? j.Ni : j.Mo, nq = swap ? j.Mo : j.Ni;
            u.ldp = pair ? 2 * np : np;                              // PAIR: a row is two pixels
            u.ldq = pair ? 2 * nq : nq;
            u.q0 = 256 * b;
            // the block's corner: not SWAP: columns q0 of D; SWAP: rows q0 of D
            u.D = swap ? j.D + (size_t)(256 * b) * j.ldd : j.D + 256 * b;
            // gy's column sums: SWAP: gy is the wide operand, every block owns 256 of its columns; else the narrow one,
            // which every block of the job reads whole: the first block alone adds
            u.bias = !j.gbias ? nullptr : (swap ? j.gbias + 256 * b : (b == 0 ? j.gbias : nullptr));
            u.ldd = j.ldd;
            u.kt_seg = (int)(j.K1 / rows);
            u.kt_total = (int)((j.K1 + j.K2) / rows);
            pl.bytes[pl.n] = (double)(j.K1 + j.K2) * 2.0 * (np + (pair ? nq : 256));
            ++pl.n;
        }
    }
    return dw_launch(pl, s);
}
