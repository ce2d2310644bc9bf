// SwinIR building blocks for gfx950 that the U-Net kernels do not already provide
// (reference: deepinv.models.SwinIR as configured at src/models/__init__.py:51-74 = the official SwinIR
// network_swinir.py; restated from the published architecture, see oracle/swinir_path.py).
//
//   sei_swin_attn_fwd / _bwd   8x8-window multi-head self-attention (WindowAttention.forward) on the qkv
//                              projections of ALL tokens in natural (b, y, x) order: the cyclic shift
//                              (torch.roll), window_partition / window_reverse, the relative-position bias
//                              lookup and the shift mask of SwinTransformerBlock.forward are index arithmetic
//                              inside the kernel, so none of those tensors is ever materialised. The backward
//                              recomputes the 64x64 probabilities from q, k (nothing but qkv is saved).
//                              This file holds the exact-f32 form (the parity mode); the bf16 MFMA form lives
//                              in swin_attn_mfma.hip.
//   sei_pad_nhwc / sei_unpad_nhwc   zero-bordered copy of an NHWC image batch (+ guard rows) and its inverse:
//                              on that padded grid a 3x3 convolution is nine row-shifted GEMMs over the SAME
//                              flat array (models/_swin_ops.py), no im2col buffer.
//   sei_rowscale               y[m, :] = s[m] x[m, :]  (stochastic depth in the backward pass)
#include "sei_common.h"

namespace {

constexpr int WS = 8, NTOK = 64;           // window side, tokens per window

struct WinGeom {
    int H, W, nwy, nwx, shift, heads, C;   // C = heads * HD
};

// token index (natural order) and mask region of window-local position i of window `win`
__device__ __forceinline__ void win_token(const WinGeom &g, int win, int i, int &tok, int &region) {
    const int per = g.nwy * g.nwx;
    const int b = win / per, w = win - b * per;
    const int wy = w / g.nwx, wx = w - wy * g.nwx;
    const int sy = wy * WS + (i >> 3), sx = wx * WS + (i & 7);          // coordinates in the shifted frame
    int oy = sy + g.shift, ox = sx + g.shift;                            // shifted = roll(x, -shift)
    if (oy >= g.H) oy -= g.H;
    if (ox >= g.W) ox -= g.W;
    tok = (b * g.H + oy) * g.W + ox;
    region = 0;
    if (g.shift > 0) {                                                   // calculate_mask: 3 x 3 regions
        const int ry = sy < g.H - WS ? 0 : (sy < g.H - g.shift ? 1 : 2);
        const int rx = sx < g.W - WS ? 0 : (sx < g.W - g.shift ? 1 : 2);
        region = 3 * ry + rx;
    }
}

__device__ __forceinline__ int bias_bin(int i, int j) {                 // relative_position_index[i][j]
    return ((i >> 3) - (j >> 3) + WS - 1) * (2 * WS - 1) + ((i & 7) - (j & 7) + WS - 1);
}

// One wave per (window, head); thread i = query i. K and V of the window in LDS, the score row in registers.
template <int HD>
__global__ __launch_bounds__(64) void swin_attn_fwd_f32_kernel(const float *__restrict__ qkv,
                                                                const float *__restrict__ table,
                                                                float *__restrict__ out, WinGeom g, float scale) {
    __shared__ float Ks[NTOK][HD + 1], Vs[NTOK][HD + 1];
    __shared__ int regs[NTOK];
    const int win = blockIdx.x, h = blockIdx.y, i = threadIdx.x;
    int tok, region;
    win_token(g, win, i, tok, region);
    const float *row = qkv + (size_t)tok * 3 * g.C + h * HD;
    float q[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) {
        q[d] = row[d] * scale;
        Ks[i][d] = row[g.C + d];
        Vs[i][d] = row[2 * g.C + d];
    }
    regs[i] = region;
    __syncthreads();
    float s[NTOK];
    float mx = -3.0e38f;
#pragma unroll
    for (int j = 0; j < NTOK; ++j) {
        float a = 0.f;
#pragma unroll
        for (int d = 0; d < HD; ++d) a = fmaf(q[d], Ks[j][d], a);
        a += table[bias_bin(i, j) * g.heads + h];
        if (regs[j] != region) a += -100.0f;
        s[j] = a;
        mx = fmaxf(mx, a);
    }
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < NTOK; ++j) {
        s[j] = __expf(s[j] - mx);
        sum += s[j];
    }
    const float inv = 1.0f / sum;
    float o[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) o[d] = 0.f;
#pragma unroll
    for (int j = 0; j < NTOK; ++j) {
        const float p = s[j] * inv;
#pragma unroll
        for (int d = 0; d < HD; ++d) o[d] = fmaf(p, Vs[j][d], o[d]);
    }
    float *orow = out + (size_t)tok * g.C + h * HD;
#pragma unroll
    for (int d = 0; d < HD; ++d) orow[d] = o[d];
}

// Backward of the same: dqkv (every slot written exactly once) and the bias-table gradient (LDS bins per
// workgroup, then one global float atomic per bin).
template <int HD>
__global__ __launch_bounds__(64) void swin_attn_bwd_f32_kernel(const float *__restrict__ qkv,
                                                                const float *__restrict__ table,
                                                                const float *__restrict__ dout,
                                                                float *__restrict__ dqkv, float *__restrict__ dtable,
                                                                WinGeom g, float scale) {
    constexpr int NB = (2 * WS - 1) * (2 * WS - 1);
    __shared__ float Ks[NTOK][HD + 1], Vs[NTOK][HD + 1], Qs[NTOK][HD + 1], Gs[NTOK][HD + 1];
    __shared__ float Pm[NTOK][NTOK + 1];
    __shared__ float bins[NB];
    __shared__ int regs[NTOK];
    const int win = blockIdx.x, h = blockIdx.y, i = threadIdx.x;
    int tok, region;
    win_token(g, win, i, tok, region);
    const float *row = qkv + (size_t)tok * 3 * g.C + h * HD;
    const float *grow = dout + (size_t)tok * g.C + h * HD;
    float q[HD], go[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) {
        q[d] = row[d] * scale;
        go[d] = grow[d];
        Ks[i][d] = row[g.C + d];
        Vs[i][d] = row[2 * g.C + d];
        Qs[i][d] = q[d];
        Gs[i][d] = go[d];
    }
    regs[i] = region;
    for (int b = i; b < NB; b += NTOK) bins[b] = 0.f;
    __syncthreads();
    float s[NTOK];
    float mx = -3.0e38f;
#pragma unroll
    for (int j = 0; j < NTOK; ++j) {
        float a = 0.f;
#pragma unroll
        for (int d = 0; d < HD; ++d) a = fmaf(q[d], Ks[j][d], a);
        a += table[bias_bin(i, j) * g.heads + h];
        if (regs[j] != region) a += -100.0f;
        s[j] = a;
        mx = fmaxf(mx, a);
    }
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < NTOK; ++j) {
        s[j] = __expf(s[j] - mx);
        sum += s[j];
    }
    const float inv = 1.0f / sum;
    // dP_j = dO_i . V_j ; delta = sum_j P_j dP_j ; dS_j = P_j (dP_j - delta)
    float dp[NTOK];
    float delta = 0.f;
#pragma unroll
    for (int j = 0; j < NTOK; ++j) {
        s[j] *= inv;
        float a = 0.f;
#pragma unroll
        for (int d = 0; d < HD; ++d) a = fmaf(go[d], Vs[j][d], a);
        dp[j] = a;
        delta = fmaf(s[j], a, delta);
        Pm[i][j] = s[j];
    }
    __syncthreads();
    // dV_j = sum_i P_ij dO_i   (thread index now plays j)
    float acc[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) acc[d] = 0.f;
    for (int ii = 0; ii < NTOK; ++ii) {
        const float p = Pm[ii][i];
#pragma unroll
        for (int d = 0; d < HD; ++d) acc[d] = fmaf(p, Gs[ii][d], acc[d]);
    }
    float *drow = dqkv + (size_t)tok * 3 * g.C + h * HD;
#pragma unroll
    for (int d = 0; d < HD; ++d) drow[2 * g.C + d] = acc[d];
    __syncthreads();
    // dS into the same LDS matrix; dQ_i = scale * sum_j dS_ij K_j ; bias-table bins
    float dq[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) dq[d] = 0.f;
#pragma unroll
    for (int j = 0; j < NTOK; ++j) {
        const float ds = s[j] * (dp[j] - delta);
        Pm[i][j] = ds;
        atomicAdd(&bins[bias_bin(i, j)], ds);
#pragma unroll
        for (int d = 0; d < HD; ++d) dq[d] = fmaf(ds, Ks[j][d], dq[d]);
    }
#pragma unroll
    for (int d = 0; d < HD; ++d) drow[d] = dq[d] * scale;
    __syncthreads();
    // dK_j = sum_i dS_ij (scale q_i)
#pragma unroll
    for (int d = 0; d < HD; ++d) acc[d] = 0.f;
    for (int ii = 0; ii < NTOK; ++ii) {
        const float ds = Pm[ii][i];
#pragma unroll
        for (int d = 0; d < HD; ++d) acc[d] = fmaf(ds, Qs[ii][d], acc[d]);
    }
#pragma unroll
    for (int d = 0; d < HD; ++d) drow[g.C + d] = acc[d];
    for (int b = i; b < NB; b += NTOK) atomicAdd(dtable + b * g.heads + h, bins[b]);
}

// padded-grid copy: dst (B, H+2, W+2, C) with a zero border, preceded and followed by `guard` zero rows of C
__global__ __launch_bounds__(256) void pad_nhwc_kernel(const float *__restrict__ src, float *__restrict__ dst, int B,
                                                        int H, int W, int C, int guard) {
    const int Hp = H + 2, Wp = W + 2;
    const size_t rows = (size_t)B * Hp * Wp + 2 * (size_t)guard;
    const int c4 = C / 4;
    const size_t total = rows * c4;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const size_t r = e / c4;
        const int c = (int)(e - r * c4);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r >= (size_t)guard && r < rows - guard) {
            const size_t p = r - guard;
            const int x = (int)(p % Wp), y = (int)((p / Wp) % Hp);
            const size_t b = p / ((size_t)Wp * Hp);
            if (x >= 1 && x <= W && y >= 1 && y <= H)
                v = reinterpret_cast<const float4 *>(src)[((b * H + (y - 1)) * W + (x - 1)) * c4 + c];
        }
        reinterpret_cast<float4 *>(dst)[e] = v;
    }
}

// interior of a padded-grid tensor back to (B, H, W, C), optionally plus a residual; act = 1: LeakyReLU(0.01)
__global__ __launch_bounds__(256) void unpad_nhwc_kernel(const float *__restrict__ src, const float *__restrict__ res,
                                                          float *__restrict__ dst, int B, int H, int W, int C, int act) {
    const int Hp = H + 2, Wp = W + 2, c4 = C / 4;
    const size_t total = (size_t)B * H * W * c4;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const size_t p = e / c4;
        const int c = (int)(e - p * c4);
        const int x = (int)(p % W), y = (int)((p / W) % H);
        const size_t b = p / ((size_t)W * H);
        float4 v = reinterpret_cast<const float4 *>(src)[((b * Hp + (y + 1)) * Wp + (x + 1)) * c4 + c];
        if (act == 1) {
            v.x = v.x > 0.f ? v.x : 0.01f * v.x;
            v.y = v.y > 0.f ? v.y : 0.01f * v.y;
            v.z = v.z > 0.f ? v.z : 0.01f * v.z;
            v.w = v.w > 0.f ? v.w : 0.01f * v.w;
        }
        if (res) {
            const float4 r = reinterpret_cast<const float4 *>(res)[e];
            v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
        }
        reinterpret_cast<float4 *>(dst)[e] = v;
    }
}

// y[m, n] = s[m] * x[m, n] * (gate ? (gate[m, n] > 0 ? 1 : 0.01) : 1)      (N % 4 == 0)
__global__ __launch_bounds__(256) void rowscale_kernel(const float *__restrict__ x, const float *__restrict__ s,
                                                        const float *__restrict__ gate, float *__restrict__ y,
                                                        size_t M, int N) {
    const int n4 = N / 4;
    const size_t total = M * n4;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const float f = s ? s[e / n4] : 1.0f;
        float4 v = reinterpret_cast<const float4 *>(x)[e];
        v.x *= f; v.y *= f; v.z *= f; v.w *= f;
        if (gate) {
            const float4 t = reinterpret_cast<const float4 *>(gate)[e];
            v.x *= t.x > 0.f ? 1.0f : 0.01f;
            v.y *= t.y > 0.f ? 1.0f : 0.01f;
            v.z *= t.z > 0.f ? 1.0f : 0.01f;
            v.w *= t.w > 0.f ? 1.0f : 0.01f;
        }
        reinterpret_cast<float4 *>(y)[e] = v;
    }
}

inline unsigned stream_grid(size_t items) {
    size_t g = sei_ceil_div(items, 256);
    return (unsigned)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

inline int check_geom(int B, int H, int W, int heads, int hd, int shift) {
    SEI_REQUIRE(B > 0 && H >= WS && W >= WS && H % WS == 0 && W % WS == 0 && heads > 0);
    SEI_REQUIRE(hd == 30 || hd == 32 || hd == 16 || hd == 8);
    SEI_REQUIRE(shift >= 0 && shift < WS);
    SEI_REQUIRE((size_t)B * H * W < ((size_t)1 << 31));
    return SEI_OK;
}

}  // namespace

extern "C" int sei_swin_attn_fwd(const float *qkv, const float *table, float *out, int B, int H, int W, int heads,
                                 int head_dim, int shift, float scale, void *stream) {
    SEI_REQUIRE(qkv && table && out && qkv != out);
    if (int rc = check_geom(B, H, W, heads, head_dim, shift)) return rc;
    WinGeom g{H, W, H / WS, W / WS, shift, heads, heads * head_dim};
    const dim3 grid((unsigned)(B * g.nwy * g.nwx), (unsigned)heads);
    hipStream_t s = (hipStream_t)stream;
    switch (head_dim) {
        case 30: hipLaunchKernelGGL(swin_attn_fwd_f32_kernel<30>, grid, dim3(64), 0, s, qkv, table, out, g, scale); break;
        case 32: hipLaunchKernelGGL(swin_attn_fwd_f32_kernel<32>, grid, dim3(64), 0, s, qkv, table, out, g, scale); break;
        case 16: hipLaunchKernelGGL(swin_attn_fwd_f32_kernel<16>, grid, dim3(64), 0, s, qkv, table, out, g, scale); break;
        default: hipLaunchKernelGGL(swin_attn_fwd_f32_kernel<8>, grid, dim3(64), 0, s, qkv, table, out, g, scale); break;
    }
    return sei_launch_status();
}

extern "C" int sei_swin_attn_bwd(const float *qkv, const float *table, const float *dout, float *dqkv, float *dtable,
                                 int B, int H, int W, int heads, int head_dim, int shift, float scale, void *stream) {
    SEI_REQUIRE(qkv && table && dout && dqkv && dtable);
    if (int rc = check_geom(B, H, W, heads, head_dim, shift)) return rc;
    WinGeom g{H, W, H / WS, W / WS, shift, heads, heads * head_dim};
    const dim3 grid((unsigned)(B * g.nwy * g.nwx), (unsigned)heads);
    hipStream_t s = (hipStream_t)stream;
    switch (head_dim) {
        case 30: hipLaunchKernelGGL(swin_attn_bwd_f32_kernel<30>, grid, dim3(64), 0, s, qkv, table, dout, dqkv, dtable, g, scale); break;
        case 32: hipLaunchKernelGGL(swin_attn_bwd_f32_kernel<32>, grid, dim3(64), 0, s, qkv, table, dout, dqkv, dtable, g, scale); break;
        case 16: hipLaunchKernelGGL(swin_attn_bwd_f32_kernel<16>, grid, dim3(64), 0, s, qkv, table, dout, dqkv, dtable, g, scale); break;
        default: hipLaunchKernelGGL(swin_attn_bwd_f32_kernel<8>, grid, dim3(64), 0, s, qkv, table, dout, dqkv, dtable, g, scale); break;
    }
    return sei_launch_status();
}

extern "C" int sei_pad_nhwc(const float *x, float *xp, int B, int H, int W, int C, int guard_rows, void *stream) {
    SEI_REQUIRE(x && xp && x != xp && B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && guard_rows >= 0);
    SEI_REQUIRE((((uintptr_t)x | (uintptr_t)xp) & 15) == 0);
    const size_t items = ((size_t)B * (H + 2) * (W + 2) + 2 * (size_t)guard_rows) * (C / 4);
    hipLaunchKernelGGL(pad_nhwc_kernel, dim3(stream_grid(items)), dim3(256), 0, (hipStream_t)stream, x, xp, B, H, W, C,
                       guard_rows);
    return sei_launch_status();
}

extern "C" int sei_unpad_nhwc(const float *xp, const float *res, float *y, int B, int H, int W, int C, int act,
                              void *stream) {
    SEI_REQUIRE(xp && y && xp != y && B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && (act == 0 || act == 1));
    SEI_REQUIRE((((uintptr_t)xp | (uintptr_t)y | (uintptr_t)res) & 15) == 0);
    const size_t items = (size_t)B * H * W * (C / 4);
    hipLaunchKernelGGL(unpad_nhwc_kernel, dim3(stream_grid(items)), dim3(256), 0, (hipStream_t)stream, xp, res, y, B, H,
                       W, C, act);
    return sei_launch_status();
}

extern "C" int sei_rowscale(const float *x, const float *row_scale, const float *leaky_gate, float *y, size_t M, int N,
                            void *stream) {
    SEI_REQUIRE(x && y && M > 0 && N > 0 && N % 4 == 0 && (row_scale || leaky_gate));
    SEI_REQUIRE((((uintptr_t)x | (uintptr_t)y | (uintptr_t)leaky_gate) & 15) == 0);
    hipLaunchKernelGGL(rowscale_kernel, dim3(stream_grid(M * (N / 4))), dim3(256), 0, (hipStream_t)stream, x, row_scale,
                       leaky_gate, y, M, N);
    return sei_launch_status();
}
