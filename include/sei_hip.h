/*
 * sei_hip.h -- C ABI of libsei_hip.so: the MI355X (gfx950) kernels behind the proposed-loss
 * training hot path of Scale-Equivariant-Imaging.
 *
 * Conventions (every entry point):
 *   - extern "C", returns int: 0 = ok, otherwise a hipError_t value (launch errors) or
 *     SEI_ERR_* (argument errors detected on the host BEFORE anything is launched). Never throws.
 *   - all tensor pointers are DEVICE pointers owned by the caller; nothing is allocated inside.
 *   - `stream` is a hipStream_t (passed as void* so that C callers need no HIP headers); all work
 *     is enqueued on it and the call returns without synchronising. No hidden globals: re-entrant.
 *   - float32 storage unless the name says bf16 (bf16 = upper 16 bits of an IEEE float32, passed as
 *     uint16_t*), float32 accumulation everywhere.
 *   - image tensors at the physics/loss boundary are NCHW planar ("planes" = B*C images of H x W);
 *     U-Net activations are NHWC ("rows" = B*H*W pixels of C contiguous channels).
 *
 * Each group cites the reference code (paths under the reference repo) whose arithmetic it replaces.
 * The reference is pure Python; it has no FFI of its own -- INTEGRATION.md shows the ctypes
 * binding a maintainer would add at each call site.
 */
#ifndef SEI_HIP_H
#define SEI_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SEI_OK 0
#define SEI_ERR_BAD_ARG 10001      /* NULL pointer, non-positive size, unsupported size */
#define SEI_ERR_TOO_LARGE 10002    /* a tile would not fit in LDS / a tap count above the limit */

/* ---------------------------------------------------------------------------------------------
 * STABLE SURFACE -- what a maintainer binding the reference's own modules needs (INTEGRATION.md section B shows the
 * ctypes stub for each at its reference call site); names, argument meaning and error behaviour are kept across ABI
 * versions:
 *   sei_abi_version, sei_build_target
 *   physics      sei_blur_sep_circ, sei_blur_dense_circ, sei_resample_sepband
 *   EI transform sei_scale_resample_fwd, sei_scale_resample_bwd
 *   loss terms   sei_axpy, sei_sure_terms, sei_mse_terms
 *   U-Net (f32)  sei_conv3x3_fwd, sei_conv3x3_bwd_weight, sei_dwconv7_fwd, sei_dwconv7_bwd_weight (+ _workspace),
 *                sei_ln_fwd, sei_ln_bwd (+ sei_ln_bwd_workspace), sei_gemm_f32, sei_colsum_f32, sei_sepmap2
 *   U-Net (bf16) sei_cast_bf16, sei_ln_fwd_bf16, sei_gemm_bf16nt
 *   optimizer    sei_adam_fused
 * INTERNAL -- everything else in this header: fused, schedule-specific (_ex, _ws, _dw2*, _plan, _eligible, _parts,
 * _count ...), SwinIR and measurement entry points that this build's own host layer (models/_ops.py, graphs.py, optim.py,
 * bench.py) calls. They are exported and documented here because that host layer sits above the C ABI, but they follow the
 * kernels: they may change with SEI_ABI_VERSION.
 * --------------------------------------------------------------------------------------------- */

/* ABI version of this header; sei_abi_version() returns the value the library was built with. */
#define SEI_ABI_VERSION 12
int sei_abi_version(void);
/* Fills name[0..n) with the gfx target the code objects were built for ("gfx950"). */
int sei_build_target(char *name, int n);

/* Events that cross a hipGraph boundary: sei_event_record_external() issued on a capturing stream becomes an
 * event-record node of the graph (hipEventRecordExternal), recorded anew at that point of every replay;
 * sei_stream_wait_event() on another (non-captured) stream, issued after the replay was enqueued, waits for it.
 * `stream` arguments are hipStream_t; `event` is an opaque handle from sei_event_create. */
int sei_event_create(void **event);
int sei_event_destroy(void *event);
int sei_event_record_external(void *event, void *stream);
int sei_stream_wait_event(void *stream, void *event);

/* Measurement aid (bench.py): node counts of a captured hipGraph_t -- kernel nodes and all nodes. */
int sei_graph_node_counts(void *graph, int *kernel_nodes, int *all_nodes);

/* ---------------------------------------------------------------------------------------------
 * Physics: circular blur.
 * Replaces BlurV2.A, src/physics/blur/__init__.py:205-223 (rfft2 * OTF -> irfft2 == circular
 * convolution) and, with transpose=1, its autograd backward / A_adjoint (:225-227) == circular
 * correlation. Also serves the legacy Blur/conv/conv_transpose, :9-194 (same arithmetic).
 *   y[p,i,j] = sum_{a,b} tv[a]*th[b] * x[p, (i-a+kv/2) mod H, (j-b+kh/2) mod W]     (transpose=0)
 *   y[p,i,j] = sum_{a,b} tv[a]*th[b] * x[p, (i+a-kv/2) mod H, (j+b-kh/2) mod W]     (transpose=1)
 * tv (kv taps) and th (kh taps) are the rank-1 factors of the kernel (device pointers).
 * kv, kh <= 63.
 * ------------------------------------------------------------------------------------------- */
int sei_blur_sep_circ(const float *x, float *y, const float *tv, const float *th, int kv, int kh,
                      int planes, int H, int W, int transpose, void *stream);
/* Non-separable kernels (loaded from a file, src/physics/__init__.py:20-22): k is (kv,kh) row-major. */
int sei_blur_dense_circ(const float *x, float *y, const float *k, int kv, int kh, int planes,
                        int H, int W, int transpose, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Physics: separable banded resampling  y[p] = Wv * x[p] * Wh^T.
 * Replaces Downsampling.A, src/physics/downsampling/__init__.py:16-19
 * (F.interpolate bicubic antialias=True), its autograd backward / true adjoint (:21-31, the
 * transposed band matrices), the deprecated adjoint (:33-34, plain bicubic upsample), and the
 * antialias pre-filter of the EI transform, src/transforms.py:46-57.
 * Band form per axis: row o of W has nb consecutive weights w[o*nb .. o*nb+nb) starting at input
 * index lo[o] (entries beyond the band are zero-padded; lo is non-decreasing; lo[o]+nb may exceed
 * the input size only on zero weights). stepv/steph = max_o (lo[o+1]-lo[o]) of each axis (the
 * caller built the bands on the host and knows it); it bounds the LDS footprint of a tile.
 * ------------------------------------------------------------------------------------------- */
int sei_resample_sepband(const float *x, float *y, int planes, int Hi, int Wi, int Ho, int Wo,
                         const float *wv, const int *lov, int nbv, int stepv,
                         const float *wh, const int *loh, int nbh, int steph, void *stream);

/* ---------------------------------------------------------------------------------------------
 * EI transform: per-image zoom-out by bicubic grid sampling with reflection.
 * Replaces padded_downsampling_transform + get_downsampling_grid, src/transforms.py:27-43,60-83
 * (F.grid_sample bicubic / reflection / align_corners=True on the grid (g - c)/rate + c); the grid
 * is generated in-kernel, never materialised. rate: (B,), center: (B,2) = (cx, cy) per image.
 * x is (B,C,Hi,Wi), y is (B,C,H,W); (Hi,Wi) = (H,W) except for --ScalingTransform__antialias, where the
 * reference samples the pre-shrunk image on the grid of the original shape (transforms.py:63-76).
 * _bwd accumulates d x (must be zero-filled by the caller) -- only needed with
 * --no-ProposedLoss__stop_gradient.
 * ------------------------------------------------------------------------------------------- */
/* The scale transform's per-image parameters from its two uniform draws (sample_from / sample_downsampling_parameters,
 * /root/reference/src/transforms.py:5-24): rate[i] = table[floor(ntable * u[i])], center[2 i + k] = 2 v[2 i + k] - 1
 * (u (B), v (B, 2) uniform in [0, 1) from the caller's generator): the reference's float32 operations, one launch. */
int sei_scale_params(const float *u, const float *v, const float *table, int ntable, int B, float *rate, float *center,
                     void *stream);
int sei_scale_resample_fwd(const float *x, float *y, const float *rate, const float *center,
                           int B, int C, int Hi, int Wi, int H, int W, void *stream);
int sei_scale_resample_bwd(const float *gy, float *gx, const float *rate, const float *center,
                           int B, int C, int Hi, int Wi, int H, int W, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Nearest-neighbour rotation about the image centre: the `Rotations` option of the equivariant loss
 * (src/losses/__init__.py:84-91 -> deepinv.transform.Rotate -> torchvision.transforms.functional.rotate with
 * its defaults: NEAREST, no expand, zero fill).  x, y: (planes, H, W).  (t00 t01; t10 t11) is the linear part
 * of torchvision's inverse affine matrix for the angle, rounded to float32 by the caller:
 * (cos a, -sin a; sin a, cos a) with a = the angle in radians.  _bwd accumulates into a zero-filled gx.
 * ------------------------------------------------------------------------------------------- */
int sei_rotate_nearest_fwd(const float *x, float *y, int planes, int H, int W, float t00, float t01, float t10,
                           float t11, void *stream);
int sei_rotate_nearest_bwd(const float *gy, float *gx, int planes, int H, int W, float t00, float t01, float t10,
                           float t11, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Loss-side streaming kernels.
 * sei_axpy: out = a + alpha*b  (GaussianNoise y + sigma*n, deepinv; SURE probe y + tau*b,
 *           src/losses/sure.py:24).
 * sei_sure_terms: src/losses/sure.py:24-31,57-62. Over the interior [m:H-m, m:W-m] of every plane:
 *   out[0] = sum b*(y2-y1)/tau ,  out[1] = sum (y1-y)^2      (sums; the caller divides)
 *   and, fused, the gradients of  L = c_mse*out[1] + c_div*out[0]  w.r.t. y1 and y2:
 *   g1 = 2*c_mse*(y1-y) - c_div*b/tau ,  g2 = c_div*b/tau   (zero outside the interior).
 *   margin_div / margin_mse select the two interiors separately (cropped_div flag, sure.py:51-54).
 *   `work` must hold 2*SEI_REDUCE_BLOCKS floats.
 * sei_mse_terms: deepinv mse metric used by EILoss (src/losses/__init__.py:117-122):
 *   out[0] = sum (a-b)^2 ; ga = scale*(a-b) (caller passes scale = 2*alpha/N). `work` as above.
 * ------------------------------------------------------------------------------------------- */
#define SEI_REDUCE_BLOCKS 256
int sei_axpy(const float *a, const float *b, float alpha, float *out, size_t n, void *stream);
/* out[0:n] = a, out[n:2n] = a + alpha b (sei_axpy's arithmetic): the 2B-image input [y, y + tau b] of the fused SURE pass
 * (src/losses/sure.py:24 + the model call on y) in one launch instead of axpy + torch.cat. n % 4 == 0, 16-byte aligned. */
int sei_stack_axpy(const float *a, const float *b, float alpha, float *out, size_t n, void *stream);
/* The eager prologue of a proposed-loss step (csrc/draws.hip).
 * sei_proposed_draws: the step's device-side draws in ONE launch, on the random stream of the torch calls they replace
 * (uniform draws bit-identical, normal draws to <= 3 ulp: csrc/draws.hip) --
 * torch.randn(B, C, H - 2m, W - 2m) into the interior of b (src/losses/sure.py:13-22; the border of b is not touched),
 * torch.rand(B) -> rate = table[floor(ntable u)], torch.rand(B, 2) -> centre = 2 u - 1 (src/transforms.py:5-24),
 * torch.randn(B, C, H, W) -> noise (deepinv GaussianNoise inside EILoss) -- for torch's CUDA generator at (seed, offset):
 * element i of each tensor = component x of rocrand_normal4 / rocrand_uniform4 of Philox4x32-10 (seed, subsequence i,
 * offset + 4 k), k = 0..3 in the order above. The caller advances the generator's offset by 16. B C H W <=
 * sei_proposed_draws_max_numel() (torch's one-element-per-thread regime), offset % 4 == 0.
 * sei_crop_window: out (planes, S, S) = the S x S window of y (planes, H, W) at (i0, j0), zero where the window leaves y
 * (src/crop.py:26-57 applied to a batch: the offset is drawn over the zero-padded extent). */
int sei_proposed_draws(unsigned long long seed, unsigned long long offset, float *b, int B, int C, int H, int W, int margin,
                       const float *table, int ntable, float *rate, float *center, float *noise, void *stream);
size_t sei_proposed_draws_max_numel(void);
int sei_crop_window(const float *y, float *out, int planes, int H, int W, int i0, int j0, int S, void *stream);
/* Glue of the step that torch ops used to provide (one launch each, so that a captured step holds no ATen kernel):
 * sei_concat2_f32: out = [a | b] (na, nb % 4 == 0; nb = 0: a plain copy) -- the joint backward's 3B-row gradient, a model
 * input into its arena buffer; sei_scale_dev_f32: out = x * (*scalar), the scalar on the device -- a loss term's stored
 * gradient times the incoming gradient of its value; sei_add_scalars: *out = *a + *b -- the sum of two loss terms. */
int sei_concat2_f32(const float *a, size_t na, const float *b, size_t nb, float *out, void *stream);
int sei_scale_dev_f32(const float *x, const float *scalar, float *out, size_t n, void *stream);
int sei_add_scalars(const float *a, const float *b, float *out, void *stream);
/* base[off_k .. off_k + len_k) = 0 for up to 8 (off, len) pairs of elements (HOST array of 2 * count values): the gaps of
 * the flat gradient bucket between the weight gradients that the captured step stores, in ONE launch (optimizer.zero_grad,
 * demo/train.py:258). */
int sei_zero_ranges(float *base, const unsigned long long *off_len_pairs, int count, void *stream);
int sei_sure_terms(const float *y, const float *y1, const float *y2, const float *b, int planes,
                   int H, int W, int margin_div, int margin_mse, float tau, float c_mse,
                   float c_div, float *out2, float *g1, float *g2, float *work, void *stream);
int sei_mse_terms(const float *a, const float *b, size_t n, float scale, float *out1, float *ga,
                  float *work, void *stream);

/* The two entry points above with the LOSS VALUE formed on the device as well (no chain of 0-dim torch kernels behind them):
 * sei_sure_loss: out3[0] = divergence sum, out3[1] = squared-error sum, out3[2] = c_mse out3[1] + c_div out3[0] - cst
 * (src/losses/sure.py:57-66: mse + 2 sigma^2 div - sigma^2 / B); y1 / y2 and g1 / g2 may be the two halves of one 2B-image
 * tensor (ProposedLoss evaluates A on both model outputs at once). sei_mse_loss: out2[0] = sum (a - b)^2, out2[1] =
 * value_scale out2[0] (deepinv's mse metric times the EI weight, src/losses/__init__.py:117-122). */
int sei_sure_loss(const float *y, const float *y1, const float *y2, const float *b, int planes, int H, int W,
                  int margin_div, int margin_mse, float tau, float c_mse, float c_div, float cst, float *out3, float *g1,
                  float *g2, float *work, void *stream);
int sei_mse_loss(const float *a, const float *b, size_t n, float grad_scale, float value_scale, float *out2, float *ga,
                 float *work, void *stream);
/* Numerator of the luma PSNR of the evaluation step (reference src/metrics.py:10-13: kornia rgb_to_ycbcr's
 * Y = 0.299 R + 0.587 G + 0.114 B, torchmetrics PSNR with data_range 1): out1[0] = sum over the npix pixels of
 * (Y(a) - Y(b))^2 for planar RGB images a, b of shape (3, npix). `work` holds SEI_REDUCE_BLOCKS floats. */
int sei_luma_sqerr(const float *a, const float *b, size_t npix, float *out1, float *work, void *stream);

/* ---------------------------------------------------------------------------------------------
 * U-Net (src/models/convolutional.py), NHWC activations ("rows" = B*H*W pixels of C channels).
 * Parameter-gradient outputs (gw, gb, ggamma, gbeta, colsum `out`) are ACCUMULATED into with float
 * atomics: zero them (or keep the running gradient in them) before the call.
 *
 * sei_conv3x3_fwd: UNet.in_conv / out_conv, :174-176 (3x3, zero 'same' padding, small channel counts).
 *   x is NCHW (nchw_in=1) or NHWC; y likewise (nchw_out); optional residual `res` in y's layout (the
 *   global residual x + x0, :246-247). transposed=1 evaluates the DATA gradient instead: pass the
 *   forward weight (Cout_f,Cin_f,3,3) with Cin=Cout_f, Cout=Cin_f, x = upstream gradient.
 * sei_conv3x3_bwd_weight: gw (Cout,Cin,3,3) += sum_p gy[p,co] x[p+tap,ci]; gb (Cout) += sum_p gy.
 * sei_dwconv7_fwd: ConvBlock.conv1, :36-38 -- depthwise 7x7, zero pad 3, w is (C,1,7,7) as torch stores
 *   it, bias may be NULL; y = conv + bias + res_scale*res when res != NULL (NHWC, same shape).
 *   flip=1 correlates with the flipped taps = the data gradient.
 * sei_dwconv7_bwd_weight: gw (C,49) += sum_p gy[p,c] x[p+tap,c]; gbias (C) += sum_p gy (may be NULL).
 * sei_ln_fwd / sei_ln_bwd: channel LayerNorm, :21-30 (eps inside the sqrt, biased variance). fwd
 *   saves per-row mean and rstd; bwd needs the fwd INPUT x. C <= 8192.
 * sei_gemm_f32: every 1x1 Conv2d (:40,42,106,143) and its gradients as a row-major GEMM on the
 *   exact-f32 matrix cores: D[M,N] = op(A)[M,K] * op(B)[K,N] with a fused epilogue.
 *   A is (M,K) row-major when transA=0, (K,M) row-major when transA=1;
 *   B is (K,N) row-major when transB=0, (N,K) row-major when transB=1. D is (M,N) row-major.
 *   SEI_EPI_ACCUM may split K across workgroups and combine with float atomics.
 *   sei_gemm_f32_ex adds a batch of `batch` independent problems (element strides) and a switch to
 *   forbid split-K (bitwise reproducible accumulation).
 * sei_sepmap2: IdealDownsample / IdealUpsample (:54-92,113-133) as the real separable rank-2 map
 *   y[b,:,:,c] = L1 X R1^T + L2 X R2^T (L: (Ho,Hi), R: (Wo,Wi), row-major), the DFT-matrix form of the
 *   reference's rfft2/fftshift/mask-or-embed/irfft2 sequence including its discarded ifftshift.
 *   Backward = the same call with the transposed matrices. `work` >= 2*B*Hi*Wo*C floats.
 * sei_colsum_f32: out[n] += sum_m X[m,n]  (bias gradients of the 1x1 convolutions).
 * sei_adam_fused: torch.optim.Adam step (demo/train.py:157-186; amsgrad=False) over one flat bucket;
 *   grad is multiplied by grad_scale first (1/world_size after a summing all-reduce). step >= 1.
 *   grad may be bf16 (grad_is_bf16: the all-reduced, bf16-compressed gradient bucket of parallel.py).
 *   param_bf16 (optional): bf16 copy of the updated parameters, written in the same pass.
 * ------------------------------------------------------------------------------------------- */
int sei_conv3x3_fwd(const float *x, const float *w, const float *bias, const float *res, float *y,
                    int B, int H, int W, int Cin, int Cout, int nchw_in, int nchw_out, int transposed,
                    void *stream);
int sei_conv3x3_bwd_weight(const float *x, const float *gy, float *gw, float *gb, int B, int H, int W,
                           int Cin, int Cout, int nchw_x, int nchw_gy, void *stream);
/* Two-stage form for the network's end convolutions (3 <-> 32 channels, UNet.in_conv / out_conv): _parts_count = the number
 * of partial rows the launch leaves (0: shape not served -- use sei_conv3x3_bwd_weight), each Cout * Cin * 9 + Cout floats:
 * the weight gradient in torch's (Cout, Cin, 3, 3) layout followed by the bias gradient. sei_fold_many adds the rows into
 * the gradients (SEI_FOLD_SPLIT with split = Cout * Cin * 9: a = gw, b = gb). No atomics. */
size_t sei_conv3x3_bwd_weight_parts_count(int B, int H, int W, int Cin, int Cout, int nchw_x, int nchw_gy);
int sei_conv3x3_bwd_weight_parts(const float *x, const float *gy, float *part, int B, int H, int W, int Cin, int Cout,
                                 int nchw_x, int nchw_gy, void *stream);

int sei_dwconv7_fwd(const float *x, const float *w, const float *bias, const float *res,
                    float res_scale, float *y, int B, int H, int W, int C, int flip, void *stream);
/* `work` holds per-workgroup partial sums (two-stage reduction, no atomics); it must have at least
 * sei_dwconv7_bwd_weight_workspace(B,H,W,C) floats. */
size_t sei_dwconv7_bwd_weight_workspace(int B, int H, int W, int C);
/* gw = gbias = NULL: the fold is left to sei_fold_many (kind SEI_FOLD_DWCONV7) -- the partial sums stay in `work`,
 * [workspace / (50 C)][50][C] floats. */
int sei_dwconv7_bwd_weight(const float *x, const float *gy, float *gw, float *gbias, int B, int H,
                           int W, int C, float *work, size_t work_floats, void *stream);
/* The same three with the caller's explicit kernel choice: seg = 0 chooses by shape (LDS-tiled, whole-image or
 * generic; what the plain entry points pass), seg = 1..64 takes the generic sliding-window kernels with that many
 * output columns per worker segment (tests compare the paths bit for bit). Per call, no library state. */
int sei_dwconv7_fwd_ex(const float *x, const float *w, const float *bias, const float *res,
                       float res_scale, float *y, int B, int H, int W, int C, int flip, int seg, void *stream);
size_t sei_dwconv7_bwd_weight_workspace_ex(int B, int H, int W, int C, int seg);
int sei_dwconv7_bwd_weight_ex(const float *x, const float *gy, float *gw, float *gbias, int B, int H,
                              int W, int C, float *work, size_t work_floats, int seg, void *stream);
/* (sei_dwconv7_fwd_ex only: seg = 65 takes the first LDS-tiled kernel, seg = 66 the pipelined LDS-DMA kernel
 * (H, W >= 8, C % 32 == 0, 16-byte aligned x / w; SEI_ERR_BAD_ARG otherwise) that seg = 0 prefers from 1024
 * (tile, channel group) stages on; all three agree bit for bit.)
 *
 * ConvBlock.conv1 -> LayerNorm (src/models/convolutional.py:36-39 with :21-30) in one call:
 *   h1 = dwconv7(x) + bias (f32, kept: the LayerNorm backward re-reads it), h2 = LN_C(h1) * gamma + beta as bf16
 *   (out16 = 1) or f32, mean / rstd per pixel. C = 32 on images of at least 8 x 8: ONE launch (a workgroup owns all
 *   channels of its pixels; LDS-DMA double-buffered halo tiles, half-wave transposing reductions); any other shape:
 *   sei_dwconv7_fwd followed by sei_ln_fwd[_bf16]. sei_dwconv7_ln_fwd_launches tells which (1 or 2).
 *   _ex: fuse = 0 as above, 1 = the fused launch (C = 32 or 128, else SEI_ERR_BAD_ARG), 2 = the two launches. */
int sei_dwconv7_ln_fwd(const float *x, const float *w, const float *bias, const float *gamma, const float *beta,
                       float *h1, void *h2, int out16, float *mean, float *rstd, int B, int H, int W, int C,
                       float eps, void *stream);
int sei_dwconv7_ln_fwd_ex(const float *x, const float *w, const float *bias, const float *gamma, const float *beta,
                          float *h1, void *h2, int out16, float *mean, float *rstd, int B, int H, int W, int C,
                          float eps, int fuse, void *stream);
size_t sei_dwconv7_ln_fwd_launches(int B, int H, int W, int C);

int sei_ln_fwd(const float *x, const float *gamma, const float *beta, float *y, float *mean,
               float *rstd, size_t rows, int C, float eps, void *stream);
/* bwd: gx is written, ggamma/gbeta are accumulated (+=). `work` holds per-row statistics and the
 * per-workgroup parameter-gradient partials that are folded in a fixed order (bitwise reproducible);
 * it needs sei_ln_bwd_workspace(rows, C) floats (0 for shapes served by the scalar kernels, which
 * accumulate with float atomics: C not 4*2^k below 512, or not a multiple of 4 above). */
size_t sei_ln_bwd_workspace(size_t rows, int C);
/* ggamma = gbeta = NULL: the fold is left to sei_fold_many -- the partial sums stay in `work`, [sei_ln_bwd_part_count]
 * [2 C] floats from float sei_ln_bwd_part_offset on (allowed where the count is > 0: not for the atomics shapes). */
size_t sei_ln_bwd_part_offset(size_t rows, int C);
size_t sei_ln_bwd_part_count(size_t rows, int C);
int sei_ln_bwd(const float *x, const float *gamma, const float *mean, const float *rstd,
               const float *gy, float *gx, float *ggamma, float *gbeta, size_t rows, int C,
               float *work, size_t work_floats, void *stream);
/* The same with a second gradient of the LayerNorm's input added in the same pass: gx = LN'(gy) + res (res may be NULL).
 * What autograd otherwise does with an extra elementwise kernel where a U-Net level's output feeds both the downsampler
 * and the skip connection (/root/reference/src/models/convolutional.py:226-232: `skips.append(x)` next to
 * `downsampling_layers[lvl](x)`). Shapes with sei_ln_bwd_part_count(rows, C) > 0 only. */
int sei_ln_bwd_res(const float *x, const float *gamma, const float *mean, const float *rstd,
                   const float *gy, const float *res, float *gx, float *ggamma, float *gbeta, size_t rows, int C,
                   float *work, size_t work_floats, void *stream);

/* The second stage of MANY two-stage reductions in one launch: the LayerNorm parameter gradients and depthwise weight
 * gradients of a whole backward pass (reference: the autograd of src/models/convolutional.py:21-39 accumulates each of
 * them into its parameter's .grad; here a reducing kernel leaves per-workgroup partial sums and a fold adds them up in a
 * fixed order). A job is ONE destination and the partial-sum arrays of up to three launches that add to it (the model
 * calls of a step that share the parameter), folded one after the other exactly as their own fold launches would have:
 * bit-identical, in 1 launch instead of ~50 (U-Net step) / ~146 (SwinIR step).
 *   SEI_FOLD_SPLIT:   part[s]: [groups[s]][ncol]; entry e adds to a[e] (e < split), b[e - split] (e < 2 split) or
 *                     c[e - 2 split] (c may be NULL: dropped) -- sei_ln_bwd (ncol = 2 C, split = C),
 *                     sei_rowgemm_lnbwd_bf16 (ncol = 3 C), a plain column sum (ncol = split);
 *   SEI_FOLD_DWCONV7: part[s]: [groups[s]][50][split = C]; entry (t, c) adds to a[c * 49 + t] for t < 49 and to b[c]
 *                     (b may be NULL) for t = 49 -- sei_dwconv7_bwd_weight.
 * Destinations of different jobs must differ (checked). The partial sums must stay untouched until this launch. */
#define SEI_FOLD_SPLIT 0
#define SEI_FOLD_DWCONV7 1
#define SEI_FOLD_MAX_JOBS 48   /* (48 x 80 bytes of job table + the count: under the 4-KiB kernel-argument block) */
typedef struct SeiFoldJob {
    float *a, *b, *c;
    int ncol, split, kind, nseg;
    const float *part[3];
    int groups[3];
    int reserved;
} SeiFoldJob;
int sei_fold_many(const SeiFoldJob *jobs, int njobs, void *stream);

#define SEI_EPI_NONE 0
#define SEI_EPI_BIAS 1            /* D = acc + bias[n]                                        */
#define SEI_EPI_BIAS_GELU 2       /* D = acc + bias[n]; D2 = gelu(D)   (two outputs)          */
#define SEI_EPI_BIAS_RES 3        /* D = acc + bias[n] + R1 (+ R2 if non-NULL)                */
#define SEI_EPI_MUL_DGELU 4       /* D = acc * gelu'(R1)                                      */
#define SEI_EPI_ACCUM 5           /* D += acc   (gradient accumulation)                       */
#define SEI_EPI_BIAS_ROWSCALE 6   /* D = acc + bias[n]*R1[m]                                  */
#define SEI_EPI_BIAS_SCALE_RES 7  /* D = R2 + R1[m]*(acc + bias[n])  (stochastic depth: one factor per row; not in the
                                     quadrant-schedule kernel: sei_gemm_bf16nt takes a 128-row tile for it) */
int sei_gemm_f32(const float *A, const float *B, float *D, int M, int N, int K, int transA,
                 int transB, int epilogue, const float *bias, const float *R1, const float *R2,
                 float *D2, void *stream);
int sei_gemm_f32_ex(const float *A, const float *B, float *D, int M, int N, int K, int transA,
                    int transB, int epilogue, const float *bias, const float *R1, const float *R2,
                    float *D2, int batch, long long strideA, long long strideB, long long strideD,
                    int allow_splitk, void *stream);

/* Same contract on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16): A and B are float32 in HBM and
 * are rounded to bf16 (round-to-nearest-even) while being staged into LDS; accumulation, epilogue
 * and D stay float32. Throughput mode (--compute_dtype bf16); parity claims are made with sei_gemm_f32. */
int sei_gemm_bf16_ex(const float *A, const float *B, float *D, int M, int N, int K, int transA,
                     int transB, int epilogue, const float *bias, const float *R1, const float *R2,
                     float *D2, int batch, long long strideA, long long strideB, long long strideD,
                     int allow_splitk, void *stream);

/* sei_gemm_bf16_ex with per-operand storage types: a_is_bf16 / b_is_bf16 = the operand is bf16 in HBM. */
int sei_gemm_bf16_mixed(const void *A, int a_is_bf16, const void *B, int b_is_bf16, float *D, int M, int N,
                        int K, int transA, int transB, int epilogue, const float *bias, const float *R1,
                        const float *R2, float *D2, int batch, long long strideA, long long strideB,
                        long long strideD, int allow_splitk, void *stream);

/* bf16-mode support (all streaming): f32 -> bf16 cast; weight shadows w16 (R,C) and its transpose
 * wt16 (C,R) from the f32 master weight w (R,C) (either output may be NULL); channel LayerNorm with a
 * bf16 output (same statistics as sei_ln_fwd); column sums of a bf16 matrix (out[n] += sum_m X[m,n]). */
int sei_cast_bf16(const float *x, uint16_t *y, size_t n, void *stream);
int sei_weight_shadow_bf16(const float *w, uint16_t *w16, uint16_t *wt16, int R, int C, void *stream);
int sei_ln_fwd_bf16(const float *x, const float *gamma, const float *beta, uint16_t *y, float *mean,
                    float *rstd, size_t rows, int C, float eps, void *stream);
int sei_colsum_bf16(const uint16_t *X, float *out, size_t M, int N, void *stream);
/* Split-bf16 GEMM mode (--compute_dtype bf16x3: models/_ops.py gemm_x3; reference src/models/convolutional.py:33-51, float32
 * 1x1 convolutions). planes: 2 n bf16 -- planes[i] = bf16(x[i]) (head), planes[n + i] = bf16(x[i] - head) (remainder);
 * n % 4 == 0. A product a b is then evaluated as a_lo b_hi + a_hi b_lo + a_hi b_hi by three sei_gemm_bf16nt launches that
 * accumulate in float32; the two epilogues that are not additive run as the element-wise passes below (exact erf forms,
 * as SEI_EPI_BIAS_GELU / SEI_EPI_MUL_DGELU of sei_gemm_f32 apply them). */
int sei_split_bf16x2(const float *x, uint16_t *planes, size_t n, void *stream);
/* Three planes, for a reduction concatenated over the three products (weight gradients: planes of a reduction-major
 * operand stack along the reduction index, so ONE launch with K' = 3 K sums them): pattern 0 = [head, head, remainder]
 * (the A side), 1 = [head, remainder, head] (the B side). planes: 3 n bf16; n % 4 == 0. */
int sei_split_bf16x3(const float *x, uint16_t *planes, size_t n, int pattern, void *stream);
int sei_gelu_f32(const float *x, float *y, size_t n, void *stream);            /* y = gelu(x) */
int sei_mul_dgelu_f32(float *d, const float *h, size_t n, void *stream);      /* d *= gelu'(h) */
/* x (R,C) float32 or bf16 -> x16 (R,C) bf16 copy (optional, float32 input only) and xt16 (C,ldt) bf16
 * transpose (optional) whose columns R..ldt-1 are zero: a K-padded operand for sei_gemm_bf16nt.
 * colsum (optional): colsum[c] += sum_r x[r][c] in float32 -- the bias gradient, from the same pass. */
int sei_cast_transpose_bf16(const void *x, int x_is_bf16, uint16_t *x16, uint16_t *xt16, int R, int C,
                            int ldt, float *colsum, void *stream);
/* The plain cast with a WEIGHTED column sum in the same pass: colsum[c] += sum_r row_weight[r] x[r][c] (un-rounded input).
 * In models/_ops.DownsampleFn16 the 1x1 convolution runs behind the ideal downsampler (src/models/convolutional.py:136-150
 * commuted), where its bias enters as bias[c] * s[row], s = the resampler's response to a constant image: the bias gradient
 * is this weighted sum of the output gradient's rows, taken while that gradient is cast for the GEMMs. C % 4 == 0. */
int sei_cast_bf16_colsum_weighted(const float *x, uint16_t *x16, const float *row_weight, float *colsum, int R, int C,
                                  void *stream);
/* The same cast with its (weighted when row_weight != NULL) column sums left as PARTIAL sums, one row of C floats per row
 * block: part[g][c], g < sei_cast_bf16_colsum_parts_count(R, C) (0: shape not taken, C % 4 != 0). The bias gradients of a
 * backward pass (src/models/convolutional.py:33-51, 96-150: every 1x1 convolution's bias) then join the pass's other
 * partial sums in ONE sei_fold_many launch (SEI_FOLD_SPLIT, ncol = split = C) instead of ending every cast in one float
 * atomic per column and workgroup -- which is what sized the atomics form's grid (512 workgroups; 2048 here). ABI 11. */
size_t sei_cast_bf16_colsum_parts_count(int R, int C);
int sei_cast_bf16_colsum_parts(const float *x, uint16_t *x16, const float *row_weight, float *part, int R, int C,
                               void *stream);

/* Several bf16 transposes in ONE launch: job k copies src (R, C) bf16 to dst (C, R). The fused pointwise MLP's backward
 * (sei_mlp_fused_bwd) reads both 1x1 weights of a ConvBlock transposed (src/models/convolutional.py:33-51: conv2 / conv3);
 * their bf16 copies change with every optimizer step, so a step re-transposes the 8 small matrices of the two fused
 * levels: one launch instead of 8 of ~9 us. */
#define SEI_TRANSPOSE_MAX_JOBS 16
typedef struct SeiTransposeJob {
    const uint16_t *src;
    uint16_t *dst;
    int R, C;
} SeiTransposeJob;
int sei_transpose_bf16_many(const SeiTransposeJob *jobs, int njobs, void *stream);

/* Large-shape bf16 GEMM with bf16 operands in HBM:  D[M,N] = op(A) * op(B),  f32 accumulation.
 * Each operand is stored either K-contiguous (a_rmajor = 0: A is (M,K) row-major, b_rmajor = 0: B is (N,K)
 * row-major -- the "NT" form) or reduction-major (a_rmajor = 1: A is (K,M) row-major; b_rmajor = 1: B is
 * (K,N) row-major). lda / ldb = row strides in elements of the operand as stored (multiples of 8).
 * Forward 1x1 convolutions: A = activations, B = bf16 weight (N,K): both K-contiguous.
 * Data gradients dX = dY W: A = dY K-contiguous, B = the SAME bf16 weight read reduction-major.
 * Weight gradients dW += dY^T X (SEI_EPI_ACCUM): both operands reduction-major (reduction over pixels).
 * Direct-to-LDS staging (global_load_lds); reduction-major tiles are consumed with the transposing LDS
 * read ds_read_b64_tr_b16, so no transposed copies exist anywhere. 8 waves, 128x128 / 192x256 / 96x256
 * tiles. D32 (float) and/or D16 (bf16) receive the epilogue result; SEI_EPI_BIAS_GELU also writes gelu(D)
 * to D2_16 (bf16). Needs K % 8 == 0 (K tails are fed from a zero page), M % 8 == 0 / N % 8 == 0 for a
 * reduction-major A / B, 16-byte aligned operands. May split K (float atomics on D32, zero-filled first
 * unless accumulating) when the output is narrow. */
int sei_gemm_bf16nt(const uint16_t *A, int lda, int a_rmajor, const uint16_t *B, int ldb, int b_rmajor,
                    float *D32, uint16_t *D16, int M, int N, int K, int epilogue, const float *bias,
                    const float *R1, const float *R2, uint16_t *D2_16, void *stream);

/* Weight gradient of two passes in one launch: D[M,N] (+)= A1^T B1 + A2^T B2, bf16 operands stored
 * reduction-major (A1 (K1,M), A2 (K2,M), row stride lda; B1 (K1,N), B2 (K2,N), row stride ldb), f32 D.
 * accumulate=0 stores D (the first write of a step into a gradient that need not be zeroed first);
 * accumulate=1 adds. One pass over the 4*M*N-byte gradient instead of two read-modify-writes.
 * (reference: the two model calls per step of ProposedLoss, src/losses/__init__.py, whose 1x1-conv weight
 * gradients torch sums in AccumulateGrad.) */
int sei_gemm_bf16nt_dw2(const uint16_t *A1, const uint16_t *A2, int lda, const uint16_t *B1, const uint16_t *B2,
                        int ldb, float *D32, int M, int N, int K1, int K2, int accumulate, void *stream);

/* The two entry points above with the caller's explicit schedule choice (tests of every tile variant, tools/):
 * tile = 0 lets the dispatcher decide (what the plain entry points pass); 1, 2, 3, 5 = 128x128, 128x256, 192x256,
 * 96x256 tiles of the 128x128-style loop; 15 / 16 = its single-stage / 128x256 reduction-major variants;
 * 30-33 = the quadrant schedule of gemm_bf16pq.h with 256x256, 288x256, 288x128, 256x128 tiles (the 128-column ones on
 * three LDS stages; 38 / 39 = 288x128 / 256x128 on two). band > 0 fixes
 * the band width of the XCD-aware tile order (0 = automatic). A choice the operands do not allow falls back to
 * the automatic one. Per call: the library keeps no mutable state. */
int sei_gemm_bf16nt_ex(const uint16_t *A, int lda, int a_rmajor, const uint16_t *B, int ldb, int b_rmajor,
                       float *D32, uint16_t *D16, int M, int N, int K, int epilogue, const float *bias,
                       const float *R1, const float *R2, uint16_t *D2_16, int tile, int band, void *stream);
int sei_gemm_bf16nt_dw2_ex(const uint16_t *A1, const uint16_t *A2, int lda, const uint16_t *B1,
                           const uint16_t *B2, int ldb, float *D32, int M, int N, int K1, int K2, int accumulate,
                           int tile, void *stream);

/* sei_gemm_bf16nt with a bf16 result (SEI_EPI_MUL_DGELU or SEI_EPI_NONE) that also accumulates the result's column sums:
 * colsum[n] += sum_m D16[m][n] (the rounded values, as sei_colsum_bf16 over D16 would add them). In a ConvBlock's backward
 * (src/models/convolutional.py:33-51) D16 = gh3 = (dY W3) gelu'(h3) and its column sums are conv2's bias gradient: where
 * the launch takes the quadrant kernel unsplit they ride in its epilogue (no pass over gh3), elsewhere the column-sum
 * kernel follows. */
int sei_gemm_bf16nt_colsum(const uint16_t *A, int lda, int a_rmajor, const uint16_t *B, int ldb, int b_rmajor,
                           uint16_t *D16, int M, int N, int K, int epilogue, const float *R1, float *colsum, void *stream);

/* The schedule sei_gemm_bf16nt would take for a GEMM of these shapes, WITHOUT launching anything (host arithmetic only:
 * callable on a machine with no GPU): (family << 48) | (tile rows << 32) | (tile columns << 16) | K splits, with
 * family 1 = gemm_bf16nt_kernel (the 128x128-style loop, whatever its tile), 2 = gemm_bf16pq_kernel (the quadrant
 * schedule); 0 = arguments the entry point would refuse. out_f32 / out_bf16: which of D32 / D16 the call passes.
 * Operands are taken as densely packed and 16-byte aligned. For tests that pin which kernels the TIMED batch runs
 * (tests/test_loss_gpu.py::test_timed_configuration_vs_oracle): a dispatch change cannot silently move the benchmarked
 * launch set away from what the oracle comparison covered. (The reference has no counterpart: its GEMMs are torch's,
 * src/models/convolutional.py:33-51.) */
size_t sei_gemm_bf16nt_plan(int a_rmajor, int b_rmajor, int out_f32, int out_bf16, int M, int N, int K, int epilogue);

/* Split-K through slabs (ABI 11). sei_gemm_bf16nt_ex and sei_gemm_bf16nt_colsum (colsum != NULL: D16 only,
 * SEI_EPI_MUL_DGELU or SEI_EPI_NONE) with a caller-owned workspace for the K slices of the quadrant kernel: `ws` points to
 * ws_bytes (>= 16 KiB) of device memory, 256-byte aligned, whose FIRST 16 KiB ARE ZERO before the first call (tile counters:
 * every call leaves them zero again; the remainder needs no initialisation) and which only one stream uses at a time.
 * Each K slice of a tile stores its accumulators into its slab (write-through), draws a ticket from the tile's counter, and
 * the slice that draws the last one adds the other slabs to its registers and runs the whole epilogue: no zero-fill
 * launch, no float atomics (1.3 TB/s chip-wide on MI355X against ~6 TB/s of plain stores), results that do not depend on
 * the order in which slices finish when there are two of them, and EVERY epilogue may split (bf16 / GELU / GELU' results
 * and riding column sums included), which the atomics form could not offer. No workgroup waits for another. ws = NULL, a
 * workspace too small for the launch (16 KiB + tiles * slices * tile bytes) or a launch off the quadrant kernel: exactly
 * the entry points above. tile / band: as sei_gemm_bf16nt_ex (0 = automatic). splitk > 0 asks for that many slices where
 * the schedule can split (tests, experiments).
 * (The reference's GEMMs are torch's: src/models/convolutional.py:33-51, 96-150; this is how their K-heavy, row-poor
 * shapes -- 288 ... 3456 rows at the two deepest U-Net levels -- fill 256 CUs.)
 * sei_gemm_bf16nt_plan_ws: the schedule with a workspace of ws_bytes (as sei_gemm_bf16nt_plan; bit 15 of the split count
 * set when the slices meet in slabs). */
int sei_gemm_bf16nt_ws(const uint16_t *A, int lda, int a_rmajor, const uint16_t *B, int ldb, int b_rmajor,
                       float *D32, uint16_t *D16, int M, int N, int K, int epilogue, const float *bias,
                       const float *R1, const float *R2, uint16_t *D2_16, float *colsum, void *ws, size_t ws_bytes,
                       int tile, int band, int splitk, void *stream);
size_t sei_gemm_bf16nt_plan_ws(int a_rmajor, int b_rmajor, int out_f32, int out_bf16, int M, int N, int K, int epilogue,
                               size_t ws_bytes);

#ifdef SEI_TUNING
/* Tools-only build (make tuning -> libsei_hip_tuning.so; not part of libsei_hip.so).
 * sei_debug_tr_probe: what ds_read_b64_tr_b16 delivers for a 64x128 LDS image (tools/probe_tr_read.py).
 * sei_debug_set_nt_tile: process-wide default for the tile / band arguments above (100 + b = band b), so the
 * experiment scripts can steer GEMMs launched by the model code; adds tile codes 4, 11-14 (LDS-ring variants),
 * 20 (the 256x256 ping-pong schedule of gemm_bf16pp.h) and 34-37 (timing-only ablations, wrong results). */
int sei_debug_tr_probe(const uint16_t *in, uint16_t *out, int r0, int c0, void *stream);
int sei_debug_set_nt_tile(int code);
#endif

/* The ideal resamplers of the DEEPEST levels in one pass (csrc/sepmap_small.hip): input extents <= 8, output extents
 * <= 24, C % 64 == 0 -- the 6 x 6 and 3 x 3 images a 48-pixel crop becomes at 2048 - 8192 channels
 * (IdealDownsample / IdealUpsample, src/models/convolutional.py:54-92,113-133). x (B, Hi, Wi, C) -> y (B, Ho, Wo, C), NHWC
 * float32, y = L1 X R1^T + L2 X R2^T per image and channel, float32 FMAs; L1, L2: (Ho, Hi), R1, R2: (Wo, Wi) row-major
 * device arrays. One workgroup item = one image x 64 channels through LDS; no HBM intermediate. _eligible: 1 / 0. */
size_t sei_sepmap2_small_eligible(int B, int Hi, int Wi, int Ho, int Wo, int C);
int sei_sepmap2_small(const float *x, void *y, int out_bf16, int B, int Hi, int Wi, int Ho, int Wo, int C,
                      const float *L1, const float *R1, const float *L2, const float *R2, void *stream);
/* out_bf16 = 1 (and sei_sepmap2_bf16_out16 below for the matrix-core kernel): y is bf16, the float32 result rounded once
 * to nearest even -- the operand of the 1x1 convolution that models/_ops.DownsampleFn16 evaluates BEHIND the ideal
 * downsampler (src/models/convolutional.py:136-150 commuted): no float32 copy of the resampled tensor, no cast pass. */
int sei_sepmap2_bf16_out16(const float *x, uint16_t *y16, int B, int Hi, int Wi, int Ho, int Wo, int C,
                           const uint16_t *packed, void *stream);

int sei_sepmap2(const float *x, float *y, int B, int Hi, int Wi, int Ho, int Wo, int C,
                const float *L1, const float *R1, const float *L2, const float *R2, float *work,
                size_t work_floats, void *stream);
/* The same map with the four matrices packed for the kernel's scalar loads (what the build's host code uses):
 * RW[j][j'][t] = R_t[j'][j] with j' padded to a multiple of 24 (zeros), LH[i][t][i'] = L_t[i'][i] with i' padded
 * to a multiple of 24. Results are bit-identical to sei_sepmap2. */
/* The same map in the bf16 throughput mode, on the matrix cores (sepmap_mfma.hip): activations rounded to bf16 (x before
 * the W product, the intermediate between the products), the four matrices as bf16 head + remainder (exact to ~2^-17),
 * f32 accumulation and output. Eligible shapes only (sei_sepmap2_bf16_eligible: 12 <= Hi, Wi <= 64 with the
 * intermediate of one image x 16 channels in LDS, C % 16 == 0); SEI_ERR_BAD_ARG otherwise -- the caller takes
 * sei_sepmap2_packed. `packed`: the matrices in the kernel's LDS image, made once per map by sei_sepmap2_bf16_pack from
 * L1, L2: (Ho, Hi), R1, R2: (Wo, Wi) float32 row-major into sei_sepmap2_bf16_pack_elems(..) uint16 elements. */
int sei_sepmap2_bf16(const float *x, float *y, int B, int Hi, int Wi, int Ho, int Wo, int C, const uint16_t *packed,
                     void *stream);
int sei_sepmap2_bf16_pack(const float *L1, const float *R1, const float *L2, const float *R2, uint16_t *packed, int Hi,
                          int Wi, int Ho, int Wo, void *stream);
size_t sei_sepmap2_bf16_eligible(int B, int Hi, int Wi, int Ho, int Wo, int C);
size_t sei_sepmap2_bf16_pack_elems(int Hi, int Wi, int Ho, int Wo);
/* The same map at LARGE extents (Hi or Wi in 65 .. 256: the x4 network's 96- and 192-pixel levels, the un-cropped 256-pixel
 * series) on the matrix cores (csrc/sepmap_big.hip): both products as one batched constant-matrix GEMM kernel -- the matrix
 * (bf16 head + remainder) in registers, the activations streamed once through LDS -- with a bf16 intermediate in `work`
 * (sei_sepmap2_big_work_elems uint16). `packed` (sei_sepmap2_big_pack_elems uint16) from sei_sepmap2_big_pack, once per
 * map. sei_sepmap2_big_eligible != 0 says which shapes are built (C % 16 == 0); others stay on sei_sepmap2_packed. */
size_t sei_sepmap2_big_eligible(int B, int Hi, int Wi, int Ho, int Wo, int C);
size_t sei_sepmap2_big_pack_elems(int Hi, int Wi, int Ho, int Wo);
size_t sei_sepmap2_big_work_elems(int B, int Hi, int Wi, int Ho, int Wo, int C);
int sei_sepmap2_big_pack(const float *L1, const float *R1, const float *L2, const float *R2, uint16_t *packed,
                         int Hi, int Wi, int Ho, int Wo, void *stream);
int sei_sepmap2_big(const float *x, float *y, int B, int Hi, int Wi, int Ho, int Wo, int C, const uint16_t *packed,
                    uint16_t *work, void *stream);
int sei_sepmap2_packed(const float *x, float *y, int B, int Hi, int Wi, int Ho, int Wo, int C,
                       const float *RW, const float *LH, float *work, size_t work_floats, void *stream);

int sei_colsum_f32(const float *X, float *out, size_t M, int N, void *stream);
/* out[n] += sum_m row_weight[m] * X[m,n]: the bias gradient of a 1x1 convolution applied AFTER the ideal
 * downsampler (models/_ops.py, DownsampleFn: the bias enters as bias[n] * s[m], s = the resampler's response
 * to a constant image). */
int sei_colsum_weighted_f32(const float *X, const float *row_weight, float *out, size_t M, int N, void *stream);

/* Fused pointwise MLP of a ConvBlock at the shallow U-Net levels (csrc/mlp_fused.hip; C = 32 or 128; reference:
 * src/models/convolutional.py:40-51): out (M, C) = res_scale * x + conv3(gelu(conv2(h2))), h2 (M, C) bf16 = the
 * LayerNorm output, W2 (4C, C) and W3 (C, 4C) the bf16 1x1-convolution weights as stored, b2 / b3 float. The 4C-wide
 * hidden activation stays in registers (accumulator tiles re-used as MFMA operands).
 * sei_mlp_fused_bwd: from go (M, C) float: gh2 (M, C) float = gradient w.r.t. h2; go16 (M, C), h4 = gelu(h3) (M, 4C) and
 * gh3 (M, 4C) in bf16 = the operands of the two weight-gradient GEMMs (h3 = conv2(h2) is recomputed). W3T (4C, C) and
 * W2T (C, 4C) are the transposed bf16 weights. (Bias gradients: column sums of go and gh3, sei_colsum_*.)
 * sei_mlp_fused_eligible(M, C) != 0 where the fused form is built to win: C = 32, and C = 128 with M a multiple of 144
 * (csrc/mlp128.hip: nine-wave workgroups of 144 pixels, the weights through an LDS-DMA ring); the entry points take any
 * M for both widths, the caller takes the GEMMs where this says 0. */
size_t sei_mlp_fused_eligible(long long M, int C);
int sei_mlp_fused_fwd(const uint16_t *h2, const uint16_t *W2, const float *b2, const uint16_t *W3, const float *b3,
                      const float *x, float res_scale, float *out, int M, int C, void *stream);
int sei_mlp_fused_bwd(const float *go, const uint16_t *h2, const uint16_t *W2, const float *b2, const uint16_t *W3T,
                      const uint16_t *W2T, float *gh2, uint16_t *go16, uint16_t *h4, uint16_t *gh3, int M, int C,
                      void *stream);

/* ---- SwinIR building blocks (csrc/swin_kernels.hip; reference: deepinv.models.SwinIR as configured at
 * src/models/__init__.py:51-74 = the official SwinIR network_swinir.py; parity unpinned, see oracle/swinir_path.py).
 *
 * sei_swin_attn_fwd: WindowAttention.forward for every 8x8 window and head of a batch of B images of HxW tokens
 * (H, W multiples of 8). qkv (B*H*W, 3*heads*head_dim) = the qkv projection of the tokens in natural (b, y, x)
 * order; out (B*H*W, heads*head_dim) in the same order. The cyclic shift by `shift` (torch.roll of
 * SwinTransformerBlock.forward), window_partition / window_reverse, the relative-position bias
 * (table ((2*8-1)^2, heads), index computed in-kernel) and the -100 mask of shifted windows are index arithmetic
 * inside the kernel. scale = head_dim^-0.5 (applied to q first, as the reference). head_dim 30 (SwinIR), 32, 16, 8.
 * sei_swin_attn_bwd: dqkv (every element written) and dtable (+=, float atomics) from dout; the probabilities are
 * recomputed from qkv. */
int sei_swin_attn_fwd(const float *qkv, const float *table, float *out, int B, int H, int W, int heads,
                      int head_dim, int shift, float scale, void *stream);
int sei_swin_attn_bwd(const float *qkv, const float *table, const float *dout, float *dqkv, float *dtable,
                      int B, int H, int W, int heads, int head_dim, int shift, float scale, void *stream);
/* The same two on the bf16 MFMA (csrc/swin_attn_mfma.hip, throughput mode): qkv (B*H*W, 3*heads*32) and out / dout
 * (B*H*W, heads*32) in bf16, every head padded from 30 to 32 dims (the pad dims must be zero in qkv; they come out
 * zero). dqkv is bf16, dtable float (+=).
 * lse (heads, B*(H/8)*(W/8) windows in partition order, 64 queries) float: the rows' log-sum-exp of the scaled, biased
 * and masked scores in log2 units, written by the forward pass when non-NULL and REQUIRED by the backward pass, which
 * rebuilds the probabilities from it (one exponential per element, no row maxima / sums) and takes
 * rowsum(dout * out) from the forward output `out` (ABI 9). */
int sei_swin_attn_fwd_bf16(const uint16_t *qkv, const float *table, uint16_t *out, float *lse, int B, int H, int W,
                           int heads, int shift, float scale, void *stream);
int sei_swin_attn_bwd_bf16(const uint16_t *qkv, const float *table, const uint16_t *out, const float *lse,
                           const uint16_t *dout, uint16_t *dqkv, float *dtable, int B, int H, int W, int heads,
                           int shift, float scale, void *stream);
/* Padded-grid form of an NHWC batch for the 3x3 convolutions with many channels (RSTB.conv, conv_after_body,
 * conv_before_upsample, upsample.*): xp = guard_rows zero rows of C floats, then (B, H+2, W+2, C) with a zero
 * border, then guard_rows zero rows. On that grid the convolution is nine row-shifted GEMMs over the same flat
 * array (row offset dy*(W+2)+dx), with no im2col buffer. sei_unpad_nhwc takes the interior of a (B, H+2, W+2, C)
 * result back to (B, H, W, C), applies LeakyReLU(0.01) when act = 1, then adds `res` when non-NULL. C % 4 == 0. */
int sei_pad_nhwc(const float *x, float *xp, int B, int H, int W, int C, int guard_rows, void *stream);
int sei_unpad_nhwc(const float *xp, const float *res, float *y, int B, int H, int W, int C, int act, void *stream);
/* y[m, n] = row_scale[m] * x[m, n] (row_scale may be NULL = 1), times 1 or 0.01 by the sign of leaky_gate[m, n]
 * when leaky_gate is non-NULL: the backward of stochastic depth (timm DropPath) and of LeakyReLU. N % 4 == 0. */
int sei_rowscale(const float *x, const float *row_scale, const float *leaky_gate, float *y, size_t M, int N,
                 void *stream);

/* ---- SwinIR throughput (bf16) path: csrc/swin_bf16_kernels.hip, csrc/gemm_bf16nt.hip ----
 * sei_pack: dst[i] = map[i] >= 0 ? src[map[i]] : 0, as bf16 (to_bf16 = 1) or float: the float32 parameter bucket into
 * the padded / permuted layouts the bf16 GEMMs read (heads 30 -> 32, channels 180 -> 192, 3x3 weights tap-major).
 * sei_unpack_add: dst[map[i]] += src[i] for map[i] >= 0: gradients in GEMM-output layout back into the bucket (a
 * parameter element appears at most once in a map). */
int sei_pack(const float *src, const int *map, void *dst, size_t n, int to_bf16, void *stream);
int sei_unpack_add(const float *src, const int *map, float *dst, size_t n, void *stream);
/* LayerNorm over C <= 256 channels (C % 4 == 0) of float32 rows -> bf16 rows of ldy elements, zeros beyond C; with
 * ones_col (ldy > C) column C holds 1.0, so that the weight gradient dY^T [LN(x) | 1] of the linear layer behind it
 * carries that layer's bias gradient in its column C (no separate column-sum pass over dY). */
int sei_ln_fwd_bf16_pad(const float *x, const float *gamma, const float *beta, uint16_t *y, float *mean,
                        float *rstd, size_t rows, int C, int ldy, float eps, int ones_col, void *stream);
/* Its backward: gx (rows, C) = LN'(gy) (+ res if non-NULL); gy float32 with row stride ldg; ggamma / gbeta += through
 * per-workgroup partial sums in `work` (>= sei_swin_partials_floats(C) floats) folded by a second launch: deterministic,
 * no atomics. */
size_t sei_swin_partials_floats(int C);
int sei_ln_bwd_pad(const float *x, const float *gamma, const float *mean, const float *rstd, const float *gy,
                   const float *res, float *gx, float *ggamma, float *gbeta, size_t rows, int C, int ldg,
                   float *work, size_t work_floats, void *stream);
/* y16[m, :C] = bf16(row_scale[m] * x[m, :]) (row_scale may be NULL), zeros up to ldy; colsum[c] += sum_m of the
 * scaled rows when non-NULL (the bias gradient of the linear layer whose output gradient this is; partial sums in
 * `work`, as sei_ln_bwd_pad). */
int sei_cast_pad_bf16(const float *x, const float *row_scale, uint16_t *y, float *colsum, size_t rows, int C,
                      int ldy, float *work, size_t work_floats, void *stream);
/* sei_pad_nhwc with bf16 output and the channel count padded from C to Cp (zeros). */
int sei_pad_nhwc_bf16(const float *x, uint16_t *xp, int B, int H, int W, int C, int Cp, int guard_rows, void *stream);
/* The same with channel C (the first padding channel, Cp > C) set to 1.0 in every row when ones_col: as the input grid
 * of a 3x3 convolution it meets zero weight columns in the forward product, and in the weight gradient
 * (sei_gemm_bf16nt_dw2_taps: D[t][n][c] = sum_rows gy[r][n] x[r + shift_t][c]) column C of every tap then holds
 * sum_rows gy[r][n] -- the convolution's BIAS gradient (nn.Conv2d.bias; reference construction
 * src/models/__init__.py:51-74), without a column-sum pass over gy. */
int sei_pad_nhwc_bf16_ones(const float *x, uint16_t *xp, int B, int H, int W, int C, int Cp, int guard_rows, int ones_col,
                           void *stream);
/* 3x3 convolution (stride 1, zero padding 1) as ONE implicit GEMM on that grid: D[r, n] = sum over taps t and
 * channels c of Ap[r + row_off9[t], c] * B[n, t * cin_pad + c] (+ bias[n]); Ap points at row 0 of the padded grid
 * (behind the guard rows), cin_pad % 64 == 0, B (N, 9 * cin_pad) bf16 tap-major. The im2col matrix exists only as
 * LDS tiles: every k-tile of the LDS-DMA stream is a 64-channel slice of one tap, fetched from the same array at that
 * tap's row shift. Rows r of the border are computed too (discard them: sei_unpad_nhwc). */
int sei_gemm_bf16nt_conv(const uint16_t *Ap, int cin_pad, const int *row_off9, const uint16_t *B, int ldb,
                         float *D32, uint16_t *D16, int M, int N, int epilogue, const float *bias, void *stream);
/* The same convolution with sei_unpad_nhwc in its epilogue: y (Bimg, H, W, N) float32 = [LeakyReLU 0.01 when act = 1]
 * (conv + bias) [+ res], Ap the zero-bordered (H + 2) x (W + 2) grids of Bimg images; the border pixels are computed and
 * dropped by the row-patch epilogue, the (grid rows, N) intermediate is never written. N % 4 == 0; res (same layout as
 * y) requires bias. Reference: nn.Conv2d(3x3, padding 1) (+ residual / LeakyReLU) of deepinv's SwinIR (RSTB.conv,
 * conv_after_body, the upsampler), forward and -- with the transposed tap-major weights -- data gradient. */
int sei_gemm_bf16nt_conv_unpad(const uint16_t *Ap, int cin_pad, const int *row_off9, const uint16_t *B, int ldb, float *y,
                               const float *res, int Bimg, int H, int W, int N, const float *bias, int act, void *stream);

int sei_adam_fused(float *param, const void *grad, int grad_is_bf16, float *exp_avg, float *exp_avg_sq,
                   size_t n, float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                   float grad_scale, uint16_t *param_bf16, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Optimizer step inside the weight-gradient GEMM (one GPU, bf16 mode; demo/train.py:157-186 + :262-270 fused):
 * sei_gemm_bf16nt_dw2_adam computes the same two-segment product as sei_gemm_bf16nt_dw2 -- the COMPLETE
 * gradient of an (M, N) weight for this step -- and, instead of storing it, applies torch.optim.Adam's update to
 * param / exp_avg / exp_avg_sq (and the bf16 shadow, optional) in the epilogue: the 4 B/parameter gradient is
 * neither written nor read back.  Identical per-element arithmetic to sei_adam_fused with grad_scale = 1.
 * hyper: DEVICE array of 6 floats for this step, {beta1, beta2, eps, weight_decay, lr / (1 - beta1^t),
 * 1 / sqrt(1 - beta2^t)}; sei_adam_scalars fills a HOST array with exactly the values sei_adam_fused derives from
 * (lr, betas, eps, weight_decay, step), so that fused and separate steps agree bit for bit.
 * ------------------------------------------------------------------------------------------- */
int sei_gemm_bf16nt_dw2_adam(const uint16_t *A1, const uint16_t *A2, int lda, const uint16_t *B1, const uint16_t *B2,
                             int ldb, float *param, float *exp_avg, float *exp_avg_sq, uint16_t *param_bf16,
                             const float *hyper, int M, int N, int K1, int K2, void *stream);
/* The same with the caller's schedule choice (tests of both kernels, tools/): tile 0 = the dispatcher's choice, 1 = the
 * 128 x 128 loop (gemm_bf16nt_kernel<..., ADAM>), 30 / 33 = the quadrant schedule's 256 x 256 / 256 x 128 tiles with the
 * same epilogue (gemm_bf16pq_kernel<..., ADAM>); a choice the operands do not allow falls back to the loop. */
int sei_gemm_bf16nt_dw2_adam_ex(const uint16_t *A1, const uint16_t *A2, int lda, const uint16_t *B1, const uint16_t *B2,
                                int ldb, float *param, float *exp_avg, float *exp_avg_sq, uint16_t *param_bf16,
                                const float *hyper, int M, int N, int K1, int K2, int tile, void *stream);
int sei_adam_scalars(float lr, float beta1, float beta2, float eps, float weight_decay, int step, float *out6_host,
                     void *stream);
/* the same values written to a DEVICE array by a launch on `stream` (what a training loop calls before every replay of
 * a captured step: stream-ordered, nothing on the host to overwrite while the GPU lags behind) */
int sei_adam_scalars_to_device(float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                               float *dev6, void *stream);

/* The same two-segment weight gradient STORED as bf16 (whole-row 8-byte quads): with several GPUs and a bf16-compressed
 * gradient exchange the gradient is written straight into the exchange buffer (parallel.FlatGradientReducer.comm) --
 * no float32 copy of it, no cast pass (2.6 GB read + 1.3 GB written per step at the default size). Same rounding
 * (round-to-nearest-even of the float32 accumulator) as sei_cast_bf16 of the stored float32 gradient. */
/* The weight gradient of a 3x3 convolution on the zero-bordered grid (sei_gemm_bf16nt_conv), all taps in ONE launch:
 * D32 + t * tap_ld (M x N, float32) (+)= sum over rows r of A[r, :]^T B[r + tap_rows[t], :] for t < ntaps <= 9 -- the same
 * dY (A1 / A2: the step's two model calls, K1 / K2 grid rows; K2 = 0 for one) against the padded input grids B1 / B2
 * shifted by each tap's row offset (negative offsets reach into the guard rows in front of the grid).
 * Reference: the nine (ky, kx) slices of Conv2d.weight.grad, torch's conv2d backward (deepinv SwinIR's 3x3 convs). */
int sei_gemm_bf16nt_dw2_taps(const uint16_t *A1, const uint16_t *A2, int lda, const uint16_t *B1, const uint16_t *B2,
                             int ldb, float *D32, int M, int N, int K1, int K2, int accumulate, int ntaps,
                             const int *tap_rows, long long tap_ld, void *stream);

int sei_gemm_bf16nt_dw2_bf16out(const uint16_t *A1, const uint16_t *A2, int lda, const uint16_t *B1, const uint16_t *B2,
                                int ldb, uint16_t *D16, int M, int N, int K1, int K2, void *stream);

/* ---------------------------------------------------------------------------------------------------------
 * Token-streaming GEMMs for layer-sized weights (token_gemm.hip; ABI 6): the linear layers of the reference's default
 * backbone, deepinv's SwinIR as built by src/models/__init__.py:51-74 (qkv / proj / fc1 / fc2 of every block:
 * nn.Linear forward, its data gradient and its weight gradient in torch's autograd), in the bf16 throughput mode.
 * One persistent workgroup per CU streams its share of the tokens once; the weight (gradient) stays in registers.
 *
 * sei_tokgrad_bf16: D (Mo, Ni; row stride ldd) += Y^T X over K1 + K2 tokens; Y1 / Y2: (K, ldy >= Mo), X1 / X2:
 * (K, ldx >= Ni) bf16, token-major as the layers store them; the two segments are the step's two model calls (K2 = 0:
 * one). Float atomics: D must hold the running gradient (or zeros). Eligible (sei_tokgrad_bf16_eligible != 0): Mo and Ni
 * multiples of 192 with at most eight 192 x 192 blocks (sei_tokgrad_bf16_blocks: sixteen), K1 and K2 multiples of 64, ldy and ldx multiples of 8, 16-byte
 * aligned operands. Same product as sei_gemm_bf16nt_dw2(..., accumulate = 1) up to the float summation order. */
#define SEI_TOKGRAD_MAX_BLOCKS 16
typedef struct SeiTokGradBlock {     /* one 192 x 192 block of a weight gradient: D += Y[:, y0:y0+192]^T X[:, x0:x0+192] */
    const uint16_t *Y1, *Y2;         /* (K1, ldy) / (K2, ldy) bf16; Y2 unused when K2 = 0 */
    const uint16_t *X1, *X2;         /* (K1, ldx) / (K2, ldx) */
    int ldy, ldx, y0, x0;            /* multiples of 8; y0 + 192 <= ldy, x0 + 192 <= ldx */
    float *D;                        /* the block's first element; row stride ldd >= 192 */
    int ldd;
} SeiTokGradBlock;
/* Up to eight blocks in ONE launch over the same K1 + K2 tokens (all four weight gradients of a Swin block = 3 + 1 + 2 + 2
 * blocks): the CUs are shared out between the blocks, so every workgroup walks nblocks times as many tokens and the
 * float atomics per output (one partial block per workgroup) shrink by the same factor. */
int sei_tokgrad_bf16_blocks(const SeiTokGradBlock *blocks, int nblocks, long long K1, long long K2, void *stream);
int sei_tokgrad_bf16(const uint16_t *Y1, const uint16_t *Y2, int ldy, const uint16_t *X1, const uint16_t *X2, int ldx,
                     float *D, int ldd, int Mo, int Ni, long long K1, long long K2, void *stream);
size_t sei_tokgrad_bf16_eligible(int Mo, int Ni, int ldy, int ldx, long long K1, long long K2);

/* Streamed weight gradients of the U-Net's shallow levels (csrc/dw_stream.hip): D (Mo, ldd >= Ni) += Y^T X over K1 + K2
 * pixels for every job of a table in ONE launch per block shape -- persistent workgroups that own a whole block of a gradient
 * in their accumulators and stream a contiguous pixel range once (LDS-DMA ring), dealt to the jobs in proportion to their
 * bytes. Replaces torch's conv2d weight gradient of the 1x1 convolutions of /root/reference/src/models/convolutional.py:40-42
 * (ConvBlock.conv2 / conv3), :106 (Upsample) and :143 (Downsample) where (Mo, Ni) is (128, 256 k) / (256 k, 128), k <= 4,
 * with K1, K2 multiples of 64, or (32, 128) / (128, 32) with K1, K2 multiples of 128; sei_dwstream_bf16_eligible() != 0
 * says so, sei_gemm_bf16nt_dw2 serves everything else. Y (K, ldy == Mo) and X (K, ldx == Ni) are bf16, pixel-major as
 * stored, in two row segments (the step's two model calls; K2 = 0: one). D is ADDED to with float atomics; so is gbias
 * (the bias gradient = the column sums of Y, from one more MFMA per fragment against a fragment of ones). */
#define SEI_DWSTREAM_MAX_JOBS 64
typedef struct SeiDwStreamJob {
    const uint16_t *Y1, *Y2;         /* (K1, ldy) / (K2, ldy) bf16; Y2 unused when K2 = 0 */
    const uint16_t *X1, *X2;         /* (K1, ldx) / (K2, ldx) */
    int ldy, ldx, Mo, Ni;
    float *D;
    int ldd, reserved;
    long long K1, K2;
    float *gbias;                    /* NULL, or (Mo): += the column sums of Y over all K1 + K2 pixels (the convolution's bias gradient) */
} SeiDwStreamJob;
int sei_dwstream_bf16_jobs(const SeiDwStreamJob *jobs, int njobs, void *stream);
size_t sei_dwstream_bf16_eligible(int Mo, int Ni, int ldy, int ldx, long long K1, long long K2);

/* sei_rowgemm_bf16: D (M, N) = epilogue(A W^T): A (M, lda >= K) bf16 rows, W (N, ldw >= K) bf16 -- the layer's matrix as
 * nn.Linear stores it (forward) or its transpose (data gradient), zero-padded to N in {192, 384, 576} rows and
 * K in {192, 384, 576} columns -- with the SEI_EPI_* epilogues of sei_gemm_bf16nt:
 *   SEI_EPI_BIAS            D16 = bf16(acc + bias)                                   (N 576, K 192: qkv)
 *   SEI_EPI_BIAS_RES        D32 = acc + bias + R1                                     (N 192, K 192 / 384: proj, fc2)
 *   SEI_EPI_BIAS_SCALE_RES  D32 = R2 + R1[row] (acc + bias)    (stochastic depth)     (the same two)
 *   SEI_EPI_BIAS_GELU       D32 = acc + bias (optional: may be NULL), D16 = bf16(gelu(acc + bias))   (N 384, K 192: fc1; nv = N)
 *   SEI_EPI_MUL_DGELU       D16 = bf16(acc gelu'(R1))                                 (N 384, K 192: fc2's data gradient)
 *   SEI_EPI_NONE            D16 = bf16(acc) (N 192, K 192)  or  D32 = acc (N 192, K 384 / 576)   (the other data gradients)
 * nv (a multiple of 4, <= N): valid columns of the float32 output, of bias and of the row-shaped R (ldr >= nv); bf16
 * outputs always receive all N columns (ld16 >= N; zeros where W's padding rows are zero). M a multiple of 64.
 * sei_rowgemm_bf16_eligible(M, N, K, epilogue, out16) != 0 says whether a combination is built; the caller takes
 * sei_gemm_bf16nt otherwise (same results up to the float summation order). */
int sei_rowgemm_bf16(const uint16_t *A, int lda, const uint16_t *W, int ldw, float *D32, int ld32, uint16_t *D16,
                     int ld16, long long M, int N, int K, int nv, int epilogue, const float *bias, const float *R1,
                     const float *R2, int ldr, void *stream);
size_t sei_rowgemm_bf16_eligible(long long M, int N, int K, int epilogue, int out16);

/* The data gradient of a linear layer that follows a LayerNorm, with the LayerNorm's backward in the epilogue (the
 * product never leaves the chip): gh = A W^T (A: (M, lda >= K) bf16; W: (192, ldw >= K) bf16, rows >= C zero; K 384 or
 * 576), then torch's native_layer_norm_backward over the C channels of every row plus the residual gradient:
 *   gx = rstd (gh gamma - mean_c(gh gamma) - xhat mean_c(gh gamma xhat)) + res,   xhat = (x - mean) rstd;
 *   ggamma += sum_rows gh xhat,  gbeta += sum_rows gh;
 * and, when y16 is given, the operand of the next weight / data gradient in the same pass:
 *   y16 (M, ldy >= 192; zeros past C) = bf16(gx row_scale[row]),  colsum += sum_rows gx row_scale[row]
 * (colsum: K = 384 only and optional -- at K = 576 no registers are left for the third column sum; the bias gradient it
 * would be then comes out of the weight gradient through sei_rowgemm_gelu_bf16's ones column).
 * x, res, gx: (M, C) float32 rows; mean, rstd, row_scale: M floats. work: sei_rowgemm_lnbwd_work_floats(C) floats (per-
 * workgroup column sums, folded by a second launch; ggamma = gbeta = NULL leaves that fold to sei_fold_many: the sums
 * stay in `work`, [min(M / 32, 256)][3][C] = for ggamma | gbeta | colsum). Replaces sei_gemm_bf16nt + sei_ln_bwd_pad (+ sei_cast_pad_bf16):
 * deepinv SwinIR's norm1 / norm2 in front of qkv / fc1 (reference construction: src/models/__init__.py:51-74). */
int sei_rowgemm_lnbwd_bf16(const uint16_t *A, int lda, const uint16_t *W, int ldw, long long M, int K, const float *x,
                           const float *gamma, const float *mean, const float *rstd, const float *res, float *gx, int C,
                           float *ggamma, float *gbeta, const float *row_scale, uint16_t *y16, int ldy, float *colsum,
                           float *work, size_t work_floats, void *stream);
size_t sei_rowgemm_lnbwd_bf16_eligible(long long M, int K, int C);
size_t sei_rowgemm_lnbwd_work_floats(int C);

/* fc2's data gradient through GELU with the GELU' input RECOMPUTED instead of stored: D16 (M, ld16 >= N) =
 * bf16((A W^T) * gelu'(A2 W2^T + bias2)), A = the gradient rows (M, lda >= K), W = fc2's matrix transposed (N, K);
 * A2 = fc1's input rows (M, lda2 >= K), W2 = fc1's matrix (N, K), bias2 its bias (nv entries) -- the second product is
 * the forward pass's pre-activation, bit for bit (same MFMA order as sei_rowgemm_bf16 with SEI_EPI_BIAS_GELU), so the
 * forward pass need not write it (D32 = NULL there) nor this pass read it: 1536 of the 2688 bytes per token of either
 * kernel. N = 384, K = 192 (deepinv SwinIR's Mlp: nn.Linear(180, 360) -> nn.GELU -> nn.Linear(360, 180), zero-padded).
 * Same result as sei_rowgemm_bf16(SEI_EPI_MUL_DGELU) on the stored pre-activation. */
int sei_rowgemm_dgelu_bf16(const uint16_t *A, int lda, const uint16_t *W, int ldw, const uint16_t *A2, int lda2,
                           const uint16_t *W2, int ldw2, const float *bias2, int nv, uint16_t *D16, int ld16, long long M,
                           int N, int K, void *stream);
size_t sei_rowgemm_dgelu_bf16_eligible(long long M, int N, int K);

/* A residual linear layer followed by a LayerNorm, in one launch: out (M, C) = res + [row_scale[row]] (A W^T + bias)
 * (SEI_EPI_BIAS_RES, or SEI_EPI_BIAS_SCALE_RES when row_scale is given: timm's DropPath) and, since a workgroup holds
 * whole rows of out, nn.LayerNorm(C, eps) of them: h16 (M, ldh >= 192) = bf16((out - mean) rstd gamma + beta), padding
 * columns zero except column C = 1.0 when ones_col (see sei_ln_fwd_bf16_pad), mean / rstd (M floats each) for the
 * backward. A (M, lda >= K) bf16, W (192, ldw >= K) bf16 with zero rows past C, K 192 or 384: deepinv SwinIR's
 * proj -> norm2 and fc2 -> the next block's norm1. Same values as sei_rowgemm_bf16 + sei_ln_fwd_bf16_pad (the LayerNorm
 * sums run over 16 lanes instead of 64: last-bit differences in mean / rstd). */
int sei_rowgemm_ln_bf16(const uint16_t *A, int lda, const uint16_t *W, int ldw, long long M, int K, int C, const float *bias,
                        const float *row_scale, const float *res, float *out, const float *gamma, const float *beta,
                        float eps, int ones_col, uint16_t *h16, int ldh, float *mean, float *rstd, void *stream);
size_t sei_rowgemm_ln_bf16_eligible(long long M, int K, int C);

/* fc1's forward alone (bf16 gelu output only, the float32 pre-activation is recomputed by sei_rowgemm_dgelu_bf16), with
 * column one_at of D16 written as 1.0 (-1: none): the weight gradient of the layer behind, D16^T-contracted over the tokens,
 * then carries that layer's BIAS gradient in column one_at (the qkv / fc1 trick of sei_ln_fwd_bf16_pad's ones_col, for
 * fc2). bias: all N entries (zeros in the padding). Same values as sei_rowgemm_bf16(SEI_EPI_BIAS_GELU, D32 = NULL). */
int sei_rowgemm_gelu_bf16(const uint16_t *A, int lda, const uint16_t *W, int ldw, const float *bias, int nv, uint16_t *D16,
                          int ld16, long long M, int N, int K, int one_at, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SEI_HIP_H */
