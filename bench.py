#!/usr/bin/env python3
"""Benchmark of the proposed-loss training step (BASELINE.json metric: training images/sec).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B_per_gpu]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch: Loss.forward (random 48-crop of the 256x256 pairs)
-> SURE + scale-equivariant loss (three U-Net evaluations, two physics operators, the EI resample)
-> backward -> gradient all-reduce (N>1) -> Adam. One "image" = one 256x256 pair entering Loss.forward.
Workload: config[1] of BASELINE.json -- deblurring, Gaussian_R2, noise 5, proposed loss, the reference's
default ConvolutionalModel (hidden 32, 5 scales, 645,063,043 parameters), per-GPU batch 32 (config[3]'s
256 / 8), synthetic inputs resident in HBM, random-init weights (torch.manual_seed(0)).

The LAST stdout line of rank 0 is the contract's ONE JSON line -- the headline only, kept under 8 kB (`compact_line`;
tests/test_bench_line.py) so that the driver parses it. Every secondary series is printed in full on an EARLIER line
(`{"series": "<name>", ...}`) and the whole uncompacted object goes to gpurun_out/bench_full.json (or $SEI_BENCH_FULL).
The headline carries these extra objects:
  roofline     -- the dominant kernel family (the bf16 MFMA GEMMs: all 1x1 convolutions and their gradients),
                  every launch of one step re-issued between HIP events on the launch stream.
  roofline_hbm -- the streaming kernel families of the same step (Adam, depthwise 7x7, LayerNorm, ideal resamplers,
                  casts + column sums, 3x3 convolutions, physics + loss terms): algorithmic bytes per step / the
                  event-timed ms of their re-issued launches / 8 TB/s.
  secondary    -- the same step in the reference's own arithmetic (float32 GEMMs; --secondary, N=1 only).
  cpu_baseline -- the oracle's torch-CPU restatement of the same step on the host cores (N=1 only).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "scale-equivariant-imaging_amd"))
sys.path.insert(1, ROOT)

# MI355X_MICROARCH.md, dense spec peaks: v_mfma_f32_32x32x2_f32 / v_mfma_f32_32x32x16_bf16
# (bf16x3: three bf16 MFMA products per algorithmic product -- priced against a third of the bf16 peak)
MFMA_PEAK_TFLOPS = {"f32": 157.3, "bf16": 2500.0, "bf16x3": 2500.0 / 3.0}
CROP, NOISE, KERNEL = 48, 5, "Gaussian_R2"


def reference_args(device, hidden=32, scales=5, task="deblurring", sr_factor=None, arch="Convolutional"):
    return argparse.Namespace(
        device=device, task=task, kernel=KERNEL, sr_factor=sr_factor, noise_level=NOISE, physics_v2=True,
        physics_true_adjoint=False, model_kind="Proposed", ProposedModel__architecture=arch,
        ConvolutionalModel__residual=True, ConvolutionalModel__inner_residual=True,
        ConvolutionalModel__num_conv_blocks=1, ConvolutionalModel__inout_convs=True,
        ConvolutionalModel__hidden_channels=hidden, ConvolutionalModel__scales=scales,
        data_parallel_devices=None, method="proposed", partial_sure=True, sure_margin=None,
        partial_sure_sr=False, sure_cropped_div=True, sure_averaged_cst=None, Loss__crop_training_pairs=True,
        Loss__crop_size=CROP, ProposedLoss__stop_gradient=True, ProposedLoss__sure_alternative=None,
        ProposedLoss__alpha_tradeoff=1.0, ProposedLoss__transforms="Scaling_Transforms",
        ScalingTransform__kind="padded", ScalingTransform__antialias=False)


def cpu_baseline(batch, hidden, scales, timed_steps=3):
    """The oracle's restatement of the same training step on the host CPU (kind "port"), by the protocol of
    BASELINE.md section 3 / SURVEY 8(d): batch 4, float32, all host cores, 1 warm-up step + 3 timed steps (~35 s)."""
    import oracle
    from oracle import torch_path as tp
    torch.manual_seed(0)
    threads = oracle.use_all_usable_cpus()                 # the cgroup's CPU quota, not every logical CPU of the host
    sd = {k: v.requires_grad_(True) for k, v in tp.unet_init_state_dict(hidden, scales).items()}
    opt = torch.optim.Adam(list(sd.values()), lr=1e-4)
    k = tp.blur_kernel(KERNEL)
    A = lambda v: tp.blur_fft(v, k)
    g = torch.Generator().manual_seed(1234)
    x = torch.rand((batch, 3, 256, 256), generator=g)
    y = tp.add_noise(A(x), NOISE / 255)
    model = lambda v: tp.unet_forward(sd, v, scales=scales)

    def step():
        opt.zero_grad()
        _, yc = tp.crop_pair(x, y, CROP, 1)
        rate, center = tp.sample_scale_params(batch)
        loss, _ = tp.proposed_loss(yc.contiguous(), A, model, NOISE / 255, margin=6, rate=rate, center=center)
        loss.backward()
        opt.step()

    t0 = time.perf_counter()
    step()                                                 # warm-up (first-call overheads, allocator, Adam state)
    t1 = time.perf_counter()
    per_step = []
    for _ in range(timed_steps):
        ts = time.perf_counter()
        step()
        per_step.append(time.perf_counter() - ts)
    dt = sum(per_step) / timed_steps
    # box-to-box spread of this figure has been +-9 % (0.457-0.540 images/s over rounds 3-4): context, not a target
    return {"value": round(batch / dt, 4), "unit": "images/s", "cores": threads, "kind": "port",
            "value_min": round(batch / max(per_step), 4), "value_max": round(batch / min(per_step), 4),
            "timed_steps": timed_steps,
            "sample": f"{timed_steps} timed proposed-loss steps after 1 warm-up ({t1 - t0:.1f} s) of the same U-Net at "
                      f"batch {batch} (crop {CROP}, 3 fwd + 3 bwd + Adam), float32, torch CPU ops in the reference's "
                      f"order (oracle/torch_path.py), {dt:.1f} s per step (min {min(per_step):.1f}, max {max(per_step):.1f}); "
                      f"{threads} intra-op threads = the CPUs this process may use (host: {os.cpu_count()} logical CPUs)"}


def cpu_baseline_swinir(batch, sr_factor=2, timed_steps=1):
    """The CPU figure beside `secondary.swinir_sr2` (SURVEY 8(d) config (1) asks for the reference's DEFAULT arguments,
    i.e. the SwinIR backbone, src/settings.py:50): the oracle's restatement of the same proposed-loss step (sr x2, training
    mode with stochastic depth, 3 forward + 3 backward passes + torch.optim.Adam) on the host cores. A bounded sample: one
    warm-up step + `timed_steps` timed steps at batch `batch` (a step is ~2 TFLOP of float32 work). PARITY UNPINNED like
    the oracle it times."""
    import oracle
    from oracle import swinir_path as sp, torch_path as tp
    torch.manual_seed(0)
    threads = oracle.use_all_usable_cpus()
    sd = {k: v.requires_grad_(True) for k, v in sp.swinir_init_state_dict(upscale=sr_factor).items()
          if v.dtype.is_floating_point}
    opt = torch.optim.Adam(list(sd.values()), lr=2e-4)
    A = lambda v: tp.downsample_aa(v, sr_factor)
    g = torch.Generator().manual_seed(1234)
    x = torch.rand((batch, 3, CROP * sr_factor, CROP * sr_factor), generator=g)
    y = tp.add_noise(A(x), NOISE / 255)
    model = lambda v: sp.swinir_forward(sd, v, upscale=sr_factor, drop_masks=sp.draw_drop_masks(v.shape[0]))

    def step():
        opt.zero_grad()
        rate, center = tp.sample_scale_params(batch)
        loss, _ = tp.proposed_loss(y, A, model, NOISE / 255, margin=0, rate=rate, center=center)
        loss.backward()
        opt.step()

    t0 = time.perf_counter()
    step()
    t1 = time.perf_counter()
    for _ in range(timed_steps):
        step()
    dt = (time.perf_counter() - t1) / timed_steps
    return {"value": round(batch / dt, 4), "unit": "images/s", "cores": threads, "kind": "port",
            "sample": f"{timed_steps} timed proposed-loss step(s) after 1 warm-up ({t1 - t0:.1f} s) of the SwinIR backbone "
                      f"(embed 180, 6 x 6 blocks) at batch {batch}, sr x{sr_factor}, pairs {CROP * sr_factor} / {CROP}, float32, "
                      f"torch CPU ops (oracle/swinir_path.py, PARITY UNPINNED), {dt:.1f} s per step; {threads} intra-op threads"}


# Algorithmic HBM bytes of one launch of the streaming entry points, from the call's own arguments (DESIGN.md
# section 4: every operand read once, every result written once; f32 = 4 B, bf16 = 2 B).
def _stream_bytes(name, a):
    if name in ("sei_dwconv7_fwd", "sei_dwconv7_fwd_ex"):
        B, H, W, C = a[6:10]
        return (8 + (4 if a[3] else 0)) * B * H * W * C
    if name == "sei_dwconv7_ln_fwd":                      # x in, h1 (f32) + h2 (bf16 / f32) out
        B, H, W, C = a[10:14]
        return (8 + (2 if a[7] else 4)) * B * H * W * C
    if name in ("sei_dwconv7_bwd_weight", "sei_dwconv7_bwd_weight_ex"):
        B, H, W, C = a[4:8]
        return 8 * B * H * W * C
    if name == "sei_ln_fwd":
        return 8 * a[6] * a[7]
    if name == "sei_ln_fwd_bf16":
        return 6 * a[6] * a[7]
    if name == "sei_ln_bwd":
        return 12 * a[8] * a[9]
    if name == "sei_ln_bwd_res":                          # + the skip connection's gradient rows
        return (16 if a[5] else 12) * a[9] * a[10]
    if name == "sei_fold_many":                           # the partial sums of every job, read once
        return sum(4 * j.ncol * j.groups[k] for j in a[0][:a[1]] for k in range(j.nseg))
    if name == "sei_ln_fwd_bf16_pad":                     # f32 in, bf16 out (padded row)
        return a[6] * (4 * a[7] + 2 * a[8])
    if name == "sei_ln_bwd_pad":                          # x, gy (strided), [res], gx
        return a[9] * (4 * a[10] * (3 if a[5] else 2) + 4 * a[11])
    if name == "sei_cast_pad_bf16":
        return a[4] * (4 * a[5] + 2 * a[6])
    if name in ("sei_pad_nhwc", "sei_unpad_nhwc"):
        B, H, W, C = (a[2:6] if name == "sei_pad_nhwc" else a[3:7])
        return 8 * B * H * W * C
    if name in ("sei_pad_nhwc_bf16", "sei_pad_nhwc_bf16_ones"):
        B, H, W, C, Cp = a[2:7]
        return B * H * W * (4 * C + 2 * Cp)
    if name == "sei_rowscale":
        return 8 * a[4] * a[5]
    if name in ("sei_pack", "sei_unpack_add"):
        return 10 * a[3]
    if name in ("sei_swin_attn_fwd", "sei_swin_attn_bwd"):
        bwd = "_bwd" in name
        B, H, W, heads = (a[5:9] if bwd else a[3:7])
        return B * H * W * heads * a[9 if bwd else 7] * 4 * (8 if bwd else 4)
    if name == "sei_swin_attn_fwd_bf16":                 # qkv in, out + the rows' log-sum-exp out
        B, H, W, heads = a[4:8]
        return B * H * W * heads * (32 * 2 * 4 + 4)
    if name == "sei_swin_attn_bwd_bf16":                 # qkv, out, dout, lse in, dqkv out
        B, H, W, heads = a[7:11]
        return B * H * W * heads * (32 * 2 * 8 + 4)
    if name == "sei_sepmap2_bf16_pack":
        return 0
    if name in ("sei_sepmap2_packed", "sei_sepmap2_bf16", "sei_sepmap2_big"):
        B, Hi, Wi, Ho, Wo, C = a[2:8]                     # (algorithmic: x in, y out; the bf16 intermediate of _big is its own)
        return 4 * B * C * (Hi * Wi + Ho * Wo)
    if name == "sei_sepmap2_bf16_out16":                  # bf16 result
        B, Hi, Wi, Ho, Wo, C = a[2:8]
        return B * C * (4 * Hi * Wi + 2 * Ho * Wo)
    if name == "sei_sepmap2_small":
        B, Hi, Wi, Ho, Wo, C = a[3:9]
        return B * C * (4 * Hi * Wi + (2 if a[2] else 4) * Ho * Wo)
    if name == "sei_sepmap2_big_pack":
        return 0
    if name == "sei_cast_transpose_bf16":
        return 6 * a[4] * a[5]
    if name == "sei_transpose_bf16_many":                  # every matrix in and out once (bf16)
        return sum(4 * a[0][k].R * a[0][k].C for k in range(a[1]))
    if name in ("sei_cast_bf16_colsum_weighted", "sei_cast_bf16_colsum_parts"):      # (x, x16, row_weight, colsum | part, R, C)
        return 6 * a[4] * a[5]
    if name == "sei_cast_bf16":
        return 6 * a[2]
    if name == "sei_colsum_bf16":
        return 2 * a[2] * a[3]
    if name == "sei_colsum_f32":
        return 4 * a[2] * a[3]
    if name == "sei_colsum_weighted_f32":
        return 4 * a[3] * a[4]
    if name == "sei_conv3x3_fwd":
        B, H, W, Ci, Co = a[5:10]
        return 4 * B * H * W * (Ci + Co + (Co if a[3] else 0))
    if name == "sei_conv3x3_bwd_weight":
        B, H, W, Ci, Co = a[4:9]
        return 4 * B * H * W * (Ci + Co)
    if name == "sei_conv3x3_bwd_weight_parts":
        B, H, W, Ci, Co = a[3:8]
        return 4 * B * H * W * (Ci + Co)
    if name == "sei_adam_fused":
        return a[5] * (24 + (2 if a[2] else 4) + (2 if a[-1] else 0))
    if name == "sei_gemm_bf16nt_dw2_adam":                # param / exp_avg / exp_avg_sq in and out, shadow out, operands
        M, Nn, K1, K2 = a[11:15]
        return M * Nn * (24 + (2 if a[9] else 0)) + 2 * (K1 + K2) * (M + Nn)
    if name == "sei_rowgemm_bf16":                        # A rows in; float32 (nv columns) and / or bf16 (N columns) rows out;
        M, Nn, K, nv, epi = a[8:13]                       # residual / GELU' rows in
        return M * (2 * K + (4 * nv if a[4] else 0) + (2 * Nn if a[6] else 0) + (4 * nv if epi in (3, 4, 7) else 0)
                    + (4 if epi == 7 else 0))
    if name == "sei_rowgemm_gelu_bf16":                   # A rows in, bf16 rows out
        M, Nn, K = a[8:11]
        return M * (2 * K + 2 * Nn)
    if name == "sei_rowgemm_ln_bf16":                     # A rows and residual rows in; new token rows, their LayerNorm (bf16) out
        M, K, C = a[4:7]
        return M * (2 * K + 8 * C + 2 * a[16] + 8 + (4 if a[8] else 0))
    if name == "sei_rowgemm_dgelu_bf16":                  # gradient rows and the layer's input rows in; bf16 rows out
        M, Nn, K = a[12:15]
        return M * (4 * K + 2 * Nn)
    if name == "sei_rowgemm_lnbwd_bf16":                  # A rows, x and residual rows in; gx (and the bf16 copy) out
        M, K, C = a[4], a[5], a[12]
        return M * (2 * K + 12 * C + 8 + ((2 * a[17] + 4) if a[16] else 0))
    if name == "sei_tokgrad_bf16_blocks":                 # every distinct 192-column operand slice once
        blocks, n, K1, K2 = a[0], a[1], a[2], a[3]
        slices = {(blocks[i].Y1, blocks[i].y0) for i in range(n)} | {(blocks[i].X1, blocks[i].x0) for i in range(n)}
        return 2 * 192 * (K1 + K2) * len(slices)
    if name == "sei_blur_sep_circ":
        return 8 * a[6] * a[7] * a[8]
    if name in ("sei_scale_resample_fwd", "sei_scale_resample_bwd"):
        B, C, Hi, Wi, H, W = a[4:10]
        return 4 * B * C * (Hi * Wi + H * W)
    if name == "sei_resample_sepband":
        return 4 * a[2] * (a[3] * a[4] + a[5] * a[6])
    if name == "sei_scale_params":
        return 24 * a[4]
    if name == "sei_axpy":
        return 12 * a[4]
    if name == "sei_stack_axpy":                          # a, b in; [a, a + alpha b] out
        return 16 * a[4]
    if name == "sei_proposed_draws":                      # written only: probe interior, rates, centres, noise
        B, C, H, W, m = a[3:8]
        return 4 * (B * C * ((H - 2 * m) * (W - 2 * m) + H * W) + 3 * B)
    if name == "sei_crop_window":
        return 8 * a[2] * a[7] * a[7]
    if name == "sei_split_bf16x2":                        # float32 in, two bf16 planes out
        return 8 * a[2]
    if name == "sei_split_bf16x3":
        return 10 * a[2]
    if name == "sei_gelu_f32":
        return 8 * a[2]
    if name == "sei_mul_dgelu_f32":
        return 12 * a[2]
    if name in ("sei_sure_terms", "sei_sure_loss"):
        return 24 * a[4] * a[5] * a[6]
    if name in ("sei_mse_terms", "sei_mse_loss"):
        return 12 * a[2]
    return None


def _gemm_bytes(name, a):
    """Algorithmic HBM bytes of one launch of a matrix-core entry point, from its own arguments: every operand read once,
    every result written once (a weight gradient's two row blocks, a fused MLP's hidden activation staying on chip).
    None = an entry point not modelled here (its launches are then left out of `algorithmic_bytes_per_launch`)."""
    if name == "sei_gemm_bf16nt_colsum":                       # operands + the GELU' input in, the bf16 result out
        M, Nn, K = a[7:10]
        return 2 * K * (M + Nn) + M * Nn * (2 + (4 if a[11] else 0))
    if name in ("sei_gemm_bf16nt", "sei_gemm_bf16nt_ex", "sei_gemm_bf16nt_ws"):      # (_ws: the same leading arguments)
        M, Nn, K, epi = a[8:12]
        extra = sum(4 for ptr in (a[13], a[14]) if ptr) if epi != 6 else 0          # R1 / R2 (BIAS_ROWSCALE: M floats)
        return 2 * K * (M + Nn) + M * Nn * ((4 if a[6] else 0) + (2 if a[7] else 0) + extra + (2 if a[15] else 0)
                                            + (4 if epi == 5 else 0))
    if name in ("sei_gemm_bf16_ex", "sei_gemm_f32_ex"):
        M, Nn, K = a[3:6]
        return 4 * (K * (M + Nn) + M * Nn * (1 + sum(1 for ptr in a[10:13] if ptr) + (1 if a[8] == 5 else 0)))
    if name == "sei_gemm_bf16_mixed":
        M, Nn, K = a[5:8]
        return K * (M * (2 if a[1] else 4) + Nn * (2 if a[3] else 4)) + 4 * M * Nn * (
            1 + sum(1 for ptr in a[12:15] if ptr) + (1 if a[10] == 5 else 0))
    if name in ("sei_gemm_bf16nt_dw2", "sei_gemm_bf16nt_dw2_bf16out", "sei_gemm_bf16nt_dw2_adam"):
        if name.endswith("_adam"):
            return _stream_bytes(name, a)
        Np, Kp, K1, K2 = a[7:11]
        out = 2 if name.endswith("_bf16out") else (8 if a[11] else 4)
        return 2 * (K1 + K2) * (Np + Kp) + out * Np * Kp
    if name == "sei_dwstream_bf16_jobs":                       # each job: both operands' row blocks once, D read + written
        return sum(2 * (j.K1 + j.K2) * (j.Mo + j.Ni) + 8 * j.Mo * j.Ni for j in a[0][:a[1]])
    if name == "sei_mlp_fused_fwd":                            # h2 (bf16) + x in, out out, the two weight matrices
        M, C = a[8:10]
        return M * C * 10 + 16 * C * C
    if name == "sei_mlp_fused_bwd":                            # go + h2 in; gh2 (f32), go16, h4, gh3 (bf16) out; 3 matrices
        M, C = a[10:12]
        return M * C * (4 + 2 + 4 + 2 + 8 + 8) + 24 * C * C
    return None


# (label, entry-point prefixes); the text before " (" is the family's short name on the compact line
_STREAM_FAMILIES = [
    ("adam_vec_kernel (fused Adam over the flat bucket)", ("sei_adam_fused",)),
    ("gemm_bf16nt_kernel<..., ADAM> (HBM-bound weight gradients whose epilogue applies the Adam step: the bottleneck pair)",
     ("sei_gemm_bf16nt_dw2_adam",)),
    ("rowgemm_* / tokgrad_* (token-streaming GEMMs of layer-sized weights: the Swin blocks' linear layers forward, data "
     "gradient (+ LayerNorm backward) and weight gradient; HBM-bound at K = 192-576, also booked in the MFMA family)",
     ("sei_rowgemm_", "sei_tokgrad_")),
    ("dwconv7_* (depthwise 7x7: forward, data and weight gradients)", ("sei_dwconv7_",)),
    ("ln_* (channel LayerNorm forward / backward)", ("sei_ln_",)),
    ("fold_many_kernel (second stage of the LayerNorm / depthwise reductions of the whole backward pass)", ("sei_fold_many",)),
    ("sepmap_* (ideal resamplers)", ("sei_sepmap2",)),
    ("swin_attn_* (8x8-window attention: qkv in, out / dqkv out)", ("sei_swin_attn_",)),
    ("pad / unpad / rowscale / pack kernels (padded-grid copies, weight re-layout)",
     ("sei_pad_nhwc", "sei_unpad_nhwc", "sei_rowscale", "sei_pack", "sei_unpack_add")),
    ("cast / colsum kernels (bf16 copies, bias gradients)", ("sei_cast_", "sei_colsum_", "sei_transpose_bf16_many")),
    ("conv3x3_* (in / out convolutions)", ("sei_conv3x3_",)),
    ("blur / scale_resample / axpy / sure / mse / draws / crop kernels (physics + loss terms, the step's prologue)",
     ("sei_blur_", "sei_scale_resample_", "sei_scale_params", "sei_axpy", "sei_sure_terms", "sei_sure_loss", "sei_mse_terms", "sei_mse_loss",
      "sei_resample_", "sei_stack_axpy", "sei_proposed_draws", "sei_crop_window")),
    ("split / gelu passes of the bf16x3 mode (bf16 head + remainder planes of every GEMM operand; GELU, GELU' element-wise)",
     ("sei_split_bf16x", "sei_gelu_f32", "sei_mul_dgelu_f32")),
]
# The depthwise 7x7 family is bound by the float32 FMA issue rate, not by HBM (PMC): one v_fma_f32 wave-instruction
# per 4 cycles per SIMD = 256 CUs x 4 SIMDs x 16 lanes x 2 FLOP x 2.4 GHz; 49 taps x 2 FLOP per element and call.
VALU_F32_FMA_PEAK_TFLOPS = 78.6


def _dwconv_flops(name, a):
    if name in ("sei_dwconv7_fwd", "sei_dwconv7_fwd_ex"):
        B, H, W, C = a[6:10]
    elif name == "sei_dwconv7_ln_fwd":
        B, H, W, C = a[10:14]
    elif name in ("sei_dwconv7_bwd_weight", "sei_dwconv7_bwd_weight_ex"):
        B, H, W, C = a[4:8]
    else:
        return 0.0
    return 98.0 * B * H * W * C
PMC_TRAFFIC_FILE = "r06_f_unet_pmc_gemm.json"      # the committed PMC pass `roofline.traffic` is read from
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)


# A weight-gradient GEMM with the Adam step in its epilogue moves 26 B per output element under 2 (K1 + K2) FLOP: the
# bottleneck pair (K = 864) sits at 66 FLOP per byte -- HBM-bound, booked with the streaming families --, the level-3
# launches (K = 3456) at 264 FLOP per byte run at ~600 TFLOP/s and belong to the MFMA family.
ADAM_GEMM_MFMA_INTENSITY = 150.0


def adam_gemm_is_mfma_bound(args):
    M, Nn, K1, K2 = args[11:15]
    return 2.0 * M * Nn * (K1 + K2) / _stream_bytes("sei_gemm_bf16nt_dw2_adam", args) > ADAM_GEMM_MFMA_INTENSITY


def stream_roofline(log, reps=3):
    """Re-issue the logged streaming launches of one step, family by family, between HIP events."""
    import _native
    out = []
    log = [(n, a) for n, a in log if not (n == "sei_gemm_bf16nt_dw2_adam" and adam_gemm_is_mfma_bound(a))]
    for label, prefixes in _STREAM_FAMILIES:
        calls = [(n, a) for n, a in log if n.startswith(prefixes) and _stream_bytes(n, a) is not None]
        if not calls:
            continue
        nbytes = float(sum(_stream_bytes(n, a) for n, a in calls))
        for n, a in calls:
            _native.call(n, *a)                           # warm
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            for n, a in calls:
                _native.call(n, *a)
        e1.record()
        e1.synchronize()
        ms = e0.elapsed_time(e1) / reps
        gbs = nbytes / (ms * 1e-3) / 1e9
        out.append({"kernel": label, "launches_per_step": len(calls), "bytes_per_step": round(nbytes),
                    "ms_per_step": round(ms, 3), "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(gbs / HBM_PEAK_GBS, 4)})
        if label.startswith("dwconv7_"):                  # its stated roof: the f32 FMA issue rate
            tf = sum(_dwconv_flops(n, a) for n, a in calls) / (ms * 1e-3) / 1e12
            out[-1]["valu"] = {"bound": "valu_f32_fma", "achieved": round(tf, 2), "peak": VALU_F32_FMA_PEAK_TFLOPS,
                               "unit": "TFLOP/s", "frac": round(tf / VALU_F32_FMA_PEAK_TFLOPS, 4)}
    return out


def dist1_child(opt, grad_comm=None, timeout=420):
    """`secondary.dist1` / `secondary.dist1_bf16`: configs[3]'s PER-RANK step on this one GPU -- the same workload with the
    whole N > 1 machinery live on RCCL at world size 1 (SEI_FORCE_EXCHANGE=1: process group, reduce-scatter of every
    gradient chunk, Adam on the share, all-gather of the updated weights, early release from inside the graph), hence
    stored gradients and no optimizer step inside the GEMMs. `grad_comm`: the exchanged bucket's dtype (f32 = train.py's
    default; bf16 = the compressed exchange with the deep levels' gradients stored as bf16 straight into the exchange
    buffer). Run as a CHILD process (a fresh interpreter whose environment selects the backend before its first GPU call;
    this process never re-execs), its JSON line embedded here."""
    import socket
    import subprocess
    grad_comm = grad_comm or opt.grad_comm
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    # (SEI_EXCHANGE_IN_PLACE=1: with one rank the in-place collectives are no-ops, which is the rehearsal's point -- the
    # out-of-place default would add whole-bucket staging copies here that shrink to 1/N of that at N ranks)
    env = dict(os.environ, SEI_FORCE_EXCHANGE="1", SEI_EXCHANGE_IN_PLACE="1", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0",
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0",
               SEI_BENCH_FULL=os.path.join(ROOT, "gpurun_out", f"bench_full_dist1_{grad_comm}.json"))
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", "10", "--warmup", "3", "--batch",
           str(opt.batch), "--no-secondary", "--no-cpu-baseline", "--grad-comm", grad_comm, "--grad-comm-mode",
           opt.grad_comm_mode]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not line:
            return {"error": f"child exited {r.returncode}", "stderr_tail": r.stderr[-600:]}
        child = json.loads(line[-1])
    except Exception as exc:                                  # the headline line must survive a failing rehearsal
        return {"error": f"{type(exc).__name__}: {exc}"[:600]}
    keep = {k: child.get(k) for k in ("value", "unit", "steps", "warmup", "ms_per_step", "dtype", "roofline", "roofline_hbm",
                                      "graph_kernel_nodes_per_step")}
    keep["workload"] = ("BASELINE configs[3], one rank of it: configs[1]'s step with the gradient exchange LIVE on RCCL at "
                        "world size 1 (reduce_scatter_tensor / all_gather_into_tensor through ProcessGroupNCCL, sharded "
                        f"FlatAdam, early release; stored {'bf16' if grad_comm == 'bf16' else 'f32'} gradients, Adam not "
                        "fused into the GEMMs): the per-rank compute + plumbing cost of the 8-GPU step, without the wire time")
    keep["grad_allreduce"] = child["config"].get("grad_allreduce")
    keep["launch"] = child["config"].get("launch")
    nparams = (child.get("config") or {}).get("parameters") or 645063043
    keep["projected_from_world1_rccl"] = project_ranks(keep, nparams, opt.batch, grad_comm)
    return keep


# xGMI on an MI355X node: every pair of GPUs is joined by ONE link (7 links per GPU, ~153.6 GB/s each in both directions
# together). The projection is computed for both readings of that figure -- 76.8 GB/s per direction, and 153.6 -- at 70 %
# of the link rate (the DESIGN section 6 assumption; RCCL has not run two ranks of this code: falsifiers listed there).
XGMI_LINK_GBS = (76.8, 153.6)
XGMI_EFFICIENCY = 0.7
EARLY_RELEASE_MS, EARLY_RELEASE_SHARE = 3.5, 0.83      # the bottleneck block's gradients leave 3.5 ms before the graph ends


def project_ranks(world1, nparams, batch, grad_comm):
    """PROJECTED, not measured: the N-rank step from the world-1 rehearsal (`secondary.dist1*`: everything but the wire
    time is live there). step_N = step_1 - Adam over the bucket x (1 - 1/N) + exposed wire time; reduce-scatter of the
    gradients (4 or 2 B per parameter) and all-gather of the updated bf16 weight copies (2 B per parameter), each moving
    1/N of its buffer over each of the N - 1 links of a rank; the early-released 83 % of the reduce-scatter hides under
    the last 3.5 ms of the backward. Returns {"n2": ..., "n8": ...} with (low, high) over the two link-rate readings."""
    step1 = world1.get("ms_per_step")
    adam = next((f["ms_per_step"] for f in world1.get("roofline_hbm") or [] if str(f.get("kernel", "")).startswith("adam_vec")), None)
    if step1 is None or adam is None:
        return None
    out = {"label": "projected from world-1 RCCL (not measured)", "inputs": {"step_ms_world1": step1, "adam_whole_bucket_ms": adam,
           "grad_bytes_per_param": 2 if grad_comm == "bf16" else 4, "weight_bytes_per_param": 2,
           "link_GBps_per_direction": list(XGMI_LINK_GBS), "link_efficiency": XGMI_EFFICIENCY}}
    for n in (2, 8):
        steps = []
        for rate in XGMI_LINK_GBS:
            per_link = lambda nbytes: nbytes / n / (rate * 1e9 * XGMI_EFFICIENCY) * 1e3         # ms: 1/N of the buffer per link
            rs = per_link((2 if grad_comm == "bf16" else 4) * nparams)
            ag = per_link(2 * nparams)
            hidden = min(EARLY_RELEASE_SHARE * rs, EARLY_RELEASE_MS)
            steps.append(step1 - adam * (1.0 - 1.0 / n) + (rs - hidden) + ag)
        slow, fast = max(steps), min(steps)
        out[f"n{n}"] = {"ms_per_step": [round(fast, 2), round(slow, 2)],
                        "images_per_s": [round(n * batch / slow * 1e3, 0), round(n * batch / fast * 1e3, 0)]}
    return out


LINE_LIMIT = 7800            # the driver keeps an 8,081-character tail of stdout: the whole headline line must fit in it

_ROOFLINE_KEEP = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch",
                  "traffic_source", "launches_per_step", "avg_launch_us", "gemm_ms_per_step", "algorithmic_gflop_per_step",
                  "gflop_per_step_inside_hbm_bound_launches")


def _short(text, limit):
    text = str(text)
    return text if len(text) <= limit else text[:limit - 3] + "..."


def compact_line(full, limit=LINE_LIMIT):
    """The contract's one line from the full result object: the headline with its `roofline`, `roofline_hbm` and
    `cpu_baseline`, and each secondary series reduced to {value, ms_per_step, steps, dtype, frac[, nodes]}. Pure (no GPU):
    tests/test_bench_line.py builds it from a recorded run and asserts the size and the required keys. If the line is
    still over `limit`, prose fields are shortened first, then dropped -- numbers never."""
    out = {k: full[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                "scaling", "vs_baseline", "dtype", "data") if k in full}
    out["config"] = dict(full.get("config") or {})
    roof = full.get("roofline")
    out["roofline"] = None if roof is None else {k: roof[k] for k in _ROOFLINE_KEEP if k in roof}
    hbm = full.get("roofline_hbm")
    if hbm is not None:
        out["roofline_hbm"] = [{("kernel" if k == "kernel" else k): (v.split(" (")[0] if k == "kernel" else v)
                                for k, v in fam.items() if k not in ("peak", "unit")} for fam in hbm]
        out["roofline_hbm_peak"] = {"peak": HBM_PEAK_GBS, "unit": "GB/s"}
    if full.get("graph_kernel_nodes_per_step") is not None:
        out["graph_kernel_nodes_per_step"] = full["graph_kernel_nodes_per_step"]
    sec = full.get("secondary")
    if sec is not None:
        out["secondary"] = {}
        for key, entry in sec.items():
            if "error" in entry:
                out["secondary"][key] = {"error": _short(entry["error"], 120)}
                continue
            small = {k: entry.get(k) for k in ("value", "ms_per_step", "steps", "dtype") if entry.get(k) is not None}
            r = entry.get("roofline")
            if r:
                small["roofline"] = {"frac": r.get("frac"), "bound": r.get("bound"), "gemm_ms_per_step": r.get("gemm_ms_per_step")}
            if entry.get("graph_kernel_nodes_per_step") is not None:
                small["nodes"] = entry["graph_kernel_nodes_per_step"]
            proj = entry.get("projected_from_world1_rccl")
            if proj:                                    # projected, not measured: labelled as such on the line too
                small["projected_not_measured"] = {k: proj[k]["images_per_s"] for k in ("n2", "n8") if k in proj}
            out["secondary"][key] = small
        out["secondary_full"] = "one earlier stdout line per series ({\"series\": name, ...}) and " + str(full.get("full_file"))
    for key in ("cpu_baseline", "cpu_baseline_swinir"):
        if key in full:
            out[key] = dict(full[key])
    line = json.dumps(out)
    # prose gives way first (never a number): samples, sources, labels, workload text
    for path, keep in ((("cpu_baseline_swinir", "sample"), 160), (("cpu_baseline", "sample"), 240),
                       (("roofline", "traffic_source"), 120), (("roofline", "kernel"), 100),
                       (("config", "workload"), 120), (("config", "optimizer"), 40), (("config", "launch"), 40),
                       (("config", "grad_allreduce"), 60)):
        if len(line) <= limit:
            break
        node = out.get(path[0])
        if isinstance(node, dict) and isinstance(node.get(path[1]), str):
            node[path[1]] = _short(node[path[1]], keep)
            line = json.dumps(out)
    for key in ("secondary_full", "cpu_baseline_swinir", "secondary", "graph_kernel_nodes_per_step"):
        if len(line) <= limit:
            break
        out.pop(key, None)
        line = json.dumps(out)
    return line


def emit(full):
    """Rank 0's output: the full series first (one line each), the full object to a side file, the headline LAST."""
    path = os.environ.get("SEI_BENCH_FULL") or os.path.join(ROOT, "gpurun_out", "bench_full.json")
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            json.dump(full, f, indent=1)
        full["full_file"] = os.path.relpath(path, ROOT)
    except OSError as exc:                                  # a read-only tree must not cost the run its line
        full["full_file"] = f"(not written: {exc})"
    for key, entry in (full.get("secondary") or {}).items():
        print(json.dumps({"series": key, **entry}), flush=True)
    print(compact_line(full), flush=True)


class Leg:
    """One configured training job on this rank: model, loss, optimizer, resident synthetic pairs, step()."""

    def __init__(self, opt, dtype, device, rank, world):
        import parallel
        from losses import get_loss
        from models import _ops, get_model
        from optim import FlatAdam
        from physics import get_physics
        self.opt, self.dtype, self.world = opt, dtype, world
        sr = opt.task == "sr"
        args = reference_args(device, opt.hidden, opt.scales, opt.task, opt.sr_factor if sr else None,
                              "Transformer" if opt.arch == "swinir" else "Convolutional")
        if opt.full256:
            args.Loss__crop_training_pairs = False
        _ops.set_compute_dtype(dtype)
        torch.manual_seed(0)
        physics = get_physics(args, device)
        self.model = model = get_model(args, physics, device)
        model.to(device)
        model.train()
        self.backbone = backbone = model.get_backbone()
        self.nparams = sum(p.numel() for p in backbone.parameters())
        exchanging = parallel.exchange_active()              # several ranks, or SEI_FORCE_EXCHANGE=1 (secondary.dist1)
        parallel.broadcast_parameters(backbone.flat_params)
        self.loss_fn = loss_fn = get_loss(args, physics)
        # "auto" = f32, train.py's own default (--grad_comm_dtype f32): the bf16-compressed exchange is opt-in in both
        self.comm_dtype = torch.bfloat16 if opt.grad_comm == "bf16" else torch.float32
        self.reducer = reducer = parallel.FlatGradientReducer(backbone.flat_grads, comm_dtype=self.comm_dtype,
                                                              mode=opt.grad_comm_mode) if exchanging else None
        self.optimizer = optimizer = FlatAdam(model, lr=1e-4, betas=(0.9, 0.999), reducer=reducer)

        # synthetic 256x256 pairs, resident in HBM before the timed region
        g = torch.Generator().manual_seed(1234 + rank)
        self.side = side = CROP * opt.sr_factor if sr else 256   # SR: the dataset's _HOTFIX crop (datasets/__init__.py:84-85)
        self.x = x = torch.rand((opt.batch, 3, side, side), generator=g).to(device)
        torch.manual_seed(4321 + rank)
        torch.cuda.manual_seed(4321 + rank)
        self.y = y = physics(x)
        self.physics = physics

        def eager_step(x=x, y=y):
            optimizer.zero_grad()
            loss = loss_fn(x=x, y=y, model=model)
            loss.backward()
            if reducer is not None:
                reducer.reduce_async()
            optimizer.step()
            return loss

        self.step, self.step_on, self.graphed, self.early_event = eager_step, eager_step, None, None
        if opt.graph:
            from graphs import GraphedLossStep
            ys = 256 if opt.full256 else CROP
            early = reducer is not None and os.environ.get("SEI_NO_EARLY_RELEASE") != "1"
            self.graphed = graphed = GraphedLossStep(loss_fn, model, optimizer, (opt.batch, 3, ys, ys),
                                                     early_release=early,
                                                     fuse_optimizer=reducer is None and opt.fuse_optimizer,
                                                     fuse_min_numel=opt.fuse_min_numel,
                                                     direct_bf16_grads=opt.direct_bf16_grads, count_nodes=True)
            if early and graphed.early_grads is not None:        # the bottleneck block's gradients leave early
                self.early_event = graphed.early_grads[0]
                reducer.set_early_range(graphed.early_grads[1:])

            def step(x=x, y=y):
                loss = graphed(x, y)
                if reducer is not None:
                    reducer.reduce_async(early=self.early_event, direct=bool(graphed.direct_views))
                optimizer.step()
                return loss

            self.step = self.step_on = step

    def feed_from_device_cache(self, pairs):
        """Every step takes the next batch of an endless epoch loop over datasets.device_cache.DeviceResidentPairs (the
        reference's per-item dataset path, src/datasets/__init__.py:67-90, produced once and kept in HBM)."""
        from datasets import SyntheticPairs
        from datasets.device_cache import DeviceResidentPairs
        cache = DeviceResidentPairs(SyntheticPairs(self.physics, self.x.device, length=pairs), self.physics, crop_size=256)
        inner, batch = self.step_on, self.opt.batch

        def epochs():
            while True:
                yield from cache.batches(batch, shuffle=True, drop_last=True)

        stream = epochs()
        self.step = lambda: inner(*next(stream))

    def timed(self, warmup, steps, fence):
        for _ in range(warmup):
            self.step()
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            last = self.step()
        fence()
        return time.perf_counter() - t0, float(last.detach())

    def snapshot_state(self):
        st = self.optimizer.state[self.backbone.flat_params]
        shadow = getattr(self.backbone, "flat_shadow", None)
        return (self.backbone.flat_params.clone(), st["exp_avg"].clone(), st["exp_avg_sq"].clone(),
                None if shadow is None else shadow.clone(), int(st["step"]))

    def restore_state(self, state):
        from models import _ops
        st = self.optimizer.state[self.backbone.flat_params]
        self.backbone.flat_params.copy_(state[0])
        st["exp_avg"].copy_(state[1])
        st["exp_avg_sq"].copy_(state[2])
        st["step"] = state[4]
        if state[3] is not None:
            self.backbone.flat_shadow.copy_(state[3])
        _ops.weights_updated(self.backbone, plain_shadow_written=state[3] is not None)
        torch.cuda.synchronize()

    def graph_nodes(self):
        """Kernel nodes of the captured step (each costs ~1.5 us of launch structure on replay), or None when eager."""
        if self.graphed is None or self.graphed.node_counts is None:
            return None
        return self.graphed.node_counts[0]

    def eager_twin_step(self):
        """The captured step's launch set issued eagerly (for rocprofv3 --pmc, which attributes counters per kernel
        dispatch and cannot see inside a graph replay): stored merged weight gradients, the deep levels' optimizer step in
        the epilogue of their GEMMs, then the Adam launch for the rest of the bucket -- what `--pmc-twin` times."""
        from models import _ops
        fused = self.graphed is not None and bool(self.graphed.fused_views)
        if fused:
            _ops.set_fused_adam(*self.graphed.fused_table, owner=self.backbone)
            self.optimizer.prepare_step()
        try:
            if self.graphed is not None and self.graphed.store_weight_grads:
                self.backbone.zero_grad_flat(store_weight_grads=True)
            else:
                self.optimizer.zero_grad()
            loss = self.loss_fn(x=self.x, y=self.y, model=self.model)
            loss.backward()
        finally:
            if fused:
                _ops.set_fused_adam(None, None, owner=self.backbone)
        if self.reducer is not None:
            self.reducer.reduce_async()
        self.optimizer.step()
        return loss

    def record_one_step(self):
        """One eager step with every native call logged (streaming families) and every GEMM launch recorded with its
        FLOPs, issued exactly as the timed (graphed) step issues them: merged weight gradients STORE."""
        import _native
        from models import _joint, _ops
        _ops.profile_gemms(True)
        _native.record_calls(True)
        keep_arena, _joint.KEEP_ARENA = _joint.KEEP_ARENA, True      # (the joint backward's 3B-row activations are operands of the re-issues)
        fused = self.graphed is not None and bool(self.graphed.fused_views)
        if fused:                              # as the captured step: the bottleneck pair is stepped inside its GEMMs
            _ops.set_fused_adam(*self.graphed.fused_table, owner=self.backbone)
            self.optimizer.prepare_step()
        try:
            if self.graphed is not None and self.graphed.store_weight_grads:
                self.backbone.zero_grad_flat(store_weight_grads=True)
            else:
                self.optimizer.zero_grad()
            keep = self.loss_fn(x=self.x, y=self.y, model=self.model)
            keep.backward(retain_graph=True)   # keeps the saved activations (GEMM operands) alive for the replay
        finally:
            _joint.KEEP_ARENA = keep_arena
            if fused:
                _ops.set_fused_adam(None, None, owner=self.backbone)
        if self.reducer is not None:
            self.reducer.reduce_async()
        self.optimizer.step()
        log = _native.record_calls(False)
        records = _ops.profile_gemms(False)
        torch.cuda.synchronize()
        return keep, records, log


def gemm_roofline(records, dtype, reps=3):
    """MFMA-bound family. The bottleneck level's weight-gradient GEMMs that carry the optimizer step in their epilogue
    move 26 bytes per output element under 1.7 kFLOP of matrix work: they are HBM-bound and are booked with the streaming
    families (stream_roofline), FLOPs and all; the level-3 ones (K = 3456) are MFMA-bound and are timed here. (Re-issuing
    any of them steps their weights again: `rooflines` restores the optimizer state afterwards.)"""
    import _native
    total_ms, flops, abytes, amodelled = 0.0, 0.0, 0.0, 0
    hbm_side = lambda r: r[1] == "sei_gemm_bf16nt_dw2_adam" and not adam_gemm_is_mfma_bound(r[2])
    riding = sum(fl for fl, entry, a in records if hbm_side((fl, entry, a)))
    records = [r for r in records if not hbm_side(r)]
    for fl, entry, cargs in records:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        _native.call(entry, *cargs)                 # warm
        e0.record()
        for _ in range(reps):
            _native.call(entry, *cargs)
        e1.record()
        e1.synchronize()
        total_ms += e0.elapsed_time(e1) / reps
        flops += fl
        nb = _gemm_bytes(entry, cargs)
        if nb is not None:
            abytes, amodelled = abytes + nb, amodelled + 1
        if os.environ.get("SEI_GEMM_TABLE"):            # per-launch table for kernel work (tools/, not the bench line)
            with open(os.environ["SEI_GEMM_TABLE"], "a") as f:
                ints = [a for a in cargs if isinstance(a, int) and 0 <= a < (1 << 24)]
                us = 1e3 * e0.elapsed_time(e1) / reps
                f.write(f"{entry} {ints} {us:.1f} us {fl / us / 1e6:.1f} TF\n")
    achieved = flops / (total_ms * 1e-3) / 1e12
    peak = MFMA_PEAK_TFLOPS[dtype]
    names = sorted({entry for _, entry, _ in records})
    if names == ["sei_gemm_bf16_ex"]:
        kernels = "gemm_bf16_kernel<*> (register-staged bf16 MFMA, f32 operands in HBM)"
    elif any(n.startswith(("sei_rowgemm_", "sei_tokgrad_")) for n in names):
        kernels = ("rowgemm_kernel<*> / tokgrad_kernel (token-streaming GEMMs of the Swin blocks' linear layers: K = 192-576, "
                   "HBM-bound -- see the same family in roofline_hbm) + gemm_bf16nt_kernel<*> (3x3 convolutions as implicit "
                   "GEMMs, their tap-batched weight gradients)")
    elif dtype == "bf16":
        kernels = ("gemm_bf16pq_kernel<*> (quadrant schedule, deep levels) + gemm_bf16nt_kernel<*> (128x128 loop, "
                   "everything else) + gemm_bf16_kernel<*> (K<64 layers)")
    else:
        kernels = "gemm_f32_kernel<*>"
    return {"bound": "mfma", "kernel": kernels, "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
            "frac": round(achieved / peak, 4), "traffic": None,
            # operands once + results once, averaged over the launches (the figure `traffic` is to be read against)
            "algorithmic_bytes_per_launch": round(abytes / amodelled) if amodelled else None,
            "algorithmic_bytes_launches_modelled": amodelled, "traffic_source": None,
            "launches_per_step": len(records), "avg_launch_us": round(1e3 * total_ms / len(records), 2),
            "gemm_ms_per_step": round(total_ms, 2), "algorithmic_gflop_per_step": round(flops / 1e9, 1),
            "gflop_per_step_inside_hbm_bound_launches": round(riding / 1e9, 1),
            "timed_with": "HIP events on the launch stream around back-to-back re-issues of every GEMM launch of one "
                          "step (recorded arguments), right after the timed region"}


def rooflines(leg, dtype, ms_step, pmc_traffic=False):
    """Roofline legs of one configured job, after its timed region: one eager step records every launch's entry point
    and arguments (the tensors stay alive in the autograd graph / the allocator's pool), then each family is re-issued
    back to back between HIP events on the launch stream: device time with a full queue, free of host-side gaps, same
    shapes and data as the timed steps. Re-issued optimizer launches (the Adam kernel, the weight-gradient GEMMs whose
    epilogue applies the step) would step the weights again with the same scalars, so the job's parameters, moments and
    bf16 shadow are put back afterwards."""
    state = leg.snapshot_state()
    keep, records, log = leg.record_one_step()
    roofline = gemm_roofline(records, dtype)
    roofline_hbm = stream_roofline(log)
    del keep
    leg.restore_state(state)
    pmc_file = os.path.join(ROOT, "profiles", PMC_TRAFFIC_FILE)
    if pmc_traffic and os.path.exists(pmc_file):
        pmc = json.load(open(pmc_file))                 # PMC counters cannot be read live; see the file
        roofline["traffic"] = round(pmc["traffic_bytes_per_launch"])
        roofline["traffic_source"] = pmc.get("traffic_source") or (
            f"bytes beyond L2 per GEMM launch from committed rocprofv3 --pmc passes (profiles/{PMC_TRAFFIC_FILE}: "
            "FETCH_SIZE x2 + WRITE_SIZE, gfx950 corrections)")
    # (the token-streaming GEMMs are listed under both roofs: counted once)
    accounted = roofline["gemm_ms_per_step"] + sum(f["ms_per_step"] for f in roofline_hbm
                                                   if "also booked in the MFMA family" not in f["kernel"])
    roofline_hbm.append({"kernel": "not attributed (torch fills / adds / copies / RNG, zero fills inside GEMM "
                                   "entry points are counted with the GEMMs, launch gaps)",
                         "ms_per_step": round(ms_step - accounted, 3)})
    return roofline, roofline_hbm


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="images per GPU per step")
    ap.add_argument("--hidden", type=int, default=32)
    ap.add_argument("--scales", type=int, default=5)
    ap.add_argument("--cpu-baseline", action=argparse.BooleanOptionalAction, default=True)
    ap.add_argument("--cpu-batch", type=int, default=4, help="batch of the CPU baseline (BASELINE.md section 3: 4)")
    ap.add_argument("--cpu-steps", type=int, default=3, help="timed CPU steps after one warm-up step")
    ap.add_argument("--profile-gemms", action=argparse.BooleanOptionalAction, default=True,
                    help="the roofline legs (GEMM family + streaming families) after the timed region")
    ap.add_argument("--secondary", action=argparse.BooleanOptionalAction, default=True,
                    help="N=1: also time a short run in the reference's own arithmetic (float32 GEMMs)")
    ap.add_argument("--grad-comm", choices=["auto", "f32", "bf16"], default="auto",
                    help="dtype of the exchanged gradient bucket (auto = f32, as train.py's --grad_comm_dtype default; bf16 is the "
                         "opt-in compressed exchange of both)")
    ap.add_argument("--grad-comm-mode", choices=["all_reduce", "rs_ag"], default="rs_ag",
                    help="N > 1: rs_ag = reduce-scatter the gradient chunks, Adam on this rank's 1 / N share, all-gather the "
                         "updated weights (bf16 copies for the GEMM weights: sharded optimizer step, train.py's default "
                         "too); all_reduce = every rank steps the whole bucket")
    ap.add_argument("--fuse-optimizer", action=argparse.BooleanOptionalAction, default=True,
                    help="one GPU, bf16, hipGraph: apply Adam to the 1x1-convolution weights of the two deepest levels (98.8 %% "
                         "of the parameters) in the epilogue of the GEMM that produces their gradient")
    ap.add_argument("--direct-bf16-grads", action=argparse.BooleanOptionalAction, default=True,
                    help="N > 1 with a bf16 exchange: the deep levels' weight-gradient GEMMs write bf16 into the exchange "
                         "buffer (no float32 copy, no cast pass)")
    ap.add_argument("--fuse-min-numel", type=int, default=1 << 24, help="smallest weight that takes its step that way")
    ap.add_argument("--graph", action=argparse.BooleanOptionalAction, default=True,
                    help="replay forward+backward as one hipGraph (the eager launch path otherwise)")
    ap.add_argument("--pmc-twin", action="store_true",
                    help="profiling aid: capture as usual, then run the steps as the captured step's EAGER twin (same launch "
                         "set: stored merged weight gradients, Adam inside the deep weight-gradient GEMMs), so that "
                         "rocprofv3 --pmc can attribute counters to the kernels of the timed configuration")
    ap.add_argument("--task", choices=["deblurring", "sr"], default="deblurring",
                    help="deblurring = BASELINE configs[1] (the headline); sr = configs[2] (x4 by default), a "
                         "secondary series: pairs (48r x 48r, 48 x 48) as the reference's dataset hands them over")
    ap.add_argument("--sr-factor", type=int, default=4)
    ap.add_argument("--arch", choices=["unet", "swinir"], default="unet",
                    help="unet = ConvolutionalModel (configs[1..3], the headline); swinir = the reference's default "
                         "backbone (configs[4]: --arch swinir --task sr --sr-factor 2)")
    ap.add_argument("--full256", action="store_true",
                    help="secondary series of SURVEY 8d: --no-Loss__crop_training_pairs, the network sees the whole "
                         "256x256 pair (28x the pixels of the default 48-crop); use a small --batch")
    ap.add_argument("--dtype", choices=["f32", "bf16", "bf16x3"], default="bf16",
                    help="arithmetic type of the 1x1-conv GEMMs (f32 = parity mode, bf16 = throughput mode, bf16x3 = split-bf16 "
                         "parity mode: three bf16 MFMA products per float32 product)")
    opt = ap.parse_args()

    import parallel
    rank, local_rank, world = parallel.init_from_env()
    if world != opt.gpus:
        raise SystemExit(f"--gpus {opt.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    device = f"cuda:{local_rank % torch.cuda.device_count()}"     # (% only matters for shared-GPU rehearsals)
    torch.cuda.set_device(device)
    import torch.distributed as dist

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    sr = opt.task == "sr"
    leg = Leg(opt, opt.dtype, device, rank, world)
    if opt.pmc_twin:
        leg.step = leg.eager_twin_step
    fused_opt = leg.graphed is not None and bool(leg.graphed.fused_views)
    elapsed, loss_value = leg.timed(opt.warmup, opt.steps, fence)
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    ms_step = 1e3 * elapsed / opt.steps

    roofline, roofline_hbm = (None, None)
    if opt.profile_gemms:
        default_cfg = (opt.dtype == "bf16" and not sr and not opt.full256 and opt.hidden == 32 and opt.scales == 5
                       and opt.batch == 32 and opt.arch == "unet")
        roofline, roofline_hbm = rooflines(leg, opt.dtype, ms_step, pmc_traffic=default_cfg)
    graph_nodes = leg.graph_nodes()

    nparams, side, comm_dtype = leg.nparams, leg.side, leg.comm_dtype
    exchanging = leg.reducer is not None
    graphed_early = leg.early_event is not None
    secondary = None
    if opt.secondary and world == 1 and not exchanging and opt.dtype == "bf16" and not opt.full256 \
            and opt.arch == "unet" and not sr:
        del leg                                             # frees the bf16 job's buckets before the next one
        torch.cuda.empty_cache()
        secondary = {}

        def short_run(key, over, dtype, warmup, steps, what, feed=None):
            """One more configured job on the same launch path: `warmup` + `steps` steps, its own roofline objects."""
            o2 = argparse.Namespace(**vars(opt))
            for k_, v_ in over.items():
                setattr(o2, k_, v_)
            leg2 = Leg(o2, dtype, device, rank, world)
            if feed is not None:
                feed(leg2)
            el2, loss2 = leg2.timed(warmup, steps, fence)
            ms2 = 1e3 * el2 / steps
            entry = {"value": round(o2.batch * steps / el2, 2), "unit": "images/s", "steps": steps, "warmup": warmup,
                     "ms_per_step": round(ms2, 2), "dtype": dtype, "workload": what, "parameters": leg2.nparams,
                     "batch": o2.batch, "final_loss": loss2}
            if opt.profile_gemms:
                entry["roofline"], entry["roofline_hbm"] = rooflines(leg2, dtype, ms2)
            entry["graph_kernel_nodes_per_step"] = leg2.graph_nodes()
            secondary[key] = entry
            del leg2
            torch.cuda.empty_cache()

        # the reference's own arithmetic, first-class: 5 warm-up + 20 timed steps, roofline against the f32 MFMA peak
        short_run("f32", {}, "f32", 5, 20,
                  "BASELINE configs[1] with exact-f32 MFMA GEMMs (v_mfma_f32_32x32x2_f32): the reference's own arithmetic, "
                  "the mode every 1e-4 parity claim is made in; same launch path (hipGraph replay, flat-bucket Adam)")
        short_run("bf16x3", {}, "bf16x3", 3, 10,
                  "BASELINE configs[1] with every float32 GEMM evaluated as three bf16 MFMA products of bf16 head / remainder "
                  "operands (csrc/bf16x3.hip): 16 mantissa bits per operand, held to the float32 mode's 1e-4 bars "
                  "(tests/test_loss_gpu.py::test_default_unet_step_vs_oracle[bf16x3]); same launch path as the f32 series")
        # SURVEY 8(d)'s other two series of configs[1]: the reference's default batch, and the un-cropped pairs
        short_run("b8", {"batch": 8}, "bf16", 2, 5,
                  "BASELINE configs[1] at the reference's default batch 8 (demo/train.py:53)")
        short_run("full256", {"batch": 16, "full256": True}, "bf16", 1, 3,
                  "SURVEY 8(d) series 'full-256': --no-Loss__crop_training_pairs (src/losses/__init__.py:203-205), the "
                  "network sees the whole 256x256 pair (28x the pixels of the default 48-crop), batch 16")
        # BASELINE configs[2] and configs[4] on the same line
        short_run("sr4", dict(task="sr", sr_factor=4, arch="unet"), "bf16", 2, 5,
                  "BASELINE configs[2]: super-resolution x4 noise=5, proposed loss, pairs 192x192 / 48x48, the same "
                  "ConvolutionalModel with its x4 pre-upsampler (the network runs at 192x192: 16x the pixels of configs[1])")
        short_run("swinir_sr2", dict(task="sr", sr_factor=2, arch="swinir"), "bf16", 2, 5,
                  "BASELINE configs[4] on one GPU: SwinIR backbone (embed 180, 6 x 6 blocks, window 8: deepinv.models.SwinIR "
                  "as src/models/__init__.py:51-74, training mode with stochastic depth), sr x2, proposed loss, pairs 96x96 / "
                  "48x48; PARITY UNPINNED (deepinv / timm absent: oracle/swinir_path.py restates the published network)")
        # N1: the headline step fed by the GPU-resident pair cache (train.py --device_cache): every step draws its pairs'
        # 256-crops from HBM-resident (x, y) and Loss.forward crops 48 out of them, instead of replaying one resident batch
        short_run("device_cache", {}, "bf16", 3, 20,
                  "BASELINE configs[1] with every step's batch gathered from the GPU-resident pair cache "
                  "(datasets/device_cache.py, 256 synthetic pairs = 8 batches per epoch; src/datasets/__init__.py:67-90)",
                  feed=lambda lg: lg.feed_from_device_cache(256))
        secondary["dist1"] = dist1_child(opt, "f32")
        secondary["dist1_bf16"] = dist1_child(opt, "bf16")

    if rank == 0:
        images = opt.batch * world * opt.steps
        out = {
            "metric": ("training images/sec, proposed-loss super-resolution" if sr else
                       "training images/sec (256x256 crops), proposed-loss deblur") +
                      (", SwinIR backbone" if opt.arch == "swinir" else ""),
            "value": round(images / elapsed, 2), "unit": "images/s", "n_gpus": world, "steps": opt.steps,
            "warmup": opt.warmup, "ms_per_step": round(ms_step, 2), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": opt.dtype, "data": "synthetic",
            "config": {"workload": (f"BASELINE configs[4]: SwinIR backbone (embed 180, 6 x 6 blocks, window 8; "
                                    f"deepinv.models.SwinIR as src/models/__init__.py:51-74), {opt.task}"
                                    + (f" x{opt.sr_factor}" if sr else "") + ", proposed loss (SURE + scale-EI), "
                                    f"pairs {side}x{side} / " + ("48x48" if sr else "cropped to 48")) if opt.arch == "swinir"
                                   else (f"BASELINE configs[2]: super-resolution x{opt.sr_factor} noise=5, proposed loss "
                                         f"(SURE + scale-EI), pairs {side}x{side} / 48x48, ConvolutionalModel "
                                         f"hidden={opt.hidden} scales={opt.scales}") if sr else
                                   ("BASELINE configs[1]: deblurring Gaussian_R2 noise=5, proposed loss (SURE + "
                                    "scale-EI), 256x256 pairs " +
                                    ("NOT cropped (full-256 series)" if opt.full256 else "cropped to 48 in Loss.forward") +
                                    f", ConvolutionalModel hidden={opt.hidden} scales={opt.scales}"),
                       "parameters": nparams, "batch_per_gpu": opt.batch, "global_batch": opt.batch * world,
                       "parallelism": f"dp{world}",
                       "optimizer": "Adam (fused, flat bucket" + ("; the deep levels' weights are stepped in the epilogue "
                                                                  "of their weight-gradient GEMMs)" if fused_opt else ")"),
                       "grad_allreduce": None if not exchanging else
                       f"{str(comm_dtype).replace('torch.', '')}, " +
                       ("reduce-scatter + Adam on 1/N shares + all-gather of the updated weights (sharded step)"
                        if opt.grad_comm_mode == "rs_ag" else "all_reduce, every rank steps the whole bucket"),
                       "launch": ("hipGraph replay of forward+backward (random draws made eagerly into static buffers)"
                                  + (", early gradient release" if graphed_early else "")) if opt.graph else "eager",
                       "final_loss": loss_value},
            "roofline": roofline,
            "roofline_hbm": roofline_hbm,
            "graph_kernel_nodes_per_step": graph_nodes,
        }
        if secondary is not None:
            out["secondary"] = secondary
        if opt.cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(opt.cpu_batch, opt.hidden, opt.scales, opt.cpu_steps)
            if secondary is not None and "swinir_sr2" in secondary:
                out["cpu_baseline_swinir"] = cpu_baseline_swinir(opt.cpu_batch)
        emit(out)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
