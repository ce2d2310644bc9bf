"""ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT.

A CPU restatement of the reference's proposed-loss training hot path,
used only as the checker: `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
`bench.py` may import it; nothing under `scale-equivariant-imaging_amd/` does.

Two layers, both pinned against golden vectors captured from the reference itself
(`tools/gen_golden.py` -> `tests/golden/`, checked by `tests/test_oracle_golden.py`):

* `oracle.torch_path`  -- the same torch CPU op sequence the reference runs (FFT blur, antialiased
  bicubic `interpolate`, `grid_sample`, FFT "ideal" resamplers, conv/LayerNorm/GELU U-Net, SURE +
  EI loss), written functionally over a state_dict. This is the timed CPU baseline ("port").
* `oracle.closed_form` -- numpy float64 closed forms of the same operators (direct circular
  convolution, explicit antialias weights, explicit bicubic/reflection sampler, DFT-matrix form of
  the FFT resamplers). These are the formulas the HIP kernels implement.

Parity pin status: PINNED for every in-tree reference function (goldens G1..G9, G11..G15 are reference
outputs: G13 = the reference's in-tree R2R / EI loss, which pins the EI arithmetic that `ei_loss` restates from deepinv;
G14 = CropPair on batches; G15 = the reference's own get_loss / Loss.forward / ProposedLoss / SURELoss / SupervisedLoss
end to end on a seeded generator, with deepinv's EILoss / GaussianNoise as flagged shells). UNPINNED, restated from
documented behaviour only: deepinv's own source for EILoss / GaussianNoise (not in the reference tree), SwinIR
(`swinir_path.py`), the PSNR-Y metric (kornia / torchmetrics) -- see DESIGN.md section 2.
"""


def usable_cpus():
    """CPUs this process may actually use: the smaller of the affinity mask and the cgroup quota (cpu.max). A GPU box
    reports 256 logical CPUs but grants 16 (cpu.max = 1600000 100000); torch's default of 128 intra-op threads then
    oversubscribes them 8-fold and the float64 oracle runs an order of magnitude slower (measured: one small U-Net
    forward + backward 9.3 s with 128 threads, 0.76 s with 16)."""
    import os
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def use_all_usable_cpus():
    """Size torch's intra-op pool to `usable_cpus()` (the tests and bench.py's cpu_baseline leg call this before the
    oracle runs); returns the thread count in effect."""
    import torch
    n = usable_cpus()
    if torch.get_num_threads() != n:
        torch.set_num_threads(n)
    return torch.get_num_threads()
