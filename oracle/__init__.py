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

Parity pin status: PINNED for every in-tree reference function (goldens G1..G9, G11 are reference
outputs). UNPINNED for the deepinv-owned glue (EILoss, GaussianNoise), which is
restated from its documented behaviour only -- see DESIGN.md.
"""
