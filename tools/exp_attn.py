"""Window attention launches of the SwinIR bench step (48 x 48 tokens, 6 heads, 64 and 32 images), timed alone.

    python tools/exp_attn.py            (on the GPU box)
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scale-equivariant-imaging_amd"))
import _native as N  # noqa: E402


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(reps):
        fn()
    t1.record()
    torch.cuda.synchronize()
    return t0.elapsed_time(t1) / reps * 1e3


def main():
    if len(sys.argv) > 1:                                # another build of the library (ablations)
        N.LIB_PATH = os.path.abspath(sys.argv[1])
        print(N.LIB_PATH)
    heads, HP, H, W = 6, 32, 48, 48
    for B in (64, 32):
        M = B * H * W
        gen = torch.Generator(device="cuda").manual_seed(1)
        qkv = torch.randn((M, 3 * heads * HP), device="cuda", generator=gen).bfloat16()
        go = torch.randn((M, heads * HP), device="cuda", generator=gen).bfloat16()
        table = torch.randn((225, heads), device="cuda", generator=gen) * 0.5
        out = torch.empty((M, heads * HP), dtype=torch.bfloat16, device="cuda")
        lse = torch.empty((heads, M), dtype=torch.float32, device="cuda")
        dqkv = torch.empty_like(qkv)
        dtable = torch.zeros_like(table)
        scale = 30 ** -0.5
        for shift in (0, 4):
            f = timed(lambda: N.call("sei_swin_attn_fwd_bf16", qkv.data_ptr(), table.data_ptr(), out.data_ptr(), lse.data_ptr(),
                                     B, H, W, heads, shift, scale))
            b = timed(lambda: N.call("sei_swin_attn_bwd_bf16", qkv.data_ptr(), table.data_ptr(), out.data_ptr(), lse.data_ptr(),
                                     go.data_ptr(), dqkv.data_ptr(), dtable.data_ptr(), B, H, W, heads, shift, scale))
            fb = M * heads * (HP * 2 * 4 + 4)
            bb = M * heads * (HP * 2 * 8 + 4)
            print(f"B={B} shift={shift}: forward {f:.1f} us ({fb / f / 1e6:.2f} TB/s), backward {b:.1f} us ({bb / b / 1e6:.2f} TB/s)",
                  flush=True)


if __name__ == "__main__":
    main()
