"""Multi-GPU readiness on one GPU: two freshly spawned ranks share the card and exchange gradients over gloo
(RCCL refuses two ranks on one device); same code path as the 8-GPU run except for the backend."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "ddp_worker.py")


def _run(out, world, extra, backend="gloo", more_env=None):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("SEI_DIST_BACKEND", None)
    if backend == "gloo":
        env["SEI_DIST_BACKEND"] = "gloo"
    env.update(more_env or {})
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    if world == 1:
        cmd = [sys.executable, WORKER, "--out", out] + extra
    else:
        port = 29600 + os.getpid() % 300
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), WORKER, "--out", out] + extra
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return [torch.load(os.path.join(out, f"rank{k}.pt")) for k in range(world)]


def relerr(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max())


@pytest.mark.parametrize("extra,tol", [
    (["--dtype", "f32", "--graph", "1", "--mode", "all_reduce"], 2e-5),
    # rs_ag + FlatAdam = the sharded optimizer step: reduce-scatter, Adam on this rank's shares, all-gather of the weights
    (["--dtype", "f32", "--graph", "0", "--mode", "rs_ag"], 2e-5),
    (["--dtype", "bf16", "--graph", "1", "--mode", "rs_ag", "--comm", "bf16", "--hidden", "32"], 2e-2),
    # ... with the big weight gradients written into the bf16 exchange buffer by their GEMM (no f32 copy, no cast pass)
    (["--dtype", "bf16", "--graph", "1", "--mode", "rs_ag", "--comm", "bf16", "--direct-min", "60000"], 2e-2),
    (["--dtype", "bf16", "--graph", "1", "--mode", "all_reduce", "--comm", "bf16", "--direct-min", "60000"], 2e-2)])
def test_two_ranks_on_one_gpu_match_the_single_process_run(tmp_path, extra, tol):
    """Two ranks x half the batch == one process x the whole batch: the summed, world-averaged gradient of the first
    step equals the single-process gradient (to split-K / atomic-order noise; to bf16 rounding when the exchange
    is compressed), the replicas stay bit-identical through the optimizer steps, and the per-step losses agree.
    Sharded step (rs_ag): every rank steps its shares only; the bf16 copies the next forward reads are identical on both
    ranks after every step, the float32 masters and the moments after `consolidate()`."""
    two = _run(str(tmp_path / "w2"), 2, extra)
    one = _run(str(tmp_path / "w1"), 1, extra)                             # (exchange flags are inert at world 1)
    assert two[0]["sharded"] == ("rs_ag" in extra) and not one[0]["sharded"]
    if two[0]["shadow"] is not None:
        assert torch.equal(two[0]["shadow"], two[1]["shadow"])
        if two[0]["sharded"]:                  # 1x1 weights with both extents % 32 == 0 (levels 1 and 2 at hidden 8, every
            assert two[0]["stale"]             # level at hidden 32) travelled as bf16 copies alone: their float32 masters
                                               # were stale on the other rank until consolidate()
    assert torch.equal(two[0]["exp_avg"], two[1]["exp_avg"])
    assert torch.equal(two[0]["params"], two[1]["params"])                  # replicas never diverge
    assert torch.equal(two[0]["grads_step0"], two[1]["grads_step0"])
    assert relerr(two[0]["grads_step0"], one[0]["grads_step0"]) < tol
    # mean of the rank means == the global mean, up to the SURE constant sigma^2 / local batch
    sigma2 = (5 / 255) ** 2
    for l2, l1 in zip(two[0]["losses"], one[0]["losses"]):
        assert abs((l2 + sigma2 / 4) - (l1 + sigma2 / 8)) < max(tol, 1e-4) * abs(l1), (two[0]["losses"], one[0]["losses"])
    assert relerr(two[0]["params"], one[0]["params"]) < 2e-3               # Adam's first steps are sign-like
    assert np.isfinite(two[0]["losses"]).all()


@pytest.mark.skipif(torch.cuda.device_count() < 2 or os.environ.get("SEI_RUN_RCCL2_TESTS") != "1",
                    reason="needs >= 2 GPUs (RCCL refuses two ranks on one device) and SEI_RUN_RCCL2_TESTS=1: this test has never "
                           "run on hardware, so it must not be able to stop a `pytest -x` tier by itself")
@pytest.mark.parametrize("extra", [
    ["--dtype", "f32", "--graph", "0", "--mode", "rs_ag"],                         # sharded optimizer step
    ["--dtype", "f32", "--graph", "0", "--mode", "rs_ag", "--shard", "0"],         # reduce-scatter + all-gather of gradients
    ["--dtype", "bf16", "--graph", "1", "--mode", "rs_ag", "--comm", "bf16", "--direct-min", "60000"]])
def test_in_place_exchange_matches_out_of_place_rccl(tmp_path, extra):
    """Two RCCL ranks on two GPUs: the in-place forms of reduce-scatter / all-gather (SEI_EXCHANGE_IN_PLACE=1: the share is
    this rank's slot of the chunk) against the default separate share buffers -- bit-identical gathered gradient,
    parameters, moments and losses (the same reductions in the same order; only where the result lands differs).
    parallel.FlatGradientReducer._in_place stays opt-in until this has passed on hardware."""
    runs = {}
    for name, env in (("out_of_place", {}), ("in_place", {"SEI_EXCHANGE_IN_PLACE": "1"})):
        runs[name] = _run(str(tmp_path / name), 2, extra, backend="nccl", more_env=env)
    for r in range(2):
        a, b = runs["out_of_place"][r], runs["in_place"][r]
        assert a["losses"] == b["losses"]
        for key in ("grads_step0", "params", "exp_avg"):
            assert torch.equal(a[key], b[key]), (r, key)
    assert torch.equal(runs["in_place"][0]["params"], runs["in_place"][1]["params"])
