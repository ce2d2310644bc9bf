"""The N > 1 training step on the REAL collective backend (RCCL, `nccl`) with one rank: SEI_FORCE_EXCHANGE=1 builds the
process group, the FlatGradientReducer and the sharded FlatAdam at WORLD_SIZE = 1 and takes none of the single-process
short cuts, so `reduce_scatter_tensor` / `all_gather_into_tensor` / `all_reduce`, the process group's stream and the
early-release side stream all run as they will on 8 GPUs (configs[3]; gloo takes another ordering branch in
parallel.FlatGradientReducer._exchange). Replaces the reference's nn.DataParallel hook, src/models/__init__.py:142-145."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "dist1_worker.py")


def _free_port():
    """A port nobody holds right now (each case starts its own rendezvous: the port of the previous child may still be in
    TIME_WAIT)."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(extra):
    env = dict(os.environ, SEI_FORCE_EXCHANGE="1", HSA_ENABLE_IPC_MODE_LEGACY="0", WORLD_SIZE="1", RANK="0",
               LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    env.pop("SEI_DIST_BACKEND", None)                         # the default on a GPU box: nccl = RCCL
    r = subprocess.run([sys.executable, WORKER] + extra, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


@pytest.mark.parametrize("extra", [
    ["--mode", "rs_ag", "--comm", "f32"],                      # bench.py --gpus N / train.py defaults: sharded step
    ["--mode", "rs_ag", "--comm", "bf16"],                     # compressed exchange, cast pass
    ["--mode", "rs_ag", "--comm", "bf16", "--direct", "1"],    # ... with the big gradients written as bf16 by their GEMM
    ["--mode", "all_reduce", "--comm", "f32"],                 # every rank steps the whole bucket
    ["--mode", "rs_ag", "--comm", "f32", "--dtype", "f32", "--early", "0"]])
def test_exchange_path_on_rccl_with_one_rank(extra):
    """Three hipGraph-replayed steps: after each, parameters, both Adam moments and the bf16 copies of the model stepped
    through reduce-scatter -> Adam on the share -> all-gather (or all-reduce -> Adam) equal, BIT FOR BIT, those of a twin
    model stepped by the plain single-process FlatAdam on the same gradients; the bucket the backend delivered equals the
    bucket that went in (world 1: the sum is the input; bf16 exchange: its rounding); the collectives were really issued."""
    out = _run(extra)
    assert out["backend"] == "nccl", out
    sharded = "rs_ag" in extra
    assert out["sharded"] == sharded and out["chunks"] >= 3
    if "--early" not in extra and "--direct" not in extra:
        assert out["early"], "the early-release range was not planned: the side-stream path did not run"
    assert out["direct"] == ("--direct" in extra)
    for rec in out["steps"]:
        assert rec["params"] and rec["exp_avg"] and rec["exp_avg_sq"] and rec["shadow"], out
        assert rec.get("exchange_exact", True), out
    calls = out["calls"]
    if sharded:
        assert out["sharded_chunks"] >= out["chunks"] - 2
        assert calls["reduce_scatter_tensor"] >= 3 * out["sharded_chunks"]
        assert calls["all_gather_into_tensor"] >= 3 * out["sharded_chunks"]
        if "bf16" in extra[3:4] or "--dtype" not in extra:    # bf16 GEMM mode: the GEMM weights travelled as bf16 copies alone
            assert any(rec["stale"] for rec in out["steps"]) and not out["stale_after_consolidate"]
    else:
        assert calls["all_reduce"] >= 3 * out["chunks"]
    assert out["grad_norm"] > 0 and all(abs(rec["loss"]) < 1e3 for rec in out["steps"])
