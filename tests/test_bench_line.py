"""bench.py's last stdout line must stay parseable by the driver: under 8 kB, with the contract's keys.

Round 4's line grew to 26.5 kB (seven secondary series, each with full roofline objects) and the driver recorded
`parsed: null`. `bench.compact_line` builds the headline from the full result object; here it is fed the recorded
full object of that very run (profiles/r04_g_bench_default_line.json) and a worst case with long error strings.
"""
import json
import os

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RECORDED = os.path.join(ROOT, "profiles", "r04_g_bench_default_line.json")

CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                 "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline")
ROOFLINE_KEYS = ("bound", "achieved", "peak", "unit", "frac", "traffic")
CPU_KEYS = ("value", "unit", "cores", "kind", "sample")


def _recorded():
    with open(RECORDED) as f:
        return json.load(f)


def _check(line, full):
    assert "\n" not in line
    assert len(line) < 8000, len(line)
    out = json.loads(line)
    for key in CONTRACT_KEYS:
        assert key in out, key
    for key in ("value", "ms_per_step", "steps", "warmup", "n_gpus", "dtype", "metric", "unit"):
        assert out[key] == full[key]
    assert "model" not in out["config"] and "workload" in out["config"]
    for key in ROOFLINE_KEYS:
        assert key in out["roofline"], key
    assert out["roofline"]["frac"] == full["roofline"]["frac"]
    for key in CPU_KEYS:
        assert key in out["cpu_baseline"], key
    assert out["cpu_baseline"]["value"] == full["cpu_baseline"]["value"]
    return out


def test_recorded_round4_run_compacts_under_8k():
    full = _recorded()
    assert len(json.dumps(full)) > 20000           # the line that broke the driver's parse
    out = _check(bench.compact_line(full), full)
    # every streaming family keeps its numbers; labels are cut to the family's short name
    assert len(out["roofline_hbm"]) == len(full["roofline_hbm"])
    for small, fam in zip(out["roofline_hbm"], full["roofline_hbm"]):
        assert small["ms_per_step"] == fam["ms_per_step"]
        assert small.get("frac") == fam.get("frac")
        assert " (" not in small["kernel"]
    # every secondary series is still on the line, reduced to its figures
    assert set(out["secondary"]) == set(full["secondary"])
    for key, small in out["secondary"].items():
        assert small["value"] == full["secondary"][key]["value"]
        assert small["ms_per_step"] == full["secondary"][key]["ms_per_step"]
        assert small["roofline"]["frac"] == full["secondary"][key]["roofline"]["frac"]
        assert "roofline_hbm" not in small and "workload" not in small


def test_oversized_prose_is_cut_before_any_number():
    full = _recorded()
    full["cpu_baseline"]["sample"] = "s" * 6000
    full["roofline"]["traffic_source"] = "t" * 6000
    full["config"]["workload"] = "w" * 6000
    full["secondary"]["dist1"] = {"error": "e" * 5000, "stderr_tail": "x" * 600}
    full["secondary"]["dist1_bf16"] = {"error": "child exited 1"}
    out = _check(bench.compact_line(full), full)
    assert out["secondary"]["dist1"]["error"].endswith("...")
    assert out["roofline"]["gemm_ms_per_step"] == full["roofline"]["gemm_ms_per_step"]


def test_line_without_optional_objects():
    """N > 1 runs (no secondary, no CPU baseline) and --no-profile-gemms runs still produce a valid line."""
    full = _recorded()
    for key in ("secondary", "cpu_baseline", "cpu_baseline_swinir"):
        full.pop(key)
    full["roofline"], full["roofline_hbm"] = None, None
    line = bench.compact_line(full)
    out = json.loads(line)
    assert len(line) < 2000 and out["roofline"] is None and "secondary" not in out


def test_emit_prints_series_first_and_the_headline_last(tmp_path, capsys, monkeypatch):
    full = _recorded()
    monkeypatch.setenv("SEI_BENCH_FULL", str(tmp_path / "full.json"))
    bench.emit(full)
    lines = capsys.readouterr().out.strip().splitlines()
    assert len(lines) == len(full["secondary"]) + 1
    for ln, key in zip(lines[:-1], full["secondary"]):
        assert json.loads(ln)["series"] == key
    _check(lines[-1], full)
    with open(tmp_path / "full.json") as f:
        assert json.load(f)["secondary"]["sr4"]["roofline_hbm"]


def test_recorded_round6_run_keeps_the_valu_roofline_and_the_projection():
    """Round 6's line (profiles/r06_h_bench_default_full.json: ten secondary series, the depthwise family's VALU roofline,
    the N = 2 / N = 8 projection of the world-1 rehearsal): under 8 kB with those objects on it, the projection labelled
    as not measured."""
    with open(os.path.join(ROOT, "profiles", "r06_h_bench_default_full.json")) as f:
        full = json.load(f)
    out = _check(bench.compact_line(full), full)
    dw = next(fam for fam in out["roofline_hbm"] if fam["kernel"].startswith("dwconv7_"))
    assert dw["valu"]["bound"] == "valu_f32_fma" and 0 < dw["valu"]["frac"] < 1 and dw["valu"]["peak"] == bench.VALU_F32_FMA_PEAK_TFLOPS
    assert set(out["secondary"]) >= {"f32", "bf16x3", "b8", "full256", "sr4", "swinir_sr2", "device_cache", "dist1", "dist1_bf16"}
    for key in ("dist1", "dist1_bf16"):
        proj = out["secondary"][key]["projected_not_measured"]
        assert set(proj) == {"n2", "n8"} and all(lo <= hi for lo, hi in proj.values())
        assert proj["n8"][0] > 3500                                  # BASELINE.json's 8-GPU target, on the pessimistic link reading
        want = bench.project_ranks(full["secondary"][key], 645063043, 32, "bf16" if key.endswith("bf16") else "f32")
        assert proj["n2"] == want["n2"]["images_per_s"] and "not measured" in want["label"]
