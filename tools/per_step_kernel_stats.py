#!/usr/bin/env python3
"""Per-STEP kernel statistics of the timed region of a bench.py run (set-up, warm-up and capture excluded), from the
rocprofv3 --kernel-trace CSV (run in the build container on the merged gpurun_out files).

The timed region is found from the trace itself: the fused Adam launches (`adam_vec_kernel`, `markers_per_step` of them,
the LAST launches of every step) mark step boundaries; the window runs from the end of the last marker of the step before
the K timed ones to the end of the trace's last marker.

    python tools/per_step_kernel_stats.py <q_kernel_trace.csv> <bench log> <out prefix> "<title>" "<command>" [K] [markers_per_step]
"""
import collections
import csv
import re
import sys

FAMILIES = [("GEMM", r"gemm_|tokgrad|dw_stream"), ("Adam (rest of the bucket)", r"adam_"), ("resampler maps", r"sepmap|cmat_gemm"),
            ("LayerNorm", r"ln_"), ("depthwise 7x7", r"dwconv7"), ("casts / column sums / transposes", r"cast|colsum|transpose"),
            ("conv3x3", r"conv3x3"), ("fused MLP", r"mlp"), ("partial folds", r"fold_"),
            ("physics / loss / prologue", r"blur|scale_resample|scale_params|axpy|sure_|mse_|finish_sums|proposed_draws|crop_window|zero_ranges|adam_scalars"),
            ("torch (ATen / rocclr)", r"at::native|rocclr|Memcpy|Memset|elementwise|CatArray|distribution")]


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*", "", name)[:90]


def main():
    trace, log, out, title, command = sys.argv[1:6]
    K = int(sys.argv[6]) if len(sys.argv) > 6 else 20
    mps = int(sys.argv[7]) if len(sys.argv) > 7 else 3
    rows = list(csv.DictReader(open(trace)))
    for r in rows:
        r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    rows.sort(key=lambda r: r["s"])
    marks = [r for r in rows if "adam_vec_kernel" in r["Kernel_Name"]]
    steps = len(marks) // mps
    if steps <= K:
        raise SystemExit(f"only {steps} steps in the trace, {K} asked for")
    t0, t1 = marks[(steps - K) * mps - 1]["e"], marks[-1]["e"]
    inside = [r for r in rows if r["s"] >= t0 and r["e"] <= t1]
    agg = collections.OrderedDict()
    for r in inside:
        a = agg.setdefault(short(r["Kernel_Name"]), [0, 0])
        a[0] += 1
        a[1] += r["e"] - r["s"]
    busy = sum(v[1] for v in agg.values())
    fam = collections.OrderedDict((f, [0, 0]) for f, _ in FAMILIES)
    fam["other"] = [0, 0]
    for name, (n, ns) in agg.items():
        key = next((f for f, pat in FAMILIES if re.search(pat, name)), "other")
        fam[key][0] += n
        fam[key][1] += ns
    line = next((l.strip() for l in open(log) if l.startswith("{")), "")
    torch_rows = [(n, v) for n, v in agg.items() if re.search(FAMILIES[-1][1], n)]
    with open(out + ".md", "w") as f:
        f.write(f"# {title}\n\nCommand (on the MI355X box): `{command}`\n\nbench.py line of the same run: {line[:1500]}\n\n")
        f.write(f"Timed region only: the last {K} of the trace's {steps} steps ({len(inside)} dispatches; set-up, warm-up and graph "
                f"capture excluded). Wall time of the region {1e-6 * (t1 - t0) / K:.3f} ms per step; kernels busy "
                f"{1e-6 * busy / K:.3f} ms per step over {len(inside) / K:.1f} dispatches per step; idle between dispatches "
                f"{1e-6 * (t1 - t0 - busy) / K:.3f} ms per step.\n\n")
        f.write("| family | dispatches per step | ms per step |\n|---|---|---|\n")
        for k, (n, ns) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
            if n:
                f.write(f"| {k} | {n / K:.1f} | {1e-6 * ns / K:.3f} |\n")
        f.write(f"\ntorch (ATen / rocclr) kernels inside the timed region: {sum(v[0] for _, v in torch_rows) / K:.1f} per step, "
                f"{1e-3 * sum(v[1] for _, v in torch_rows) / K:.1f} us per step"
                + ("" if torch_rows else " -- none") + ".\n\n")
        f.write("| kernel | calls per step | us per step | avg us |\n|---|---|---|---|\n")
        for name, (n, ns) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            f.write(f"| `{name}` | {n / K:.2f} | {1e-3 * ns / K:.1f} | {1e-3 * ns / n:.1f} |\n")
    with open(out + ".csv", "w") as f:
        f.write("kernel,calls_per_step,us_per_step,avg_us\n")
        for name, (n, ns) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            f.write(f"\"{name}\",{n / K:.3f},{1e-3 * ns / K:.2f},{1e-3 * ns / n:.2f}\n")
    print(f"{out}.md: {1e-6 * (t1 - t0) / K:.3f} ms per step wall, {1e-6 * busy / K:.3f} busy, torch kernels per step "
          f"{sum(v[0] for _, v in torch_rows) / K:.1f}")


if __name__ == "__main__":
    main()
