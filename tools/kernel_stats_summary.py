#!/usr/bin/env python3
"""rocprofv3 --kernel-trace --stats CSV of a bench.py run -> profiles/<name>.csv + .md (run in the build container).

    python tools/kernel_stats_summary.py gpurun_out/prof_q/q_kernel_stats.csv gpurun_out/bench_prof.log \
        profiles/r01_f_bf16_graph_bench_kernel_stats "<title>" "<command>"
"""
import csv
import re
import shutil
import sys

FAMILIES = [("GEMM", r"gemm_|tokgrad|dw_stream"), ("Adam", r"adam_"), ("resampler maps", r"sepmap"), ("LayerNorm", r"ln_"),
            ("depthwise 7x7", r"dwconv7"), ("casts / column sums", r"cast|colsum"), ("fills", r"fill|Fill"),
            ("conv3x3", r"conv3x3"), ("window attention", r"swin_attn"),
            ("pad / pack / partial folds", r"pad_nhwc|unpad|pack_kernel|unpack|rowscale|fold_partials|fold_many"), ("fused MLP", r"mlp_")]


def main():
    stats, log, out, title, command = sys.argv[1:6]
    rows = list(csv.DictReader(open(stats)))
    shutil.copyfile(stats, out + ".csv")
    line = next((l.strip() for l in open(log) if l.startswith("{")), "")
    total = sum(float(r["TotalDurationNs"]) for r in rows) / 1e6
    fam = {}
    for r in rows:
        name = r["Name"]
        key = next((f for f, pat in FAMILIES if re.search(pat, name)), "other")
        fam[key] = fam.get(key, 0.0) + float(r["TotalDurationNs"]) / 1e6
    with open(out + ".md", "w") as f:
        f.write(f"# {title}\n\nCommand (on the MI355X box): `{command}`\n\nbench.py line of the same run: {line}\n\n")
        f.write(f"Total kernel time in the trace {total:.1f} ms. By family (ms over the whole trace): " +
                ", ".join(f"{k} {v:.1f}" for k, v in sorted(fam.items(), key=lambda kv: -kv[1])) + "\n\n")
        f.write("| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|\n")
        for r in rows[:60]:
            name = re.sub(r"\(anonymous namespace\)::", "", r["Name"])
            name = re.sub(r"\(.*", "", name)[:80]
            f.write(f"| `{name}` | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.2f} | "
                    f"{float(r['AverageNs']) / 1e3:.1f} | {float(r['Percentage']):.1f} |\n")


if __name__ == "__main__":
    main()
