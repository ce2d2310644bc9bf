"""rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/pmc_step_gemms.py -> per-shape beyond-L2 traffic against the
algorithmic bytes of the launch (operands once, results once), gfx950 corrections as tools/pmc_summary.py (KiB, FETCH x2).
    python tools/pmc_step_gemms_summary.py gpurun_out/X/fetch/pmc_counter_collection.csv gpurun_out/X/write/pmc_counter_collection.csv out.md"""
import csv, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from exp_tile_sweep_shapes import SHAPES
EPI_NONE, EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_RES, EPI_MUL_DGELU, EPI_ACCUM, EPI_ROWSCALE = range(7)


def per_shape(path, counter):
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    groups, cur = [], None
    for r in rows:
        if "axpy_kernel" in r["Kernel_Name"]:
            cur = []
            groups.append(cur)
        elif cur is not None and "gemm_" in r["Kernel_Name"]:
            cur.append((float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"]))
    return groups


def main():
    fetch, write, out = sys.argv[1:4]
    F, W = per_shape(fetch, "FETCH_SIZE"), per_shape(write, "WRITE_SIZE")
    assert len(F) == len(W) == len(SHAPES), (len(F), len(W), len(SHAPES))
    lines = ["| M x N x K | B stored | epilogue | kernel launches per call | algorithmic MB | fetched MB (x2 corrected) | written MB | beyond-L2 / algorithmic | us |",
             "|---|---|---|---|---|---|---|---|---|"]
    tot_alg = tot_tr = 0.0
    for (M, N, K, brm, epi, count), f, w in zip(SHAPES, F, W):
        reps = 3
        per_call = len(f) // reps
        fb = sum(v for v, _, _ in f[-per_call:]) * 2048.0
        wb = sum(v for v, _, _ in w[-per_call:]) * 1024.0
        us = sum(t for _, t, _ in f[-per_call:]) / 1e3
        out_b = {EPI_BIAS_GELU: 6, EPI_MUL_DGELU: 2}.get(epi, 4)
        in_extra = {EPI_BIAS_RES: 4, EPI_MUL_DGELU: 4}.get(epi, 0)
        alg = 2.0 * K * (M + N) + M * N * (out_b + in_extra)
        tot_alg += count * alg
        tot_tr += count * (fb + wb)
        lines.append(f"| {M} x {N} x {K} | {'(K, N)' if brm else '(N, K)'} | {epi} | {per_call} | {alg / 1e6:.1f} | {fb / 1e6:.1f} | {wb / 1e6:.1f} | "
                     f"{(fb + wb) / alg:.2f} | {us:.1f} |")
    lines.append(f"\nWeighted by launches per step: {tot_tr / 1e9:.2f} GB beyond L2 against {tot_alg / 1e9:.2f} GB algorithmic = {tot_tr / tot_alg:.2f}x.")
    open(out, "w").write("# Beyond-L2 traffic per forward / data-gradient GEMM launch of the timed step (rocprofv3 --pmc, third of three back-to-back launches per shape)\n\n" + "\n".join(lines) + "\n")
    print("\n".join(lines[-12:]))


if __name__ == "__main__":
    main()
