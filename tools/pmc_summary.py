#!/usr/bin/env python3
"""Summarise the rocprofv3 --pmc passes of bench.py into profiles/ (run in the build container).

    rocprofv3 --pmc FETCH_SIZE  --output-format csv -d gpurun_out/pmc_FETCH_SIZE  -o pmc -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-graph --no-profile-gemms
    rocprofv3 --pmc WRITE_SIZE  ...        (one counter group per pass; FETCH_SIZE and WRITE_SIZE do not fit one pass)
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE ...
    python tools/pmc_summary.py gpurun_out profiles/r01_c_pmc

gfx950 corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE and WRITE_SIZE are in KiB; FETCH_SIZE
reports exactly half of the bytes of a wide coalesced stream (16 B per lane, global_load and
global_load_lds alike) -> doubled; WRITE_SIZE is exact for 16-byte-per-lane stores and float atomics.
The counters sit on the L2's memory side, so Infinity-Cache hits are included ("beyond-L2 traffic").
"""
import collections
import csv
import json
import re
import sys


WINDOW = None      # "start=<regex>": dispatches from the first kernel matching on; "endmark=<regex>": whole steps, each ENDING
                   # with a kernel matching (the run's earlier dispatches -- warm-up, capture preparation -- are left out)


def _first_dispatch(rows):
    if not WINDOW:
        return 0
    kind, pat = WINDOW.split("=", 1)
    ids = sorted({int(r["Dispatch_Id"]) for r in rows})
    hits = sorted({int(r["Dispatch_Id"]) for r in rows if re.search(pat, r["Kernel_Name"])})
    if not hits:
        return 0
    if kind == "start":
        return hits[0]
    ends = [h for k, h in enumerate(hits) if k + 1 == len(hits) or hits[k + 1] - h > 8]     # last launch of each group
    period = ends[1] - ends[0] if len(ends) > 1 else ends[0] - ids[0] + 1
    return ends[0] - period + 1


LONG = None        # (regex, microseconds): dispatches of matching kernels that take longer are booked under "<name> [long]"
                   # (the HBM-bound bottleneck launches of a kernel whose other launches are MFMA-bound)


def load(path, counter):
    d = collections.defaultdict(lambda: [0.0, 0, 0.0])
    rows = list(csv.DictReader(open(path)))
    first = _first_dispatch(rows)
    for r in rows:
        if r["Counter_Name"] != counter or int(r["Dispatch_Id"]) < first:
            continue
        k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        k = re.sub(r"\(.*", "", k)
        if LONG and re.search(LONG[0], k) and int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > 1e3 * LONG[1]:
            k += " [long]"
        d[k][0] += float(r["Counter_Value"])
        d[k][1] += 1
        d[k][2] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    return d


def main():
    src, out = sys.argv[1], sys.argv[2]
    title = sys.argv[3] if len(sys.argv) > 3 else "PMC passes of bench.py (bf16, eager launches, 3 steps)"
    exclude = re.compile(sys.argv[4]) if len(sys.argv) > 4 and sys.argv[4] else None   # kernels kept out of the GEMM family
    global WINDOW, LONG
    WINDOW = sys.argv[5] if len(sys.argv) > 5 and sys.argv[5] else None
    if len(sys.argv) > 6:                                   # "regex@us"
        pat, us = sys.argv[6].rsplit("@", 1)
        LONG = (pat, float(us))
    F = load(f"{src}/pmc_FETCH_SIZE/pmc_counter_collection.csv", "FETCH_SIZE")
    W = load(f"{src}/pmc_WRITE_SIZE/pmc_counter_collection.csv", "WRITE_SIZE")
    Mb = load(f"{src}/pmc_SQ_VALU_MFMA_BUSY_CYCLES/pmc_counter_collection.csv", "SQ_VALU_MFMA_BUSY_CYCLES")
    G = load(f"{src}/pmc_SQ_VALU_MFMA_BUSY_CYCLES/pmc_counter_collection.csv", "GRBM_GUI_ACTIVE")
    gem = [k for k in F if ("gemm_" in k or "tokgrad" in k or "mlp_fwd" in k or "mlp_bwd" in k or "mlp128_" in k or "dw_stream" in k)
           and not (exclude and exclude.search(k))]
    n = sum(F[k][1] for k in gem)
    fetch = sum(F[k][0] for k in gem) * 1024 * 2          # KiB -> B, x2 (gfx950 wide-stream correction)
    write = sum(W[k][0] for k in gem if k in W) * 1024
    t_ns = sum(F[k][2] for k in gem)
    busy = sum(Mb[k][0] for k in gem if k in Mb)
    gui = sum(G[k][0] for k in gem if k in G)             # summed over the 8 XCDs
    util = busy / (gui / 8 * 256 * 4) if gui else None
    res = {"kernel_family": "gemm_*_kernel<*> (all GEMM launches)", "launches": n,
           "fetch_bytes_per_launch": fetch / n, "write_bytes_per_launch": write / n,
           "traffic_bytes_per_launch": (fetch + write) / n, "avg_launch_us_under_pmc": t_ns / n / 1e3,
           "mfma_busy_fraction": util,
           "method": "rocprofv3 --pmc, one pass per counter group (FETCH_SIZE; WRITE_SIZE; SQ_VALU_MFMA_BUSY_CYCLES + "
                     "GRBM_GUI_ACTIVE); FETCH_SIZE x2 and KiB->B per MI355X_MICROARCH.md; beyond-L2 traffic "
                     "(Infinity-Cache hits included)" + (f"; kernels matching /{exclude.pattern}/ excluded" if exclude else "")}
    json.dump(res, open(out + "_gemm.json", "w"), indent=1)
    with open(out + "_per_kernel.md", "w") as f:
        f.write("# " + title + "\n\n" + __doc__.split("gfx950")[0] + "\n")
        f.write("GB/s = (FETCH_SIZE x 2 + WRITE_SIZE) / kernel time: beyond-L2 bytes per second (Infinity-Cache hits "
                "included), to set against the 8 TB/s HBM3E peak for the streaming kernels (dwconv7_*, ln_*, sepmap_*, "
                "cast / colsum, adam_vec_kernel). Times are under the profiler (lower clocks than the bench).\n\n")
        f.write("| kernel | launches | fetch GB (x2 corrected) | write GB | time ms | GB/s | frac of 8 TB/s | MFMA busy fraction |\n"
                "|---|---|---|---|---|---|---|---|\n")
        for k in sorted(F, key=lambda k: -F[k][2])[:40]:
            mb, ga = Mb.get(k, [0])[0], G.get(k, [0])[0]
            u = f"{mb / (ga / 8 * 1024):.3f}" if ga and mb else "-"
            fb, wb = F[k][0] * 2048, W.get(k, [0])[0] * 1024
            gbs = (fb + wb) / max(F[k][2], 1)
            f.write(f"| `{k[:70]}` | {F[k][1]} | {fb / 1e9:.2f} | {wb / 1e9:.2f} | "
                    f"{F[k][2] / 1e6:.2f} | {gbs:.0f} | {gbs / 8000:.2f} | {u} |\n")
        f.write("\nGEMM family: " + json.dumps(res) + "\n")
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
