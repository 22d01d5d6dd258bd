#!/usr/bin/env python3
"""profiles/issue.json from rocprofv3 --pmc passes of bench.py (or tests/manual/gpu_kbench.py): per launch of dg_raster_tiles, the
wave-instructions by class, the vector instructions that were paired into an issue slot (SQ_ACTIVE_INST_VALU2), the LDS cycles and the
clock the launch ran at (SQ_BUSY_CU_CYCLES per CU / launch duration, both from the same pass).  bench.py turns them into `roofline.issue`.

    tools/issue_counters.py <dir with rocprofv3 --pmc csv output> <W>x<H>x<B> [--out profiles/issue.json] [--grid <lanes>]

Only dispatches of dg_raster_tiles with B frames (grid z) count: the grid size of such a launch is 512 threads x strips x segments x B.
"""
import collections, csv, glob, json, os, sys

def main():
    src, key = sys.argv[1], sys.argv[2]
    out = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "issue.json")
    W, H, B = (int(v) for v in key.split("x"))
    vals, dur = collections.defaultdict(list), []
    grids = collections.Counter()
    rows = []
    for fn in glob.glob(src + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            if "dg_raster_tiles" not in r["Kernel_Name"]:
                continue
            rows.append(r)
            grids[int(r["Grid_Size"])] += 1
    # the launches of B frames are the ones with the largest grid (warm-up / parity launches are smaller)
    want = max(grids)
    for r in rows:
        if int(r["Grid_Size"]) != want:
            continue
        vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
        if r["Counter_Name"] == "SQ_BUSY_CU_CYCLES":
            dur.append((float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    mean = {k: sum(v) / len(v) for k, v in vals.items()}
    need = ["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU2", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM", "SQ_LDS_IDX_ACTIVE", "SQ_BUSY_CU_CYCLES"]
    missing = [k for k in need if k not in mean]
    if missing:
        sys.exit(f"missing counters: {missing} (have {sorted(mean)})")
    clock_ghz = sum(c / 256.0 / ns for c, ns in dur) / len(dur)          # cycles per CU / nanoseconds
    entry = {"valu": mean["SQ_INSTS_VALU"], "valu_paired": mean["SQ_ACTIVE_INST_VALU2"], "salu": mean["SQ_INSTS_SALU"], "lds": mean["SQ_INSTS_LDS"],
             "vmem": mean["SQ_INSTS_VMEM_RD"] + mean["SQ_INSTS_VMEM_WR"], "smem": mean["SQ_INSTS_SMEM"], "lds_cycles": mean["SQ_LDS_IDX_ACTIVE"],
             "busy_cu_cycles": mean["SQ_BUSY_CU_CYCLES"], "clock_ghz": clock_ghz, "launches": len(dur), "grid": want,
             "launch_ns_under_the_counters": sum(ns for _, ns in dur) / len(dur)}
    j = json.load(open(out)) if os.path.exists(out) else {}
    j[key] = entry
    j["_what"] = "per launch of dg_raster_tiles, mean over the launches of one rocprofv3 --pmc run per counter set (tools/issue_counters.py); bench.py: roofline.issue"
    json.dump(j, open(out, "w"), indent=1, sort_keys=True)
    chunks = B * ((W + 63) // 64) * 64 * ((H + 63) // 64)
    print(f"{key}: per chunk VALU {entry['valu'] / chunks:.1f} (paired {entry['valu_paired'] / chunks:.1f}) SALU {entry['salu'] / chunks:.1f} LDS {entry['lds'] / chunks:.1f} "
          f"VMEM {entry['vmem'] / chunks:.2f} SMEM {entry['smem'] / chunks:.2f}; clock {clock_ghz:.2f} GHz; launch {entry['launch_ns_under_the_counters'] / 1e6:.3f} ms under the counters")

main()
