#!/bin/bash
# What do SQ_ACTIVE_INST_VALU / _VALU2 count?  The issue-rate microbenchmark under the counters: pure simple-class, pure complex-class and mixed streams.
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05/calib; mkdir -p $OUT
timeout -k 10 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VALU2 SQ_INSTS_VALU SQ_BUSY_CU_CYCLES --output-format csv -d $OUT/a -- $GRAFT_REPO_ROOT/tools/microbench/issue_rates > $OUT/a.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/b -- $GRAFT_REPO_ROOT/tools/microbench/issue_rates > $OUT/b.log 2>&1
python3 - <<'PY'
import csv, glob, collections, os
acc = collections.defaultdict(dict)
for fn in glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r05/calib/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        acc[(r["Kernel_Name"].split("(")[0], int(r["Grid_Size"]), r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
seen = set()
for (k, g, d), c in sorted(acc.items()):
    if not any(k.endswith(n) for n in ("k_fma", "k_cvti", "k_mixhf", "k_mix21", "k_mix12", "k_pkmul", "k_readlane", "k_sadd", "k_mixhs", "k_muls")): continue
    key = (k, g, tuple(sorted(c)))
    if key in seen: continue
    seen.add(key)
    print(f"{k:12s} grid {g:8d} " + "  ".join(f"{n}={v:.3e}" for n, v in sorted(c.items())))
PY
