#!/bin/bash
# Round 5's profile of the default benchmark command: kernel trace, HBM traffic, the issue counters (profiles/issue.json), and kernel traces of
# configs 2 and 3 through the device seg walk.  tools/r05_profile.sh <tag>; output under gpurun_out/r05/prof_<tag>/ (one rocprofv3 run per counter set,
# --pmc passes without any trace domain, each under its own timeout).
TAG=${1:-final}
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05/prof_$TAG
BENCH="python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-resident --no-host-frames --no-latency --no-side-legs"
export TMPDIR=/tmp
mkdir -p $OUT
run() { name=$1; shift; timeout -k 10 240 "$@" > $OUT/$name.log 2>&1 && echo "$name ok" || echo "$name FAILED ($?)"; }
run trace rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH
run fetch rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $BENCH
run write rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $BENCH
run sq1 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU2 SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES --output-format csv -d $OUT/pmc_sq1 -- $BENCH
run sq2 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_BRANCH --output-format csv -d $OUT/pmc_sq2 -- $BENCH
run tcc rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_tcc -- $BENCH
for c in 2 3; do
  run trace_segs_c$c rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_segs_c$c -- python3 bench.py --config $c --front-end segs --steps 20 --warmup 2 --no-cpu-baseline --no-resident --no-host-frames --no-latency --no-side-legs
  cat $OUT/trace_segs_c$c/*/*_kernel_stats.csv | cut -d, -f1-4,6,7 | head -12
done
# config 2 with the raster launch on the kernel stream (no overlap): the front-end kernels' own durations
DOOMGPU_RASTER_OVERLAP=0 timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_segs_c2_serial -- python3 bench.py --config 2 --front-end segs --steps 20 --warmup 2 --no-cpu-baseline --no-resident --no-host-frames --no-latency --no-side-legs > $OUT/trace_segs_c2_serial.log 2>&1 && { echo "config 2, one stream"; cat $OUT/trace_segs_c2_serial/*/*_kernel_stats.csv | cut -d, -f1-4,6,7 | head -9; }
# the doom2-scale map through the seg walk
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_segs_doom2 -- python3 bench.py --config 3 --synth-map 2002:32x24:500 --front-end segs --steps 10 --warmup 1 --no-cpu-baseline --no-resident --no-host-frames --no-latency --no-side-legs > $OUT/trace_segs_doom2.log 2>&1 && { echo "doom2-scale map"; cat $OUT/trace_segs_doom2/*/*_kernel_stats.csv | cut -d, -f1-4,6,7 | head -9; }
python3 tools/issue_counters.py $OUT 1280x800x1000 --out $OUT/issue.json
python3 tools/summarize_prof.py $OUT "default bench (round 5)" > $OUT/summary.md
head -20 $OUT/summary.md
