#!/bin/bash
# seg-walk variants: phase clocks of dg_fs_frame (instrumented build), then kernel traces of build/variants/libdoomgpu_<v>.so on the doom2-scale map and config 3
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
B="python3 bench.py --config 3 --steps 10 --warmup 1 --no-cpu-baseline --no-resident --no-host-frames --no-latency --no-side-legs --front-end segs"
echo "== phase clocks, doom2-scale map"; DOOMGPU_LIB=$GRAFT_REPO_ROOT/build/variants/libdoomgpu_fstiming.so timeout -k 10 300 $B --synth-map 2002:32x24:500 2>&1 >/dev/null | grep -A24 "fs timing"
for v in "$@"; do
  export DOOMGPU_LIB=$GRAFT_REPO_ROOT/build/variants/libdoomgpu_$v.so
  [ "$v" = base ] && unset DOOMGPU_LIB
  for m in "--synth-map 2002:32x24:500" ""; do
    OUT=gpurun_out/r05/fsvar/$v$(echo $m | tr -dc 0-9 | head -c4); mkdir -p $OUT
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- $B $m > $OUT.json 2> $OUT.err || { echo "$v FAILED"; tail -3 $OUT.err; exit 1; }
    echo "== $v ${m:-config 3}"; cat $OUT/*/*_kernel_stats.csv | cut -d, -f1-4 | grep "dg_fs"
  done
done
