#!/bin/bash
# non-temporal stores in the column walk, one stream at a time: variants ntscat (dg_fe_scatter's output) and ntcols (dg_fe_columns' scratch) against the product;
# end-to-end frames/s (device front end, config 3) and the kernels under the tracer
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for round in 1 2; do for v in base ntscat ntcols; do
  export DOOMGPU_LIB=$GRAFT_REPO_ROOT/build/variants/libdoomgpu_$v.so; [ "$v" = base ] && unset DOOMGPU_LIB
  echo -n "$v: "; python3 bench.py --config 3 --front-end device --steps 20 --warmup 2 --no-cpu-baseline --no-resident --no-host-frames --no-latency --no-side-legs 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']; print(round(d['value']), 'fps | raster', round(r['mean_launch_ms'],4), 'fe', round(r['front_end_kernels_mean_ms'],4))"
done; done
for v in ntscat ntcols; do
  OUT=gpurun_out/r05/ntfe/$v; mkdir -p $OUT
  DOOMGPU_LIB=$GRAFT_REPO_ROOT/build/variants/libdoomgpu_$v.so timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 bench.py --config 3 --front-end device --steps 20 --warmup 2 --no-cpu-baseline --no-resident --no-host-frames --no-latency --no-side-legs > $OUT.json 2> $OUT.err || exit 1
  echo "== $v"; cat $OUT/*/*_kernel_stats.csv | cut -d, -f1-4 | grep "dg_"
done
