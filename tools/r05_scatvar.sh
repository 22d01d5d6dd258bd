#!/bin/bash
# dg_fe_scatter variants: parity (the GPU tier's front-end tests under DOOMGPU_LIB), then kernel traces at config 3 and 2 through the device column walk, against the working-tree library
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for v in "$@"; do
  export DOOMGPU_LIB=$GRAFT_REPO_ROOT/build/variants/libdoomgpu_$v.so
  [ "$v" = base ] && unset DOOMGPU_LIB
  if [ "$v" != base ]; then timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "full_camera_path or heavy_map or vanilla or widths or capacity" 2>&1 | tail -2 || exit 1; fi
  for c in 3 2; do
    OUT=gpurun_out/r05/scatvar/$v$c; mkdir -p $OUT
    DOOMGPU_RASTER_OVERLAP=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 bench.py --config $c --front-end device --steps 20 --warmup 2 --no-cpu-baseline --no-resident --no-host-frames --no-latency --no-side-legs > $OUT.json 2> $OUT.err || { echo "$v FAILED"; tail -3 $OUT.err; exit 1; }
    echo "== $v config $c"; cat $OUT/*/*_kernel_stats.csv | cut -d, -f1-4 | grep "dg_fe"
  done
done
