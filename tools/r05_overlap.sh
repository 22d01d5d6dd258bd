#!/bin/bash
cd $GRAFT_REPO_ROOT
for ov in 0 1; do echo "== DOOMGPU_RASTER_OVERLAP=$ov"; DOOMGPU_RASTER_OVERLAP=$ov tools/fs_bench.sh "segs auto" "2 3" --no-side-legs; done
DOOMGPU_RASTER_OVERLAP=1 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "path_320 or heavy_map or overlapped or checksum_of or rerendering or auto_front" 2>&1 | tail -3
