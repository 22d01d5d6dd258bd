#!/bin/bash
cd $GRAFT_REPO_ROOT
tools/fs_bench.sh "segs" "2 3" --no-side-legs
python3 -m pytest tests/test_gpu_parity.py tests/test_seg_walk_limits.py tests/test_hand_wad.py -m gpu -x -q -k "path_320 or heavy_map or per_view or doom2 or more_ or widest or round_room or random_views" 2>&1 | tail -3
