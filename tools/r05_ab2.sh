#!/bin/bash
# tools/r05_ab2.sh <variants...>: kbench parity line of each variant, then three interleaved timing rounds (1280x800 and 320x200, 1 000-frame launches)
cd $GRAFT_REPO_ROOT
export KBENCH_SIZES=${KBENCH_SIZES:-1280x800x1000,320x200x1000}
for v in "$@"; do [ "$v" = base ] && continue; echo -n "$v parity: "; DOOMGPU_LIB=$GRAFT_REPO_ROOT/build/variants/libdoomgpu_$v.so python3 tests/manual/gpu_kbench.py 2>&1 | grep "B=" | sed 's/|.*//' | tr '\n' ' '; echo; done
tools/ab_variants.sh "$@"
