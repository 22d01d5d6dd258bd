#!/bin/bash
# tools/r05_ab.sh [variants...]: kbench parity + timing of the working-tree library ("base") against build/variants/libdoomgpu_<v>.so, interleaved three times
cd $GRAFT_REPO_ROOT
export KBENCH_SIZES=${KBENCH_SIZES:-1280x800x1000,1280x800x250,320x200x1000}
python3 tests/manual/gpu_kbench.py 2>&1 | grep "B="
tools/ab_variants.sh "$@" base
