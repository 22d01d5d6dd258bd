#!/bin/bash
# Build the COMMITTED kernels (git HEAD, or $1 = a commit) as build/variants/libdoomgpu_head.so for same-box A/B runs (tools/ab_variants.sh head base).
set -e
REV=${1:-HEAD}
ROOT=$(git rev-parse --show-toplevel)
TMP=$(mktemp -d)
git -C $ROOT archive $REV doom-rust-renderer_amd/csrc include data | tar -x -C $TMP
make -s -C $TMP/doom-rust-renderer_amd/csrc OUT=$TMP/libdoomgpu.so
mkdir -p $ROOT/build/variants && cp $TMP/libdoomgpu.so $ROOT/build/variants/libdoomgpu_${2:-head}.so
rm -rf $TMP
echo "built $REV -> build/variants/libdoomgpu_${2:-head}.so"
