#!/bin/bash
# builds the kernels of a git revision (default HEAD) into tools/_ab/librfgpu_A.so for tools/ab.sh
# (tools/_ab/ is git-ignored; it travels to the GPU box with the gpurun snapshot -- delete it when done)
REV=${1:-HEAD}
R=$(cd $(dirname $0)/.. && pwd)
W=$(mktemp -d)
git -C $R worktree add -f $W $REV > /dev/null 2>&1 || exit 1
make -C $W/rf_inv_amd/csrc -s 2>&1 | grep -E "error"
mkdir -p $R/tools/_ab
cp $W/rf_inv_amd/lib/librfgpu.so $R/tools/_ab/librfgpu_A.so
git -C $R worktree remove --force $W; git -C $R worktree prune
git -C $R log --oneline -1 $REV
