#!/bin/bash
# builds the kernels of a git revision (default HEAD) into rf_inv_amd/lib/librfgpu_A.so for tools/ab.sh
REV=${1:-HEAD}
R=$(cd $(dirname $0)/.. && pwd)
W=$(mktemp -d)
git -C $R worktree add -f $W $REV > /dev/null 2>&1 || exit 1
make -C $W/rf_inv_amd/csrc -s 2>&1 | grep -E "error"
cp $W/rf_inv_amd/lib/librfgpu.so $R/rf_inv_amd/lib/librfgpu_A.so
git -C $R worktree remove --force $W; git -C $R worktree prune
git -C $R log --oneline -1 $REV
