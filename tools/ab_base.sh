#!/bin/bash
# Build the library of another commit beside the working tree's, for same-box A/B runs:
#   bash tools/ab_base.sh <commit> <name>   ->  musicgeneration_amd/libmgx_<name>.so   (load with MGX_LIB_PATH=...)
# The commit must have the same MGX_ABI_VERSION as the working tree's Python side.
set -e
C=${1:?commit}; N=${2:?name}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
WT=/tmp/mgx_wt_$N
rm -rf $WT; git -C $ROOT worktree prune; git -C $ROOT worktree add -f --detach $WT $C > /dev/null
(cd $WT && python3 -m musicgeneration_amd._build --force > /dev/null)
cp $WT/musicgeneration_amd/libmgx.so $ROOT/musicgeneration_amd/libmgx_$N.so
git -C $ROOT worktree remove --force $WT
echo $ROOT/musicgeneration_amd/libmgx_$N.so
