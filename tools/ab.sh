#!/bin/bash
# A/B of library variants on one GPU box: tools/ab.sh "<variant> <variant> ..." <sweep args...>
# (variant "" = the shipped library; others are granne_amd/lib/libgranne_hip_<variant>.so from tools/build_variant.sh)
cd "$(dirname "$0")/.."
VARIANTS=$1; shift
export TMPDIR=/tmp
for v in $VARIANTS; do
  if [ "$v" = "ship" ]; then unset GRANNE_HIP_LIB; else export GRANNE_HIP_LIB=$PWD/granne_amd/lib/libgranne_hip_$v.so; fi
  echo "== variant $v"
  python tools/sweep.py "$@" 2>&1 | grep -v amdgpu.ids
done
