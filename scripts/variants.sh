#!/bin/bash
# Compare tuning builds on one GPU box: scripts/variants.sh <lib.so> ... (paths relative to the repo root; "base" =
# the shipped libartp.so).  Headline step + per-kernel times of the 2^22-state launches for each.
cd $GRAFT_REPO_ROOT
for L in "$@"; do
  echo "== $L"
  if [ "$L" = base ]; then unset ARTP_LIB; else export ARTP_LIB=$GRAFT_REPO_ROOT/$L; fi
  bash scripts/quick_perf.sh 2>&1 | grep -v amdgpu.ids
done
