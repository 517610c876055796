#!/bin/bash
# usage: tools/time_variants.sh NAME...   -- timing probe only (ablation builds produce wrong bytes by construction)
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  export AGMV_HIP_LIB=$GRAFT_REPO_ROOT/tools/variants/libagmv_hip_$v.so
  for k in synth noise3; do timeout -k 10 120 python tools/probe_enc.py $k 2>&1 | grep encode; done
done
