#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in "$@"; do
export AGMV_HIP_LIB=$GRAFT_REPO_ROOT/tools/variants/libagmv_hip_$v.so
for k in flat synth; do echo "== $v $k"; timeout -k 10 120 python tools/probe_enc.py $k 2>&1 | grep -v amdgpu.ids; done
done | tee gpurun_out/b5_prof.txt
