#!/bin/bash
cd $GRAFT_REPO_ROOT
export AGMV_HIP_LIB=$GRAFT_REPO_ROOT/tools/variants/libagmv_hip_PROF.so
for k in synth flat noise3; do timeout -k 10 120 python tools/probe_enc.py $k 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/b2_prof.txt
