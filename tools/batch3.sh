#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in "$@"; do
export AGMV_HIP_LIB=$GRAFT_REPO_ROOT/tools/variants/libagmv_hip_$v.so
timeout -k 10 400 python -m pytest tests/test_gpu_hotpath.py -x -q -m gpu -k "encode or roundtrip or lut or fuzz" > gpurun_out/b3_${v}_pytest.log 2>&1
echo "$v pytest rc=$? $(tail -1 gpurun_out/b3_${v}_pytest.log)"
done
unset AGMV_HIP_LIB
timeout -k 10 500 python tools/probe_multi.py synth,noise3,flat,noise BASE "$@" $EXTRA 2>&1 | grep -v amdgpu.ids | tee gpurun_out/b3_probe.txt
