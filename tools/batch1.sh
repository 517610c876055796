#!/bin/bash
# round 2, batch 1: TSKIP parity, then timing of the look-up / cache-policy variants in one process
cd $GRAFT_REPO_ROOT
export AGMV_HIP_LIB=$GRAFT_REPO_ROOT/tools/variants/libagmv_hip_TSKIP.so
timeout -k 10 400 python -m pytest tests/test_gpu_hotpath.py -x -q -m gpu -k "encode or roundtrip or lut or fuzz" > gpurun_out/b1_tskip_pytest.log 2>&1
echo "TSKIP pytest rc=$? $(tail -1 gpurun_out/b1_tskip_pytest.log)"
unset AGMV_HIP_LIB
timeout -k 10 500 python tools/probe_multi.py synth,noise3,flat BASE CUR TSKIP A_NOGATHER GMASK1 GMASK2 GMASK3 PIX0 PIX16 PIX17 PIX18 PIX19 LUT1 LUT2 LUT16 BASE 2>&1 | grep -v amdgpu.ids | tee gpurun_out/b1_probe.txt
