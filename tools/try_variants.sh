#!/bin/bash
# usage: tools/try_variants.sh NAME...   -- encode parity tests + timing probe for tools/variants/libagmv_hip_NAME.so
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  export AGMV_HIP_LIB=$GRAFT_REPO_ROOT/tools/variants/libagmv_hip_$v.so
  timeout -k 10 300 python -m pytest tests/test_gpu_hotpath.py -x -q -m gpu -k "encode or roundtrip or lut" > gpurun_out/var_$v.log 2>&1
  rc=$?
  echo "$v pytest rc=$rc: $(tail -1 gpurun_out/var_$v.log)"
  if [ $rc -ne 0 ]; then tail -30 gpurun_out/var_$v.log; exit 1; fi
  for k in synth noise3; do timeout -k 10 120 python tools/probe_enc.py $k 2>&1 | grep encode; done
done
