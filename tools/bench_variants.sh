#!/bin/bash
# the default bench (c3, kernels only) for several library builds in ONE call: boxes of the pool differ by a few percent, so
# only numbers of the same call compare.  usage: tools/bench_variants.sh NAME...   ("BASE" = the product build)
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  if [ "$v" = BASE ]; then unset AGMV_HIP_LIB; else export AGMV_HIP_LIB=$GRAFT_REPO_ROOT/tools/variants/libagmv_hip_$v.so; fi
  timeout -k 10 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-12s' % '$v', d['kernels_ms'], 'k_encode frac', d['roofline']['frac'])"
done
