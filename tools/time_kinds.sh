#!/bin/bash
# usage: tools/time_kinds.sh VARIANT KIND...
cd $GRAFT_REPO_ROOT
export AGMV_HIP_LIB=$GRAFT_REPO_ROOT/tools/variants/libagmv_hip_$1.so; shift
for k in "$@"; do timeout -k 10 120 python tools/probe_enc.py $k 2>&1 | grep encode; done
