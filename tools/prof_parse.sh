#!/bin/bash
# kernel-trace of the decode-side probe: per-kernel average durations
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pp_$1 -o p -- python3 $R/tools/probe_dec.py $1 > $R/gpurun_out/pp_$1.log 2>&1
f=$(find $R/gpurun_out/pp_$1 -name 'p_kernel_stats.csv' | sort | sed -n 1p)
cut -d, -f1-4 "$f" | sed -n 1,14p
grep "decode" $R/gpurun_out/pp_$1.log
