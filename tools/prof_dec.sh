#!/bin/bash
# per-kernel times of parser + k_decode, both forms (tools/probe_dec.py), on T x 1080p of the given content kind
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf $R/gpurun_out/prof_dec && mkdir -p $R/gpurun_out/prof_dec
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_dec -- python3 $R/tools/probe_dec.py ${1:-synth} > $R/gpurun_out/prof_dec/run.log 2>&1
f=$(find $R/gpurun_out/prof_dec -name '*kernel_stats.csv' | head -1)
cat "$f" | cut -c1-150
tail -2 $R/gpurun_out/prof_dec/run.log
find $R/gpurun_out/prof_dec -name '*.csv' -size +300k -delete
