#!/bin/bash
# per-kernel times of the parser (fast path + gated robust kernels) on 256 x 1080p of the given content kind
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf $R/gpurun_out/prof_parse2 && mkdir -p $R/gpurun_out/prof_parse2
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_parse2 -- python3 $R/tools/probe_parse2.py ${1:-synth} > $R/gpurun_out/prof_parse2/run.log 2>&1
f=$(find $R/gpurun_out/prof_parse2 -name '*kernel_stats.csv' | head -1)
cat "$f" | cut -c1-160
find $R/gpurun_out/prof_parse2 -name '*.csv' -size +300k -delete
