#!/bin/bash
# kernel-trace stats of one short bench run (per-kernel average durations)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pb -o p -- python3 $R/bench.py --no-cpu-baseline --steps 2 --warmup 1 > $R/gpurun_out/pb.log 2>&1
f=$(find $R/gpurun_out/pb -name 'p_kernel_stats.csv' | sort | sed -n 1p)
cut -d, -f1-4 "$f" | grep -v "at::native\|rocclr" | sed -n 1,14p
