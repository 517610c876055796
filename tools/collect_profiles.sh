#!/bin/bash
# Round profile collection on the GPU box: kernel-trace stats of the default bench, then two --pmc passes
# (FETCH_SIZE, WRITE_SIZE: separate runs, counters only).  Output under gpurun_out/prof_<tag>/.
set -e
tag=${1:-r01}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python3 $R/bench.py --no-cpu-baseline --no-normal-heavy --no-secondary --steps 5 --warmup 2 > $O/bench_trace.log 2>&1
echo trace done
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o p -- python3 $R/bench.py --no-cpu-baseline --no-normal-heavy --no-secondary --steps 2 --warmup 1 > $O/bench_pmc_fetch.log 2>&1
echo fetch done
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o p -- python3 $R/bench.py --no-cpu-baseline --no-normal-heavy --no-secondary --steps 2 --warmup 1 > $O/bench_pmc_write.log 2>&1
echo write done
python3 $R/tools/pmc_summarise.py $O/pmc_fetch_write.json $O/pmc_fetch $O/pmc_write
f=$(find $O/trace -name 't_kernel_stats.csv' | sort | sed -n 1p)
cp "$f" $O/bench_kernel_stats.csv
# the raw per-dispatch traces are large: keep the summaries only
rm -rf $O/trace $O/pmc_fetch $O/pmc_write
tail -1 $O/bench_trace.log
