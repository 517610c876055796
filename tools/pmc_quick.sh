#!/bin/bash
# one counter pass over the encode probe: tools/pmc_quick.sh KIND VARIANT "COUNTERS"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmcq; rm -rf $O; mkdir -p $O
export AGMV_HIP_LIB=$R/tools/variants/libagmv_hip_$2.so
timeout -k 10 150 rocprofv3 --pmc $3 --output-format csv -d $O/p -o p -- python3 $R/tools/probe_enc.py $1 > $O/log 2>&1 || tail -3 $O/log
grep encode $O/log | tail -1
python3 $R/tools/pmc_summarise.py $O/s.json $O/p > /dev/null
python3 -c "
import json
d=json.load(open('$O/s.json')).get('k_encode', {})
for k,v in sorted(d.items()):
    if k.endswith('_per_launch'): print('%-48s %.5g' % (k[:-11], v))
"
rm -rf $O
