#!/bin/bash
# counter passes over the encode probe (k_encode, 256 frames 1080p): what the CU, the TA and the L1 are doing
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_enc; rm -rf $O; mkdir -p $O
i=0
# (the TA_* counters hang rocprofv3 on this pool -- 7 minutes of silence, then the run is killed -- and are left out;
#  every pass runs under its own timeout and prints a line so that a stuck pass cannot take the call with it)
for set in "GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCP_LATENCY_sum TCP_GATE_EN1_sum" \
           "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VALU"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --pmc $set --output-format csv -d $O/p$i -o p -- python3 $R/tools/probe_enc.py ${1:-synth} > $O/p$i.log 2>&1 || { echo pass $i failed; tail -3 $O/p$i.log; }
  echo pass $i done
done
python3 $R/tools/pmc_summarise.py $O/summary.json $O/p1 $O/p2 $O/p3 > /dev/null
python3 - <<PY
import json
d=json.load(open("$O/summary.json"))["k_encode"]
for k,v in sorted(d.items()):
    if k.endswith("_per_launch"): print("%-44s %.4g" % (k[:-11], v))
PY
rm -rf $O/p1 $O/p2 $O/p3
