#!/bin/bash
# counter passes over the encode probe (k_encode, 256 frames 1080p): what the CU and its L1 (TCP) are doing.
# usage: tools/pmc_enc.sh [kind=synth] [variant]      output: gpurun_out/pmc_enc_<kind>/summary.json + a table on stdout
# At most 4 TCP counters per pass (7 in one pass abort rocprofv3: "Request exceeds the capabilities of the hardware to
# collect"); no TA_* counters (a TA pass went silent for 7 minutes in round 1); every pass under its own timeout.
cd /tmp && export TMPDIR=/tmp
kind=${1:-synth}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_enc_$kind; rm -rf $O; mkdir -p $O
[ -n "$2" ] && export AGMV_HIP_LIB=$R/tools/variants/libagmv_hip_$2.so
i=0
for set in "GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCP_TA_ADDR_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
           "TCP_TD_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TOTAL_READ_sum TCP_TOTAL_ACCESSES_sum" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --pmc $set --output-format csv -d $O/p$i -o p -- python3 $R/tools/probe_enc.py $kind > $O/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $O/p$i.log; }
  echo "pass $i done: $(grep encode $O/p$i.log | tail -1)"
done
python3 $R/tools/pmc_summarise.py $O/summary.json $O/p1 $O/p2 $O/p3 $O/p4 $O/p5 $O/p6 > /dev/null
python3 - <<PY
import json
d=json.load(open("$O/summary.json")).get("k_encode", {})
for k,v in sorted(d.items()):
    if k.endswith("_per_launch"): print("%-48s %.5g" % (k[:-11], v))
PY
rm -rf $O/p1 $O/p2 $O/p3 $O/p4 $O/p5 $O/p6
