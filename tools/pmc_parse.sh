#!/bin/bash
# SQ counters of the parser kernels (k_fp_walk / k_fp_expand) on 256 x 1080p: is the walk bound by vector instruction issue?
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/pmc_parse; rm -rf $O; mkdir -p $O
kind=${1:-synth}
timeout -k 10 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $O/p1 -o p -- python3 $R/tools/probe_parse2.py $kind > $O/p1.log 2>&1 || tail -3 $O/p1.log
python3 $R/tools/pmc_summarise.py $O/summary.json $O/p1 > /dev/null
python3 - "$O/summary.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k in ("k_fp_walk", "k_fp_expand", "k_fp_finish"):
    v = d.get(k, {})
    if not v: continue
    g = v["GRBM_GUI_ACTIVE_per_launch"] / 8          # summed over the 8 XCDs
    print("%-12s %.0f k cycles per launch; per CU cycle: VALU %.3f  SALU %.3f  LDS %.3f instructions; VALU active %.2f of CU-busy cycles; waves waiting %.2f of wave cycles"
          % (k, g / 1e3, v["SQ_INSTS_VALU_per_launch"] / (g * 256), v["SQ_INSTS_SALU_per_launch"] / (g * 256), v["SQ_INSTS_LDS_per_launch"] / (g * 256),
             v["SQ_ACTIVE_INST_VALU_per_launch"] / max(1.0, v["SQ_BUSY_CU_CYCLES_per_launch"]), v["SQ_WAIT_INST_ANY_per_launch"] / max(1.0, v["SQ_WAVE_CYCLES_per_launch"])))
PY
cp $O/summary.json $R/gpurun_out/pmc_parse_$kind.json; rm -rf $O/p1
