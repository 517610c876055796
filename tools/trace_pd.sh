#!/bin/bash
# kernel start/end stamps of the overlapped parse || decode call (is the parse of range k+1 really running beside k_decode of range k?)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf $R/gpurun_out/trace_pd && mkdir -p $R/gpurun_out/trace_pd
T=256 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_pd -- python3 $R/tools/probe_pd.py ${1:-4} > $R/gpurun_out/trace_pd/run.log 2>&1
f=$(find $R/gpurun_out/trace_pd -name '*kernel_trace.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
sel = [r for r in rows if r["Kernel_Name"].startswith(("k_parse", "void k_parse", "void k_decode", "void k_fixup"))]
for r in sel[-40:]:
    print("%-40s q=%s start %10.1f us  dur %8.1f us" % (r["Kernel_Name"][:40], r.get("Queue_Id"), (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
PY
find $R/gpurun_out/trace_pd -name '*.csv' -size +200k -delete
