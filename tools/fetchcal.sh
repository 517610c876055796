#!/bin/bash
# FETCH_SIZE calibration (tools/micro/fetchcal.hip): event times, then the same kernels under --pmc FETCH_SIZE
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/fetchcal; rm -rf $O; mkdir -p $O
timeout -k 10 120 $R/tools/micro/fetchcal > $O/times.txt 2>&1; cat $O/times.txt
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc -o p -- $R/tools/micro/fetchcal > $O/pmc.log 2>&1
python3 - "$O" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/pmc/**/*counter_collection.csv", recursive=True):
    per = collections.defaultdict(float); nm = {}
    for r in csv.DictReader(open(f)):
        if "k_cal" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
            per[r["Dispatch_Id"]] += float(r["Counter_Value"]); nm[r["Dispatch_Id"]] = r["Kernel_Name"][:40]
    for d, v in per.items(): acc[nm[d]].append(v)
for k in sorted(acc):
    v = sum(acc[k]) / len(acc[k])
    print("%-40s FETCH_SIZE %.0f KiB per launch = %.3f of the 2 GiB buffer" % (k, v, v * 1024 / (2 << 30)))
PY
rm -rf $O/pmc
