#!/bin/bash
# kernel trace of 'mgm_multi' tiles, 4 per library call (1000 x 1000 x 256): which kernels make up the call
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
mkdir -p $R/gpurun_out/r04/ms_trace
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04/ms_trace -o ms4 -- python tools/ms_batch_stages.py 1000 256 mgm_multi 4 > gpurun_out/r04/ms_trace/run.txt 2>&1
tail -2 gpurun_out/r04/ms_trace/run.txt
f=$(find gpurun_out/r04/ms_trace -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("kernels: %d names, %.3f ms in total" % (len(rows), tot / 1e6))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:22]:
    print("%8.3f ms %6d calls %9.1f us avg  %s" % (float(r["TotalDurationNs"]) / 1e6, int(r["Calls"]), float(r["AverageNs"]) / 1e3, r["Name"][:110]))
PY
find gpurun_out/r04/ms_trace -name "*.csv" ! -name "*kernel_stats.csv" -delete; find gpurun_out/r04/ms_trace -name "*.db" -delete
