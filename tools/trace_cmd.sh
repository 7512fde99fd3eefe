#!/bin/bash
# tools/trace_cmd.sh NAME CMD... -- rocprofv3 kernel stats of CMD, summary of the top kernels printed and kept under gpurun_out/r04/trace_NAME/
name=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
cd $R; mkdir -p gpurun_out/r04/trace_$name
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04/trace_$name -o t -- "$@" > gpurun_out/r04/trace_$name/run.txt 2>&1
grep -v "rocprofv3\|SQLite3\|^W2026\|^E2026" gpurun_out/r04/trace_$name/run.txt | tail -3 | cut -c1-600
f=$(find gpurun_out/r04/trace_$name -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("kernels: %d names, %.3f ms in total" % (len(rows), tot / 1e6))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:16]:
    print("%8.3f ms %6d calls %9.1f us avg  %s" % (float(r["TotalDurationNs"]) / 1e6, int(r["Calls"]), float(r["AverageNs"]) / 1e3, r["Name"][:100]))
PY
find gpurun_out/r04/trace_$name -name "*.csv" ! -name "*kernel_stats.csv" -delete; find gpurun_out/r04/trace_$name -name "*.db" -delete
