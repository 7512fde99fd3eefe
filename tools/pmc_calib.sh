#!/bin/bash
# tools/pmc_calib.sh -- on the GPU box: known-byte calibration of FETCH_SIZE / WRITE_SIZE for the access widths of the matcher's kernels
# (tools/probes/pmc_calib.hip: every kernel moves exactly 1 GiB once).  Separate --pmc passes, as the guide
# prescribes.  Writes gpurun_out/profiles/<ROUND>/pmc_calibration.json (copy into profiles/<ROUND>/): per kernel the counter's bytes, the known
# bytes and their ratio -- bench.py's pmc_traffic() reads the factors from there.   Usage: tools/pmc_calib.sh [ROUND]   (default r06)
cd "$(dirname "$0")/.."
OUT=gpurun_out/profiles/${1:-r06}
mkdir -p $OUT
export TMPDIR=/tmp
BIN=$PWD/tools/probes/pmc_calib
[ -x $BIN ] && [ $BIN -nt tools/probes/pmc_calib.hip ] || hipcc --offload-arch=gfx950 -O3 -o $BIN tools/probes/pmc_calib.hip     # (built here or on the box; the binary is not tracked)
$BIN > $OUT/pmc_calibration_timing.txt 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_calib_$c
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pmc_calib_$c -- $BIN > /dev/null 2>&1
done
python - "$OUT" <<'EOP'
import csv, glob, json, sys, collections
out = sys.argv[1]
known = 8192 * 1024 * 128
res = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("gpurun_out/pmc_calib_%s/*/*counter_collection.csv" % c)[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == c and r["Kernel_Name"].startswith("calib_"):
            acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        res[k]["%s_KiB_per_launch" % c] = [round(x, 1) for x in v]
        res[k]["%s_bytes_over_known" % c] = round(sum(v) / len(v) * 1024.0 / known, 4)
res = dict(sorted(res.items()))
res["_known_bytes_per_launch"] = known
res["_what"] = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, KiB) of tools/probes/pmc_calib: each kernel moves exactly 1 GiB once "
                "(4 x the Infinity Cache); *_bytes_over_known = counter bytes / known bytes, i.e. multiply a counter by 1 / that to get bytes")
json.dump(res, open("%s/pmc_calibration.json" % out, "w"), indent=1)
print(json.dumps(res, indent=1))
EOP
rm -rf gpurun_out/pmc_calib_FETCH_SIZE gpurun_out/pmc_calib_WRITE_SIZE
cat $OUT/pmc_calibration_timing.txt
