#!/usr/bin/env python3
"""tools/inflight_union.py <kernel_trace.csv> <out.json> [kernel substring] -- what one k_mgm_bands launch costs when tiles
are in flight, MEASURED: the union of the busy intervals of all its launches in a rocprofv3 --kernel-trace of
`bench.py --no-cpu --no-job ...` divided by the number of launches (replaces the derived
`achieved_3_in_flight`).  The launches that overlap at least one other launch of the same kernel (tiles in flight) and
those that do not (the single-stream passes of the same run) are reported separately."""
import csv
import json
import sys


def main():
    path, out = sys.argv[1], sys.argv[2]
    key = sys.argv[3] if len(sys.argv) > 3 else "k_mgm_bands"
    iv = []
    with open(path) as f:
        for r in csv.DictReader(f):
            if key in r["Kernel_Name"]:
                iv.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    iv.sort()
    n = len(iv)
    if not n:
        raise SystemExit("no %s launches in %s" % (key, path))
    overl = [False] * n
    for i in range(n):                                       # sorted by start: i overlaps a later j iff start_j < end_i
        j = i + 1
        while j < n and iv[j][0] < iv[i][1]:
            overl[i] = overl[j] = True
            j += 1

    def union(items):
        tot, cs, ce = 0, None, None
        for s, e in items:
            if cs is None:
                cs, ce = s, e
            elif s <= ce:
                ce = max(ce, e)
            else:
                tot += ce - cs
                cs, ce = s, e
        if cs is not None:
            tot += ce - cs
        return tot
    fl = [v for v, o in zip(iv, overl) if o]
    al = [v for v, o in zip(iv, overl) if not o]
    res = {"kernel": key, "trace": path.split("/")[-1], "launches_total": n}
    if fl:
        res.update(launches=len(fl), union_ms=round(union(fl) / 1e6, 4), union_ms_per_launch=round(union(fl) / 1e6 / len(fl), 5),
                   mean_duration_ms_in_flight=round(sum(e - s for s, e in fl) / 1e6 / len(fl), 5))
    if al:
        res.update(launches_alone=len(al), mean_duration_ms_alone=round(sum(e - s for s, e in al) / 1e6 / len(al), 5))
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
