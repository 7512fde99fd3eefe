#!/bin/bash
# round 4: arenas that outlive their workers (S2P_HIP_BROKER_RECYCLE, default on) against mapping + page-locking every Pool's arenas anew.
# Three successive Pools per run, as the reference's steps are: the first one creates the arenas either way.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r04
fmt='
import json,sys
d=json.loads(sys.stdin.read())
print(" | ".join("f2j %s steady %s cold med/max %s/%s attach med/max %s/%s new/recycled %s/%s" % (p["tiles_per_s_fork_to_join"], p["steady"]["tiles_per_s"], p["cold_start_s"]["median"], p["cold_start_s"]["max"], (p.get("broker_connect_attach_ms") or {}).get("median"), (p.get("broker_connect_attach_ms") or {}).get("max"), (p.get("broker") or {}).get("arenas_new"), (p.get("broker") or {}).get("arenas_recycled")) for p in d["pools"]))
print("errors", d["errors"])'
for rep in 1 2; do for R in 1 0; do for W in 64 16; do
  T=$((W * 24))
  echo "recycle=$R workers=$W: $(S2P_HIP_BROKER_RECYCLE=$R timeout 300 python bench_pool.py --workers $W,$W,$W --tiles $T 2>gpurun_out/r04/recycle_probe.err | python -c "$fmt")"
done; done; done
