# tools/c4_probe.sh -- (GPU box) the multi-level jobs against build/variants/* (latency-bound: a host sync per pyramid level)
cp s2p_amd/lib/libs2p_hip.so /tmp/orig.so
for v in ${VARIANTS:-old cur nw4}; do
  cp build/variants/$v/libs2p_hip.so s2p_amd/lib/libs2p_hip.so
  python bench.py --workload config4 --steps 60 --tile-algo mgm_multi --no-cpu 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v config4 mgm_multi ms/tile', d['ms_per_step'])"
  python bench.py --workload config4 --steps 60 --tile-algo mgm --no-cpu 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v config4 mgm ms/tile', d['ms_per_step'])"
  python tools/config2_time.py 2>/dev/null | tail -2
  python tools/shim_time.py 2>/dev/null | grep -v amdgpu
done
cp /tmp/orig.so s2p_amd/lib/libs2p_hip.so
