#!/bin/bash
# tools/issue_probe.sh -- GPU box: 16 extra s_nop / 16 extra v_mov in every MGM step (135 instructions): does the launch pay for issue slots?
cd "$(dirname "$0")/.."
run() {
  for ARGS in "--batch 48 --batch-launch 1 --streams 1" "--batch 48 --batch-launch 8 --streams 1"; do
    echo "[$1] $ARGS: $(python bench.py --no-cpu --no-job --steps 3 $ARGS 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4f ms per tile | aggregate launch %.4f ms' % (d['ms_per_tile'], d['stage_ms']['aggregate']))")"
  done
}
run "shipped"
for F in "-DS2P_MGM_PROBE_NOP=16" "-DS2P_MGM_PROBE_VMOV=16"; do
  S2P_HIP_EXTRA_FLAGS="$F" python -m s2p_amd.build --force > /dev/null 2>&1
  run "$F"
done
python -m s2p_amd.build --force > /dev/null 2>&1
