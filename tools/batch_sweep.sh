#!/bin/bash
# tools/batch_sweep.sh -- GPU box: the MGM headline through the batched entry (tiles per library call x tile streams x stagger)
for NB in ${NBS:-1 2 4 8}; do for S in ${STREAMS:-1 2 3}; do for ST in ${STAGGERS:-128}; do
  echo "tiles/call $NB streams $S stagger $ST: $(S2P_MGM_STAGGER=$ST python bench.py --no-cpu --no-job --steps 3 --batch 96 --streams $S --batch-launch $NB 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%.4f ms per tile | aggregate launch %.4f ms, frac %.3f | cost %.3f wta %.3f' % (d['ms_per_tile'], d['stage_ms']['aggregate'], r['frac'], d['stage_ms']['cost'], d['stage_ms']['wta']))")"
done; done; done
