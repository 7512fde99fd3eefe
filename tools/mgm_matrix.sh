# tools/mgm_matrix.sh -- (GPU box) MGM aggregation time over tile sizes x disparity counts for build/variants/$VARIANTS
cp s2p_amd/lib/libs2p_hip.so /tmp/orig.so
for v in ${VARIANTS:-old cur}; do
  cp build/variants/$v/libs2p_hip.so s2p_amd/lib/libs2p_hip.so
  for sz in ${SIZES:-256 512 1024}; do for nd in ${NDISP:-32 64 128 256}; do for st in 1 3; do
    python bench.py --algo census --recursion 1 --streams $st --size $sz --ndisp $nd --steps 20 --warmup 3 --no-cpu 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('${TAG:-}$v size $sz ndisp $nd streams $st ms/tile %.4f agg %.4f' % (d['ms_per_step'], d['stage_ms']['aggregate']))"
  done; done; done
done
cp /tmp/orig.so s2p_amd/lib/libs2p_hip.so
