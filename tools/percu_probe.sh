# tools/percu_probe.sh -- (GPU box) launch time of k_mgm_bands against the cap on workgroups per CU (S2P_MGM_PER_CU; unset = the library's rule)
for sz in ${SIZES:-512 768 1024 1536 2048}; do
  for pc in "" 1 2 3; do
    [ -z "$pc" ] && unset S2P_MGM_PER_CU || export S2P_MGM_PER_CU=$pc
    for st in 1 3; do
    timeout 120 python bench.py --algo census --recursion 1 --streams $st --size $sz --ndisp ${NDISP:-128} --steps 20 --warmup 4 --no-cpu 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('size $sz x ${NDISP:-128} per_cu=${pc:-rule} streams=$st ms/tile', d['ms_per_step'], 'agg', d['stage_ms']['aggregate'])"
    done
  done
done
