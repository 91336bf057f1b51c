# A/B of the unroll depth of the activation kernels (rows in flight per thread): builds the library with -DDB1_ACT_UNROLL=n on the GPU box
for u in 2 4 8; do
  DB1_EXTRA_HIPCC_FLAGS="-DDB1_ACT_UNROLL=$u" python -m bdm_db1_amd.build > /dev/null 2>&1
  python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-decode 2>/dev/null | python -c "
import sys, json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']; print('unroll $u', d['ms_per_step'], 'act_fwd', k['ffn_act_fwd']['avg_us'], 'act_bwd', k['ffn_act_bwd']['avg_us'])"
done
