# usage: bash tools/exp/q2_nt.sh  -- builds the library with DB1_Q2_NT = 0..3 on the GPU box and times the stored-probabilities backward
for m in 0 1 2 3; do
  DB1_EXTRA_HIPCC_FLAGS="-DDB1_Q2_NT=$m" python -m bdm_db1_amd.build > /dev/null 2>&1
  for r in 1 2; do echo "NT=$m $(python tools/bench_kernels.py flash 64 2>&1 | grep 'forward-stored')"; done
done
