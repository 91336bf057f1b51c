cd ${GRAFT_REPO_ROOT:-.}
for a in "" "dma,store" "dma,store,barrier"; do
  FW2_ABLATE=$a python tools/gen_flash_fwd.py > /dev/null
  DB1_EXTRA_HIPCC_FLAGS=-DDB1_EXPERIMENT python -m bdm_db1_amd.build > /dev/null 2>&1
  export DB1_ALLOW_EXPERIMENT=1 DB1_EXTRA_HIPCC_FLAGS=-DDB1_EXPERIMENT
  echo "== ablate '$a': 8 waves / 4 waves"
  timeout 120 python tools/exp/check_fwd2.py time 2>&1 | grep "fwd2=True"
  FW2_HALF=1 timeout 120 python tools/exp/check_fwd2.py time 2>&1 | grep "fwd2=True"
done
python tools/gen_flash_fwd.py > /dev/null
unset DB1_EXTRA_HIPCC_FLAGS DB1_ALLOW_EXPERIMENT
python -m bdm_db1_amd.build > /dev/null 2>&1
