# timing-only ablations of the hand-scheduled 4-wave flash forward (results wrong by construction)
cd ${GRAFT_REPO_ROOT:-.}
for a in "" $1; do
  FW3_ABLATE=$a python tools/gen_flash_fwd3.py > /dev/null
  python -m bdm_db1_amd.build > /dev/null 2>&1
  echo "== ablate: '$a'"
  timeout 120 python tools/exp/check_fwd2.py time 2>&1 | grep "fwd2=1"
done
python tools/gen_flash_fwd3.py > /dev/null
python -m bdm_db1_amd.build > /dev/null 2>&1
