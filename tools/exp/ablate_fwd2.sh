# timing-only ablations of the hand-scheduled flash forward (results wrong by construction): regenerate the loop without one kind of
# instruction, rebuild, time B = 64.   bash tools/exp/ablate_fwd2.sh "dma store mfma lds valu barrier dma,store,barrier"
cd ${GRAFT_REPO_ROOT:-.}
for a in "" $1; do
  FW2_ABLATE=$a python tools/gen_flash_fwd.py > /dev/null
  python -m bdm_db1_amd.build > /dev/null 2>&1
  echo "== ablate: '$a'"
  timeout 120 python tools/exp/check_fwd2.py time 2>&1 | grep "fwd2=True"
done
python tools/gen_flash_fwd.py > /dev/null
python -m bdm_db1_amd.build > /dev/null 2>&1
