# fixed cost of a workgroup of the hand-scheduled flash forward: empty workgroup / prologue only / loop skeleton without epilogue
cd ${GRAFT_REPO_ROOT:-.}
FW2_ABLATE=dma,store,mfma,lds,valu,barrier python tools/gen_flash_fwd.py > /dev/null
for st in 1 2 3 0; do
  DB1_ALLOW_EXPERIMENT=1 DB1_EXTRA_HIPCC_FLAGS="-DDB1_EXPERIMENT -DFW2_STOP=$st" python -m bdm_db1_amd.build > /dev/null 2>&1
  echo "== FW2_STOP=$st (loop body ablated)"
  DB1_ALLOW_EXPERIMENT=1 DB1_EXTRA_HIPCC_FLAGS="-DDB1_EXPERIMENT -DFW2_STOP=$st" timeout 120 python tools/exp/check_fwd2.py time 2>&1 | grep "fwd2=True"
done
python tools/gen_flash_fwd.py > /dev/null
python -m bdm_db1_amd.build > /dev/null 2>&1
