# what the GELU arithmetic costs inside the two GEGLU epilogues: the library rebuilt with the arithmetic replaced by a copy (WRONG results, timing only)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06p; mkdir -p $O; cd $R
echo "shipped build" | tee $O/geglu_epi_ablate.txt
python tools/exp/exp_geglu_epi.py 2>&1 | grep round | tee -a $O/geglu_epi_ablate.txt
export DB1_ALLOW_EXPERIMENT=1 DB1_EXTRA_HIPCC_FLAGS="-DDB1_EXPERIMENT -DW4_GEGLU_ABLATE=1"
python -m bdm_db1_amd.build > $O/build.log 2>&1
echo "W4_GEGLU_ABLATE=1 (no GELU arithmetic)" | tee -a $O/geglu_epi_ablate.txt
python tools/exp/exp_geglu_epi.py 2>&1 | grep round | tee -a $O/geglu_epi_ablate.txt
