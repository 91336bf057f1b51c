R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06l; mkdir -p $O; cd $R
for k in 0 1; do
DB1_TRI_SPLIT=$k PROF_STEPS=4 bash tools/prof_step.sh > $O/prof_$k.log 2>&1
cp gpurun_out/step_table.txt $O/step_table_tri_split$k.txt; cp gpurun_out/step_table.json $O/step_table_tri_split$k.json
echo "tri_split=$k"; grep -E "region|0, 4>|splitk_reduce_kernel<float, float>" $O/step_table_tri_split$k.txt | cut -c1-150
done
