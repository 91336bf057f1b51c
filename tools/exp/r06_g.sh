R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06g; mkdir -p $O; cd $R
timeout 2400 python -m pytest tests -q -m gpu > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log
tail -6 $O/tests.log
TAG=r06g bash tools/evidence.sh pmc > $O/evidence.log 2>&1
tail -5 $O/evidence.log
