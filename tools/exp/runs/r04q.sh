cd $GRAFT_REPO_ROOT
run() { env $1 python tools/gen_gemm_w4.py > /dev/null; python -m bdm_db1_amd.build > /dev/null 2>&1; echo "== $1"; python tools/exp/sweep_w4n.py 2>&1 | grep sum; }
run "W4N_SP=4"
run "W4N_SP=3.5"
run "W4N_SP=4.3"
run "W4N_A=11"
run "W4N_A=15 W4N_B=20 W4N_RD=20"
run "W4N_B=15 W4N_RD=16"
run "W4N_B=16 W4N_RD=19"
run "W4N_A=12 W4N_SP=4.2"
