# evidence of DESIGN 10 "one persistent launch per layer": hand-off microbench, single-launch streaming, stage times inside the graphed call,
# the graphed 1-token call with and without the chain  ->  gpurun_out/r04_decode_chain.txt
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
O=gpurun_out/r04_decode_chain.txt
{
echo "== tools/exp/sync_bench (hand-off between 256 resident workgroups: mode 0 counter, mode 1 tagged words)"
timeout 60 ./tools/exp/sync_bench 96 | sed -n '2,5p;7,10p'
echo "== tools/exp/stream_once (one launch of 256 workgroups streaming a layer-sized buffer)"
timeout 60 ./tools/exp/stream_once
echo "== tools/exp/stream_first (W0 then W1 requested up front: when are they issued / landed)"
timeout 60 ./tools/exp/stream_first | sed -n '1,2p;5,6p;9,10p'
echo "== stage times of db1_decode_chain inside the graphed 1-token call (layer 10, workgroup 0 + distribution over workgroups; us)"
timeout 200 python tools/exp/dbg_chain_inmodel.py 10 2>&1 | grep -v "worker wave\|amdgpu.ids" | tail -9
echo "== tools/bench_decode.py, DB1_DECODE_CHAIN=1 (default)"
timeout 300 python tools/bench_decode.py 2>&1 | grep -v amdgpu.ids
echo "== tools/bench_decode.py, DB1_DECODE_CHAIN=0"
DB1_DECODE_CHAIN=0 timeout 300 python tools/bench_decode.py 2>&1 | grep -v amdgpu.ids
} > $O 2>&1
cat $O
