cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 600 python -m pytest tests/test_decode_gpu.py tests/test_geglu_epilogue_gpu.py tests/test_gradsync_gpu.py -x -q 2>&1 | tail -2
timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-decode --no-mixture 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print(d['value'], d['ms_per_step'])"
