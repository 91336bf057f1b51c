"""bench.py with library A/B knobs set first (the knobs are thread-local and bench.py itself never reads the environment for them):
    python tools/exp/bench_knob.py flash_kv3=0 -- --batch 4 --ga 16 --graph --steps 3 --warmup 1 --no-cpu-baseline --no-decode"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
i = sys.argv.index("--")
from bdm_db1_amd import lib
for kv in sys.argv[1:i]:
    k, v = kv.split("=")
    lib.set_knob(k, int(v))
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[i + 1:]
import bench
bench.main()
