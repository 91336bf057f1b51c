import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bdm_db1_amd import ops
from bdm_db1_amd import lib; lib.apply_env_knobs()   # DB1_* A/B switches of this script -> the library's thread-local knobs
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60160
dy = torch.randn(n * 256, 64, device="cuda").bfloat16(); x = torch.randn(n * 256, 64, device="cuda").bfloat16()
gp = torch.zeros(64, 576, device="cuda")
for _ in range(3): ops.conv3x3_implicit_wgrad(dy, x, gp, n)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): ops.conv3x3_implicit_wgrad(dy, x, gp, n)
e1.record(); torch.cuda.synchronize()
print(f"conv wgrad n_patches={n} ks={os.environ.get('DB1_CONV_WGRAD_KS', 'default')}: {e0.elapsed_time(e1) / 10 * 1e3:.0f} us")
