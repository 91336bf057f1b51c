"""same-box A/B of the two forms of the 64 -> 64 channel 3x3 convolution (forward / data gradient) at the RL step's size: 60 160 patches"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from bdm_db1_amd import lib, ops
dev = torch.device("cuda", 0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 60160
x = torch.randn(N * 256, 64, device=dev).to(torch.bfloat16)
w = (torch.randn(64, 64, 3, 3, device=dev) * 0.1).to(torch.bfloat16)
bias = torch.randn(64, device=dev).to(torch.bfloat16)
res = torch.randn(N * 256, 64, device=dev).to(torch.bfloat16)
w_op = torch.empty(64, 576, device=dev, dtype=torch.bfloat16)
ops.conv_weight_permute(w, w_op, 64, 64)
y = torch.empty(N * 256, 64, device=dev, dtype=torch.bfloat16)
def timeit(fn, n=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for rnd in range(2):
    for knob in (0, 1):
        lib.set_knob("conv_patch", knob)
        t0 = timeit(lambda: ops.conv3x3_implicit_fwd(x, w_op, bias, y, N, sign=1))
        t1 = timeit(lambda: ops.conv3x3_implicit_fwd(x, w_op, bias, y, N, sign=1, res=res))
        t2 = timeit(lambda: ops.conv3x3_implicit_fwd(x, w_op, None, y, N, sign=-1))
        gp = torch.zeros(64, 576, device=dev)
        gbias = torch.zeros(64, device=dev)
        t3 = timeit(lambda: ops.conv3x3_implicit_wgrad(res, x, gp, N, gbias_acc=gbias))
        gb = 2 * N * 256 * 64 * 2 / 1e9
        print(f"round {rnd} conv_patch={knob}: fwd {t0:.0f} us ({gb / t0 * 1e6 / 1e3:.2f} TB/s)  fwd+res {t1:.0f} us  dgrad {t2:.0f} us  wgrad {t3:.0f} us   ({N} patches)")
