"""debug: db1_decode_chain stage by stage against torch fp32 arithmetic (random partials)"""
import torch, math
from bdm_db1_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
d, dff, H, D, nunit = 2048, 4096, 16, 128, 5
bf = torch.bfloat16
r = lambda *s, sc=1.0: (torch.randn(*s, device=dev) * sc).to(bf)
x = r(1, d)
Wo, W1, W2, Wq = r(d, d, sc=0.02), r(2 * dff, d, sc=0.02), r(d, dff, sc=0.02), r(3 * d, d, sc=0.02)
b1, b2 = r(2 * dff, sc=0.1), r(d, sc=0.1)
g1, be1, g2, be2 = (1 + 0.1 * torch.randn(d, device=dev)).to(bf), r(d, sc=0.1), (1 + 0.1 * torch.randn(d, device=dev)).to(bf), r(d, sc=0.1)
part = torch.zeros(H, nunit, 64, D + 2, device=dev)
part[:, :, 0, :D] = torch.randn(H, nunit, D, device=dev)
part[:, :, 0, D] = torch.randn(H, nunit, device=dev)
part[:, :, 0, D + 1] = torch.rand(H, nunit, device=dev) + 0.5
alpha, eps = 1.3, 1e-5
fo = torch.zeros(1, d, device=dev, dtype=bf)
h1o, xn, qn = torch.zeros(1, d, device=dev, dtype=bf), torch.zeros(1, d, device=dev, dtype=bf), torch.zeros(1, 3 * d, device=dev, dtype=bf)
ops.decode_chain(part, nunit * 128, H, x, Wo, W1, b1, W2, b2, Wq, g1, be1, g2, be2, alpha, eps, h1o, fo, xn, qn, 0)
torch.cuda.synchronize()
buf = ops.decode_chain_scratch(dev)
def rows():
    w = buf[: (2 * d + dff) * 4].view(torch.int32)
    val = (w >> 16).to(torch.int16).view(bf).float()
    return val[:d], val[d:d + dff], val[d + dff:], (w & 0xffff)
y_o, act, f, tags = rows()
print("err flag", ops.decode_chain_error(dev), "tags", tags.unique().tolist())
m, l, o = part[:, :, 0, D], part[:, :, 0, D + 1], part[:, :, 0, :D]
mx = m.max(1, keepdim=True).values
wt = torch.exp(m - mx)
merged = ((o * wt[..., None]).sum(1) / (l * wt).sum(1, keepdim=True)).reshape(1, d).to(bf).float()
rb = lambda t: t.to(bf).float()
y_ref = rb(merged @ Wo.float().t())
def ln(s, g, b):
    s = rb(s)
    mu = s.mean(-1, keepdim=True); var = ((s - mu) ** 2).mean(-1, keepdim=True)
    return rb((s - mu) * torch.rsqrt(var + eps) * g.float() + b.float())
h1 = ln(alpha * x.float() + y_ref, g1, be1)
z = rb(h1 @ W1.float().t() + b1.float())
act_ref = rb(z[:, :dff] * torch.nn.functional.gelu(z[:, dff:]))
act_alt = rb(torch.nn.functional.gelu(z[:, :dff]) * z[:, dff:])
f_ref = rb(act_ref @ W2.float().t() + b2.float())
xn_ref = ln(alpha * h1 + f_ref, g2, be2)
q_ref = rb(xn_ref @ Wq.float().t())
e = lambda a, b: ((a.reshape(-1) - b.reshape(-1)).abs().max() / b.abs().max()).item()
print("y_o", e(y_o, y_ref)); print("h1", e(h1o.float(), h1)); print("act", e(act, act_ref), "alt", e(act, act_alt)); print("f", e(f, f_ref))
print("x_next", e(xn.float(), xn_ref)); print("qkv_next", e(qn.float(), q_ref))
# last-layer form
ops.decode_chain(part, nunit * 128, H, x, Wo, W1, b1, W2, b2, None, g1, be1, g2, be2, alpha, eps, h1o, fo, None, None, 1)
torch.cuda.synchronize()
print("last: h1", e(h1o.float(), h1), "f", e(rows()[2], f_ref), "f_out", e(fo.float(), f_ref), "err", ops.decode_chain_error(dev))
# stage times of workgroup 0 + whole-launch time
import ctypes
import numpy as np
from bdm_db1_amd import lib
ts = torch.zeros(16 + 2048, dtype=torch.int64, device=dev)
L = lib.load()
L.db1_test_decode_chain_timestamps.restype = None
L.db1_test_decode_chain_timestamps(ctypes.c_void_p(ts.data_ptr()), -1)
names = ["start", "A0", "B0", "wait0", "LN1", "A1", "B1", "wait1", "actLDS", "A2", "B2", "wait2", "LN2", "A3", "B3"]
for rep in range(4):
    ops.decode_chain(part, nunit * 128, H, x, Wo, W1, b1, W2, b2, Wq, g1, be1, g2, be2, alpha, eps, h1o, fo, xn, qn, 2 + rep)
    torch.cuda.synchronize()
    t = ts.cpu().numpy()
    print("rep", rep, " ".join(f"{n}={(t[i] - t[0]) / 100:.2f}" for i, n in enumerate(names)))
    perw = (t[1040:].reshape(256, 4) - t[0]) / 100
    print("      worker wave 0:  " + "  ".join(f"{n} min {perw[:, i].min():.2f} med {np.median(perw[:, i]):.2f} max {perw[:, i].max():.2f}" for i, n in enumerate(["W0 issued", "merge requested", "W1 issued", "merged row"])))
    per = (t[16:1040].reshape(256, 4) - t[0]) / 100
    print("      all workgroups: " + "  ".join(f"{n} min {per[:, i].min():.2f} med {np.median(per[:, i]):.2f} max {per[:, i].max():.2f}" for i, n in enumerate(names[:4])))
L.db1_test_decode_chain_timestamps(ctypes.c_void_p(0), -1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for nrep in (1, 20):
    torch.cuda.synchronize()
    e0.record()
    for i in range(nrep):
        ops.decode_chain(part, nunit * 128, H, x, Wo, W1, b1, W2, b2, Wq, g1, be1, g2, be2, alpha, eps, h1o, fo, xn, qn, 10 + nrep + i)
    e1.record(); torch.cuda.synchronize()
    print("launches", nrep, "us per launch", e0.elapsed_time(e1) * 1e3 / nrep)
