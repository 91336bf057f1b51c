"""stage times of db1_decode_chain inside the graphed 1-token call (layer 10 of the 1.3B model)"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from bdm_db1_amd import TransformerXL, GraphedRingStep, synth, lib
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = TransformerXL(synth.db1_config("1.3B"), device=dev); model.eval()
ts = torch.zeros(16 + 2048, dtype=torch.int64, device=dev)
L = lib.load(); L.db1_test_decode_chain_timestamps.restype = None
L.db1_test_decode_chain_timestamps(ctypes.c_void_p(ts.data_ptr()), int(sys.argv[1]) if len(sys.argv) > 1 else 10)
step = GraphedRingStep(model, batch_size=1, n_new=1)
ids = torch.randint(0, 32000, (1, 1), device=dev)
names = ["start", "A0", "B0", "wait0", "LN1", "A1", "B1", "wait1", "actLDS", "A2", "B2", "wait2", "LN2", "A3", "B3"]
for rep in range(5):
    for _ in range(3): step(ids)
    torch.cuda.synchronize()
    t = ts.cpu().numpy()
    print("rep", rep, " ".join(f"{n}={(t[i] - t[0]) / 100:.2f}" for i, n in enumerate(names)))
    perw = (t[1040:].reshape(256, 4) - t[0]) / 100
    print("      worker wave 0:  " + "  ".join(f"{n} min {perw[:, i].min():.2f} med {np.median(perw[:, i]):.2f} max {perw[:, i].max():.2f}" for i, n in enumerate(["W0 issued", "merge requested", "W1 issued", "merged row"])))
    per = (t[16:1040].reshape(256, 4) - t[0]) / 100
    order = np.argsort(-per[:, 2])[:6]
    print("      latest B0: " + "  ".join(f"wg {int(b)} (xcd {int(b) % 8}): A0 {per[b, 1]:.2f} B0 {per[b, 2]:.2f} y_o {per[b, 3]:.2f}" for b in order), " | B0 > 5 us:", int((per[:, 2] > 5).sum()))
    print("      all workgroups: " + "  ".join(f"{n} min {per[:, i].min():.2f} med {np.median(per[:, i]):.2f} max {per[:, i].max():.2f}" for i, n in enumerate(names[:4])))
