import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bdm_db1_amd import TransformerXL, synth, GraphedRingStep
dev = torch.device("cuda", 0); torch.manual_seed(0)
model = TransformerXL(synth.db1_config("1.3B"), device=dev); model.eval()
q = int(sys.argv[1]) if len(sys.argv) > 1 else 1
step = GraphedRingStep(model, batch_size=1, n_new=q)
ids = torch.randint(0, 32000, (1, q), device=dev)
for _ in range(60): step(ids)
torch.cuda.synchronize()
