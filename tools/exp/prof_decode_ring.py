"""eager 1-token calls on the K / V ring at the 1.3B geometry, for a kernel trace (rocprofv3 --kernel-trace --stats)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bdm_db1_amd import TransformerXL, RingMemory, synth
from bdm_db1_amd.data import NLPTaskInput
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = TransformerXL(synth.db1_config("1.3B"), device=dev)
model.eval()
mems = RingMemory(model, 1)
ids = torch.randint(0, 32000, (1, 1), device=dev)
with torch.no_grad():
    for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 30):
        x = NLPTaskInput(position_id=None, attention_mask=None, loss_mask=None, label=None, text_seq=ids, text_len=None)
        _, _, mems = model([x], compute_loss=False, mems=mems)
torch.cuda.synchronize()
