import os, sys, cProfile, pstats, io, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from types import SimpleNamespace
from bdm_db1_amd import TransformerXL, synth, initialize
dev = torch.device("cuda", 0); torch.manual_seed(0)
cfg = synth.db1_config("1.3B", drop=0.1, embd_pdrop=0.1)
model = TransformerXL(cfg, device=dev)
engine, _, _, _ = initialize(SimpleNamespace(lr=1e-4, weight_decay=0.01, clip_grad=1.0, optimizer="adamw", keep_logits=False, fuse_head_loss=True, gradient_accumulation_steps=16), model)
engine.train()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
batch = [synth.text_batch(B, cfg.n_position, 0, dev)]
def micro(n):
    for _ in range(n):
        _, loss = engine(batch); engine.backward(loss); engine.step()
    torch.cuda.synchronize()
micro(4)
t0 = time.perf_counter(); micro(16); print(f"B={B}: {(time.perf_counter() - t0) / 16 * 1e3:.2f} ms per micro-step (wall)")
pr = cProfile.Profile(); pr.enable(); micro(16); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22); print(s.getvalue()[:4200])
