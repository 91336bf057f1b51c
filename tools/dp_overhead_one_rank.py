"""What the data-parallel path costs a rank BESIDES the wire: the 64-sequence DB1-1.3B step with the engine built for two ranks over RCCL
itself -- one rank on the one GPU of a test box, the world size read as 2 while the engine is built (tests/test_dp_gpu.py:
test_engine_steps_over_rccl_with_one_rank checks that this path computes the single-rank step) -- against the plain single-rank step, same
process, alternating.  The hooks launch 25 + 1 bucket all-reduces from the backward (bf16 staging casts on the compute stream, RCCL launches on
its high-priority stream, per-bucket norms on the side stream behind stream-level waits), Adam reads the staged copy.  A one-rank all-reduce
moves nothing over xGMI, so this is the rank-side floor of the scaling loss, not the scaling loss."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT", "29533"), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
from types import SimpleNamespace
import torch
import torch.distributed as dist
from bdm_db1_amd import TransformerXL, initialize, mpu, synth, engine as E
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
E.init_distributed(dist_backend="nccl")
mpu.initialize_model_parallel()
dev = torch.device("cuda", 0)
cfg = synth.db1_config("1.3B", drop=0.1, embd_pdrop=0.1)
torch.manual_seed(1234)
model = TransformerXL(cfg, device=dev)
eargs = SimpleNamespace(lr=1e-4, weight_decay=0.01, clip_grad=1.0, optimizer="adam", keep_logits=False, fuse_head_loss=True, gradient_accumulation_steps=1)
plain, _, _, _ = initialize(eargs, model, mpu=None)
real = dist.get_world_size
dist.get_world_size = lambda group=None: 2
try:
    dp, _, _, _ = initialize(eargs, model, mpu=mpu)      # (a second engine over the same model: the arena, its optimizer state of its own)
finally:
    dist.get_world_size = real
assert dp.dp_world == 2 and dp.sync._side is not None and plain.dp_world == 1
batch = [synth.text_batch(64, cfg.n_position, 1234, dev)]
def run(engine, n):
    engine.train()
    for _ in range(n):
        _, loss = engine(batch)
        engine.backward(loss)
        engine.step()
    return loss
def timed(engine, n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    run(engine, n)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
run(plain, 2); run(dp, 2)
rec = {"plain_ms": [], "dp_path_ms": [], "exposed_wait_ms": []}
for r in range(rounds):
    rec["plain_ms"].append(round(timed(plain, steps), 3))
    dp.time_comm = True
    rec["dp_path_ms"].append(round(timed(dp, steps), 3))
    rec["exposed_wait_ms"].append(round(dp.exposed_comm_ms() / steps, 3))
    dp.time_comm = False
p, d = min(rec["plain_ms"]), min(rec["dp_path_ms"])
rec.update(steps_per_sample=steps, overhead_ms=round(d - p, 3), overhead_frac=round((d - p) / p, 5),
           buckets=len(dp.sync.order), bucket_mb_bf16=round(sum(e - s for s, e in dp.sync.buckets.values()) * 2 / 1e6 / len(dp.sync.order), 1),
           note="one-rank RCCL all-reduces (nothing crosses xGMI): staging casts + RCCL launches + per-bucket norms + hooks only")
print(json.dumps(rec))
dist.barrier(); dist.destroy_process_group()
