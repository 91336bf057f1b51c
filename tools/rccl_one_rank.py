"""RCCL with ONE rank on the one GPU of a test box: the engine's own initialiser (high-priority communication stream, eager communicator),
then GradSync's bucket path -- bf16 staging cast, async all-reduce, the per-bucket norm on a side stream behind the collective's
STREAM-level wait, finish() -- with the world size reported as 2 so that nothing is skipped (a sum over one rank is the identity)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT", "29531"), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
import torch
import torch.distributed as dist
from bdm_db1_amd import engine as E, ops
t0 = time.time()
E.init_distributed(dist_backend="nccl")
dev = torch.device("cuda", 0)
x = torch.ones(8, device=dev)
dist.all_reduce(x, op=dist.ReduceOp.MAX)
dist.barrier()
torch.cuda.synchronize()
print(f"communicator up in {time.time() - t0:.1f} s, backend {dist.get_backend()}", flush=True)
real_ws = E.dist.get_world_size
E.dist.get_world_size = lambda group=None: 2
nb, per = 25, 4_600_000                                   # 25 buckets of 4.6 M gradients (the 1.3B model's: 48 M per layer)
g = torch.randn(nb * per, device=dev)
buckets = [(f"h.{nb - 1 - i}", i * per, (i + 1) * per) for i in range(nb)]
stage = torch.zeros(nb * per, device=dev, dtype=torch.bfloat16)
sync = E.GradSync(g, buckets, None, stage=stage, cast=ops.cast, norm_sq=ops.grad_norm_sq)
E.dist.get_world_size = real_ws
assert sync._side is not None and sync.world == 2
sync.time_waits = True
want = torch.zeros(1, device=dev)
for step in range(3):
    with ops.stream_scope():
        for name, _, _ in buckets[:20]:
            sync.launch(name)
        sync.finish()
        got = sync.reduced_norm_sq()
        ops.grad_norm_sq(stage, want)
    torch.cuda.synchronize()
    assert torch.equal(stage, g.to(torch.bfloat16)), "the reduced staging copy is not the cast gradient"
    assert abs(got.item() - want.item()) <= 1e-5 * want.item(), (got.item(), want.item())
    ms = sync.wait_events[-1][0].elapsed_time(sync.wait_events[-1][1])
    print(f"step {step}: 25 bf16 buckets of {per * 2 / 1e6:.1f} MB all-reduced over one rank, norm {got.item() ** 0.5:.3f}, finish() waited {ms:.2f} ms", flush=True)
    g.normal_()
# fp32 on the wire
sync32 = None
E.dist.get_world_size = lambda group=None: 2
sync32 = E.GradSync(g, buckets, None, norm_sq=ops.grad_norm_sq)
E.dist.get_world_size = real_ws
g0 = g.clone()
with ops.stream_scope():
    sync32.finish()
torch.cuda.synchronize()
assert torch.equal(g, g0)
print("fp32 buckets ok", flush=True)
dist.barrier()
dist.destroy_process_group()
print("RCCL one-rank path ok")
