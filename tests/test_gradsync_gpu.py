"""GradSync's per-bucket norm on device tensors behind a collective whose wait() is a STREAM-level wait (what RCCL's is): one GPU cannot
run two RCCL ranks, so the collective is a stand-in that doubles the bucket on a communication stream of its own."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def test_per_bucket_norm_of_the_reduced_gradients_behind_stream_level_waits(monkeypatch):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from bdm_db1_amd import engine as E, ops
    dev = torch.device("cuda", torch.cuda.current_device())
    comm = torch.cuda.Stream(device=dev)

    class Handle:
        def wait(self):   # the CURRENT stream waits for the collective; the host does not
            torch.cuda.current_stream(dev).wait_stream(comm)

    def all_reduce(buf, op=None, group=None, async_op=False):
        comm.wait_stream(torch.cuda.current_stream(dev))   # the collective starts behind the launching stream's work (the cast)
        with torch.cuda.stream(comm):
            torch.cuda._sleep(2_000_000)                    # (long enough that a missing wait would read the bucket too early)
            buf.mul_(2)
        return Handle()
    monkeypatch.setattr(E.dist, "is_initialized", lambda: True)
    monkeypatch.setattr(E.dist, "get_world_size", lambda group=None: 2)
    monkeypatch.setattr(E.dist, "get_backend", lambda group=None: "nccl")
    monkeypatch.setattr(E.dist, "all_reduce", all_reduce)
    nb, per = 25, 46_000
    g = torch.randn(nb * per, device=dev)
    buckets = [(f"h.{nb - 1 - i}", i * per, (i + 1) * per) for i in range(nb)]
    stage = torch.zeros(nb * per, device=dev, dtype=torch.bfloat16)
    sync = E.GradSync(g, buckets, None, stage=stage, cast=ops.cast, norm_sq=ops.grad_norm_sq)
    assert sync._side is not None and sync.norm_parts.numel() == nb
    want = torch.zeros(1, device=dev)
    for step in range(2):                                   # the object is re-used every optimizer step
        with ops.stream_scope():                            # (the hooks fire inside the model's backward, which runs under a stream scope)
            for name, _, _ in buckets[:20]:
                sync.launch(name)
            sync.finish()
            got = sync.reduced_norm_sq()
            ops.grad_norm_sq(stage, want)                   # one pass over the whole reduced tensor
        ref = (2.0 * g.to(torch.bfloat16).float()).pow(2).sum()
        torch.cuda.synchronize()
        assert torch.equal(stage.float(), 2.0 * g.to(torch.bfloat16).float())
        assert abs(got.item() - want.item()) <= 1e-5 * want.item() and abs(got.item() - ref.item()) <= 1e-4 * ref.item()
        for i in (0, 7, nb - 1):
            assert abs(sync.norm_parts[i].item() - stage[i * per:(i + 1) * per].float().pow(2).sum().item()) <= 1e-4 * sync.norm_parts[i].item()
        g.normal_()
