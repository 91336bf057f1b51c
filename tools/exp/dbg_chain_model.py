import numpy as np, torch, sys
from bdm_db1_amd import RingMemory, TransformerXL, synth, ops
from bdm_db1_amd.data import NLPTaskInput
DEV = "cuda:0"
poison = len(sys.argv) > 1
for nl in (3,):
    cfg = synth.db1_config("1.3B", n_layer=nl)
    torch.manual_seed(11)
    model = TransformerXL(cfg, device=torch.device(DEV), compute_dtype=torch.bfloat16)
    model.eval()
    if poison:
        def _new(*shape, dtype=None):
            dt = model.compute_dtype if dtype is None else dtype
            return torch.full(shape, float("nan"), device=model.dev, dtype=dt)
        model._new = _new
    rng = np.random.default_rng(4)
    calls = [rng.integers(0, 32000, (1, q)) for q in (22, 1, 1, 1, 9, 1, 1)]
    def run(chain):
        model.use_decode_chain = chain
        mems = RingMemory(model, 1)
        outs = []
        with torch.no_grad():
            for ids in calls:
                x = NLPTaskInput(position_id=None, attention_mask=None, loss_mask=None, label=None, text_seq=torch.from_numpy(ids).to(DEV), text_len=None)
                logits, _, mems = model([x], compute_loss=False, mems=mems)
                outs.append(logits.float().cpu().numpy())
        return outs
    ref, got, ref2 = run(False), run(True), run(False)
    print("layers", nl, "poison", poison, "err flag", ops.decode_chain_error(model.dev))
    for a, b, c in zip(got, ref, ref2):
        print("   q", a.shape[1], "chain vs separate", np.abs(a - b).max() / np.abs(b).max(), " separate twice", np.abs(c - b).max() / np.abs(b).max(), "nan", np.isnan(a).sum(), np.isnan(b).sum())
