"""Developer check of the 256x256 ping-pong GEMM (DB1_GEMM_TILE=512): all three operand forms, odd/even k-tile counts."""
import os, sys
os.environ.setdefault("DB1_GEMM_TILE", "512")
import os, sys
_T = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))   # tools/
sys.path.insert(0, _T); sys.path.insert(0, os.path.dirname(_T))   # tools/ (bench_kernels) and the repository root
import torch
from bdm_db1_amd import ops
from bdm_db1_amd import lib; lib.apply_env_knobs()   # DB1_* A/B switches of this script -> the library's thread-local knobs
torch.manual_seed(0)
bad = 0
for (M, N, K) in [(256, 256, 64), (512, 768, 128), (512, 256, 320), (768, 512, 1024)]:
    for form in ("nt", "nn", "tn"):
        for od in (torch.float32, torch.bfloat16):
            A = torch.randn(M, K, device="cuda").bfloat16(); B = torch.randn(K, N, device="cuda").bfloat16()
            a_store = A if form != "tn" else A.t().contiguous()
            b_store = B.t().contiguous() if form == "nt" else B
            a = a_store if form != "tn" else a_store.t()
            b = b_store.t() if form == "nt" else b_store
            C0 = torch.randn(M, N, device="cuda").to(od)
            out = C0.clone()
            ops.gemm(a, b, out, alpha=0.5, beta=1.0)
            ref = 0.5 * (A.float() @ B.float()) + C0.float()
            err = ((out.float() - ref).abs().max() / ref.abs().max()).item()
            ok = err < (2e-6 if od == torch.float32 else 6e-3)
            bad += not ok
            print(f"{form} {M}x{N}x{K} {str(od)[6:]:9s} rel err {err:.2e} {'ok' if ok else 'FAIL'}")
print("FAILED" if bad else "ALL OK")
