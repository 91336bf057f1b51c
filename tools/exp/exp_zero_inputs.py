import torch, sys
import os, sys
_T = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))   # tools/
sys.path.insert(0, _T); sys.path.insert(0, os.path.dirname(_T))   # tools/ (bench_kernels) and the repository root
from bdm_db1_amd import ops
DEV='cuda:0'
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n*1e-3
M,N,K=65536,8192,2048
for kind in ['randn','zeros','small_int']:
    if kind=='randn':
        x=torch.randn(M,K,device=DEV).to(torch.bfloat16); w=(torch.randn(N,K,device=DEV)*0.02).to(torch.bfloat16)
    elif kind=='zeros':
        x=torch.zeros(M,K,device=DEV,dtype=torch.bfloat16); w=torch.zeros(N,K,device=DEV,dtype=torch.bfloat16)
    else:
        x=torch.randint(0,2,(M,K),device=DEV).to(torch.bfloat16); w=torch.randint(0,2,(N,K),device=DEV).to(torch.bfloat16)
    y=torch.empty(M,N,device=DEV,dtype=torch.bfloat16)
    t=timeit(lambda: ops.gemm(x,w.t(),y))
    print(kind,'NT ff1', f'{t*1e6:.1f} us {2.0*M*N*K/t/1e12:.1f} TF')
    dy=x.new_empty(M,N).copy_(y) if kind!='randn' else torch.randn(M,N,device=DEV).to(torch.bfloat16)
    dx=torch.empty(M,K,device=DEV,dtype=torch.bfloat16)
    t=timeit(lambda: ops.gemm(dy,w,dx))
    print(kind,'NN ff1 dx', f'{t*1e6:.1f} us {2.0*M*N*K/t/1e12:.1f} TF')
    t=timeit(lambda: torch.matmul(x, w.t(), out=y))
    print(kind,'vendor NT', f'{t*1e6:.1f} us {2.0*M*N*K/t/1e12:.1f} TF')
