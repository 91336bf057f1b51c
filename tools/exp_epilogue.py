import torch, sys
sys.path.insert(0, '.')
from bdm_db1_amd import ops
sys.path.insert(0, 'tools')
from bench_kernels import timeit
DEV='cuda'
M,N,K=65536,8192,2048
x=torch.randn(M,K,device=DEV).to(torch.bfloat16); w=(torch.randn(N,K,device=DEV)*0.02).to(torch.bfloat16)
y=torch.empty(M,N,device=DEV,dtype=torch.bfloat16)
for alpha in (1.0, 12345.0, 1.0, 12345.0):
    t=timeit(lambda: ops.gemm(x,w.t(),y,alpha=alpha))
    print('NT ff1 alpha',alpha, f'{t*1e3:.1f} us')
yf=torch.empty(M,N,device=DEV,dtype=torch.float32)
for alpha in (1.0, 12345.0):
    t=timeit(lambda: ops.gemm(x,w.t(),yf,alpha=alpha))
    print('NT ff1 f32 out alpha',alpha, f'{t*1e3:.1f} us')
