import os, sys
sys.path.insert(0, 'tools'); sys.path.insert(0, '.')
import torch
from bdm_db1_amd import ops
DEV='cuda'
T,di,d=65536,8192,2048
dz=torch.randn(T,di,device=DEV).to(torch.bfloat16)
fin=torch.randn(T,d,device=DEV).to(torch.bfloat16)
W1=(torch.randn(di,d,device=DEV)*0.02).to(torch.bfloat16)
G=torch.zeros(di,d,device=DEV)
dfin=torch.empty(T,d,device=DEV,dtype=torch.bfloat16)
def tn(): ops.gemm(dz.t(), fin, G, beta=1.0)
def nn(): ops.gemm(dz, W1, dfin)
def seq(name, fns, reps=30):
    for f in fns: f()
    torch.cuda.synchronize()
    n=len(fns); acc=[0.0]*n
    evs=[[torch.cuda.Event(enable_timing=True) for _ in range(n+1)] for _ in range(reps)]
    for r in range(reps):
        evs[r][0].record()
        for i,f in enumerate(fns):
            f(); evs[r][i+1].record()
    torch.cuda.synchronize()
    for r in range(reps):
        for i in range(n): acc[i]+=evs[r][i].elapsed_time(evs[r][i+1])
    print(name, ' '.join(f'{a/reps*1e3:.0f}' for a in acc))
seq('tn only     ', [tn])
seq('nn only     ', [nn])
seq('tn, nn      ', [tn, nn])
seq('nn, tn      ', [nn, tn])
x=torch.randn(T,di,device=DEV).to(torch.bfloat16); h=torch.empty(T,di//2,device=DEV,dtype=torch.bfloat16)
def act(): ops.ffn_act_fwd(x,h,"geglu")
seq('act, tn, nn ', [act, tn, nn])
seq('act, nn, tn ', [act, nn, tn])
W1b=(torch.randn(di,d+64,device=DEV)*0.02).to(torch.bfloat16)[:, :d]
def nn_pad(): ops.gemm(dz, W1b, dfin)
seq('tn, nn (W ld 2112)', [tn, nn_pad])
