import torch, sys
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
def timeit(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/iters
T,d=16384,2048
for name,M,N,K in [("qkv",T,3*d,d),("ff1",T,4*d,d),("ff2",T,d,2*d),("head",T,33280,d)]:
    x=torch.randn(M,K,device='cuda').bfloat16(); w=(torch.randn(N,K,device='cuda')*0.02).bfloat16(); dy=torch.randn(M,N,device='cuda').bfloat16()
    t=timeit(lambda: torch.matmul(x,w.t())); print(f"vendor {name} NT: {t*1e3:8.1f} us {2*M*N*K/t/1e9:7.1f} TF")
    t=timeit(lambda: torch.matmul(dy,w)); print(f"vendor {name} NN: {t*1e3:8.1f} us {2*M*N*K/t/1e9:7.1f} TF")
    t=timeit(lambda: torch.matmul(dy.t(),x)); print(f"vendor {name} TN: {t*1e3:8.1f} us {2*M*N*K/t/1e9:7.1f} TF")
