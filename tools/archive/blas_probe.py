import torch, time
dev="cuda:0"
def timeit(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
M=16000
for (N,K,name) in [(3072,1024,"qkv"),(1024,1024,"out"),(4096,1024,"ff1"),(1024,4096,"ff2"),(3072,3072,"qkv K'=3K"),(4096,3072,"ff1 K'=3K"),(1024,12288,"ff2 K'=3K")]:
    for dt in (torch.float16, torch.bfloat16, torch.float32):
        a=torch.randn(M,K,device=dev,dtype=dt); w=torch.randn(N,K,device=dev,dtype=dt)
        t=timeit(lambda: torch.matmul(a, w.T))
        print(f"torch.matmul {name:12s} {str(dt)[6:]:9s} M={M} N={N} K={K}: {t:8.1f} us  {2*M*N*K/t/1e6:7.1f} TF")
