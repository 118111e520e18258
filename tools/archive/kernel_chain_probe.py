import torch, time
x = torch.zeros(512, device="cuda")
def chain(n):
    for _ in range(n): x.add_(1.0)
chain(10); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g): chain(544)
for _ in range(3): g.replay()
torch.cuda.synchronize(); t0=time.perf_counter()
for _ in range(20): g.replay()
torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/20
print(f"graph: 544 dependent tiny kernels: {dt*1e3:.2f} ms -> {dt/544*1e6:.2f} us per kernel")
torch.cuda.synchronize(); t0=time.perf_counter()
for _ in range(20): chain(544)
torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/20
print(f"eager: {dt/544*1e6:.2f} us per kernel")
