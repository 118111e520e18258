"""One library GEMM shape for PMC runs: the three-term product as a plain fp16 GEMM with K' = 3K (ff2 shape)."""
import os, torch
dev = "cuda:0"
M, N, K = 16000, 1024, 12288
a = torch.randn(M, K, device=dev, dtype=torch.float16); w = torch.randn(N, K, device=dev, dtype=torch.float16)
if os.environ.get("ZERO") == "1":
    a.zero_(); w.zero_()
for _ in range(30):
    c = torch.matmul(a, w.T)
torch.cuda.synchronize()
