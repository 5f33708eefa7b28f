"""the library's rates on the Linear layers' forward (x W^T + b) and input-gradient (dY W) products at the step's shapes -- what
an own GEMM would have to beat.  python3 tools/lib_gemm_rates.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench_predictor as bp
torch.manual_seed(0)
dev = "cuda"
for M in (16384, 5120):
    for (N, K, name) in ((2304, 768, "qkv"), (768, 768, "out"), (3072, 768, "ffn up"), (768, 3072, "ffn down"), (1536, 768, "cross kv")):
        x = torch.randn(M, K, device=dev).to(torch.bfloat16); w = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
        b = torch.randn(N, device=dev).to(torch.bfloat16); dy = torch.randn(M, N, device=dev).to(torch.bfloat16)
        f = sorted(bp.timeit(lambda: torch.nn.functional.linear(x, w, b), iters=30) for _ in range(3))[1]
        g = sorted(bp.timeit(lambda: torch.matmul(dy, w), iters=30) for _ in range(3))[1]
        fl = 2.0 * M * N * K
        print("M %5d %-9s N %4d K %4d  forward %.1f us %.0f TFLOP/s   dX %.1f us %.0f TFLOP/s" % (M, name, N, K, f * 1e3, fl / f / 1e9, g * 1e3, fl / g / 1e9))
