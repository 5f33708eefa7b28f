"""the weight-gradient GEMM's main launch at 252 tiles without a split, as a function of the token count: slope = the loop's
rate per 64-row step with the whole chip busy, intercept = prologue + the store of the accumulators"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench_predictor as bp
from textreact_amd.predictor import ops
torch.manual_seed(0)
N, K = 16128, 1024
for M in (1024, 2368, 4736, 9472, 16384):
    dy = torch.randn(M, N, device="cuda").to(torch.bfloat16); x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    ts = sorted(bp.timeit(lambda: ops.gemm_tn(dy, x, colsum=True, out_dtype=torch.float32), iters=20) for _ in range(5))
    print(M, "steps", (M + 63) // 64, "median %.1f us" % (ts[2] * 1e3), "%.0f TFLOP/s" % (2.0 * M * N * K / ts[2] / 1e9))
