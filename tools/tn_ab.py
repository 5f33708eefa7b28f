"""weight-gradient GEMM (trx_gemm_tn_bf16 + its reduction launch) at the encoder's and decoder's Linear shapes; run once per library
build (TRX_NN_LIB=...) on the same box: python3 tools/tn_ab.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench_predictor as bp
from textreact_amd.predictor import ops
torch.manual_seed(0)
for (M, N, K) in ((16384, 2304, 768), (16384, 768, 768), (16384, 3072, 768), (16384, 768, 3072), (5120, 2304, 768), (5120, 3072, 768), (16384, 1536, 768)):
    dy = torch.randn(M, N, device="cuda").to(torch.bfloat16); x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    ts = sorted(bp.timeit(lambda: ops.gemm_tn(dy, x, colsum=True, out_dtype=torch.float32), iters=30) for _ in range(5))
    dw, db = ops.gemm_tn(dy, x, colsum=True, out_dtype=torch.float32)
    ref = dy.float().t() @ x.float()
    err = float((dw - ref).abs().max() / ref.abs().max())
    print(M, N, K, "median %.1f us min %.1f us" % (ts[2] * 1e3, ts[0] * 1e3), "%.0f TFLOP/s" % (2.0 * M * N * K / ts[2] / 1e9), "rel err %.2e" % err)
