"""one pass of the weight-gradient GEMM over the step's shapes, for rocprofv3 --kernel-trace --stats (TRX_NN_LIB selects the build)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from textreact_amd.predictor import ops
torch.manual_seed(0)
for (M, N, K) in ((16384, 2304, 768), (16384, 768, 768), (16384, 3072, 768), (16384, 768, 3072), (5120, 2304, 768), (5120, 3072, 768), (16384, 1536, 768)):
    dy = torch.randn(M, N, device="cuda").to(torch.bfloat16); x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    for _ in range(20):
        ops.gemm_tn(dy, x, colsum=True, out_dtype=torch.float32)
    torch.cuda.synchronize()
