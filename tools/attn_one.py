"""one MFMA attention forward shape, repeated: target for rocprofv3 (python3 tools/attn_one.py [B H L causal])"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from textreact_amd.predictor import ops
B, H, L = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (32, 12, 512)
causal = bool(int(sys.argv[4])) if len(sys.argv) > 4 else False
g = torch.Generator(device='cuda').manual_seed(0)
q, k, v = (torch.randn(B, L, H, 64, device='cuda', generator=g).bfloat16() for _ in range(3))
mask = torch.zeros(B, L, device='cuda')
for _ in range(20):
    o = ops.attention(q, k, v, mask=mask, causal=causal, backend="hip")
torch.cuda.synchronize()
