"""round 5: the encoder self-attention shape alone (B 32, 12 heads, 512 x 512, bf16, key mask), N forward + backward calls:
the target of the LDS / wait counter passes of tools/r05/eighteenth.sh"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from textreact_amd.predictor import ops
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10
B, H, L = 32, 12, 512
g = torch.Generator(device="cuda").manual_seed(0)
q = torch.randn(B, L, H, 64, device="cuda", generator=g).bfloat16().requires_grad_()
k = torch.randn(B, L, H, 64, device="cuda", generator=g).bfloat16().requires_grad_()
v = torch.randn(B, L, H, 64, device="cuda", generator=g).bfloat16().requires_grad_()
mask = torch.zeros(B, L, device="cuda")
for _ in range(N):
    o = ops.attention(q, k, v, mask=mask, causal=False)
    o.backward(torch.ones_like(o))
torch.cuda.synchronize()
print("done")
