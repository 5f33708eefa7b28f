"""what the in-kernel dropout costs: attention forward / backward at the encoder shape with p = 0 and p = 0.1"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from textreact_amd.predictor import ops
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s = torch.cuda.Event(True); e = torch.cuda.Event(True); s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n
B, H, L = 32, 12, 512
qkv = torch.randn(B, L, 3, H, 64, device="cuda").bfloat16().requires_grad_(True)
m = torch.zeros(B, L, device="cuda")
for p in (0.0, 0.1):
    fwd = t(lambda: ops.attention_qkv(qkv.detach(), mask=m, dropout_p=p, seed=5))
    o = ops.attention_qkv(qkv, mask=m, dropout_p=p, seed=5)
    do = torch.randn_like(o)
    bwd = t(lambda: torch.autograd.grad(o, qkv, do, retain_graph=True))
    print("p = %.1f: forward %.1f us, backward %.1f us" % (p, fwd * 1e3, bwd * 1e3))
