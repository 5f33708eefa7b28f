"""time the mixed-storage add+LayerNorm backward at the training shape"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from textreact_amd.predictor import ops
def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); s = torch.cuda.Event(True); e = torch.cuda.Event(True); s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n
rows, cols = 16384, 768
x = torch.randn(rows, cols, device="cuda").bfloat16().requires_grad_(True); r = torch.randn(rows, cols, device="cuda").requires_grad_(True)
g = torch.ones(cols, device="cuda", requires_grad=True); b = torch.zeros(cols, device="cuda", requires_grad=True); xb = torch.zeros(cols, device="cuda", requires_grad=True)
y, ylow = ops.add_layernorm(x, r, g, b, 1e-12, dropout_p=0.1, seed=1, dual=True, bias=xb)
d32, d16 = torch.randn_like(y), torch.randn_like(ylow)
ms = t(lambda: torch.autograd.grad((y, ylow), (x, r, g, b, xb), (d32, d16), retain_graph=True))
print("add_ln_bwd mixed %.1f us  (%.2f TB/s over 225 MB)" % (ms * 1e3, 225e6 / (ms * 1e-3) / 1e12))
