"""host cost per call of the library products the step makes (tiny operands: the GPU is never the limit)"""
import time, torch
dev = "cuda"
x = torch.randn(256, 768, device=dev).bfloat16(); w = torch.randn(768, 768, device=dev).bfloat16(); b = torch.randn(768, device=dev).bfloat16()
wt = w.t().contiguous(); out = torch.empty(256, 768, device=dev, dtype=torch.bfloat16)
def t(fn, n=2000):
    for _ in range(50): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
print("F.linear(x, w, b)      %.1f us" % t(lambda: torch.nn.functional.linear(x, w, b)))
print("F.linear(x, w)         %.1f us" % t(lambda: torch.nn.functional.linear(x, w)))
print("torch.matmul(x, wt)    %.1f us" % t(lambda: torch.matmul(x, wt)))
print("torch.mm(x, wt)        %.1f us" % t(lambda: torch.mm(x, wt)))
print("torch.mm(x, wt, out=)  %.1f us" % t(lambda: torch.mm(x, wt, out=out)))
print("torch.addmm(b, x, wt)  %.1f us" % t(lambda: torch.addmm(b, x, wt)))
print("x + x (elementwise)    %.1f us" % t(lambda: x + x))
print("torch.empty            %.1f us" % t(lambda: torch.empty(256, 768, device=dev, dtype=torch.bfloat16)))
