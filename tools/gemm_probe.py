"""is one packed QKV GEMM faster than three? (python3 tools/gemm_probe.py)"""
import torch, time
def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); s = torch.cuda.Event(True); e = torch.cuda.Event(True); s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n
for M in (16384, 5120):
    x = torch.randn(M, 768, device="cuda").bfloat16()
    w = [torch.randn(768, 768, device="cuda").bfloat16() for _ in range(3)]
    b = [torch.randn(768, device="cuda").bfloat16() for _ in range(3)]
    wp, bp = torch.cat(w), torch.cat(b)
    w2, b2 = torch.cat(w[:2]), torch.cat(b[:2])
    print(M, "3 x N=768:", round(t(lambda: [torch.nn.functional.linear(x, w[i], b[i]) for i in range(3)]), 4), "ms",
          " packed N=2304:", round(t(lambda: torch.nn.functional.linear(x, wp, bp)), 4), "ms",
          " packed N=1536:", round(t(lambda: torch.nn.functional.linear(x, w2, b2)), 4), "ms")
    g = torch.randn(M, 2304, device="cuda").bfloat16()
    gs = [g[:, i * 768:(i + 1) * 768].contiguous() for i in range(3)]
    print("   dgrad 3x:", round(t(lambda: [gs[i] @ w[i] for i in range(3)]), 4), " packed:", round(t(lambda: g @ wp), 4),
          "  wgrad 3x:", round(t(lambda: [gs[i].t() @ x for i in range(3)]), 4), " packed:", round(t(lambda: g.t() @ x), 4))
