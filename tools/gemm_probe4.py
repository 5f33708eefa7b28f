import torch
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s = torch.cuda.Event(True); e = torch.cuda.Event(True); s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n
g = torch.Generator(device="cuda").manual_seed(0)
for (M, N, K) in [(16384, 3072, 768), (16384, 768, 3072), (16384, 2304, 768), (16384, 768, 768), (5120, 3072, 768)]:
    x = torch.randn(M, K, device="cuda", generator=g).bfloat16(); dy = torch.randn(M, N, device="cuda", generator=g).bfloat16()
    ref = (dy.float().t() @ x.float())
    variants = {
        "dy.t() @ x": lambda: dy.t() @ x,
        "(x.t() @ dy).t()": lambda: (x.t() @ dy).t(),
        "transpose dy, then NN": lambda: dy.t().contiguous() @ x,
        "both transposed, NT": lambda: torch.nn.functional.linear(dy.t().contiguous(), x.t().contiguous()),
        "fp32 accumulate split-M x4": lambda: sum((dy[i::4].t() @ x[i::4]) for i in range(4)),
    }
    print("wgrad M %d N %d K %d" % (M, N, K))
    for name, fn in variants.items():
        out = fn()
        err = float((out.float() - ref).abs().max()) / float(ref.abs().max())
        print("   %-28s %.1f us  rel err %.4f" % (name, t(fn) * 1e3, err))
