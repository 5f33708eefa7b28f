"""trx_gemm_tn_bf16 (dW = dY^T X) against torch: correctness and speed on the predictor's shapes"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from textreact_amd.predictor import ops
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s = torch.cuda.Event(True); e = torch.cuda.Event(True); s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n
g = torch.Generator(device="cuda").manual_seed(0)
bad = 0
for (M, N, K) in [(64, 256, 256), (128, 256, 512), (448, 512, 256), (5120, 768, 768), (16384, 3072, 768), (16384, 768, 3072), (16384, 2304, 768)]:
    dy = torch.randn(M, N, device="cuda", generator=g).bfloat16(); x = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    c = ops.gemm_tn(dy, x)
    ref = dy.float().t() @ x.float()
    err = float((c.float() - ref).abs().max()) / max(1.0, float(ref.abs().max()))
    ok = err < 1e-2; bad += not ok
    mine = t(lambda: ops.gemm_tn(dy, x)); lib_ = t(lambda: dy.t() @ x)
    fl = 2.0 * M * N * K
    print("M %5d N %4d K %4d  rel err %.5f %s   ours %.1f us (%.0f TF/s)  library %.1f us (%.0f TF/s)" %
          (M, N, K, err, "OK" if ok else "FAIL", mine * 1e3, fl / mine / 1e9, lib_ * 1e3, fl / lib_ / 1e9))
big = torch.randn(1024, 2304, device="cuda", generator=g).bfloat16(); a = big[:, 768:1536]; x = torch.randn(1024, 768, device="cuda", generator=g).bfloat16()
c = ops.gemm_tn(a, x); ref = a.float().t() @ x.float()
print("strided A rel err", float((c.float() - ref).abs().max()) / float(ref.abs().max()))
print("failures", bad)
