"""the library's bf16 GEMM (hipBLASLt through torch.matmul) on the scan kernel's contraction shape, for reference:
65,536 queries x N corpus rows x 768, output discarded (python3 tools/gemm_ref.py)"""
import torch, time
torch.manual_seed(0)
q = torch.randn(65536, 768, device="cuda").bfloat16()
for n in (8192, 32768):
    y = torch.randn(n, 768, device="cuda").bfloat16()
    out = torch.empty(65536, n, device="cuda", dtype=torch.bfloat16)
    for _ in range(3): torch.matmul(q, y.t(), out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps): torch.matmul(q, y.t(), out=out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    fl = 2.0 * 65536 * n * 768
    print("N %6d: %.3f ms  %.0f TFLOP/s (writes %.1f GB of bf16 output per call); scaled to 1M rows: %.1f ms" % (n, ms, fl / ms / 1e9, 65536 * n * 2 / 1e9, ms * 1e6 / n))
