"""round 5: the persistent forward kernel (attn_fwd_persist.h, TRX_NN_ATTN_PERSIST=1) against attention_fwd_mfma_kernel: the switch
is read once per process, so this prints one sha256 per case of (output, log-sum-exp) and the caller runs it twice -- with and
without the switch -- and compares (tests/test_predictor_gpu.py::test_persistent_forward_kernel; the per-query arithmetic is the
same, so the bits are).  python3 tools/r05/persist_check.py"""
import ctypes
import hashlib
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
vp, i32, f32, u64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_uint64
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
L = ctypes.CDLL(os.path.join(root, "textreact_amd", "csrc", os.environ.get("TRX_NN_LIB", "libtrxnn.so")))
L.trx_attention_fwd_dropout.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, f32, i32, f32, u64, vp, vp, vp]
dev = torch.device("cuda", 0)
P = lambda t: vp(t.data_ptr()) if t is not None else None
out = {}
# (B, H, Lq, Lk, mask mode, dropout): more than 768 items of 128 queries each (else the plain kernel runs either way); 780, 792,
# 840, 960 items: workgroups with one item and workgroups with two
for (B, H, Lq, Lk, mm, p) in [(32, 12, 512, 512, 1, 0.0), (32, 12, 512, 512, 1, 0.1), (32, 12, 512, 512, 0, 0.0), (40, 12, 256, 256, 1, 0.1), (13, 12, 640, 384, 1, 0.0),
                              (70, 12, 128, 512, 0, 0.1), (33, 8, 384, 128, 1, 0.3), (32, 12, 512, 448, 1, 0.0)]:
    g = torch.Generator(device=dev); g.manual_seed(Lq + Lk)
    q = torch.randn(B, Lq, H, 64, device=dev, generator=g).bfloat16(); k = torch.randn(B, Lk, H, 64, device=dev, generator=g).bfloat16()
    v = torch.randn(B, Lk, H, 64, device=dev, generator=g).bfloat16()
    m = torch.zeros(B, Lk, device=dev); m[::3, Lk * 3 // 4:] = -1e4; m[1::5, : Lk // 8] = torch.finfo(torch.float32).min
    o = torch.full((B, Lq, H * 64), float("nan"), device=dev, dtype=torch.bfloat16); lse = torch.full((B, H, Lq), float("nan"), device=dev)
    rc = L.trx_attention_fwd_dropout(P(q), P(k), P(v), P(m) if mm else None, mm, 0, B, H, Lq, Lk, 0.125, 1, p, 1234, P(o), P(lse), None)
    torch.cuda.synchronize()
    assert rc == 0, rc
    assert not bool(torch.isnan(o.float()).any()) and not bool(torch.isnan(lse).any())
    h = hashlib.sha256(o.view(torch.int16).cpu().numpy().tobytes()); h.update(lse.cpu().numpy().tobytes())
    out["%d x %d x %d x %d mask %d p %.1f" % (B, H, Lq, Lk, mm, p)] = h.hexdigest()
print(json.dumps(out))
