"""GPU numerics of the predictor kernels (through the C ABI) against a plain fp32 PyTorch statement
of the same op, and the whole model against the golden logits of the reference's get_model.
Tolerances: fp32 kernels 2e-5 abs (same math, different summation order); bf16 storage 2e-2;
model logits 1e-3 (BASELINE.json north_star)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import nn_ref
from textreact_amd.predictor import ops
from textreact_amd.predictor.model import Config, TextReactModel, random_state_dict

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "predictor_small.npz")


def _rand(*shape, dtype=torch.float32, seed=0):
    g = torch.Generator(device="cuda"); g.manual_seed(seed)
    return torch.randn(shape, generator=g, device="cuda", dtype=torch.float32).to(dtype)


@pytest.mark.parametrize("rows,cols", [(1, 768), (777, 768), (16384, 768), (33, 600), (5, 31090 // 10), (64, 4096)])
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 3e-2)])
@pytest.mark.parametrize("with_res", [True, False])
def test_add_layernorm_forward(rows, cols, dtype, tol, with_res):
    x, r = _rand(rows, cols, dtype=dtype, seed=1), (_rand(rows, cols, dtype=dtype, seed=2) if with_res else None)
    g, b = _rand(cols, seed=3) * 0.1 + 1, _rand(cols, seed=4) * 0.1
    for eps in (1e-12, 1e-5):
        y = ops.add_layernorm(x, r, g, b, eps)
        ref = nn_ref.add_layernorm(x.float(), None if r is None else r.float(), g, b, eps)
        assert y.dtype == dtype and float((y.float() - ref).abs().max()) <= tol


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 3e-2)])
def test_add_layernorm_parameters_at_an_odd_offset(dtype, tol):
    """gamma / beta as views of a flattened parameter buffer that start 4 bytes into a 16-byte line (round 5 made the vector
    kernel read them 16 bytes at a time): the call takes the narrow kernel and gives the same answer, forward and backward"""
    x, r = _rand(300, 768, dtype=dtype, seed=1), _rand(300, 768, dtype=dtype, seed=2)
    flat = _rand(2 * 768 + 8, seed=3) * 0.1
    g, b = (flat[1:769] + 1).detach().clone(), None
    buf = torch.empty(2 * 768 + 8, device="cuda"); buf[1:769] = g; buf[773:1541] = flat[773:1541]
    gv, bv = buf[1:769], buf[773:1541]
    assert gv.data_ptr() % 16 == 4 and bv.data_ptr() % 16 == 4
    y = ops.add_layernorm(x, r, gv, bv, 1e-5)
    ref = nn_ref.add_layernorm(x.float(), r.float(), gv, bv, 1e-5)
    assert float((y.float() - ref).abs().max()) <= tol
    gp = gv.clone().requires_grad_(True); xp = x.clone().requires_grad_(True)
    ops.add_layernorm(xp, r, gp[:], bv, 1e-5).float().square().sum().backward()
    gq = gv.clone().requires_grad_(True); xq = x.float().clone().requires_grad_(True)
    nn_ref.add_layernorm(xq, r.float(), gq, bv, 1e-5).square().sum().backward()
    assert float((gp.grad - gq.grad).abs().max()) <= tol * max(1.0, float(gq.grad.abs().max()))


def test_attention_refuses_operands_that_are_not_16_byte_aligned():
    """the matrix-core kernels move rows in 16-byte pieces: an output (or q, k, v) view at an 8-byte offset is an EINVAL, not a
    misaligned store"""
    import ctypes
    L = ops.lib()
    q = torch.zeros(1 * 128 * 2 * 64 + 8, device="cuda", dtype=torch.bfloat16)
    o = torch.zeros(1 * 128 * 2 * 64 + 8, device="cuda", dtype=torch.bfloat16)
    vp = ctypes.c_void_p
    st = vp(torch.cuda.current_stream().cuda_stream)
    args = lambda qq, oo: (vp(qq.data_ptr()), vp(qq.data_ptr()), vp(qq.data_ptr()), None, 0, 0, 1, 2, 128, 128, ctypes.c_float(0.125), 1, vp(oo.data_ptr()), st)
    L.trx_attention_fwd.argtypes = [vp, vp, vp, vp] + [ctypes.c_int] * 6 + [ctypes.c_float, ctypes.c_int, vp, vp]
    assert L.trx_attention_fwd(*args(q, o)) == 0
    assert L.trx_attention_fwd(*args(q, o[4:])) == -1 and L.trx_attention_fwd(*args(q[4:], o)) == -1
    torch.cuda.synchronize()


@pytest.mark.parametrize("rows,cols", [(1000, 768), (37, 130)])
def test_add_layernorm_backward(rows, cols):
    x, r = _rand(rows, cols, seed=1), _rand(rows, cols, seed=2)
    g, b = _rand(cols, seed=3) * 0.1 + 1, _rand(cols, seed=4) * 0.1
    dy = _rand(rows, cols, seed=5)
    outs = []
    for backend in ("hip", "torch"):
        with nn_ref.implementation(backend):
            xs, rs, gs, bs = (t.clone().requires_grad_(True) for t in (x, r, g, b))
            y = ops.add_layernorm(xs, rs, gs, bs, 1e-5)
            y.backward(dy)
            outs.append((xs.grad, rs.grad, gs.grad, bs.grad))
    for a, c in zip(*outs):
        assert float((a - c).abs().max()) <= 2e-4 * max(1.0, float(c.abs().max()))
    # deterministic: two runs give the same bits (two-stage reduction, no float atomics)
    xs = x.clone().requires_grad_(True); gs = g.clone().requires_grad_(True)
    ops.add_layernorm(xs, r, gs, b, 1e-5).backward(dy)
    assert torch.equal(gs.grad, outs[0][2])


@pytest.mark.parametrize("B,H,Lq,Lk,mask,causal", [
    (2, 12, 512, 512, "key", False),      # encoder self-attention at the scripts' length
    (3, 2, 37, 37, "none", False),
    (2, 4, 70, 70, "full", False),        # --unattend_nonbonds style 2-D mask
    (2, 12, 7, 7, "key", True),           # RCR decoder: BOS + 5 + EOS
    (2, 12, 160, 160, "none", True),      # retro decoder self-attention
    (2, 12, 160, 512, "key", False),      # cross-attention over the encoder states
    (4, 2, 1, 129, "none", True),         # one decode step against a KV prefix
    (1, 1, 65, 200, "key", True),
    (1, 2, 130, 2100, "key", False),      # key mask longer than the 1024 keys the kernel keeps in LDS at a time
    (1, 2, 300, 1100, "full", True),
    (2, 12, 7, 512, "key", False),        # RCR cross-attention: one 32-query unit, its keys split four ways inside the workgroup
    (2, 3, 40, 200, "full", False),       # two units, keys split two ways
    (1, 2, 161, 161, "key", True),        # a one-query tail block under causality with three key tiles: too few to split
    (1, 2, 161, 600, "key", True),        # ... with ten: split four ways, the waves' last tiles cut by the diagonal
    (1, 2, 20, 140, "none", False),       # three tiles for one unit: not split (fewer than two per wave)
])
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 2e-2)])
def test_attention_forward(B, H, Lq, Lk, mask, causal, dtype, tol):
    q, k, v = _rand(B, Lq, H, 64, dtype=dtype, seed=1), _rand(B, Lk, H, 64, dtype=dtype, seed=2), _rand(B, Lk, H, 64, dtype=dtype, seed=3)
    m = None
    neg = torch.finfo(torch.float32).min
    if mask == "key":
        keep = torch.ones(B, Lk, device="cuda"); keep[0, Lk // 2 + 1:] = 0
        m = (1 - keep) * neg
    elif mask == "full":
        keep = (torch.rand(B, Lq, Lk, device="cuda") > 0.3).float(); keep[:, :, 0] = 1
        m = (1 - keep) * neg
    out = ops.attention(q, k, v, mask=m, causal=causal)
    ref = nn_ref.attention(q.float(), k.float(), v.float(), mask=m, causal=causal)
    assert out.shape == (B, Lq, H * 64) and out.dtype == dtype
    assert float((out.float() - ref).abs().max()) <= tol


@pytest.mark.parametrize("Lq,Lk,masked", [
    (20, 600, (64, 600)),      # one 32-query unit, its keys split over four waves: three of them see masked keys only
    (20, 77, (64, 77)),        # (too few tiles to split: the second tile of the one wave is all masked)
    (40, 300, (64, 300)),      # two units, keys split two ways: the second wave of a unit meets all-masked tiles only
    (128, 200, (0, 70)),       # no split: left padding, the FIRST tile of every wave is all masked (latent since round 1)
    (7, 512, (128, 512)),      # RCR cross-attention over a short encoder input
])
def test_attention_tiles_whose_keys_are_all_masked(Lq, Lk, masked):
    """a key tile in which every key is masked (finfo.min, as Hugging Face builds its masks) opens the softmax of a wave with a
    huge negative reference; the exponent's argument must stay within rounding of it -- with the -1e30 floor of rounds 1-3 it
    came out as +-1e21 and the tile produced inf / NaN (found when the key-split waves of round 4 began at later tiles)"""
    bf = torch.bfloat16
    q, k, v = _rand(2, Lq, 3, 64, dtype=bf, seed=1), _rand(2, Lk, 3, 64, dtype=bf, seed=2), _rand(2, Lk, 3, 64, dtype=bf, seed=3)
    dout = _rand(2, Lq, 3 * 64, dtype=bf, seed=4)
    m = torch.zeros(2, Lk, device="cuda")
    m[0, masked[0]:masked[1]] = torch.finfo(torch.float32).min
    with nn_ref.implementation("hip"):
        qs, ks, vs = (t.clone().requires_grad_(True) for t in (q, k, v))
        out = ops.attention(qs, ks, vs, mask=m)
        out.backward(dout)
    qr, kr, vr = (t.float().requires_grad_(True) for t in (q, k, v))
    ref = nn_ref.attention(qr, kr, vr, mask=m)
    ref.backward(dout.float())
    assert torch.isfinite(out).all() and float((out.float() - ref).abs().max()) <= 2e-2
    for name, a, c in (("dq", qs.grad, qr.grad), ("dk", ks.grad, kr.grad), ("dv", vs.grad, vr.grad)):
        assert torch.isfinite(a).all(), name
        assert float((a.float() - c).abs().max()) <= 2e-2 * max(1.0, float(c.abs().max())), name


def test_model_logits_match_reference_golden_on_gpu():
    z = np.load(G)
    enc, dec = json.loads(str(z["enc_cfg"])), json.loads(str(z["dec_cfg"]))
    m = TextReactModel(Config(**enc), Config(is_decoder=True, **dec))
    m.load_state_dict(random_state_dict(m, int(z["seed"])))
    m = m.cuda().eval()
    t = lambda k_: torch.from_numpy(z[k_]).cuda()
    with torch.no_grad():
        logits, encs = m(t("input_ids"), t("attention_mask"), t("decoder_input_ids"), t("decoder_attention_mask"))
    assert float((logits.cpu() - torch.from_numpy(z["logits"])).abs().max()) <= 1e-3
    assert float((encs.cpu() - torch.from_numpy(z["encoder_last_hidden_state"])).abs().max()) <= 1e-3


def _full_size_case(T, device):
    """the model and inputs tests/golden/predictor_full.npz was generated with (make_predictor_golden.py: main_full)"""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    from make_predictor_golden import full_inputs
    z = np.load(os.path.join(os.path.dirname(G), "predictor_full.npz"))
    enc, dec = json.loads(str(z["enc_cfg"])), json.loads(str(z["dec_cfg"]))
    m = TextReactModel(Config(**enc), Config(is_decoder=True, **dec))
    m.load_state_dict(random_state_dict(m, int(z["seed"])))
    return z, m.to(device).eval(), [t.to(device) for t in full_inputs(T)]


@pytest.mark.parametrize("T", [7, 160])
def test_full_size_model_matches_the_reference_golden(T):
    """scripts' shapes -- BERT-base encoder (SciBERT vocabulary), the bert_l6.json decoder, L = 512, T = 7 (RCR) and
    160 (RetroSyn) -- through the HIP fp32 path against logits of the REFERENCE's own get_model (the full-size
    golden): <= 1e-3, the north-star tolerance"""
    z, m, (ids, am, dids) = _full_size_case(T, "cuda")
    with torch.no_grad():
        logits, enc = m(ids, am, dids)
    logits, enc = logits.cpu().numpy(), enc.cpu().numpy()
    if T == 7:
        assert np.abs(logits - z["logits_T7"]).max() <= 1e-3
        pos = z["enc_pos"]
        assert np.abs(enc[pos[:, 0], pos[:, 1]] - z["enc_at"]).max() <= 1e-3
    else:
        pos = z["logits_pos_T160"]
        assert np.abs(logits[pos[:, 0], pos[:, 1]] - z["logits_at_T160"]).max() <= 1e-3


@pytest.mark.parametrize("cols", [64, 256, 512, 520, 768, 1024, 1280, 1536, 2040, 2048])
@pytest.mark.parametrize("variant,tol", [("fp32", 3e-5), ("bf16", 3e-2), ("mixed", 2e-2)])
def test_add_layernorm_every_chunk_count(cols, variant, tol):
    """the vector kernels are compiled per number of 16-byte chunks a lane holds (1 .. 4: nn_ops.hip NC); widths on both sides of
    every boundary, each storage variant, forward and backward with dropout, against the PyTorch statement.  (Widths the
    vector path does not take -- fp32 / mixed beyond 1024 columns -- run the scalar kernels or the widened path: same answer.)"""
    rows = 301
    xd = torch.float32 if variant == "fp32" else torch.bfloat16
    rd = torch.bfloat16 if variant == "bf16" else torch.float32
    x, r = _rand(rows, cols, dtype=xd, seed=1), _rand(rows, cols, dtype=rd, seed=2)
    g, b = _rand(cols, seed=3) * 0.2 + 1, _rand(cols, seed=4) * 0.2
    dy = _rand(rows, cols, dtype=rd, seed=5)
    res = []
    for backend in ("hip", "torch"):
        with nn_ref.implementation(backend):
            hip = backend == "hip"
            xs = (x.clone() if hip else x.float().clone()).requires_grad_(True)
            rs = (r.clone() if hip else r.float().clone()).requires_grad_(True)
            gs, bs = g.clone().requires_grad_(True), b.clone().requires_grad_(True)
            y = ops.add_layernorm(xs, rs, gs, bs, 1e-12, dropout_p=0.1, seed=5)
            y.backward(dy if hip else dy.float())
            res.append((y.detach().float(), xs.grad.float(), rs.grad.float(), gs.grad, bs.grad))
    for a, c in zip(*res):
        assert float((a - c).abs().max()) <= tol * max(1.0, float(c.abs().max())), (cols, variant)


@pytest.mark.parametrize("B,H,Lq,Lk,mask,causal", [
    (2, 3, 70, 70, "key", False), (2, 2, 33, 33, "none", True), (2, 2, 20, 77, "key", False),
    (1, 2, 65, 65, "full", False), (1, 1, 5, 37, "none", True),
])
def test_attention_backward(B, H, Lq, Lk, mask, causal):
    q, k, v = _rand(B, Lq, H, 64, seed=1), _rand(B, Lk, H, 64, seed=2), _rand(B, Lk, H, 64, seed=3)
    dout = _rand(B, Lq, H * 64, seed=4)
    neg = torch.finfo(torch.float32).min
    m = None
    if mask == "key":
        keep = torch.ones(B, Lk, device="cuda"); keep[0, Lk // 2 + 1:] = 0
        m = (1 - keep) * neg
    elif mask == "full":
        keep = (torch.rand(B, Lq, Lk, device="cuda") > 0.3).float(); keep[:, :, 0] = 1
        m = (1 - keep) * neg
    grads = []
    for backend in ("hip", "torch"):
        with nn_ref.implementation(backend):
            qs, ks, vs = (t.clone().requires_grad_(True) for t in (q, k, v))
            out = ops.attention(qs, ks, vs, mask=m, causal=causal)
            out.backward(dout)
            grads.append((out.detach(), qs.grad, ks.grad, vs.grad))
    for a, c in zip(*grads):
        assert float((a - c).abs().max()) <= 5e-5 * max(1.0, float(c.abs().max()))


@pytest.mark.parametrize("B,H,Lq,Lk,mask,causal", [
    (2, 3, 70, 70, "key", False), (2, 2, 33, 33, "none", True), (2, 2, 20, 77, "key", False),
    (1, 2, 65, 65, "full", False), (1, 1, 5, 37, "none", True),
    (2, 4, 512, 512, "key", False),      # encoder self-attention, several tiles either way
    (2, 4, 160, 160, "none", True),      # retro decoder self-attention
    (2, 4, 160, 512, "key", False),      # cross-attention
    (1, 2, 300, 1100, "key", False),     # key mask spans more than one 1024-key chunk
    (1, 2, 257, 129, "none", False), (1, 2, 129, 257, "none", True), (1, 1, 200, 200, "full", True),
    # query blocks of one or two 32-query units split their KEYS over the idle waves (round 4): RCR's decoder (7 positions)
    # over the encoder states, two units over several key tiles, a tail unit with a per-element mask, causal with idle key waves
    (2, 12, 7, 512, "key", False), (2, 2, 40, 200, "key", False), (2, 2, 160, 512, "full", False), (1, 2, 7, 7, "key", True),
    (1, 3, 161, 300, "key", True), (1, 2, 161, 600, "key", True),
])
def test_attention_backward_bf16_matrix_cores(B, H, Lq, Lk, mask, causal):
    """bf16 in / bf16 out through the MFMA backward (attn_bwd_mfma.h) against fp32 autograd on the
    same bf16-rounded inputs; tolerance = bf16 rounding of P, dS and of the outputs"""
    bf = torch.bfloat16
    q, k, v = _rand(B, Lq, H, 64, dtype=bf, seed=1), _rand(B, Lk, H, 64, dtype=bf, seed=2), _rand(B, Lk, H, 64, dtype=bf, seed=3)
    dout = _rand(B, Lq, H * 64, dtype=bf, seed=4)
    neg = torch.finfo(torch.float32).min
    m = None
    if mask == "key":
        keep = torch.ones(B, Lk, device="cuda"); keep[0, Lk // 2 + 1:] = 0
        m = (1 - keep) * neg
    elif mask == "full":
        keep = (torch.rand(B, Lq, Lk, device="cuda") > 0.3).float(); keep[:, :, 0] = 1
        m = (1 - keep) * neg
    qs, ks, vs = (t.clone().requires_grad_(True) for t in (q, k, v))
    out = ops.attention(qs, ks, vs, mask=m, causal=causal)
    out.backward(dout)
    qr, kr, vr = (t.float().clone().requires_grad_(True) for t in (q, k, v))
    ref = nn_ref.attention(qr, kr, vr, mask=m, causal=causal)
    ref.backward(dout.float())
    for name, a, c in (("dq", qs.grad, qr.grad), ("dk", ks.grad, kr.grad), ("dv", vs.grad, vr.grad)):
        assert a.dtype == bf and a.shape == c.shape
        err = float((a.float() - c).abs().max())
        assert err <= 2e-2 * max(1.0, float(c.abs().max())), (name, err, float(c.abs().max()))


# ---- training mode: dropout -------------------------------------------------------------------------
def test_dropout_decisions_statistics_and_determinism():
    keep = ops.dropout_keep_mask(1234, 0.1, 6, 300, 257, "cuda")
    assert keep.shape == (6, 300, 257)
    frac = float(keep.float().mean())
    assert abs(frac - 0.9) < 3e-3, frac                        # 462,600 draws: sigma = 4.4e-4
    assert torch.equal(keep, ops.dropout_keep_mask(1234, 0.1, 6, 300, 257, "cuda"))
    other = ops.dropout_keep_mask(1235, 0.1, 6, 300, 257, "cuda")
    assert 0.15 < float((keep != other).float().mean()) < 0.21     # independent: 2 * 0.9 * 0.1 = 0.18
    # no structure along any axis: neighbouring rows / columns / streams agree as often as chance says
    for a, b in ((keep[:, 1:], keep[:, :-1]), (keep[:, :, 1:], keep[:, :, :-1]), (keep[1:], keep[:-1])):
        assert abs(float((a == b).float().mean()) - 0.82) < 6e-3
    assert bool(ops.dropout_keep_mask(7, 0.0, 1, 4, 64, "cuda").all())
    assert abs(float(ops.dropout_keep_mask(7, 0.5, 1, 512, 512, "cuda").float().mean()) - 0.5) < 4e-3


@pytest.mark.parametrize("rows,cols", [(777, 768), (33, 600), (5, 3109)])
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 3e-5), (torch.bfloat16, 3e-2)])
@pytest.mark.parametrize("with_res", [True, False])
def test_add_layernorm_dropout_forward_backward(rows, cols, dtype, tol, with_res):
    x, r = _rand(rows, cols, dtype=dtype, seed=1), (_rand(rows, cols, dtype=dtype, seed=2) if with_res else None)
    g, b = _rand(cols, seed=3), _rand(cols, seed=4)
    dy = _rand(rows, cols, dtype=dtype, seed=5)
    res = []
    for backend in ("hip", "torch"):
        with nn_ref.implementation(backend):
            xs = x.clone().requires_grad_(True) if backend == "hip" else x.float().clone().requires_grad_(True)
            rs = None if r is None else (r.clone() if backend == "hip" else r.float().clone()).requires_grad_(True)
            gs, bs = g.clone().requires_grad_(True), b.clone().requires_grad_(True)
            y = ops.add_layernorm(xs, rs, gs, bs, 1e-5, dropout_p=0.1, seed=99)
            y.backward(dy if backend == "hip" else dy.float())
            res.append((y.detach().float(), xs.grad.float(), None if rs is None else rs.grad.float(), gs.grad, bs.grad))
    keep = ops.dropout_keep_mask(99, 0.1, 1, rows, cols, "cuda").view(rows, cols)
    assert bool((res[0][1][~keep] == 0).all())                 # dropped elements get no gradient
    for a, c in zip(*res):
        if a is None:
            continue
        assert float((a - c).abs().max()) <= tol * max(1.0, float(c.abs().max()))
    # p = 0 and eval are the plain op; a different seed drops different elements
    y0 = ops.add_layernorm(x, r, g, b, 1e-5, dropout_p=0.0)
    assert torch.equal(y0, ops.add_layernorm(x, r, g, b, 1e-5))
    assert not torch.equal(ops.add_layernorm(x, r, g, b, 1e-5, dropout_p=0.1, seed=1), ops.add_layernorm(x, r, g, b, 1e-5, dropout_p=0.1, seed=2))


@pytest.mark.parametrize("B,H,Lq,Lk,mask,causal", [
    (2, 3, 70, 70, "key", False), (2, 2, 33, 33, "none", True), (2, 2, 20, 77, "key", False),
    (1, 2, 65, 65, "full", False), (2, 4, 200, 300, "key", False), (1, 2, 160, 160, "none", True),
    (2, 4, 160, 512, "key", False), (2, 3, 7, 512, "key", False), (1, 2, 40, 200, "full", False),     # key-split query blocks
])
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 5e-5), (torch.bfloat16, 2e-2)])
def test_attention_dropout_forward_backward(B, H, Lq, Lk, mask, causal, dtype, tol):
    """fp32 goes through the VALU kernels, bf16 through the matrix-core kernels; both must take the same
    decisions as trx_dropout_keep_mask in the forward AND in both backward passes"""
    q, k, v = _rand(B, Lq, H, 64, dtype=dtype, seed=1), _rand(B, Lk, H, 64, dtype=dtype, seed=2), _rand(B, Lk, H, 64, dtype=dtype, seed=3)
    dout = _rand(B, Lq, H * 64, dtype=dtype, seed=4)
    neg = torch.finfo(torch.float32).min
    m = None
    if mask == "key":
        keep = torch.ones(B, Lk, device="cuda"); keep[0, Lk // 2 + 1:] = 0
        m = (1 - keep) * neg
    elif mask == "full":
        keep = (torch.rand(B, Lq, Lk, device="cuda") > 0.3).float(); keep[:, :, 0] = 1
        m = (1 - keep) * neg
    res = []
    for backend in ("hip", "torch"):
        with nn_ref.implementation(backend):
            qs, ks, vs = ((t if backend == "hip" else t.float()).clone().requires_grad_(True) for t in (q, k, v))
            out = ops.attention(qs, ks, vs, mask=m, causal=causal, dropout_p=0.1, seed=4242)
            out.backward(dout if backend == "hip" else dout.float())
            res.append((out.detach().float(), qs.grad.float(), ks.grad.float(), vs.grad.float()))
    for name, a, c in zip(("out", "dq", "dk", "dv"), *res):
        err = float((a - c).abs().max())
        assert err <= tol * max(1.0, float(c.abs().max())), (name, err)
    plain = ops.attention(q, k, v, mask=m, causal=causal)
    assert torch.equal(plain, ops.attention(q, k, v, mask=m, causal=causal, dropout_p=0.0))
    assert not torch.equal(plain.float(), res[0][0])


def test_model_backward_hip_vs_torch():
    # one training-style step on the small config: loss = CE over decoder logits (main.py:129-131)
    z = np.load(G)
    enc, dec = json.loads(str(z["enc_cfg"])), json.loads(str(z["dec_cfg"]))
    t = lambda k_: torch.from_numpy(z[k_]).cuda()
    grads = {}
    for backend in ("hip", "torch"):
        with nn_ref.implementation(backend):
            m = TextReactModel(Config(**enc), Config(is_decoder=True, **dec))
            m.load_state_dict(random_state_dict(m, int(z["seed"])))
            m = m.cuda().eval()
            logits, _ = m(t("input_ids"), t("attention_mask"), t("decoder_input_ids"), t("decoder_attention_mask"))
            labels = t("decoder_input_ids")[:, 1:]
            loss = torch.nn.functional.cross_entropy(logits[:, :-1].reshape(-1, logits.shape[-1]), labels.reshape(-1), ignore_index=0)
            loss.backward()
            grads[backend] = {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}
            grads[backend]["__loss__"] = loss.detach()
    assert set(grads["hip"]) == set(grads["torch"])
    for n in grads["hip"]:
        a, c = grads["hip"][n], grads["torch"][n]
        assert float((a - c).abs().max()) <= 1e-4 * max(1.0, float(c.abs().max())), n


def test_model_training_mode_dropout_hip_vs_torch():
    """train(): dropout on, the same seeds on both backends (torch.manual_seed fixes the seed stream the ops
    draw from) -> the same elements are dropped, so loss and every gradient must agree"""
    z = np.load(G)
    enc, dec = json.loads(str(z["enc_cfg"])), json.loads(str(z["dec_cfg"]))
    t = lambda k_: torch.from_numpy(z[k_]).cuda()
    grads = {}
    for backend in ("hip", "torch"):
        with nn_ref.implementation(backend):
            m = TextReactModel(Config(**enc), Config(is_decoder=True, **dec))
            m.load_state_dict(random_state_dict(m, int(z["seed"])))
            m = m.cuda().train()
            torch.manual_seed(11)
            logits, _ = m(t("input_ids"), t("attention_mask"), t("decoder_input_ids"), t("decoder_attention_mask"))
            labels = t("decoder_input_ids")[:, 1:]
            loss = torch.nn.functional.cross_entropy(logits[:, :-1].reshape(-1, logits.shape[-1]), labels.reshape(-1), ignore_index=0)
            loss.backward()
            grads[backend] = {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}
            grads[backend]["__loss__"] = loss.detach()
    m.eval()
    with torch.no_grad():
        ev, _ = m(t("input_ids"), t("attention_mask"), t("decoder_input_ids"), t("decoder_attention_mask"))
    assert not torch.allclose(ev, logits.detach(), atol=1e-4)      # dropout did change the forward
    for n in grads["hip"]:
        a, c = grads["hip"][n], grads["torch"][n]
        assert float((a - c).abs().max()) <= 1e-4 * max(1.0, float(c.abs().max())), n


def test_dense_producer_feeds_the_flat_index_on_device(tmp_path):
    """encoder -> [N, hidden] bf16 embeddings in HBM -> flat index -> neighbor file, nothing through the host;
    the ids must be what the oracle finds on exactly those embeddings"""
    from textreact_amd import dense, neighbors
    from oracle import flat_knn as oracle
    cfg = Config(vocab_size=500, num_hidden_layers=2, max_position_embeddings=64)
    torch.manual_seed(0)
    enc = dense.DenseEncoder(cfg).cuda().eval()
    g = torch.Generator().manual_seed(3)
    c_ids = torch.randint(1, 500, (1500, 40), generator=g); c_am = torch.ones_like(c_ids); c_am[::4, 25:] = 0
    q_ids = torch.randint(1, 500, (70, 33), generator=g); q_am = torch.ones_like(q_ids)
    corpus = dense.encode(enc, c_ids, c_am, batch_size=512)
    assert corpus.is_cuda and corpus.dtype == torch.bfloat16 and corpus.shape == (1500, 768)
    index = dense.build_index(corpus)
    assert index.ntotal == 1500
    ckeys = ["c%d" % i for i in range(1500)]; qkeys = ["q%d" % i for i in range(70)]
    result = dense.retrieve(enc, index, q_ids, q_am, qkeys, ckeys, k=10)
    q = dense.encode(enc, q_ids, q_am)
    _, Ir = oracle.knn_canonical(0, q.float().cpu().numpy(), corpus.float().cpu().numpy(), 10)
    assert [r["id"] for r in result] == qkeys
    assert [r["nn"] for r in result] == [[ckeys[j] for j in row] for row in Ir]
    neighbors.write_neighbors(str(tmp_path / "val.json"), result)
    assert neighbors.read_neighbors(str(tmp_path / "val.json"))["q3"] == result[3]["nn"]
    # the bf16 autocast path agrees with the fp32 statement of the ops to bf16 accuracy
    with nn_ref.reference_ops():
        ref = dense.encode(enc, c_ids[:64], c_am[:64], autocast=False, out_dtype=torch.float32)
    assert float((corpus[:64].float() - ref).abs().max()) <= 5e-2 * max(1.0, float(ref.abs().max()))


def test_packed_projection_weights_live_for_one_encode_call_only(monkeypatch):
    """inside dense.encode, ops.linear_multi keeps the packed bf16 query / key / value weight of a layer between the batches;
    nothing survives the call -- a fused optimizer updates parameters WITHOUT moving their autograd version, so no cache may
    outlive the block that vouches for them -- and the embeddings are those of the concatenate-every-call form, bit for bit"""
    from textreact_amd import dense
    cfg = Config(vocab_size=500, num_hidden_layers=2, max_position_embeddings=64)
    torch.manual_seed(0)
    enc = dense.DenseEncoder(cfg).cuda().eval()
    g = torch.Generator().manual_seed(3)
    ids = torch.randint(1, 500, (300, 40), generator=g).cuda(); am = torch.ones_like(ids); am[::3, 17:] = 0
    lens = am.sum(dim=1)
    built = []
    real_cat = torch.cat
    monkeypatch.setattr(torch, "cat", lambda ts, *a, **k: (built.append(len(ts)), real_cat(ts, *a, **k))[1])
    a = dense.encode(enc, ids, am, lengths=lens, batch_tokens=2048)          # several batches
    assert built.count(3) == 2 * 2 and ops._packed_cache is None             # (weights + biases) x 2 layers, once; nothing left behind
    opt = torch.optim.AdamW(enc.parameters(), lr=0.05, fused=True)
    for p_ in enc.parameters():
        p_.grad = torch.ones_like(p_)
    w = enc.encoder.encoder.layer[0].attention.self.query.weight
    v0 = w._version
    opt.step()
    b = dense.encode(enc, ids, am, lengths=lens, batch_tokens=2048)
    assert not torch.equal(a, b)                                              # the update is seen (whatever _version says: w._version - v0 may be 0)
    monkeypatch.setattr(torch, "cat", real_cat)
    plain = lambda x, ws, bs=None: ops.linear(x, torch.cat(list(ws)), None if bs is None else torch.cat(list(bs)))
    monkeypatch.setattr(ops, "linear_multi", plain)
    c = dense.encode(enc, ids, am, lengths=lens, batch_tokens=2048)
    assert torch.equal(b, c)


@pytest.mark.parametrize("train_mode", [False, True])
def test_model_under_bf16_autocast_hip_vs_torch(train_mode):
    """--precision 16-mixed style: Linear layers emit bf16, the residual stream stays fp32 (mixed operand
    types reach add_layernorm), attention runs on the matrix cores.  Loss and gradients against the
    PyTorch statement of the ops under the same autocast (and the same dropout decisions)."""
    z = np.load(G)
    enc, dec = json.loads(str(z["enc_cfg"])), json.loads(str(z["dec_cfg"]))
    t = lambda k_: torch.from_numpy(z[k_]).cuda()
    res = {}
    for backend in ("hip", "torch"):
        with nn_ref.implementation(backend):
            m = TextReactModel(Config(**enc), Config(is_decoder=True, **dec))
            m.load_state_dict(random_state_dict(m, int(z["seed"])))
            m = m.cuda().train(train_mode)
            torch.manual_seed(5)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                logits, _ = m(t("input_ids"), t("attention_mask"), t("decoder_input_ids"), t("decoder_attention_mask"))
                labels = t("decoder_input_ids")[:, 1:]
                loss = torch.nn.functional.cross_entropy(logits[:, :-1].reshape(-1, logits.shape[-1]).float(), labels.reshape(-1), ignore_index=0)
            loss.backward()
            assert bool(torch.isfinite(logits).all()) and bool(torch.isfinite(loss))
            res[backend] = (logits.detach().float(), loss.detach(), {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None})
    assert float((res["hip"][0] - res["torch"][0]).abs().max()) <= 6e-2 * max(1.0, float(res["torch"][0].abs().max()))
    assert abs(float(res["hip"][1] - res["torch"][1])) <= 3e-2 * max(1.0, abs(float(res["torch"][1])))
    for n in res["hip"][2]:
        a, c = res["hip"][2][n], res["torch"][2][n]
        assert bool(torch.isfinite(a).all()), n
        assert float((a - c).abs().max()) <= 8e-2 * max(1e-2, float(c.abs().max())), n


@pytest.mark.parametrize("rows,cols", [(777, 768), (33, 600), (5000, 1024)])
@pytest.mark.parametrize("p", [0.0, 0.1])
def test_add_layernorm_mixed_storage(rows, cols, p):
    """x bf16, residual stream fp32 (autocast): forward and backward against the PyTorch statement"""
    x = _rand(rows, cols, dtype=torch.bfloat16, seed=1); r = _rand(rows, cols, seed=2)
    g, b = _rand(cols, seed=3), _rand(cols, seed=4)
    dy = _rand(rows, cols, seed=5)
    res = []
    for backend in ("hip", "torch"):
        with nn_ref.implementation(backend):
            xs, rs = x.clone().requires_grad_(True), r.clone().requires_grad_(True)
            gs, bs = g.clone().requires_grad_(True), b.clone().requires_grad_(True)
            y = ops.add_layernorm(xs, rs, gs, bs, 1e-12, dropout_p=p, seed=77)
            assert y.dtype == torch.float32
            y.backward(dy)
            assert xs.grad.dtype == torch.bfloat16 and rs.grad.dtype == torch.float32
            res.append((y.detach(), xs.grad.float(), rs.grad, gs.grad, bs.grad))
    tols = (3e-5, 1e-2, 3e-5, 3e-5, 3e-5)     # dx is rounded to bf16 once
    for a, c, tol in zip(*res, tols):
        assert float((a - c).abs().max()) <= tol * max(1.0, float(c.abs().max()))


def test_beam_search_on_the_hip_ops_matches_the_huggingface_golden():
    """the decode loop (KV cache, one-token attention against the cache, cross-attention over the encoder
    states) through the HIP ops reproduces what `generate` returned for the reference model"""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_generate_cpu import check_case, golden_model
    z, dec, m = golden_model()
    m = m.cuda()
    for graph in (False, True):               # the eager loop, and the step replayed from a captured HIP graph
        for i, c in enumerate(json.loads(str(z["cases"]))):
            check_case(z, dec, m, i, c, dev="cuda", tol=1e-3, graph=graph)


@pytest.mark.parametrize("autocast", [False, True])
def test_graph_decode_equals_the_eager_loop(autocast):
    """full-width model, long inputs with padding, beams that finish at different steps: the captured step (static
    buffers, key mask over the whole cache, parents applied at the head of the next replay) returns what the
    eager loop returns"""
    from textreact_amd.predictor.generate import generate
    torch.manual_seed(3)
    m = TextReactModel(Config(vocab_size=300, num_hidden_layers=2), Config(vocab_size=40, num_hidden_layers=2, type_vocab_size=1,
                       layer_norm_eps=1e-5, is_decoder=True))
    m.load_state_dict(random_state_dict(m, 5))      # BERT-style init: logits small enough for bf16 to resolve
    m = m.cuda().eval()
    g = torch.Generator().manual_seed(1)
    ids = torch.randint(1, 300, (3, 70), generator=g).cuda()
    am = torch.ones(3, 70, dtype=torch.long).cuda()
    am[1, 50:] = 0
    for nb in (1, 5):
        res = []
        for graph in (False, True):
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
                res.append(generate(m, ids, am, num_beams=nb, num_return_sequences=nb, max_length=24, length_penalty=0,
                                    bos_token_id=1, eos_token_id=2, pad_token_id=0, graph=graph))
        (s0, c0), (s1, c1) = res
        if not autocast:
            assert torch.equal(s0, s1)
            assert nb == 1 or float((c0 - c1).abs().max()) < 1e-4
        else:       # bf16 rounding may reorder near-ties; the score scale agrees (the eager loop rounds its logits to bf16,
            # ~0.04 per position; the captured step keeps the vocabulary projection in fp32 -- the step-level test below
            # bounds both against the fp32 step)
            assert s0.shape[0] == s1.shape[0]
            assert nb == 1 or float((c0 - c1).abs().max()) < 0.3


def test_template_based_branch_on_the_hip_ops():
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_template_cpu import check, load
    z, m, batch = load()
    m = m.cuda()
    batch = {k: (v.cuda() if torch.is_tensor(v) else ([t.cuda() for t in v] if k == "atom_indices" else v)) for k, v in batch.items()}
    check(z, m, batch, 1e-3)


def test_repeated_graph_decodes_release_their_memory():
    """graphs, cache sets and pools die with the call; the warm-up stream is reused (a fresh stream per call pinned one
    BLAS workspace of 76 MB each)"""
    from textreact_amd.predictor.generate import generate
    torch.manual_seed(0)
    m = TextReactModel(Config(vocab_size=300, num_hidden_layers=1), Config(vocab_size=40, num_hidden_layers=1, type_vocab_size=1,
                       layer_norm_eps=1e-5, is_decoder=True)).cuda().eval()
    g = torch.Generator().manual_seed(0)
    seen = []
    for it in range(8):
        ids = torch.randint(1, 300, (4, 64), generator=g).cuda()
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=it % 2 == 0):
            generate(m, ids, None, num_beams=4, num_return_sequences=4, max_length=16, length_penalty=0, bos_token_id=1,
                     eos_token_id=2, pad_token_id=0, graph=True)
        torch.cuda.synchronize()
        seen.append(torch.cuda.memory_allocated())
    assert max(seen[2:]) - min(seen[2:]) < (8 << 20), seen


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_graph_decode_step_on_bf16_weights_is_as_close_to_fp32_as_eager_autocast(dtype):
    """the captured step under bf16 autocast (prepared bf16 weights, packed key/value cache, biases added inside the
    LayerNorm kernel) against the fp32 step on the same tokens and the same beam re-ordering, position by position:
    its error is bf16 rounding of the logits, the same size as the eager autocast loop's"""
    from textreact_amd.predictor.generate import _DecoderState
    torch.manual_seed(3)
    m = TextReactModel(Config(vocab_size=300, num_hidden_layers=2), Config(vocab_size=40, num_hidden_layers=2, type_vocab_size=1,
                       layer_norm_eps=1e-5, is_decoder=True))
    m.load_state_dict(random_state_dict(m, 5))
    m = m.cuda().eval()
    g = torch.Generator().manual_seed(1)
    ids = torch.randint(1, 300, (3, 70), generator=g).cuda()
    am = torch.ones(3, 70, dtype=torch.long).cuda()
    am[1, 50:] = 0
    nb, T = 5, 12
    with torch.no_grad():
        with torch.autocast("cuda", dtype=dtype):
            eager = _DecoderState(m, ids, am, nb, T, graph=False)
            fast = _DecoderState(m, ids, am, nb, T, graph=True)
        truth = _DecoderState(m, ids, am, nb, T, graph=False)
        assert fast.fast and fast.graph is not None and truth.graph is None
        for t in range(T - 1):
            tok = torch.randint(3, 40, (3 * nb,), generator=g).cuda()
            with torch.autocast("cuda", dtype=dtype):
                le, lf = eager.step(tok, t).clone(), fast.step(tok, t).clone()
            lt = truth.step(tok, t)
            par = torch.stack([torch.randint(0, nb, (nb,), generator=g) + b * nb for b in range(3)]).view(-1).cuda()   # shared parents
            for st in (eager, fast, truth):
                st.reorder(par, t)
            ee, ef = float((le - lt).abs().max()), float((lf - lt).abs().max())
            assert ef < 0.1 and ef < 2 * ee + 0.01, (t, ee, ef)


@pytest.mark.parametrize("n,T,H", [(12, 40, 12), (7, 160, 2), (33, 256, 3), (2, 1, 12)])
def test_decode_attention_follows_the_ancestor_table(n, T, H):
    """trx_attention_decode_gather: keys / values of beam i at position s are read from cache row anc[i][s]; the length
    comes from a device scalar; q may be a strided slice of a packed projection"""
    g = torch.Generator().manual_seed(n + T)
    kv = _rand(n, T, 2, H, 64, dtype=torch.bfloat16, seed=1)
    anc = torch.randint(0, n, (n, T), generator=g, dtype=torch.int32).cuda()
    qkv = _rand(n, 1, 3, H, 64, dtype=torch.bfloat16, seed=2)
    for t in sorted({0, T // 3, T - 1}):
        t_dev = torch.tensor([t], dtype=torch.int64, device="cuda")
        for q in (qkv[:, :, 0], qkv[:, :, 0].contiguous()):
            out = ops.attention_decode_gather(q, kv, anc, t_dev)
            idx = anc[:, :t + 1].long()                                            # [n, t + 1]
            pos = torch.arange(t + 1, device="cuda")[None].expand_as(idx)
            k = kv[idx, pos, 0]                                                    # [n, t + 1, H, 64]: what beam i's history holds
            v = kv[idx, pos, 1]
            ref = nn_ref.attention(q.float(), k.float(), v.float())
            assert out.shape == (n, 1, H * 64)
            assert float((out.float() - ref).abs().max()) <= 2e-2


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 2e-2)])
def test_attention_reads_a_key_value_cache_in_place(dtype, tol):
    B, H, Lmax = 5, 12, 40
    kc, vc = _rand(B, Lmax, H, 64, dtype=dtype, seed=2), _rand(B, Lmax, H, 64, dtype=dtype, seed=3)
    for Lk in (1, 7, 33, 40):
        q = _rand(B, 1, H, 64, dtype=dtype, seed=10 + Lk)
        k, v = kc[:, :Lk], vc[:, :Lk]
        assert Lk == Lmax or not k.is_contiguous()
        with torch.no_grad():
            out = ops.attention(q, k, v)                                   # strided: the kv-cache entry point
            ref = nn_ref.attention(q.float(), k.float(), v.float())
            same = ops.attention(q, k.contiguous(), v.contiguous())        # dense entry point
        assert float((out.float() - ref).abs().max()) <= tol
        assert torch.equal(out, same)


@pytest.mark.parametrize("p", [0.0, 0.1])
def test_add_layernorm_dual_output_and_two_gradient_paths(p):
    """autocast shape of the op: x bf16, stream fp32, outputs (y fp32, y bf16 copy); gradients arriving through
    either output, or both, are summed in the kernel"""
    rows, cols = 700, 768
    x = _rand(rows, cols, dtype=torch.bfloat16, seed=1); r = _rand(rows, cols, seed=2)
    g, b = _rand(cols, seed=3), _rand(cols, seed=4)
    d32, d16 = _rand(rows, cols, seed=5), _rand(rows, cols, dtype=torch.bfloat16, seed=6)
    for use32, use16 in ((True, True), (True, False), (False, True)):
        res = []
        for backend in ("hip", "torch"):
            with nn_ref.implementation(backend):
                xs, rs = x.clone().requires_grad_(True), r.clone().requires_grad_(True)
                gs, bs = g.clone().requires_grad_(True), b.clone().requires_grad_(True)
                y, ylow = ops.add_layernorm(xs, rs, gs, bs, 1e-12, dropout_p=p, seed=5, dual=True)
                if backend == "hip":
                    assert y.dtype == torch.float32 and ylow.dtype == torch.bfloat16
                    assert torch.equal(ylow, y.to(torch.bfloat16))
                else:
                    ylow = y.to(torch.bfloat16)            # the statement: the low copy is a cast of y
                loss = (y * d32).sum() * (1.0 if use32 else 0.0) + (ylow.float() * d16.float()).sum() * (1.0 if use16 else 0.0)
                if backend == "hip":                       # really leave one output out of the graph
                    loss = ((y * d32).sum() if use32 else 0.0) + ((ylow.float() * d16.float()).sum() if use16 else 0.0)
                loss.backward()
                res.append((xs.grad.float(), rs.grad, gs.grad, bs.grad))
        for a, c, tol in zip(*res, (1e-2, 3e-5, 3e-5, 3e-5)):
            assert float((a - c).abs().max()) <= tol * max(1.0, float(c.abs().max())), (use32, use16)


@pytest.mark.parametrize("B,H,L,Lk,mask,causal,p", [
    (2, 12, 512, 512, "key", False, 0.0), (2, 3, 70, 70, "key", False, 0.1), (2, 4, 160, 160, "none", True, 0.1),
    (1, 2, 129, 129, "full", False, 0.0), (2, 12, 160, 512, "key", False, 0.1), (1, 2, 33, 300, "none", False, 0.0),
])
def test_packed_projection_operands_equal_separate_tensors(B, H, L, Lk, mask, causal, p):
    """q, k, v read as slices of one packed GEMM output (row strides), gradients written into the packed shape:
    bit-identical to the kernels run on separate contiguous tensors"""
    bf = torch.bfloat16
    neg = torch.finfo(torch.float32).min
    m = None
    if mask == "key":
        keep = torch.ones(B, Lk, device="cuda"); keep[0, Lk // 2 + 1:] = 0
        m = (1 - keep) * neg
    elif mask == "full":
        keep = (torch.rand(B, L, Lk, device="cuda") > 0.3).float(); keep[:, :, 0] = 1
        m = (1 - keep) * neg
    dout = _rand(B, L, H * 64, dtype=bf, seed=4)
    if L == Lk:   # self-attention: [B, L, 3, H, 64]
        qkv = _rand(B, L, 3, H, 64, dtype=bf, seed=1)
        a = qkv.clone().requires_grad_(True)
        out = ops.attention_qkv(a, mask=m, causal=causal, dropout_p=p, seed=9)
        out.backward(dout)
        q, k, v = (t.contiguous().clone().requires_grad_(True) for t in qkv.unbind(dim=2))
        ref = ops.attention(q, k, v, mask=m, causal=causal, dropout_p=p, seed=9)
        ref.backward(dout)
        assert torch.equal(out, ref)
        for i, g in enumerate((q.grad, k.grad, v.grad)):
            assert torch.equal(a.grad[:, :, i], g), i
    else:         # cross-attention: q dense, [B, Lk, 2, H, 64] packed keys / values
        q0, kv0 = _rand(B, L, H, 64, dtype=bf, seed=1), _rand(B, Lk, 2, H, 64, dtype=bf, seed=2)
        q1, kv1 = q0.clone().requires_grad_(True), kv0.clone().requires_grad_(True)
        out = ops.attention_q_kv(q1, kv1, mask=m, causal=causal, dropout_p=p, seed=9)
        out.backward(dout)
        q, k, v = q0.clone().requires_grad_(True), *(t.contiguous().clone().requires_grad_(True) for t in kv0.unbind(dim=2))
        ref = ops.attention(q, k, v, mask=m, causal=causal, dropout_p=p, seed=9)
        ref.backward(dout)
        assert torch.equal(out, ref) and torch.equal(q1.grad, q.grad)
        assert torch.equal(kv1.grad[:, :, 0], k.grad) and torch.equal(kv1.grad[:, :, 1], v.grad)


@pytest.mark.parametrize("p", [0.0, 0.1])
def test_add_layernorm_with_the_producing_linears_bias(p):
    """x + bias formed inside the mixed-storage kernel, bias gradient = column sums of dx from its backward"""
    rows, cols = 1500, 768
    x = _rand(rows, cols, dtype=torch.bfloat16, seed=1); r = _rand(rows, cols, seed=2)
    g, b, xb = _rand(cols, seed=3), _rand(cols, seed=4), _rand(cols, seed=7)
    dy = _rand(rows, cols, seed=5)
    res = []
    for backend in ("hip", "torch"):
        with nn_ref.implementation(backend):
            xs, rs = x.clone().requires_grad_(True), r.clone().requires_grad_(True)
            gs, bs, xbs = (t.clone().requires_grad_(True) for t in (g, b, xb))
            if backend == "hip":
                y, ylow = ops.add_layernorm(xs, rs, gs, bs, 1e-12, dropout_p=p, seed=3, dual=True, bias=xbs)
                assert ylow.dtype == torch.bfloat16
            else:   # the statement: bias added in fp32, then the plain op
                y = nn_ref.add_layernorm(xs.float() + xbs, rs, gs, bs, 1e-12, dropout_p=p, seed=3)
            y.backward(dy)
            res.append((y.detach(), xs.grad.float(), rs.grad, gs.grad, bs.grad, xbs.grad))
    for a, c, tol in zip(*res, (3e-5, 1e-2, 3e-5, 3e-5, 3e-5, 2e-3)):   # dx (and so its column sums) is rounded to bf16
        assert float((a - c).abs().max()) <= tol * max(1.0, float(c.abs().max()))


def test_ddp_wrapper_single_rank_rccl_on_the_hip_ops():
    """the predictor under DistributedDataParallel with the RCCL ("nccl") backend, one rank: gradients must equal
    the unwrapped module's (bucketing, hooks and our autograd Functions get along), bf16 autocast"""
    import socket
    import torch.distributed as dist
    from textreact_amd.predictor import train
    z = np.load(G)
    enc, dec = json.loads(str(z["enc_cfg"])), json.loads(str(z["dec_cfg"]))
    for c in (enc, dec):
        c["hidden_dropout_prob"] = c["attention_probs_dropout_prob"] = 0.0
    t = lambda k_: torch.from_numpy(z[k_]).cuda()
    batch = {k_: t(k_) for k_ in ("input_ids", "attention_mask", "decoder_input_ids", "decoder_attention_mask")}
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        grads = []
        for wrap in (False, True):
            p = train.Predictor(Config(**enc), Config(is_decoder=True, **dec), mlm=False).cuda().train()
            p.model.load_state_dict(random_state_dict(p.model, int(z["seed"])))

            class Step(torch.nn.Module):
                def __init__(self, pred):
                    super().__init__(); self.pred = pred

                def forward(self, **b):
                    return self.pred.training_step(b)[0]
            mod = Step(p)
            if wrap:
                mod = torch.nn.parallel.DistributedDataParallel(mod, device_ids=[0], find_unused_parameters=True)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                loss = mod(**batch)
            loss.backward()
            grads.append({n: q.grad.detach().clone() for n, q in p.named_parameters() if q.grad is not None})
        assert set(grads[0]) == set(grads[1])
        for n in grads[0]:
            assert torch.equal(grads[0][n], grads[1][n]), n
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("M,N,K", [(64, 256, 256), (448, 512, 256), (5120, 768, 768), (16384, 3072, 768), (16384, 768, 3072), (16384, 2304, 768),
                                   (65, 256, 256), (127, 256, 512), (1000, 768, 768), (13984, 768, 3072), (4256, 2304, 768), (1, 256, 256)])
def test_split_contraction_weight_gradient_gemm(M, N, K):
    """trx_gemm_tn_bf16: dW = dY^T X with the token rows split over workgroups, against fp32 matmul; token counts that are
    not multiples of the 64-row step included (the rows past the end of the last step are read as zeros)"""
    dy, x = _rand(M, N, dtype=torch.bfloat16, seed=1), _rand(M, K, dtype=torch.bfloat16, seed=2)
    c = ops.gemm_tn(dy, x)
    ref = dy.float().t() @ x.float()
    assert c.dtype == torch.bfloat16 and c.shape == (N, K)
    assert float((c.float() - ref).abs().max()) <= 6e-3 * max(1.0, float(ref.abs().max()))     # one bf16 rounding of the result
    assert torch.equal(c, ops.gemm_tn(dy, x))                                                   # fixed summation order
    c2, cs = ops.gemm_tn(dy, x, colsum=True)                                                    # db from the same pass
    assert torch.equal(c2, c)
    assert float((cs.float() - dy.float().sum(0)).abs().max()) <= 6e-3 * max(1.0, float(dy.float().sum(0).abs().max()))
    big = _rand(M, N + 512, dtype=torch.bfloat16, seed=3)                                       # a slice of a packed gradient
    sl = big[:, 256:256 + N]
    assert torch.equal(ops.gemm_tn(sl, x), ops.gemm_tn(sl.contiguous(), x))


def test_grouped_weight_gradient_gemm():
    """trx_gemm_tn_grouped_*: many dW = dY^T X (+ db) in one persistent launch, nothing split, straight into the gradient:
    the step's shapes, ragged token counts, a slice of a packed gradient; against fp32 matmul and against the per-call form"""
    shapes = [(16384, 2304, 768), (1000, 768, 768), (5120, 3072, 768), (437, 256, 512), (64, 256, 256), (1, 256, 256), (13984, 768, 3072),
              (4256, 1536, 768), (130, 512, 256), (5120, 600, 768), (333, 8, 256), (200, 264, 256)]      # the last three: N % 256 != 0 (a vocabulary projection)
    probs, refs = [], []
    for i, (M, N, K) in enumerate(shapes):
        dy, x = _rand(M, N + (256 if i % 3 == 0 else 0), dtype=torch.bfloat16, seed=10 + i), _rand(M, K, dtype=torch.bfloat16, seed=50 + i)
        dy = dy[:, :N] if i % 3 == 0 else dy
        dw = torch.full((N, K), float("nan"), device="cuda"); db = torch.full((N,), float("nan"), device="cuda") if i % 2 == 0 else None
        probs.append((dy, x, dw, db)); refs.append((dy.float().t() @ x.float(), dy.float().sum(0)))
    ops.gemm_tn_grouped(probs)
    first = [(p[2].clone(), None if p[3] is None else p[3].clone()) for p in probs]
    for (dy, x, dw, db), (rw, rb), (M, N, K) in zip(probs, refs, shapes):
        assert float((dw - rw).abs().max()) <= 2e-5 * max(1.0, float(rw.abs().max())) * max(1.0, M / 512), (M, N, K)      # fp32 sums, two orders
        if db is not None:
            assert float((db - rb).abs().max()) <= 2e-5 * max(1.0, float(rb.abs().max())) * max(1.0, M / 512), (M, N, K)
        if M >= 64 and N % 256 == 0:
            c, cs = ops.gemm_tn(dy, x, colsum=True, out_dtype=torch.float32)                # the split form: other order, same sums
            assert float((dw - c).abs().max()) <= 1e-5 * max(1.0, float(rw.abs().max())) * max(1.0, M / 512)
    ops.gemm_tn_grouped(probs)                                                               # no split, no atomics: reproducible
    for (dy, x, dw, db), (w0, b0) in zip(probs, first):
        assert torch.equal(dw, w0) and (db is None or torch.equal(db, b0))


def test_deferred_weight_gradients_equal_the_per_layer_calls():
    """ops.backward (deferred_wgrad): the gradients a backward pass leaves in .grad with the one grouped launch at its end =
    those of the per-layer calls, also when a .grad is already there (gradient accumulation) and for packed projections"""
    x = _rand(3, 171, 768, dtype=torch.bfloat16, seed=1)
    ws = [(_rand(768, 768, seed=2 + i) * 0.05) for i in range(3)]; bs = [_rand(768, seed=7 + i) for i in range(3)]
    w2, dy = _rand(256, 2304, seed=11) * 0.05, _rand(3, 171, 256, dtype=torch.bfloat16, seed=12)
    res = []
    hooked = []
    for mode in ("deferred", "percall", "deferred, flushed in groups", "deferred, a hook on one weight"):
        xs = x.clone().requires_grad_(True)
        pw = [w.clone().requires_grad_(True) for w in ws]; pb = [b.clone().requires_grad_(True) for b in bs]
        pw2 = w2.clone().requires_grad_(True)
        if "hook" in mode:                     # a parameter with a hook keeps the per-call path: autograd hands it the gradient
            pw2.register_hook(lambda g: hooked.append(tuple(g.shape)))
        budget = ops._DEFERRED_BUDGET
        if "groups" in mode:                   # TRX_NN_WGRAD_BUDGET_MB: past 1 MB of held operands the pending problems run
            ops._DEFERRED_BUDGET = 1 << 20
        try:
            for rep in range(2):               # the second pass adds to the first's .grad
                with torch.autocast("cuda", dtype=torch.bfloat16):
                    h = ops.linear_multi(xs, pw, pb)
                    y = ops.linear_multi(h, [pw2])
                if mode != "percall":
                    ops.backward((y.float() * dy.float()).sum())
                else:
                    (y.float() * dy.float()).sum().backward()
        finally:
            ops._DEFERRED_BUDGET = budget
        res.append([xs.grad] + [p.grad for p in pw + pb + [pw2]])
    assert hooked == [(256, 2304)] * 2
    for other in res[1:]:
        for a, b in zip(res[0], other):
            assert a is not None and a.shape == b.shape and a.dtype == b.dtype
            assert float((a.float() - b.float()).abs().max()) <= 2e-5 * max(1.0, float(b.float().abs().max()))


def test_deferred_layernorm_column_sums_equal_the_per_call_second_stage():
    """ops.backward: the add+LayerNorm backward calls of a pass run their first stage only and ONE launch at the end
    (trx_add_layernorm_bwd_reduce_many) makes every call's dgamma / dbeta / fused-bias gradient -- bit for bit the per-call second
    stage's sums, also when a .grad is already there (gradient accumulation), with and without the fused bias, and a parameter
    with a hook keeps the per-call path"""
    rows, cols = 3 * 171, 768
    x = _rand(rows, cols, dtype=torch.bfloat16, seed=1); res = _rand(rows, cols, seed=2)
    dy = _rand(rows, cols, seed=3)
    seen = []
    res_grads = []
    for mode in ("deferred", "percall", "deferred, hook on gamma"):
        xs, rs = x.clone().requires_grad_(True), res.clone().requires_grad_(True)
        ps = [[(_rand(cols, seed=10 + 3 * i + j) * 0.1 + (1.0 if j == 0 else 0.0)).requires_grad_(True) for j in range(3)] for i in range(3)]
        if "hook" in mode:
            ps[0][0].register_hook(lambda g: seen.append(tuple(g.shape)))
        for rep in range(2):
            h = rs
            for i, (g, b, bias) in enumerate(ps):
                h = ops.add_layernorm(xs, h, g, b, 1e-12, dropout_p=0.1, seed=77 + i, bias=bias if i != 1 else None)
            loss = (h * dy).sum()
            if mode != "percall":
                ops.backward(loss)
            else:
                loss.backward()
        res_grads.append([xs.grad, rs.grad] + [t.grad for trio in ps for t in trio if t.grad is not None])
    assert seen == [(cols,)] * 2
    for other in res_grads[1:]:
        assert len(other) == len(res_grads[0]) == 2 + 3 + 2 + 3
        for a, b in zip(res_grads[0], other):
            assert torch.equal(a, b)


@pytest.mark.parametrize("B,L", [(4, 128), (4, 131), (3, 437), (1, 63)])
def test_linear_with_our_weight_gradient_equals_autograd(B, L):
    """token counts that are not multiples of 64 included (batches padded to their longest sequence); fewer than 64
    tokens: the library"""
    x = _rand(B, L, 768, dtype=torch.bfloat16, seed=1); w = _rand(2304, 768, seed=2) * 0.05; b = _rand(2304, seed=3)
    dy = _rand(B, L, 2304, dtype=torch.bfloat16, seed=4)
    res = []
    for mine in (True, False):
        xs, ws, bs = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = ops.linear(xs, ws, bs) if mine else torch.nn.functional.linear(xs, ws, bs)
        y.backward(dy)
        res.append((y.detach().float(), xs.grad.float(), ws.grad, bs.grad))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    for a, c in zip(res[0][2:], res[1][2:]):
        assert float((a - c).abs().max()) <= 1e-2 * max(1.0, float(c.abs().max()))


def test_short_training_run_learns_and_tracks_the_pytorch_statement():
    """60 optimisation steps on a toy copy task (bf16 autocast, dropout 0.1, AdamW, every fused path of the
    training step): the loss must fall, and fall like it does with the PyTorch statement of the ops"""
    from textreact_amd.predictor import train
    enc = dict(vocab_size=40, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
               max_position_embeddings=64, type_vocab_size=2, layer_norm_eps=1e-12)
    dec = dict(vocab_size=40, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
               max_position_embeddings=64, type_vocab_size=1, layer_norm_eps=1e-5)
    g = torch.Generator().manual_seed(0)
    src = torch.randint(4, 40, (64, 12), generator=g)
    batch = {"input_ids": src.cuda(), "attention_mask": torch.ones_like(src).cuda(),
             "decoder_input_ids": torch.cat([torch.full((64, 1), 2), src[:, :8], torch.full((64, 1), 3)], 1).cuda(),
             "decoder_attention_mask": torch.ones(64, 10, dtype=torch.long).cuda()}
    curves = {}
    for backend in ("hip", "torch"):
        with nn_ref.implementation(backend):
            torch.manual_seed(1)
            p = train.Predictor(Config(**enc), Config(is_decoder=True, **dec), mlm=False).cuda().train()
            opt, _ = train.configure_optimizer(p, 2e-3, 0.0, 1000, 0.0, scheduler="constant")
            losses = []
            for _ in range(60):
                with torch.autocast("cuda", dtype=torch.bfloat16):
                    loss, _ = p.training_step(batch)
                loss.backward(); opt.step(); opt.zero_grad(set_to_none=True)
                losses.append(float(loss))
            assert all(np.isfinite(losses)), backend
            curves[backend] = losses
    for backend, l in curves.items():
        assert np.mean(l[-5:]) < 0.5 * np.mean(l[:3]), (backend, l[:3], l[-5:])
    assert abs(np.mean(curves["hip"][-5:]) - np.mean(curves["torch"][-5:])) < 0.25 * np.mean(curves["torch"][:3])


def test_model_under_fp16_autocast_runs_on_the_bf16_kernels():
    """the reference's --precision 16-mixed is fp16 autocast: the HIP ops take fp16 activations (on their bf16 kernels)
    and the model agrees with the PyTorch statement under the same autocast to bf16 accuracy"""
    z = np.load(G)
    enc, dec = json.loads(str(z["enc_cfg"])), json.loads(str(z["dec_cfg"]))
    t = lambda k_: torch.from_numpy(z[k_]).cuda()
    out = {}
    for backend in ("hip", "torch"):
        with nn_ref.implementation(backend):
            m = TextReactModel(Config(**enc), Config(is_decoder=True, **dec))
            m.load_state_dict(random_state_dict(m, int(z["seed"])))
            m = m.cuda().eval()
            with torch.autocast("cuda", dtype=torch.float16):
                logits, _ = m(t("input_ids"), t("attention_mask"), t("decoder_input_ids"), t("decoder_attention_mask"))
                loss = torch.nn.functional.cross_entropy(logits[:, :-1].reshape(-1, logits.shape[-1]).float(),
                                                         t("decoder_input_ids")[:, 1:].reshape(-1), ignore_index=0)
            loss.backward()
            assert bool(torch.isfinite(logits).all())
            out[backend] = (logits.detach().float(), {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None})
    assert float((out["hip"][0] - out["torch"][0]).abs().max()) <= 6e-2 * max(1.0, float(out["torch"][0].abs().max()))
    for n in out["hip"][1]:
        a, c = out["hip"][1][n], out["torch"][1][n]
        assert bool(torch.isfinite(a).all()), n
        assert float((a - c).abs().max()) <= 8e-2 * max(1e-2, float(c.abs().max())), n


def test_fp16_autocast_training_step_through_the_weight_gradient_gemm():
    """--precision 16-mixed (fp16 autocast + GradScaler) at layer widths that route the Linear layers' weight gradients
    to trx_gemm_tn_bf16 (multiples of 256): forward, backward and an optimizer step run, the gradients are finite and
    agree with the PyTorch statement under the same autocast to bf16 / fp16 accuracy"""
    from textreact_amd.predictor import train
    torch.manual_seed(0)
    g = torch.Generator().manual_seed(0)
    B, L, T = 4, 64, 32
    batch = {"input_ids": torch.randint(1, 300, (B, L), generator=g).cuda(), "attention_mask": torch.ones(B, L, dtype=torch.long).cuda(),
             "decoder_input_ids": torch.randint(3, 50, (B, T), generator=g).cuda(),
             "decoder_attention_mask": torch.ones(B, T, dtype=torch.long).cuda()}
    grads = {}
    for backend in ("hip", "torch"):
        with nn_ref.implementation(backend):
            enc = Config(vocab_size=300, hidden_size=256, num_hidden_layers=1, num_attention_heads=4, intermediate_size=512,
                         hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
            dec = Config(vocab_size=50, hidden_size=256, num_hidden_layers=1, num_attention_heads=4, intermediate_size=512, type_vocab_size=1,
                         layer_norm_eps=1e-5, is_decoder=True, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
            p = train.Predictor(enc, dec, mlm=False)
            p.model.load_state_dict(random_state_dict(p.model, 4))
            p = p.cuda().train()
            opt, _ = train.configure_optimizer(p, 1e-4, 0.01, 100, 0.02)
            scaler = torch.amp.GradScaler("cuda", init_scale=1024.0)
            with torch.autocast("cuda", dtype=torch.float16):
                loss, _ = p.training_step(batch)
            scaler.scale(loss).backward()
            grads[backend] = {n: q.grad.detach().float() / 1024.0 for n, q in p.named_parameters() if q.grad is not None}
            scaler.step(opt); scaler.update()
            assert bool(torch.isfinite(loss)) and all(bool(torch.isfinite(q).all()) for q in p.parameters())
    assert set(grads["hip"]) == set(grads["torch"])
    for n, a in grads["hip"].items():
        c = grads["torch"][n]
        assert bool(torch.isfinite(a).all()), n
        assert float((a - c).abs().max()) <= 8e-2 * max(1e-3, float(c.abs().max())), n


def test_packed_projection_without_concatenation_equals_autograd():
    """ops.linear_multi: query / key / value as ONE product whose weight gradient comes back as row blocks, with and
    without a WeightShadows registry installed"""
    x = _rand(3, 131, 768, dtype=torch.bfloat16, seed=1)
    ws = [(_rand(768, 768, seed=10 + i) * 0.05) for i in range(3)]
    bs = [_rand(768, seed=20 + i) for i in range(3)]
    dy = _rand(3, 131, 2304, dtype=torch.bfloat16, seed=4)
    res = []
    for mode in ("shadows", "casts", "autograd"):
        xs = x.clone().requires_grad_(True)
        w_ = [w.clone().requires_grad_(True) for w in ws]
        b_ = [b.clone().requires_grad_(True) for b in bs]
        with torch.autocast("cuda", dtype=torch.bfloat16):
            if mode == "autograd":
                y = torch.nn.functional.linear(xs, torch.cat(w_), torch.cat(b_))
            else:
                with ops.use_shadows(ops.WeightShadows() if mode == "shadows" else None):
                    y = ops.linear_multi(xs, tuple(w_), tuple(b_))
        y.backward(dy)
        res.append((y.detach().float(), xs.grad.float(), torch.cat([w.grad for w in w_]), torch.cat([b.grad for b in b_])))
    for r in res[:2]:
        assert torch.equal(r[0], res[2][0]) and torch.equal(r[1], res[2][1])
        for a, c in zip(r[2:], res[2][2:]):
            assert float((a - c).abs().max()) <= 1e-2 * max(1.0, float(c.abs().max()))
    assert torch.equal(res[0][2], res[1][2]) and torch.equal(res[0][3], res[1][3])


def test_weight_shadows_follow_the_optimizer_and_allow_two_forwards_before_one_backward():
    from textreact_amd.predictor import train
    enc = Config(vocab_size=300, hidden_size=256, num_hidden_layers=1, num_attention_heads=4, intermediate_size=512,
                 hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    dec = Config(vocab_size=50, hidden_size=256, num_hidden_layers=1, num_attention_heads=4, intermediate_size=512, type_vocab_size=1,
                 layer_norm_eps=1e-5, is_decoder=True, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    g = torch.Generator().manual_seed(0)
    mk = lambda B: {"input_ids": torch.randint(1, 300, (B, 70), generator=g).cuda(), "attention_mask": torch.ones(B, 70, dtype=torch.long).cuda(),
                    "decoder_input_ids": torch.randint(3, 50, (B, 33), generator=g).cuda(),
                    "decoder_attention_mask": torch.ones(B, 33, dtype=torch.long).cuda()}
    b1, b2 = mk(4), mk(3)
    out = {}
    for backend in ("hip", "torch"):
        with nn_ref.implementation(backend):
            p = train.Predictor(enc, dec, mlm=False)
            p.model.load_state_dict(random_state_dict(p.model, 4))
            p = p.cuda().train()
            opt = torch.optim.AdamW(p.parameters(), lr=2e-3, fused=True)      # fused: updates parameters without moving their autograd version
            losses = []
            for _ in range(3):
                with torch.autocast("cuda", dtype=torch.bfloat16):
                    l1, _ = p.training_step(b1)
                    l2, _ = p.training_step(b2)          # a second forward before the backward: the shadows must not be rewritten
                (l1 + l2).backward()
                opt.step(); opt.zero_grad(set_to_none=True)
                losses.append(float(l1) + float(l2))
            out[backend] = losses
            if backend == "hip":
                reg = p.model.__dict__["_weight_shadows"]
                assert len(reg.groups) >= 8
                for ws_, bs_, w16, b16 in reg.groups.values():           # after the last step the NEXT forward refreshes them
                    pass
                with torch.autocast("cuda", dtype=torch.bfloat16):
                    p.training_step(b1)
                for ws_, bs_, w16, b16 in reg.groups.values():
                    assert torch.equal(w16, torch.cat([w.detach() for w in ws_]).to(torch.bfloat16))
    assert out["hip"][-1] < out["hip"][0] - 0.05                      # the optimizer moves the weights: stale shadows would not learn
    for a, c in zip(out["hip"], out["torch"]):
        assert abs(a - c) <= 3e-2 * max(1.0, abs(c)), (out["hip"], out["torch"])


@pytest.mark.gpu
def test_a_backward_across_a_parameter_update_is_refused():
    """the bf16 weight copies `linear` shares between forwards (ops.WeightShadows) carry a generation: two forwards before
    their backwards are fine, a backward whose forward predates train.mark_parameters_updated raises.  (Widths of 256:
    `linear` routes only such layers through the shared copies.)"""
    from textreact_amd.predictor import train
    enc = dict(vocab_size=40, hidden_size=256, num_hidden_layers=1, num_attention_heads=4, intermediate_size=512,
               max_position_embeddings=64, type_vocab_size=2, layer_norm_eps=1e-12)
    dec = dict(vocab_size=40, hidden_size=256, num_hidden_layers=1, num_attention_heads=4, intermediate_size=512,
               max_position_embeddings=64, type_vocab_size=1, layer_norm_eps=1e-5)
    g = torch.Generator().manual_seed(0)
    src = torch.randint(4, 40, (64, 12), generator=g)
    batch = {"input_ids": src.cuda(), "attention_mask": torch.ones_like(src).cuda(),
             "decoder_input_ids": torch.cat([torch.full((64, 1), 2), src[:, :8], torch.full((64, 1), 3)], 1).cuda(),
             "decoder_attention_mask": torch.ones(64, 10, dtype=torch.long).cuda()}
    torch.manual_seed(1)
    p = train.Predictor(Config(**enc), Config(is_decoder=True, **dec), mlm=False).cuda().train()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        l1, _ = p.training_step(batch)
        l2, _ = p.training_step(batch)
    (l1 + l2).backward()                                   # two forwards, one generation: legal
    assert all(torch.isfinite(q.grad).all() for q in p.parameters() if q.grad is not None)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        l3, _ = p.training_step(batch)
    train.mark_parameters_updated(p)
    with pytest.raises(RuntimeError, match="between this forward and its backward"):
        l3.backward()


def _graph_case(seed=0):
    from textreact_amd.predictor import train
    enc = dict(vocab_size=300, hidden_size=256, num_hidden_layers=2, num_attention_heads=4, intermediate_size=512, max_position_embeddings=64)
    dec = dict(vocab_size=40, hidden_size=256, num_hidden_layers=2, num_attention_heads=4, intermediate_size=512, max_position_embeddings=64,
               type_vocab_size=1, layer_norm_eps=1e-5)
    torch.manual_seed(seed)
    p = train.Predictor(Config(**enc), Config(is_decoder=True, **dec), mlm=True, mlm_lambda=0.1).cuda().train()
    g = torch.Generator().manual_seed(1)
    batches = []
    for _ in range(9):
        ids = torch.randint(1, 300, (6, 40), generator=g).cuda(); dids = torch.randint(3, 40, (6, 9), generator=g).cuda()
        batches.append(({"input_ids": ids, "attention_mask": torch.ones_like(ids), "decoder_input_ids": dids, "decoder_attention_mask": torch.ones_like(dids)},
                        {"mlm_labels": torch.randint(0, 300, (6, 5), generator=g).cuda()}))
    return train, p, batches


def test_device_seed_dropout_hip_equals_the_reference_statement():
    """ops.set_seed_device: a launch's seed only numbers the site, the kernels mix it with a device counter -- the materialised
    decisions (trx_dropout_keep_mask) follow the same rule, so the PyTorch statement still drops the same elements"""
    seed_t = torch.tensor([123456789], dtype=torch.int64, device="cuda")
    ops.set_seed_device(seed_t)
    try:
        q, k, v = (_rand(2, 70, 3, 64, dtype=torch.bfloat16, seed=s_) for s_ in (1, 2, 3))
        x, r = _rand(50, 768, seed=4), _rand(50, 768, seed=5)
        gm, bt = _rand(768, seed=6) * 0.1 + 1, _rand(768, seed=7) * 0.1
        outs = {}
        for step_value in (5, 6):
            seed_t.fill_(step_value)
            for backend in ("hip", "torch"):
                with nn_ref.implementation(backend):
                    a = ops.attention(q if backend == "hip" else q.float(), k if backend == "hip" else k.float(),
                                      v if backend == "hip" else v.float(), dropout_p=0.3, seed=7)
                    y = ops.add_layernorm(x, r, gm, bt, 1e-12, dropout_p=0.3, seed=8)
                outs[(step_value, backend)] = (a.float(), y)
            assert float((outs[(step_value, "hip")][0] - outs[(step_value, "torch")][0]).abs().max()) <= 3e-2
            assert float((outs[(step_value, "hip")][1] - outs[(step_value, "torch")][1]).abs().max()) <= 1e-4
        assert float((outs[(5, "hip")][1] - outs[(6, "hip")][1]).abs().max()) > 0.1       # the counter moved: other decisions
    finally:
        ops.set_seed_device(None)


def test_graphed_step_equals_the_eager_steps():
    """train.GraphedStep: 3 eager steps, a capture, 5 replays -- against the same 9 steps all run eagerly with the same
    device-side seeds: the losses agree step by step and the parameters end up the same.  Runs in a child process: the
    runtime switch GraphedStep needs (train.GRAPH_RUNTIME_ENV) is read when the HIP runtime starts."""
    import subprocess, sys
    from textreact_amd.predictor import train
    env = dict(os.environ); env[train.GRAPH_RUNTIME_ENV[0]] = train.GRAPH_RUNTIME_ENV[1]
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "graphed_step_case.py")],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "graphed step: ok" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def test_graphed_step_refuses_a_runtime_that_was_started_without_its_switch():
    from textreact_amd.predictor import train
    if os.environ.get(train.GRAPH_RUNTIME_ENV[0]) == train.GRAPH_RUNTIME_ENV[1]:
        pytest.skip("this process was started with the switch")
    _, p, _ = _graph_case()
    opt, _ = train.configure_optimizer(p, 1e-3, 0.01, 100, 0.1, capturable=True)
    with pytest.raises(ops.TrxNNError, match=train.GRAPH_RUNTIME_ENV[0]):
        train.GraphedStep(p, opt)


def test_deferred_layernorm_sums_survive_a_refused_grouped_launch(monkeypatch):
    """ops._flush_deferred_ln (round 6, advisor): when the one-launch-per-pass reduction is refused, the calls are reduced one by one
    by the same kernel -- same gradients bit for bit -- instead of losing every LayerNorm's dgamma / dbeta of the step"""
    def grads(refuse):
        torch.manual_seed(0)
        x = [_rand(500, 768, seed=i) for i in range(3)]      # (the deferred second stage belongs to the trainer's mixed storage: bf16 x, fp32 stream)
        g = [(_rand(768, seed=10 + i) * 0.1 + 1).requires_grad_(True) for i in range(3)]
        b = [(_rand(768, seed=20 + i) * 0.1).requires_grad_(True) for i in range(3)]
        real = ops.lib().trx_add_layernorm_bwd_reduce_many
        calls = []
        if refuse:
            class Lib:
                def __getattr__(self, name):
                    if name == "trx_add_layernorm_bwd_reduce_many":
                        def f(arr, n, cols, st):
                            calls.append(n)
                            return -1 if n > 1 else real(arr, n, cols, st)
                        return f
                    return getattr(ops._LIB_REAL, name)
            ops._LIB_REAL = ops.lib()
            monkeypatch.setattr(ops, "lib", lambda: Lib())
        with ops.deferred_wgrad():
            loss = sum(ops.add_layernorm(x[i].bfloat16().requires_grad_(True), x[(i + 1) % 3], g[i], b[i], 1e-5).square().sum() for i in range(3))
            loss.backward()
        if refuse:
            monkeypatch.undo()
            assert calls[0] == 3 and calls[1:] == [1, 1, 1], calls
        return [t.grad.clone() for t in g + b]
    a, c = grads(False), grads(True)
    for u, v in zip(a, c):
        assert torch.equal(u, v)
