"""torch.ops.trx.*: the predictor kernels as torch.library ops.  CPU: they are registered with the
documented schemas, trace under fake tensors, and refuse CPU tensors (no fallback).  GPU: same numbers and gradients as
the autograd Functions the model calls."""
import pytest
import torch

from textreact_amd.predictor import ops


def test_registered_with_schemas():
    s = str(torch.ops.trx.attention_fwd.default._schema)
    assert s.startswith("trx::attention_fwd(Tensor q, Tensor k, Tensor v, Tensor? mask, bool causal, float scale, float p, int seed)")
    for name in ("attention_fwd", "attention_bwd", "add_layernorm_fwd", "add_layernorm_bwd"):
        assert hasattr(torch.ops.trx, name)


def test_fake_tensor_shapes():
    from torch._subclasses.fake_tensor import FakeTensorMode
    with FakeTensorMode():
        q = torch.empty(2, 9, 3, 64, dtype=torch.bfloat16, device="cuda")
        k = torch.empty(2, 17, 3, 64, dtype=torch.bfloat16, device="cuda")
        out, lse = torch.ops.trx.attention_fwd(q, k, k, None, False, 0.125, 0.0, 0)
        assert out.shape == (2, 9, 192) and lse.shape == (2, 3, 9) and lse.dtype == torch.float32
        x = torch.empty(5, 7, 768, device="cuda")
        g = torch.empty(768, device="cuda")
        y, mean, rstd = torch.ops.trx.add_layernorm_fwd(x, x, g, g, 1e-12, 0.0, 0)
        assert y.shape == x.shape and mean.shape == (35,) and rstd.shape == (35,)


def test_cpu_tensors_are_refused():
    q = torch.zeros(1, 4, 1, 64)
    with pytest.raises((NotImplementedError, RuntimeError)):       # no CPU kernel is registered: the dispatcher says so
        torch.ops.trx.attention_fwd(q, q, q, None, False, 0.125, 0.0, 0)


@pytest.mark.gpu
def test_attention_op_matches_the_function_path():
    g = torch.Generator(device="cuda").manual_seed(0)
    B, Lq, Lk, H = 3, 70, 130, 4
    for dt in (torch.bfloat16, torch.float32):
        q = torch.randn(B, Lq, H, 64, device="cuda", generator=g).to(dt)
        k = torch.randn(B, Lk, H, 64, device="cuda", generator=g).to(dt)
        v = torch.randn(B, Lk, H, 64, device="cuda", generator=g).to(dt)
        mask = torch.zeros(B, Lk, device="cuda"); mask[:, -11:] = torch.finfo(torch.float32).min
        go = torch.randn(B, Lq, H * 64, device="cuda", generator=g).to(dt)
        res = []
        for use_op in (False, True):
            a, b, c = (t.clone().requires_grad_() for t in (q, k, v))
            if use_op:
                out, _ = torch.ops.trx.attention_fwd(a, b, c, mask, False, 0.125, 0.1, 1234)
            else:
                out = ops.attention(a, b, c, mask=mask, scale=0.125, dropout_p=0.1, seed=1234)
            out.backward(go)
            res.append((out.detach(), a.grad, b.grad, c.grad))
        for x, y in zip(*res):
            assert torch.equal(x, y)


@pytest.mark.gpu
def test_add_layernorm_op_matches_the_function_path():
    g = torch.Generator(device="cuda").manual_seed(1)
    for dt in (torch.bfloat16, torch.float32):
        x = torch.randn(6, 33, 768, device="cuda", generator=g).to(dt)
        r = torch.randn(6, 33, 768, device="cuda", generator=g).to(dt)
        gm = torch.randn(768, device="cuda", generator=g); bt = torch.randn(768, device="cuda", generator=g)
        go = torch.randn(6, 33, 768, device="cuda", generator=g).to(dt)
        res = []
        for use_op in (False, True):
            a, b, c, d = (t.clone().requires_grad_() for t in (x, r, gm, bt))
            if use_op:
                y, _, _ = torch.ops.trx.add_layernorm_fwd(a, b, c, d, 1e-12, 0.1, 77)
            else:
                y = ops.add_layernorm(a, b, c, d, 1e-12, dropout_p=0.1, seed=77)
            y.backward(go)
            res.append((y.detach(), a.grad, b.grad, c.grad, d.grad))
        for p_, q_ in zip(*res):
            assert torch.equal(p_, q_)
