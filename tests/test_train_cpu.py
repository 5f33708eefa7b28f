"""Training step, losses, checkpoint layout and 2-rank DDP of the predictor on CPU (on the PyTorch statement of the
two ops, oracle/nn_ref.py; the HIP kernels' gradients are checked against the same statement on the GPU)."""
import json
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from textreact_amd.predictor.model import Config, random_state_dict
from textreact_amd.predictor import train

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.usefixtures("reference_ops")
G = os.path.join(ROOT, "tests", "golden", "predictor_small.npz")


def _predictor(mlm=True, dropout=0.0):
    z = np.load(G)
    enc, dec = json.loads(str(z["enc_cfg"])), json.loads(str(z["dec_cfg"]))
    for c in (enc, dec):   # these tests compare numbers across runs: dropout off unless asked for
        c["hidden_dropout_prob"] = c["attention_probs_dropout_prob"] = dropout
    p = train.Predictor(Config(**enc), Config(is_decoder=True, **dec), mlm=mlm)
    g = torch.Generator().manual_seed(5)
    with torch.no_grad():
        for t in p.parameters():
            t.copy_(torch.randn(t.shape, generator=g) * 0.05)
    batch = {k: torch.from_numpy(z[k]) for k in ("input_ids", "attention_mask", "decoder_input_ids", "decoder_attention_mask")}
    return z, p, batch


def test_state_dict_prefixes_match_the_lightning_module():
    _, p, _ = _predictor()
    keys = list(p.state_dict().keys())
    assert all(k.startswith("model.") or k.startswith("mlm_head.") for k in keys)
    for k in ("model.encoder.embeddings.word_embeddings.weight", "model.decoder.lm_head.decoder.weight",
              "mlm_head.bias", "mlm_head.transform.dense.weight", "mlm_head.transform.LayerNorm.weight", "mlm_head.decoder.weight"):
        assert k in keys, k


def test_losses_follow_main_py():
    _, p, batch = _predictor()
    p.eval()
    logits, enc = p.model(**batch)
    loss = p.compute_loss(logits, batch)
    labels = batch["decoder_input_ids"][:, 1:]
    ref = torch.nn.functional.cross_entropy(logits[:, :-1].reshape(-1, logits.shape[-1]), labels.reshape(-1), ignore_index=0)
    assert torch.equal(loss, ref)
    per = p.compute_loss(logits, batch, reduction="none")
    assert per.shape == (3,)
    mlm_labels = torch.randint(0, 120, (3, 5))
    total, logs = p.training_step(batch, {"mlm_labels": mlm_labels})
    assert torch.allclose(total, logs["train_loss"] + logs["mlm_loss"]) and set(logs) == {"train_loss", "mlm_loss", "total_loss"}
    assert 0.0 <= float(p.compute_acc(logits, batch)) <= 1.0


def test_checkpoint_layout_round_trip(tmp_path):
    _, p, batch = _predictor()
    opt, sch = train.configure_optimizer(p, 1e-3, 0.01, 100, 0.02)
    total, _ = p.training_step(batch, {"mlm_labels": torch.randint(0, 120, (3, 5))})
    total.backward(); opt.step(); sch.step()
    path = train.save_checkpoint(str(tmp_path / "run" / "best.ckpt"), p, opt, sch, epoch=3, global_step=77, monitor="val_loss")
    raw = torch.load(path, map_location="cpu", weights_only=False)
    assert tuple(raw.keys()) == train.CKPT_KEYS and raw["epoch"] == 3 and raw["global_step"] == 77
    assert list(raw["state_dict"].keys()) == list(p.state_dict().keys())
    _, q, _ = _predictor()
    ck, missing, unexpected = train.load_checkpoint(path, q, strict=True)
    assert not missing and not unexpected
    for (n, a), (_, b) in zip(p.state_dict().items(), q.state_dict().items()):
        assert torch.equal(a, b), n
    # a transformers-4.27.3 era checkpoint also carries position_ids buffers: tolerated (strict=False, main.py:404)
    raw["state_dict"]["model.encoder.embeddings.position_ids"] = torch.arange(64)[None]
    torch.save(raw, path)
    train.load_checkpoint(path, q, strict=True)
    train.clear_checkpoints(str(tmp_path / "run"))
    assert not os.listdir(tmp_path / "run")


def test_checkpoints_are_interchangeable_between_graphed_and_eager_runs(tmp_path):
    """a --hip_graph_step run keeps the learning rate in a tensor and AdamW `capturable`; its checkpoint holds what an eager
    run (and the reference's Lightning run) writes, and loading an eager checkpoint into an optimizer built for the graph
    keeps the optimizer's form: the SAME rate tensor (the captured update reads it) holding the loaded value"""
    _, p, batch = _predictor()
    opt, sch = train.configure_optimizer(p, 1e-3, 0.01, 100, 0.02)
    total, _ = p.training_step(batch, {"mlm_labels": torch.randint(0, 120, (3, 5))})
    total.backward(); opt.step(); sch.step(); sch.step()
    eager_lr = opt.param_groups[0]["lr"]
    path = train.save_checkpoint(str(tmp_path / "last.ckpt"), p, opt, sch, epoch=0, global_step=2)
    # the graphed form of the same optimizer (built by hand: configure_optimizer only makes it for parameters on a GPU)
    _, q, _ = _predictor()
    lr_t = torch.tensor(1e-3)
    gopt = torch.optim.AdamW(list(q.parameters()), lr=lr_t, weight_decay=0.01, capturable=True)
    gsch = torch.optim.lr_scheduler.LambdaLR(gopt, lambda step: 1.0)
    train.load_checkpoint(path, q, gopt, gsch)
    g = gopt.param_groups[0]
    assert g["lr"] is lr_t and abs(float(lr_t) - eager_lr) < 1e-9 and g["capturable"] is True
    assert gsch._last_lr[0] is lr_t and gsch.last_epoch == 2
    # ... and what it writes back is the portable form again
    back = train.save_checkpoint(str(tmp_path / "graphed.ckpt"), q, gopt, gsch, epoch=1, global_step=3)
    raw = torch.load(back, map_location="cpu", weights_only=False)
    grp = raw["optimizer_states"][0]["param_groups"][0]
    assert isinstance(grp["lr"], float) and grp["capturable"] is False and abs(grp["lr"] - eager_lr) < 1e-9
    assert all(isinstance(v, float) for v in raw["lr_schedulers"][0]["_last_lr"] + raw["lr_schedulers"][0]["base_lrs"])
    assert gopt.param_groups[0]["lr"] is lr_t          # saving did not touch the live optimizer
    # an eager optimizer resumes from it without inheriting the graph's flags
    _, r, _ = _predictor()
    eopt, esch = train.configure_optimizer(r, 1e-3, 0.01, 100, 0.02)
    train.load_checkpoint(back, r, eopt, esch)
    assert isinstance(eopt.param_groups[0]["lr"], float) and not eopt.param_groups[0].get("capturable", False)


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); pt = s.getsockname()[1]; s.close(); return pt


def _ddp_worker2(rank, world, port, ret):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import nn_ref
    nn_ref.install()                 # this child process runs the module tree on the PyTorch statement of the ops
    _, p, batch = _predictor(mlm=False)

    class Step(torch.nn.Module):
        def __init__(self, pred):
            super().__init__(); self.pred = pred

        def forward(self, **b):
            return self.pred.training_step(b)[0]
    ddp = torch.nn.parallel.DistributedDataParallel(Step(p), find_unused_parameters=True)
    lo = 0 if rank == 0 else 2
    hi = 2 if rank == 0 else 3
    shard = {k: v[lo:hi] for k, v in batch.items()}
    loss = ddp(**shard)
    loss.backward()
    grads = {n: t.grad.clone() for n, t in p.named_parameters() if t.grad is not None}
    merged = train.gather_outputs({rank: float(loss)})
    ret[rank] = (grads, merged)
    dist.destroy_process_group()


def test_two_rank_ddp_gradients_are_the_rank_average():
    mgr = mp.Manager(); ret = mgr.dict()
    mp.spawn(_ddp_worker2, args=(2, _free_port(), ret), nprocs=2, join=True)
    g0, m0 = ret[0]; g1, m1 = ret[1]
    assert m0 == m1 and set(m0) == {0, 1}                      # all_gather_object merge (main.py:259-268)
    for n in g0:
        assert torch.allclose(g0[n], g1[n], atol=1e-7), n     # DDP all-reduce left identical gradients
    # and they equal the average of the two ranks' local gradients
    _, p, batch = _predictor(mlm=False)
    local = []
    for lo, hi in ((0, 2), (2, 3)):
        p.zero_grad()
        p.training_step({k: v[lo:hi] for k, v in batch.items()})[0].backward()
        local.append({n: t.grad.clone() for n, t in p.named_parameters() if t.grad is not None})
    for n in g0:
        assert torch.allclose(g0[n], (local[0][n] + local[1][n]) / 2, atol=1e-6), n


def test_training_mode_applies_dropout_and_eval_mode_does_not():
    _, p, batch = _predictor(mlm=False, dropout=0.1)
    p.eval()
    a = p.model(**batch)[0]
    assert torch.equal(a, p.model(**batch)[0])
    p.train()
    torch.manual_seed(0); b1 = p.model(**batch)[0]
    torch.manual_seed(0); b2 = p.model(**batch)[0]
    torch.manual_seed(1); b3 = p.model(**batch)[0]
    assert torch.equal(b1, b2) and not torch.equal(b1, b3) and not torch.equal(a, b1)


def test_merge_predictions_per_neighbor():
    # main.py:239-240 / utils.py:55-64: samples 0..5 = reactions 0, 1 with 3 neighbours each, outputs concatenated per key
    from textreact_amd.predictor import train
    outs = {i: {"prediction": ["p%d" % i], "score": [float(i)]} for i in (3, 0, 5, 1, 4, 2)}
    m = train.merge_predictions_per_neighbor(outs, 3)
    assert m == {0: {"prediction": ["p0", "p1", "p2"], "score": [0.0, 1.0, 2.0]},
                 1: {"prediction": ["p3", "p4", "p5"], "score": [3.0, 4.0, 5.0]}}


def test_mark_parameters_updated_sees_a_submodule_added_later():
    """the module list mark_parameters_updated caches is rebuilt when any module was registered since: a layer added after the
    first step gets its weight shadows marked stale like the others"""
    import torch
    from textreact_amd.predictor import train, ops

    class Holder(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.a = torch.nn.Linear(4, 4)

    m = Holder()
    m.a.__dict__["_weight_shadows"] = ops.WeightShadows()
    train.mark_parameters_updated(m)
    assert m.a.__dict__["_weight_shadows"].generation == 1
    m.b = torch.nn.Linear(4, 4)                       # added after the cache was made
    m.b.__dict__["_weight_shadows"] = ops.WeightShadows()
    train.mark_parameters_updated(m)
    assert m.b.__dict__["_weight_shadows"].generation == 1 and m.a.__dict__["_weight_shadows"].generation == 2
