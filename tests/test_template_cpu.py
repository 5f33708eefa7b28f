"""Template-based branch (textreact_amd/predictor/template.py) against outputs of the reference's own
TemplateBasedModel / loss / accuracy / combined_edit (tests/golden/template_small.npz,
tests/golden/make_template_golden.py)."""
import json
import os

import numpy as np
import pytest
import torch

from textreact_amd.predictor import template as T
from textreact_amd.predictor.model import Config

pytestmark = pytest.mark.usefixtures("reference_ops")
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "template_small.npz")


def _tup(x):
    """json turned the tuples into lists: ('a', idx, t) / ('b', (i, j), t) back"""
    return (x[0], tuple(x[1]) if isinstance(x[1], list) else x[1], x[2])


def load():
    z = np.load(G)
    enc = json.loads(str(z["enc_cfg"]))
    m = T.TemplateBasedModel(Config(**enc), int(z["n_atom_t"]), int(z["n_bond_t"]))
    g = torch.Generator().manual_seed(int(z["seed"]))
    sd = {}
    for name, t in m.state_dict().items():
        w = torch.randn(t.shape, generator=g) * 0.05
        sd[name] = w + 1.0 if name.endswith("LayerNorm.weight") else w
    m.load_state_dict(sd)
    batch = {"input_ids": torch.from_numpy(z["input_ids"]), "attention_mask": torch.from_numpy(z["attention_mask"]),
             "atom_indices": [torch.tensor(a) for a in json.loads(str(z["atom_indices"]))],
             "decoder_atom_template_labels": torch.from_numpy(z["atom_labels"]),
             "decoder_bond_template_labels": torch.from_numpy(z["bond_labels"]),
             "bonds": [[tuple(p) for p in b] for b in json.loads(str(z["bonds"]))],
             "decoder_raw_template_labels": [[_tup(x) for x in r] for r in json.loads(str(z["raw"]))]}
    return z, m.eval(), batch


def check(z, m, batch, tol):
    with torch.no_grad():
        logits, enc = m(**batch)
    assert float((logits[0].cpu() - torch.from_numpy(z["atom_logits"])).abs().max()) <= tol
    assert float((logits[1].cpu() - torch.from_numpy(z["bond_logits"])).abs().max()) <= tol
    assert float((enc.cpu() - torch.from_numpy(z["encoder_last_hidden_state"])).abs().max()) <= tol
    assert abs(float(T.template_loss(logits, batch)) - float(z["loss_mean"])) <= tol
    assert np.allclose(T.template_loss(logits, batch, reduction="none").cpu().numpy(), z["loss_none"], atol=tol)
    assert np.allclose(T.template_acc(logits, batch, reduction="none").numpy(), z["acc"])          # 1, 1/2, 0
    want = json.loads(str(z["edits"]))
    ap, bp = T.masked_probabilities(logits, batch)
    for a_, b_, bonds, w in zip(ap, bp, batch["bonds"], want):
        pred, prob = T.combined_edit(a_, b_, bonds, top_num=6)
        assert pred == [_tup(x) for x in w["pred"]]
        assert np.allclose(prob, w["prob"], atol=tol)
    return logits


def test_template_model_loss_accuracy_and_edit_ranking_match_the_reference():
    z, m, batch = load()
    check(z, m, batch, 2e-5)
    # state dict names are the reference's
    assert sorted(k for k in m.state_dict()) == [k for k in json.loads(str(z["state_dict_keys"]))
                                                  if "position_ids" not in k and "token_type_ids" not in k or k in m.state_dict()]


def test_template_test_step_structure():
    z, m, batch = load()
    out = T.template_test_step(m, [5, 6, 7], batch, top_num=4)
    assert sorted(out) == [5, 6, 7] and len(out[5]["prediction"]) == 4
    assert out[5]["top1_template_match"] is True and out[7]["top1_template_match"] is False
    assert out[6]["prediction"][0] in out[6]["raw_template_labels"]
