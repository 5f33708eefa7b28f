"""Generates tests/golden/template_small.npz: outputs of the REFERENCE template-based model
(textreact.model.TemplateBasedModel + TemplatePredictionHead over a Hugging Face BertModel, eager attention,
eval mode), its losses (main.py:112-123), accuracy (main.py:138-150) and edit ranking
(textreact.utils.combined_edit) for seeded weights and inputs.

Run HERE (needs /root/reference and transformers; rdkit is stubbed):
    python tests/golden/make_template_golden.py
"""
import json
import os
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
from make_predictor_golden import ENC  # noqa: E402

SEED, N_ATOM_T, N_BOND_T = 4321, 7, 5


def seeded_state_dict(module, seed):
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for name, t in module.state_dict().items():
        w = torch.randn(t.shape, generator=g) * 0.05
        if name.endswith("LayerNorm.weight"):
            w = w + 1.0
        sd[name] = w
    return sd


def inputs():
    g = torch.Generator().manual_seed(7)
    ids = torch.randint(1, ENC["vocab_size"], (3, 30), generator=g)
    am = torch.ones(3, 30, dtype=torch.long)
    am[1, 22:] = 0; ids[1, 22:] = 0
    atom_indices = [torch.tensor([1, 2, 3, 5, 8, 9]), torch.tensor([2, 4, 6, 7]), torch.tensor([1, 3, 4, 10, 11])]
    amax = 6
    atom_labels = torch.full((3, amax), -100, dtype=torch.long)
    bond_labels = torch.full((3, amax, amax), -100, dtype=torch.long)
    bonds = [[(0, 1), (1, 0), (1, 2), (2, 1), (3, 4), (4, 3)], [(0, 1), (1, 0), (2, 3), (3, 2)], [(0, 2), (2, 0), (1, 4), (4, 1)]]
    for b, idx in enumerate(atom_indices):
        atom_labels[b, :len(idx)] = torch.randint(0, N_ATOM_T + 1, (len(idx),), generator=g)
        for (i, j) in bonds[b]:
            bond_labels[b, i, j] = int(torch.randint(0, N_BOND_T + 1, (1,), generator=g))
    raw = [[("a", 1, 3), ("b", (1, 2), 2)], [("b", (0, 1), 1)], []]
    return ids, am, atom_indices, atom_labels, bond_labels, bonds, raw


def main():
    for mod in ("rdkit", "rdkit.Chem"):
        sys.modules.setdefault(mod, types.ModuleType(mod))
    sys.modules["rdkit"].Chem = sys.modules["rdkit.Chem"]
    sys.path.insert(0, "/root/reference")
    from transformers import BertConfig, BertModel
    from textreact.model import TemplateBasedModel, TemplatePredictionHead
    from textreact import utils
    from textreact_amd.predictor.model import Config
    from textreact_amd.predictor import template as T

    enc = BertModel(BertConfig(**{k: v for k, v in ENC.items() if k != "model_type"}))
    try:
        enc.config._attn_implementation = "eager"
    except Exception:
        pass
    ref = TemplateBasedModel(enc, TemplatePredictionHead(ENC["hidden_size"], N_ATOM_T, N_BOND_T)).eval()
    mine = T.TemplateBasedModel(Config(**ENC), N_ATOM_T, N_BOND_T, backend="torch").eval()
    sd = seeded_state_dict(mine, SEED)
    missing, unexpected = ref.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all("position_ids" in k or "token_type_ids" in k for k in missing), missing
    mine.load_state_dict(sd)
    ids, am, atom_indices, atom_labels, bond_labels, bonds, raw = inputs()
    batch = {"decoder_atom_template_labels": atom_labels, "decoder_bond_template_labels": bond_labels, "bonds": bonds,
             "decoder_raw_template_labels": raw}
    with torch.no_grad():
        out = ref(input_ids=ids, attention_mask=am, atom_indices=atom_indices)
    atom_logits, bond_logits = out.logits

    # the reference's own loss / accuracy code, main.py:112-123 and :138-150, on its own logits
    def ref_loss(reduction):
        la = F.cross_entropy(atom_logits.reshape(-1, atom_logits.shape[-1]), atom_labels.reshape(-1), reduction=reduction)
        lb = F.cross_entropy(bond_logits.reshape(-1, bond_logits.shape[-1]), bond_labels.reshape(-1), reduction=reduction)
        if reduction == "none":
            la, lb = la.view(3, -1).mean(dim=1), lb.view(3, -1).mean(dim=1)
        return la + lb
    ap, bp = F.softmax(atom_logits, dim=-1), F.softmax(bond_logits, dim=-1)
    ap[atom_labels == -100] = 0; bp[bond_labels == -100] = 0
    # raw template labels that exercise hit / partial hit / empty: built from the reference's own top-1 edits
    top1 = [utils.combined_edit(a_, b_, bd, 1)[0][0] for a_, b_, bd in zip(ap, bp, bonds)]
    raw = [[top1[0]], [top1[1], ("a", 0, 1)], []]
    batch["decoder_raw_template_labels"] = raw
    acc, edits = [], []
    for a_, b_, bd, rw in zip(ap, bp, bonds, raw):
        pred = utils.combined_edit(a_, b_, bd, 1)[0][0]
        acc.append(float(pred in rw) / max(len(rw), 1))
        e, p = utils.combined_edit(a_, b_, bd, top_num=6)
        edits.append({"pred": [list(x) if not isinstance(x, tuple) else [x[0], list(x[1]) if isinstance(x[1], tuple) else x[1], x[2]] for x in e], "prob": p})
    np.savez_compressed(os.path.join(HERE, "template_small.npz"), seed=SEED, enc_cfg=json.dumps(ENC), n_atom_t=N_ATOM_T,
                        n_bond_t=N_BOND_T, input_ids=ids.numpy(), attention_mask=am.numpy(),
                        atom_indices=json.dumps([t.tolist() for t in atom_indices]), atom_labels=atom_labels.numpy(),
                        bond_labels=bond_labels.numpy(), bonds=json.dumps(bonds), raw=json.dumps(raw),  # tuples become lists: the test turns them back
                       
                        atom_logits=atom_logits.numpy(), bond_logits=bond_logits.numpy(),
                        encoder_last_hidden_state=out.encoder_last_hidden_state.numpy(),
                        loss_mean=float(ref_loss("mean")), loss_none=ref_loss("none").numpy(), acc=np.array(acc),
                        edits=json.dumps(edits), state_dict_keys=json.dumps(sorted(ref.state_dict().keys())))
    with torch.no_grad():
        (ma, mb), menc = mine(ids, am, atom_indices)
    print("max |atom - ref| =", float((ma - atom_logits).abs().max()), " max |bond - ref| =", float((mb - bond_logits).abs().max()),
          " loss", float(ref_loss("mean")), "mine", float(T.template_loss((ma, mb), batch)), "acc", acc)
    print(edits[0])


if __name__ == "__main__":
    main()
