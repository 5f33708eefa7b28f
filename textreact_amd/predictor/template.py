"""Template-based retrosynthesis branch (`--template_based`, scripts/train_RetroSyn_tb.sh).

Reference: textreact/model.py:11-19,49-90 -- `TemplateBasedModel(encoder, TemplatePredictionHead)`: the
BERT encoder (no decoder), the hidden states of the atom tokens gathered per reaction and padded, an
atom-template classifier (Linear) and a bond-template classifier over every ordered atom pair
(`BondTemplatePredictor`: Linear on the concatenation [x_i ; x_j]); losses and the edit ranking are
main.py:112-123,138-150,201-216 and utils.py:68-108.

Parameter names are the reference's (`encoder.*`, `template_head.atom_template_head.*`,
`template_head.bond_template_head.linear.*`), so its checkpoints load.  One deliberate difference in
the arithmetic: the reference materialises the [.., L, L, 2d] tensor of concatenated pairs and runs one
Linear over it; W [x_i ; x_j] + b = W_left x_i + (W_right x_j + b), so here it is two [L, d] x [d, n]
GEMMs and a broadcast add -- L times less memory traffic, the same numbers up to fp32 rounding.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn.utils.rnn import pad_sequence

from . import ops
from .model import BertEncoder, additive_key_mask, weight_shadows


class BondTemplatePredictor(nn.Module):
    def __init__(self, input_size, num_bond_templates):
        super().__init__()
        self.linear = nn.Linear(2 * input_size, num_bond_templates + 1)

    def forward(self, x):
        d = x.shape[-1]
        w = self.linear.weight
        left = F.linear(x, w[:, :d])                        # depends on atom i (dim -3 of the result)
        right = F.linear(x, w[:, d:], self.linear.bias)     # depends on atom j (dim -2)
        return left.unsqueeze(-2) + right.unsqueeze(-3)


class TemplatePredictionHead(nn.Module):
    """[B x] L x d -> ([B x] L x (n_a + 1), [B x] L x L x (n_b + 1))"""

    def __init__(self, input_size, num_atom_templates, num_bond_templates):
        super().__init__()
        self.atom_template_head = nn.Linear(input_size, num_atom_templates + 1)
        self.bond_template_head = BondTemplatePredictor(input_size, num_bond_templates)

    def forward(self, x):
        return self.atom_template_head(x), self.bond_template_head(x)


class TemplateBasedModel(nn.Module):
    """forward(input_ids, attention_mask, atom_indices=[LongTensor per reaction]) ->
    ((atom_logits, bond_logits), encoder_last_hidden_state)"""

    def __init__(self, enc_cfg, num_atom_templates, num_bond_templates):
        super().__init__()
        self.encoder = BertEncoder(enc_cfg)
        self.template_head = TemplatePredictionHead(enc_cfg.hidden_size, num_atom_templates, num_bond_templates)

    def forward(self, input_ids, attention_mask=None, atom_indices=None, position_ids=None, token_type_ids=None, **_):
        if attention_mask is None:
            attention_mask = torch.ones_like(input_ids)
        if self.training:
            with ops.seed_scope(), ops.use_shadows(weight_shadows(self, input_ids)):
                enc = self.encoder(input_ids, additive_key_mask(attention_mask), position_ids, token_type_ids, None)
        else:
            enc = self.encoder(input_ids, additive_key_mask(attention_mask), position_ids, token_type_ids, None)
        atoms = pad_sequence([h[idx] for h, idx in zip(enc, atom_indices)], batch_first=True)
        return self.template_head(atoms), enc


# ---- losses and metrics (main.py:112-123, 138-150) ----------------------------------------------------
def template_loss(logits, batch, reduction="mean"):
    atom_logits, bond_logits = logits
    b = atom_logits.shape[0]
    la = F.cross_entropy(atom_logits.reshape(-1, atom_logits.shape[-1]), batch["decoder_atom_template_labels"].reshape(-1),
                         reduction=reduction)
    lb = F.cross_entropy(bond_logits.reshape(-1, bond_logits.shape[-1]), batch["decoder_bond_template_labels"].reshape(-1),
                         reduction=reduction)
    if reduction == "none":
        la, lb = la.view(b, -1).mean(dim=1), lb.view(b, -1).mean(dim=1)
    return la + lb


def masked_probabilities(logits, batch):
    """softmax over templates, rows whose label is -100 (padding / non-bonds) zeroed (main.py:140-143)"""
    atom_p, bond_p = F.softmax(logits[0], dim=-1), F.softmax(logits[1], dim=-1)
    atom_p = atom_p.masked_fill((batch["decoder_atom_template_labels"] == -100).unsqueeze(-1), 0.0)
    bond_p = bond_p.masked_fill((batch["decoder_bond_template_labels"] == -100).unsqueeze(-1), 0.0)
    return atom_p, bond_p


def template_acc(logits, batch, reduction="mean"):
    atom_p, bond_p = masked_probabilities(logits, batch)
    acc = []
    for ap, bp, bonds, raw in zip(atom_p, bond_p, batch["bonds"], batch["decoder_raw_template_labels"]):
        pred = combined_edit(ap, bp, bonds, 1)[0][0]
        acc.append(float(pred in raw) / max(len(raw), 1))
    acc = torch.tensor(acc)
    return acc.mean() if reduction == "mean" else acc


# ---- edit ranking (utils.py:68-108, adapted there from LocalRetro) ---------------------------------------
def _id_template(a, class_n, num_atoms, edit_type):
    idx, template = int(a) // class_n, int(a) % class_n
    return ((idx // num_atoms, idx % num_atoms) if edit_type == "b" else idx), template


def output2edit(out, top_num, edit_type, bonds=None):
    """the top_num most probable (site, template != 0) pairs; bond sites must be real bonds"""
    num_atoms, class_n = out.shape[-2:]
    readout = out.detach().cpu().numpy().reshape(-1)
    picked = []
    for r in np.flip(np.argsort(readout)):
        idx, template = _id_template(r, class_n, num_atoms, edit_type)
        if (bonds is None or idx in bonds) and template != 0:
            picked.append(r)
            if len(picked) == top_num:
                break
    return [_id_template(a, class_n, num_atoms, edit_type) for a in picked], [readout[a].item() for a in picked]


def combined_edit(atom_out, bond_out, bonds, top_num=None):
    ids_a, p_a = output2edit(atom_out, top_num, "a")
    ids_b, p_b = output2edit(bond_out, top_num, "b", bonds=bonds)
    ids, kinds, probs = ids_a + ids_b, ["a"] * len(p_a) + ["b"] * len(p_b), p_a + p_b
    rank = np.flip(np.argsort(probs))
    if top_num is not None:
        rank = rank[:top_num]
    return [(kinds[r], *ids[r]) for r in rank], [probs[r] for r in rank]


def template_test_step(model, indices, batch_in, top_num=500):
    """main.py:201-216: {idx: {'prediction', 'score', 'raw_template_labels', 'top1_template_match'}}"""
    logits, _ = model(**batch_in)
    atom_p, bond_p = masked_probabilities(logits, batch_in)
    out = {}
    for idx, ap, bp, bonds, raw in zip(indices, atom_p, bond_p, batch_in["bonds"], batch_in["decoder_raw_template_labels"]):
        pred, prob = combined_edit(ap, bp, bonds, top_num=top_num)
        out[int(idx)] = {"prediction": pred, "score": prob, "raw_template_labels": raw,
                         "top1_template_match": pred[0] in raw}
    return out


# ---- the LightningModule's three steps for this branch (main.py:165-233 with args.template_based) -----------------
class TemplatePredictor(nn.Module):
    """what `textreact_amd.main` drives under --template_based: the same step interface as predictor/train.py's Predictor,
    parameters under `model.` like the reference's LightningModule (so the checkpoint keys are the reference's)"""

    def __init__(self, enc_cfg, num_atom_templates, num_bond_templates):
        super().__init__()
        self.model = TemplateBasedModel(enc_cfg, num_atom_templates, num_bond_templates)

    @staticmethod
    def _model_inputs(batch_in):
        return {k: batch_in[k] for k in ("input_ids", "attention_mask", "atom_indices") if k in batch_in}

    def training_step(self, batch_in, batch_out=None):
        logits, _ = self.model(**self._model_inputs(batch_in))
        loss = template_loss(logits, batch_in)
        return loss, {"train_loss": loss.detach()}

    @torch.no_grad()
    def validation_step(self, indices, batch_in, val_metric="val_loss"):
        logits, _ = self.model(**self._model_inputs(batch_in))
        if val_metric == "val_loss":
            scores = template_loss(logits, batch_in, reduction="none")
        elif val_metric == "val_acc":
            scores = template_acc(logits, batch_in, reduction="none")
        else:
            raise ValueError(val_metric)
        return {int(i): float(s) for i, s in zip(indices, scores)}

    @torch.no_grad()
    def test_step(self, indices, batch_in, top_num=500):
        return template_test_step(lambda **kw: self.model(**self._model_inputs(kw)), indices, batch_in, top_num=top_num)
