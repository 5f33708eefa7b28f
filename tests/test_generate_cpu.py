"""Beam search / greedy decoding (textreact_amd/predictor/generate.py) against what Hugging Face `generate`
returned for the REFERENCE model, called as textreact/main.py:218-226 calls it
(tests/golden/generate_small.npz, made by tests/golden/make_generate_golden.py)."""
import json
import os

import numpy as np
import pytest
import torch

from textreact_amd.predictor.generate import generate
from textreact_amd.predictor.model import Config, TextReactModel, random_state_dict

pytestmark = pytest.mark.usefixtures("reference_ops")
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "generate_small.npz")


def golden_model():
    z = np.load(G)
    enc, dec = json.loads(str(z["enc_cfg"])), json.loads(str(z["dec_cfg"]))
    m = TextReactModel(Config(**enc), Config(is_decoder=True, **dec))
    sd = random_state_dict(m, int(z["seed"]))
    sd["decoder.lm_head.bias"] = sd["decoder.lm_head.bias"].clone()
    sd["decoder.lm_head.bias"][dec["eos_token_id"]] += float(z["eos_boost"])
    sd["decoder.lm_head.decoder.bias"] = sd["decoder.lm_head.bias"]
    sd["decoder.lm_head.layer_norm.weight"] = sd["decoder.lm_head.layer_norm.weight"] * float(z["head_gain"])
    m.load_state_dict(sd)
    return z, dec, m.eval()


def until_eos(row, eos):
    """the hypothesis itself: tokens up to and including the first end token (what follows is padding, whose
    value differs between transformers versions and is erased by batch_decode(skip_special_tokens=True))"""
    row = list(int(t) for t in row)
    return row[: row.index(eos) + 1] if eos in row else row


def check_case(z, dec, m, i, case, dev="cpu", tol=1e-4, graph=None):
    ids, am = torch.from_numpy(z["input_ids"]).to(dev), torch.from_numpy(z["attention_mask"]).to(dev)
    seq, sc = generate(m, ids, am, num_beams=case["num_beams"], num_return_sequences=case["num_beams"],
                       max_length=case["max_length"], length_penalty=0, bos_token_id=dec["bos_token_id"],
                       eos_token_id=dec["eos_token_id"], pad_token_id=dec["pad_token_id"], graph=graph)
    want_seq, want_sc = z["sequences_%d" % i], z["scores_%d" % i]
    assert seq.shape[0] == want_seq.shape[0] == ids.shape[0] * case["num_beams"]
    got = [until_eos(r, dec["eos_token_id"]) for r in seq.cpu().numpy()]
    want = [until_eos(r, dec["eos_token_id"]) for r in want_seq]
    # a hypothesis cut off by max_length has no end token; the golden's rows are max_length wide in that case
    assert got == want, (case, got, want)
    if case["num_beams"] > 1:
        assert np.allclose(sc.cpu().numpy(), want_sc, atol=tol), (sc, want_sc)
    else:
        assert sc is None
    return seq


def test_beam_search_and_greedy_match_huggingface_generate_on_the_reference_model():
    z, dec, m = golden_model()
    cases = json.loads(str(z["cases"]))
    assert any(c["num_beams"] == 1 for c in cases) and any(c["num_beams"] >= 8 for c in cases)
    for i, c in enumerate(cases):
        seq = check_case(z, dec, m, i, c)
        assert int(seq.max()) < dec["vocab_size"] and bool((seq[:, 0] == dec["bos_token_id"]).all())


def test_generate_restores_training_mode_and_pads_with_the_pad_token():
    z, dec, m = golden_model()
    m.train()
    ids, am = torch.from_numpy(z["input_ids"]), torch.from_numpy(z["attention_mask"])
    seq, sc = generate(m, ids, am, num_beams=4, num_return_sequences=2, max_length=12, length_penalty=0,
                       bos_token_id=dec["bos_token_id"], eos_token_id=dec["eos_token_id"], pad_token_id=dec["pad_token_id"])
    assert m.training and seq.shape[0] == 6 and sc.shape == (6,)
    assert bool((sc.view(3, 2)[:, 0] >= sc.view(3, 2)[:, 1]).all())          # best first
    for row in seq.numpy():
        row = list(row)
        if dec["eos_token_id"] in row:
            assert all(t == dec["pad_token_id"] for t in row[row.index(dec["eos_token_id"]) + 1:])


def test_test_step_output_structure_follows_main_py():
    from textreact_amd.predictor import train
    z, dec, m = golden_model()

    class P:      # the Predictor surface test_step needs
        model = m
    ids, am = torch.from_numpy(z["input_ids"]), torch.from_numpy(z["attention_mask"])
    out = train.test_step(P, [7, 8, 9], {"input_ids": ids, "attention_mask": am}, num_beams=3, max_dec_length=8,
                          bos_token_id=dec["bos_token_id"], eos_token_id=dec["eos_token_id"], pad_token_id=dec["pad_token_id"])
    assert sorted(out) == [7, 8, 9]
    want = [until_eos(r, dec["eos_token_id"]) for r in z["sequences_1"]]
    strip = lambda r: [t for t in r if t not in (dec["bos_token_id"], dec["eos_token_id"], dec["pad_token_id"])]
    assert out[7]["prediction"] == [strip(r) for r in want[:3]] and len(out[9]["score"]) == 3
    assert np.allclose(out[8]["score"], z["scores_1"][3:6], atol=1e-4)
