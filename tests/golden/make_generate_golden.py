"""Generates tests/golden/generate_small.npz: beam-search output of the REFERENCE model through Hugging Face
`generate`, called the way textreact/main.py:218-226 calls it (num_beams = num_return_sequences,
length_penalty = 0, bos / eos / pad ids, scores returned), for the seeded weights of predictor_small.npz
plus a bias on the EOS logit (so that hypotheses finish at different lengths).

Run HERE (needs /root/reference and transformers; rdkit is stubbed):
    python tests/golden/make_generate_golden.py
"""
import json
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
from make_predictor_golden import DEC, ENC, SEED, inputs  # noqa: E402

EOS_BOOST = 1.0
HEAD_GAIN = 6.0     # sharpens the output distribution so that hypotheses of different lengths compete
CASES = [dict(num_beams=5, max_length=14), dict(num_beams=3, max_length=8), dict(num_beams=1, max_length=10), dict(num_beams=8, max_length=20)]


def main():
    for mod in ("rdkit", "rdkit.Chem"):
        sys.modules.setdefault(mod, types.ModuleType(mod))
    sys.modules["rdkit"].Chem = sys.modules["rdkit.Chem"]
    sys.path.insert(0, "/root/reference")
    from textreact.model import get_model
    from textreact_amd.predictor.model import Config, TextReactModel, random_state_dict

    tmp = tempfile.mkdtemp()
    for name, cfg in (("enc", ENC), ("dec", DEC)):
        os.makedirs(os.path.join(tmp, name))
        json.dump(cfg, open(os.path.join(tmp, name, "config.json"), "w"))

    class Args:
        template_based = False; encoder = os.path.join(tmp, "enc"); decoder = os.path.join(tmp, "dec")
        encoder_pretrained = False; decoder_pretrained = False; max_length = 64; encoder_tokenizer = "text"
    ref = get_model(Args())
    ref.eval()
    for m in (ref, ref.encoder, ref.decoder):
        try:
            m.config._attn_implementation = "eager"
        except Exception:
            pass
    mine = TextReactModel(Config(**ENC), Config(is_decoder=True, **DEC), backend="torch")
    sd = random_state_dict(mine, SEED)
    sd["decoder.lm_head.bias"] = sd["decoder.lm_head.bias"].clone()
    sd["decoder.lm_head.bias"][DEC["eos_token_id"]] += EOS_BOOST
    sd["decoder.lm_head.decoder.bias"] = sd["decoder.lm_head.bias"]
    sd["decoder.lm_head.layer_norm.weight"] = sd["decoder.lm_head.layer_norm.weight"] * HEAD_GAIN
    ref.load_state_dict(sd, strict=False)
    ids, am, _, _ = inputs()
    out = dict(seed=SEED, eos_boost=EOS_BOOST, head_gain=HEAD_GAIN, enc_cfg=json.dumps(ENC), dec_cfg=json.dumps(DEC),
               input_ids=ids.numpy(), attention_mask=am.numpy(), cases=json.dumps(CASES))
    for i, c in enumerate(CASES):
        with torch.no_grad():
            o = ref.generate(input_ids=ids, attention_mask=am, num_beams=c["num_beams"], num_return_sequences=c["num_beams"],
                             max_length=c["max_length"], length_penalty=0, bos_token_id=DEC["bos_token_id"],
                             eos_token_id=DEC["eos_token_id"], pad_token_id=DEC["pad_token_id"],
                             return_dict_in_generate=True, output_scores=True, do_sample=False)
        seq = o.sequences.numpy()
        sc = o.sequences_scores.numpy() if getattr(o, "sequences_scores", None) is not None else np.zeros(len(seq), np.float32)
        print(c, "sequences", seq.shape, "lengths", (seq != 0).sum(1).tolist())
        print(seq[:c["num_beams"]]); print(sc[:c["num_beams"]])
        out["sequences_%d" % i] = seq
        out["scores_%d" % i] = sc
    np.savez_compressed(os.path.join(HERE, "generate_small.npz"), **out)


if __name__ == "__main__":
    main()
