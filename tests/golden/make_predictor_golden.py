"""Generates tests/golden/predictor_small.npz: logits of the REFERENCE model
(textreact.model.get_model from /root/reference -> Hugging Face EncoderDecoderModel, eager
attention, eval mode, fp32) for seeded weights and inputs.  The weights come from
textreact_amd.predictor.model.random_state_dict(seed), which is deterministic, so the fixture
carries only the seed, the two configs, the inputs and the expected outputs.

Run HERE (needs /root/reference and transformers; rdkit is stubbed):
    python tests/golden/make_predictor_golden.py
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

ENC = dict(model_type="bert", vocab_size=120, hidden_size=128, num_hidden_layers=2, num_attention_heads=2,
           intermediate_size=256, max_position_embeddings=64, type_vocab_size=2, layer_norm_eps=1e-12,
           hidden_act="gelu", hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1, pad_token_id=0)
DEC = dict(model_type="roberta", vocab_size=60, hidden_size=128, num_hidden_layers=2, num_attention_heads=2,
           intermediate_size=256, max_position_embeddings=64, type_vocab_size=1, layer_norm_eps=1e-5,
           hidden_act="gelu", hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1,
           bos_token_id=12, eos_token_id=13, pad_token_id=0)
SEED = 1234


def inputs():
    g = torch.Generator().manual_seed(99)
    ids = torch.randint(1, ENC["vocab_size"], (3, 24), generator=g)
    am = torch.ones(3, 24, dtype=torch.long)
    am[1, 17:] = 0; ids[1, 17:] = 0
    am[2, 5:] = 0; ids[2, 5:] = 0
    dids = torch.randint(14, DEC["vocab_size"], (3, 9), generator=g)
    dids[:, 0] = 12
    dam = torch.ones(3, 9, dtype=torch.long)
    dids[1, 6:] = 0; dam[1, 6:] = 0
    return ids, am, dids, dam


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
    missing, unexpected = ref.load_state_dict(sd, strict=False)
    ref_keys = set(ref.state_dict().keys())
    print("ref keys not provided:", sorted(missing))
    print("provided keys unknown to ref:", sorted(unexpected))
    assert not unexpected, unexpected
    assert all("position_ids" in k or "token_type_ids" in k for k in missing), missing
    ids, am, dids, dam = inputs()
    with torch.no_grad():
        out = ref(input_ids=ids, attention_mask=am, decoder_input_ids=dids, decoder_attention_mask=dam)
    np.savez_compressed(os.path.join(HERE, "predictor_small.npz"), seed=SEED, enc_cfg=json.dumps(ENC), dec_cfg=json.dumps(DEC),
                        input_ids=ids.numpy(), attention_mask=am.numpy(), decoder_input_ids=dids.numpy(),
                        decoder_attention_mask=dam.numpy(), logits=out.logits.numpy(),
                        encoder_last_hidden_state=out.encoder_last_hidden_state.numpy(),
                        state_dict_keys=json.dumps(sorted(ref_keys)))
    # cross-check right here with the torch backend of our own tree
    mine.load_state_dict(sd)
    mine.eval()
    with torch.no_grad():
        lg, enc = mine(ids, am, dids, dam)
    print("max |logits - ref| =", float((lg - out.logits).abs().max()), " max |enc - ref| =",
          float((enc - out.encoder_last_hidden_state).abs().max()))


if __name__ == "__main__":
    main()
