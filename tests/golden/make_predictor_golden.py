"""Generates tests/golden/predictor_small.npz and tests/golden/predictor_full.npz: logits of the REFERENCE model
(textreact.model.get_model from /root/reference -> Hugging Face EncoderDecoderModel, eager
attention, eval mode, fp32) for seeded weights and inputs.  The weights come from
textreact_amd.predictor.model.random_state_dict(seed), which is deterministic, so the fixture
carries only the seed, the two configs, the inputs and the expected outputs.

predictor_full.npz is the same at the scripts' size ("hashes for the full-size model"): BERT-base encoder
with the SciBERT vocabulary (31090), the decoder of textreact/configs/bert_l6.json (read from the reference), B = 2,
L = 512, decoder lengths T = 7 (RCR: scripts/train_RCR.sh) and T = 160 (train_RetroSyn_tf.sh:33).  It stays compact: all
logits at T = 7, the logits of 32 sampled (sample, position) pairs at T = 160, the encoder states of 32 sampled
positions, and the SHA-256 of every full fp32 array.

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


def reference_and_mine(enc_cfg, dec_cfg, seed, max_length):
    """the reference's get_model(...) (eager attention, eval) and our module tree, both holding random_state_dict(seed)"""
    for mod in ("rdkit", "rdkit.Chem"):
        sys.modules.setdefault(mod, types.ModuleType(mod))
    sys.modules["rdkit"].Chem = sys.modules["rdkit.Chem"]
    if "/root/reference" not in sys.path:
        sys.path.insert(0, "/root/reference")
    from textreact.model import get_model
    from textreact_amd.predictor.model import Config, TextReactModel, random_state_dict

    tmp = tempfile.mkdtemp()
    for name, cfg in (("enc", enc_cfg), ("dec", dec_cfg)):
        os.makedirs(os.path.join(tmp, name))
        json.dump(cfg, open(os.path.join(tmp, name, "config.json"), "w"))

    class Args:
        template_based = False; encoder = os.path.join(tmp, "enc"); decoder = os.path.join(tmp, "dec")
        encoder_pretrained = False; decoder_pretrained = False; encoder_tokenizer = "text"
    Args.max_length = max_length
    ref = get_model(Args())
    ref.eval()
    for m in (ref, ref.encoder, ref.decoder):
        try:
            m.config._attn_implementation = "eager"
        except Exception:
            pass
    mine = TextReactModel(Config(**enc_cfg), Config(is_decoder=True, **dec_cfg))
    sd = random_state_dict(mine, seed)
    missing, unexpected = ref.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all("position_ids" in k or "token_type_ids" in k for k in missing), missing
    mine.load_state_dict(sd)
    return ref, mine.eval(), sd


FULL_ENC = dict(model_type="bert", vocab_size=31090, hidden_size=768, num_hidden_layers=12, num_attention_heads=12,
                intermediate_size=3072, max_position_embeddings=512, type_vocab_size=2, layer_norm_eps=1e-12,
                hidden_act="gelu", hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1, pad_token_id=0)   # BERT-base, SciBERT vocabulary
FULL_SEED = 7


def full_inputs(T):
    g = torch.Generator().manual_seed(1)
    ids = torch.randint(1, 31090, (2, 512), generator=g)
    am = torch.ones(2, 512, dtype=torch.long)
    am[1, 300:] = 0; ids[1, 300:] = 0
    dids = torch.randint(14, 600, (2, T), generator=g)
    dids[:, 0] = 12
    return ids, am, dids


def sha(a):
    import hashlib
    return hashlib.sha256(np.ascontiguousarray(a, dtype=np.float32).tobytes()).hexdigest()


def main_full():
    from oracle import nn_ref
    dec_cfg = json.load(open("/root/reference/textreact/configs/bert_l6.json"))
    ref, mine, _ = reference_and_mine(FULL_ENC, dec_cfg, FULL_SEED, 512)
    out = {"seed": FULL_SEED, "enc_cfg": json.dumps(FULL_ENC), "dec_cfg": json.dumps(dec_cfg)}
    rng = np.random.default_rng(0)
    for T in (7, 160):
        ids, am, dids = full_inputs(T)
        with torch.no_grad():
            r = ref(input_ids=ids, attention_mask=am, decoder_input_ids=dids)
            with nn_ref.reference_ops():
                lg, enc = mine(ids, am, dids)
        logits, states = r.logits.numpy(), r.encoder_last_hidden_state.numpy()
        print("T = %d: max |logits - ref| = %.3g, max |enc - ref| = %.3g (our tree on the PyTorch statement of the ops)"
              % (T, float((lg - r.logits).abs().max()), float((enc - r.encoder_last_hidden_state).abs().max())))
        out["sha_logits_T%d" % T], out["sha_enc_T%d" % T] = sha(logits), sha(states)
        if T == 7:
            out["logits_T7"] = logits
            pos = np.stack([rng.integers(0, 2, 32), rng.integers(0, 512, 32)], 1)
            pos[pos[:, 0] == 1, 1] %= 300                     # sample 1 is padded from 300 on
            out["enc_pos"], out["enc_at"] = pos, states[pos[:, 0], pos[:, 1]]
        else:
            pos = np.stack([rng.integers(0, 2, 32), rng.integers(0, T, 32)], 1)
            out["logits_pos_T160"], out["logits_at_T160"] = pos, logits[pos[:, 0], pos[:, 1]]
    np.savez_compressed(os.path.join(HERE, "predictor_full.npz"), **out)
    print("predictor_full.npz: %d bytes" % os.path.getsize(os.path.join(HERE, "predictor_full.npz")))


def main():
    from oracle import nn_ref
    ref, mine, sd = reference_and_mine(ENC, DEC, SEED, 64)
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
    # cross-check right here with our own tree on the PyTorch statement of the ops
    with torch.no_grad(), nn_ref.reference_ops():
        lg, enc = mine(ids, am, dids, dam)
    print("max |logits - ref| =", float((lg - out.logits).abs().max()), " max |enc - ref| =",
          float((enc - out.encoder_last_hidden_state).abs().max()))


if __name__ == "__main__":
    main()
    main_full()
