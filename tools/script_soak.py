"""the reference's script configurations at full model size on the HIP ops, a few steps each, under the precision
the scripts pass (16-mixed = fp16 autocast + GradScaler): RCR (condition task, --mlm mlp head, train_RCR.sh), RetroSyn
template-free (train_RetroSyn_tf.sh) incl. validation and a beam-search test step, RetroSyn template-based
(train_RetroSyn_tb.sh).  Prints the losses of the HIP ops and of their PyTorch statement (oracle/nn_ref.py) side by side."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import nn_ref  # noqa: E402  (the PyTorch statement of the ops: the side the kernels are compared with)
from textreact_amd.predictor import template, train  # noqa: E402
from textreact_amd.predictor.model import Config  # noqa: E402

dev = "cuda"
g = torch.Generator().manual_seed(0)


def run(name, make, step_fn, steps=3):
    out = {}
    for backend in ("hip", "torch"):
        with nn_ref.implementation(backend):
            torch.manual_seed(0)
            p = make(backend).to(dev).train()
            opt = torch.optim.AdamW(p.parameters(), lr=1e-4, fused=True)
            scaler = torch.amp.GradScaler("cuda")
            losses, scales = [], []
            for _ in range(steps):
                with torch.autocast("cuda", dtype=torch.float16):
                    loss = step_fn(p)
                scaler.scale(loss).backward()
                scaler.step(opt); scaler.update(); opt.zero_grad(set_to_none=True)
                losses.append(float(loss)); scales.append(scaler.get_scale())
            assert all(l == l and abs(l) < 1e4 for l in losses), (name, backend, losses)
            out[backend] = (losses, p, scales)
    print(name, "hip", [round(l, 4) for l in out["hip"][0]], "torch", [round(l, 4) for l in out["torch"][0]],
          "loss scales", out["hip"][2], out["torch"][2], flush=True)
    # the trajectories must agree step by step (stale weights, a wrong gradient ... show here) -- as long as the two GradScalers
    # took the same decisions: the HIP ops carry fp16-autocast activations and gradients in bf16, which does not overflow where
    # fp16 does, so the PyTorch statement may skip a step (and halve its scale) that the HIP run takes
    for i, (a, c) in enumerate(zip(out["hip"][0], out["torch"][0])):
        if i > 0 and out["hip"][2][:i] != out["torch"][2][:i]:
            print(name, "  (the loss scalers part ways at step %d: later losses are not compared)" % i, flush=True)
            break
        assert abs(a - c) <= 5e-3 * max(1.0, abs(c)), (name, out["hip"][0], out["torch"][0])
    return out


B, L, T = 16, 512, 160
enc = dict(vocab_size=31090)
dec = dict(vocab_size=600, num_hidden_layers=6, type_vocab_size=1, layer_norm_eps=1e-5, is_decoder=True)
batch = {"input_ids": torch.randint(1, 31090, (B, L), generator=g).to(dev), "attention_mask": torch.ones(B, L, dtype=torch.long, device=dev),
         "decoder_input_ids": torch.randint(14, 600, (B, T), generator=g).to(dev),
         "decoder_attention_mask": torch.ones(B, T, dtype=torch.long, device=dev)}
batch["attention_mask"][::3, 400:] = 0
mlm_labels = torch.randint(0, 31090, (B, 77), generator=g).to(dev)

# 1. RCR: --mlm --mlm_layer mlp --mlm_lambda 0.1
run("RCR mlm", lambda be: train.Predictor(Config(**enc), Config(**dec), mlm=True, mlm_layer="mlp", mlm_lambda=0.1),
    lambda p: p.training_step(batch, {"mlm_labels": mlm_labels})[0])

# 2. RetroSyn template-free + validation + test step (beam 20 in the script; 5 here, short)
res = run("RetroSyn tf", lambda be: train.Predictor(Config(**enc), Config(**dec), mlm=False), lambda p: p.training_step(batch)[0])
p = res["hip"][1].eval()
small = {k: v[:4] for k, v in batch.items()}
with torch.autocast("cuda", dtype=torch.float16):
    val = p.validation_step(list(range(4)), small)
    out = train.test_step(p, list(range(4)), {"input_ids": small["input_ids"], "attention_mask": small["attention_mask"]}, num_beams=5,
                          max_dec_length=24, bos_token_id=12, eos_token_id=13, pad_token_id=0)
print("validation scores", [round(v, 3) for v in val.values()], "test step beams", len(out[0]["prediction"]), flush=True)

# 3. template-based (num templates as in USPTO-50k LocalRetro tables: a few hundred)
n_atoms = 40
tb = {"input_ids": batch["input_ids"], "attention_mask": batch["attention_mask"],
      "atom_indices": [torch.arange(1, 1 + n_atoms, device=dev) for _ in range(B)],
      "decoder_atom_template_labels": torch.randint(0, 157, (B, n_atoms), generator=g).to(dev),
      "decoder_bond_template_labels": torch.randint(0, 500, (B, n_atoms, n_atoms), generator=g).to(dev)}
tb["decoder_bond_template_labels"][:, :, 20:] = -100
run("RetroSyn tb", lambda be: template.TemplateBasedModel(Config(**enc), 156, 499),
    lambda m: template.template_loss(m(**{k: tb[k] for k in ("input_ids", "attention_mask", "atom_indices")})[0], tb))
print("soak ok")
