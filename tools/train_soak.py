"""200 optimisation steps of the full-size predictor (bf16 autocast, dropout 0.1, fused AdamW, batches of varying ragged
length) on the HIP ops and on their PyTorch statement (oracle/nn_ref.py) under the same seeds: loss every 20 steps, and the memory the hip run holds"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import nn_ref  # noqa: E402  (the PyTorch statement of the ops: the side the kernels are compared with)
from textreact_amd.predictor import train, ops  # noqa: E402
from textreact_amd.predictor.model import Config, random_state_dict  # noqa: E402

dev, steps = "cuda", int(sys.argv[1]) if len(sys.argv) > 1 else 200
res = {}
for backend in ("hip", "torch"):
    with nn_ref.implementation(backend):
        torch.manual_seed(0)
        g = torch.Generator().manual_seed(0)
        p = train.Predictor(Config(vocab_size=31090), Config(vocab_size=600, num_hidden_layers=6, type_vocab_size=1, layer_norm_eps=1e-5,
                                                             is_decoder=True), mlm=False)
        p.model.load_state_dict(random_state_dict(p.model, 1))
        p = p.to(dev).train()
        opt, sched = train.configure_optimizer(p, 1e-4, 0.01, steps, 0.02)
        losses = []
        for it in range(steps):
            L, T = int(torch.randint(200, 513, (1,), generator=g)), int(torch.randint(40, 161, (1,), generator=g))
            src = torch.randint(14, 600, (16, T), generator=g)
            batch = {"input_ids": torch.cat([src, torch.randint(1, 31090, (16, L - T), generator=g)], 1).to(dev),
                     "attention_mask": torch.ones(16, L, dtype=torch.long, device=dev),
                     "decoder_input_ids": src.to(dev), "decoder_attention_mask": torch.ones(16, T, dtype=torch.long, device=dev)}
            with torch.autocast("cuda", dtype=torch.bfloat16):
                loss, _ = p.training_step(batch)
            ops.backward(loss); opt.step(); opt.zero_grad(set_to_none=True)      # what the trainer calls: the weight gradients as one grouped launch
            train.mark_parameters_updated(p)
            if sched is not None:
                sched.step()
            if it % 20 == 19:
                losses.append(round(float(loss), 3))
                if backend == "hip":
                    print(it, "hip loss", losses[-1], "allocated MB %.0f reserved MB %.0f" % (torch.cuda.memory_allocated() / 2**20, torch.cuda.memory_reserved() / 2**20), flush=True)
        res[backend] = losses
        del p, opt
        torch.cuda.empty_cache()
print("hip  ", res["hip"])
print("torch", res["torch"])
