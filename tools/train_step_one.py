#!/usr/bin/env python3
"""N optimisation steps of the full-size predictor at the scripts' per-GPU shapes, for tools/prof_train.sh (rocprofv3 kernel trace):
python3 tools/train_step_one.py N hip L T   (bf16 autocast, dropout 0.1, fused AdamW; the first step allocates)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from textreact_amd.predictor import train, ops
from textreact_amd.predictor.model import Config

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
L = int(sys.argv[3]) if len(sys.argv) > 3 else 512
T = int(sys.argv[4]) if len(sys.argv) > 4 else 160
dev, B = torch.device("cuda", 0), 32
g = torch.Generator().manual_seed(0)
batch = {"input_ids": torch.randint(1, 31090, (B, L), generator=g).to(dev), "attention_mask": torch.ones(B, L, dtype=torch.long, device=dev),
         "decoder_input_ids": torch.randint(14, 600, (B, T), generator=g).to(dev), "decoder_attention_mask": torch.ones(B, T, dtype=torch.long, device=dev)}
batch["attention_mask"][::3, L * 4 // 5:] = 0
torch.manual_seed(0)
p = train.Predictor(Config(vocab_size=31090), Config(vocab_size=600, num_hidden_layers=6, type_vocab_size=1, layer_norm_eps=1e-5, is_decoder=True), mlm=False).to(dev).train()
opt, _ = train.configure_optimizer(p, 1e-4, 0.01, 1000, 0.02)
for i in range(steps):
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss, _ = p.training_step(batch)
    ops.backward(loss); opt.step(); opt.zero_grad(set_to_none=True)
    train.mark_parameters_updated(p)
torch.cuda.synchronize()
print("steps", steps, "loss %.4f" % float(loss))
