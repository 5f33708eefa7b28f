#!/usr/bin/env python3
"""the full-size predictor's optimisation step, timed alone (bench_predictor.train_step_bench's hip row only):
python3 tools/train_step_time.py [T [rounds]]   -- median / min of `rounds` timings of 10 steps each"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench_predictor as bp
from textreact_amd.predictor import train, ops
from textreact_amd.predictor.model import Config
T = int(sys.argv[1]) if len(sys.argv) > 1 else 160
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev, B, L = torch.device("cuda", 0), 32, 512
g = torch.Generator().manual_seed(0)
batch = {"input_ids": torch.randint(1, 31090, (B, L), generator=g).to(dev), "attention_mask": torch.ones(B, L, dtype=torch.long, device=dev),
         "decoder_input_ids": torch.randint(14, 600, (B, T), generator=g).to(dev), "decoder_attention_mask": torch.ones(B, T, dtype=torch.long, device=dev)}
batch["attention_mask"][::3, L * 4 // 5:] = 0
torch.manual_seed(0)
p = train.Predictor(Config(vocab_size=31090), Config(vocab_size=600, num_hidden_layers=6, type_vocab_size=1, layer_norm_eps=1e-5, is_decoder=True), mlm=False).to(dev).train()
opt, _ = train.configure_optimizer(p, 1e-4, 0.01, 1000, 0.02)
def step():
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss, _ = p.training_step(batch)
    ops.backward(loss); opt.step(); opt.zero_grad(set_to_none=True)
    return loss
ts = sorted(bp.timeit(step, iters=10, warm=4 if i == 0 else 0) for i in range(rounds))
print("T=%d train step: median %.2f ms  min %.2f ms  (loss %.4f)  env %s" % (T, ts[len(ts) // 2], ts[0], float(step()), {k: v for k, v in os.environ.items() if k.startswith("TRX_")}))
