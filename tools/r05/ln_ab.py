"""round 5: the training step with the add+LayerNorm column sums finished by one launch at the end of the backward pass
(ops.backward: trx_add_layernorm_bwd_reduce_many) against the per-call second stage, ONE process, one model, alternating blocks
of 10 steps.   python3 tools/r05/ln_ab.py [T]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from textreact_amd.predictor import ops, train  # noqa: E402
from textreact_amd.predictor.model import Config  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 160
dev, B, L = "cuda", 32, 512
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
    ops.backward(loss)
    opt.step()
    opt.zero_grad(set_to_none=True)


deferrable = ops._ln_deferrable
modes = {"deferred": deferrable, "per call": lambda params, needs: False}
for _ in range(6):
    step()
times = {k: [] for k in modes}
for rnd in range(6):
    for name, fn in modes.items():
        ops._ln_deferrable = fn
        step()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); a.record()
        for _ in range(10):
            step()
        b.record(); torch.cuda.synchronize()
        times[name].append(a.elapsed_time(b) / 10)
print(json.dumps({"what": "train step, B32 x L512 x T%d, bf16 autocast, ms per step; 6 interleaved rounds of 10 steps" % T,
                  **{k: {"median": sorted(v)[len(v) // 2], "min": min(v), "all": [round(x, 3) for x in v]} for k, v in times.items()}}))
