"""round 5: who launches the ~144 FillFunctor kernels of a training step (profiles/r05_train_step_kernel_stats.csv)?  One step
under torch.profiler with stacks; aten::fill_ / aten::zero_ / aten::zeros* calls grouped by their Python caller."""
import collections
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.profiler import profile, ProfilerActivity
from textreact_amd.predictor import train, ops
from textreact_amd.predictor.model import Config
dev, B, L, T = torch.device("cuda", 0), 32, 512, 160
g = torch.Generator().manual_seed(0)
batch = {"input_ids": torch.randint(1, 31090, (B, L), generator=g).to(dev), "attention_mask": torch.ones(B, L, dtype=torch.long, device=dev),
         "decoder_input_ids": torch.randint(14, 600, (B, T), generator=g).to(dev), "decoder_attention_mask": torch.ones(B, T, dtype=torch.long, device=dev)}
torch.manual_seed(0)
p = train.Predictor(Config(vocab_size=31090), Config(vocab_size=600, num_hidden_layers=6, type_vocab_size=1, layer_norm_eps=1e-5, is_decoder=True), mlm=False).to(dev).train()
opt, _ = train.configure_optimizer(p, 1e-4, 0.01, 1000, 0.02)
def step():
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss, _ = p.training_step(batch)
    ops.backward(loss); opt.step(); opt.zero_grad(set_to_none=True)
    train.mark_parameters_updated(p)
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
cnt = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::fill_", "aten::zero_", "aten::zeros", "aten::zeros_like", "aten::new_zeros", "aten::full", "aten::ones_like", "aten::ones"):
        st = [s for s in (ev.stack or []) if "textreact_amd" in s or "torch/optim" in s or "autograd" in s]
        cnt[(ev.name, st[0] if st else (ev.stack[0] if ev.stack else "?"))] += 1
for (n, s), c in cnt.most_common(30):
    print(c, n, s)
