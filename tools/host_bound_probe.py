"""wall time of a training step at B = 32 and at B = 4 (same kernel count, 1/8 of the GPU work): the small batch's
time is (an upper bound of) what the host needs to issue a step"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from textreact_amd.predictor.model import Config
from textreact_amd.predictor import train
dev = "cuda"
for backend in ("hip", "torch"):
    p = train.Predictor(Config(vocab_size=31090), Config(vocab_size=600, num_hidden_layers=6, type_vocab_size=1, layer_norm_eps=1e-5, is_decoder=True),
                        mlm=False, backend=backend).to(dev).train()
    opt, _ = train.configure_optimizer(p, 1e-4, 0.01, 1000, 0.02)
    for B in (32, 4):
        L, T = 512, 160
        g = torch.Generator().manual_seed(0)
        batch = {"input_ids": torch.randint(1, 31090, (B, L), generator=g).to(dev), "attention_mask": torch.ones(B, L, dtype=torch.long, device=dev),
                 "decoder_input_ids": torch.randint(14, 600, (B, T), generator=g).to(dev), "decoder_attention_mask": torch.ones(B, T, dtype=torch.long, device=dev)}
        def step():
            with torch.autocast("cuda", dtype=torch.bfloat16):
                loss, _ = p.training_step(batch)
            loss.backward(); opt.step(); opt.zero_grad(set_to_none=True)
        for _ in range(4): step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): step()
        torch.cuda.synchronize(); t1 = time.perf_counter()
        print(backend, "B", B, "%.2f ms/step" % ((t1 - t0) * 100))
    del p, opt; torch.cuda.empty_cache()
