"""The dense encoder's forward at the size of the C4 refresh (BERT-base, passages of 16 .. 128 tokens + [CLS] [SEP], batches of
256 cut to their longest row, bf16 autocast): passages/s and the effective FLOP rate; the target of
    rocprofv3 --kernel-trace --stats -d gpurun_out/encprof -- python3 tools/encode_profile.py [n_passages [batch]]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from textreact_amd.dense import DenseEncoder, Config, encode
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40960
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 256
torch.manual_seed(0)
enc = DenseEncoder(Config(vocab_size=31090)).cuda().eval()
g = torch.Generator().manual_seed(1)
lengths = torch.randint(16, 129, (n,), generator=g) + 2
ids = torch.randint(5, 31090, (n, 130), generator=g)
mask = (torch.arange(130)[None, :] < lengths[:, None]).long()
ids = ids * mask
encode(enc, ids[:1024], mask[:1024], bs, lengths=lengths[:1024])
torch.cuda.synchronize(); t0 = time.perf_counter()
e = encode(enc, ids, mask, bs, lengths=lengths)
torch.cuda.synchronize(); t1 = time.perf_counter()
tok = float(lengths.sum())
order = torch.argsort(lengths, descending=True)
padded = float(sum(int(w) * min(bs, n - i * bs) for i, w in enumerate(lengths[order][::bs].tolist())))
flop_tok = 12 * 2 * (4 * 768 * 768 + 2 * 768 * 3072)
print(json.dumps({"passages": n, "batch": bs, "ms": round((t1 - t0) * 1e3, 1), "passages_per_s": round(n / (t1 - t0)), "tokens": tok, "padded_tokens": padded,
                  "gemm_PFLOPs_effective_on_padded_tokens": round(padded * flop_tok / (t1 - t0) / 1e15, 3)}))
