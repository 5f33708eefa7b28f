"""The dense encoder's forward at the size of the C4 refresh (BERT-base, passages of 16 .. 128 tokens + [CLS] [SEP], batches of
256 cut to their longest row, bf16 autocast): passages/s and the effective FLOP rate; the target of
    rocprofv3 --kernel-trace --stats -d gpurun_out/encprof -- python3 tools/encode_profile.py [n_passages [batch [batch_tokens]]]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from textreact_amd.dense import DenseEncoder, Config, encode
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40960
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 256
bt = int(sys.argv[3]) if len(sys.argv) > 3 else 65536      # 0: `batch` rows per batch whatever the width
torch.manual_seed(0)
enc = DenseEncoder(Config(vocab_size=31090)).cuda().eval()
g = torch.Generator().manual_seed(1)
lengths = torch.randint(16, 129, (n,), generator=g) + 2
ids = torch.randint(5, 31090, (n, 130), generator=g)
mask = (torch.arange(130)[None, :] < lengths[:, None]).long()
ids = (ids * mask).cuda(); mask = mask.cuda(); lengths = lengths.cuda()      # (on the device, as live.py holds them)
if os.environ.get("ENC_NO_PACK"):      # A/B: the packed-weight cache of ops.linear_multi off
    from textreact_amd.predictor import ops
    ops.linear_multi = lambda x, ws, bs_=None: ops.linear(x, torch.cat(list(ws)), None if bs_ is None else torch.cat(list(bs_)))
encode(enc, ids, mask, bs, lengths=lengths, batch_tokens=bt)      # (every batch width is a new GEMM shape to the library: ~20 ms of host time each, once per process)
torch.cuda.synchronize(); t0 = time.perf_counter()
e = encode(enc, ids, mask, bs, lengths=lengths, batch_tokens=bt)
torch.cuda.synchronize(); t1 = time.perf_counter()
tok = float(lengths.sum().item())
flop_tok = 12 * 2 * (4 * 768 * 768 + 2 * 768 * 3072)
print(json.dumps({"passages": n, "batch": bs, "ms": round((t1 - t0) * 1e3, 1), "passages_per_s": round(n / (t1 - t0)), "batch_tokens": bt, "tokens": tok,
                  "gemm_PFLOPs_effective_on_real_tokens": round(tok * flop_tok / (t1 - t0) / 1e15, 3)}))
