"""BASELINE.json configs[4] on the hardware a test box has: on-the-fly retrieval through the HIP encoder (attention /
add+LayerNorm kernels), the HIP flat index per rank and the HIP merge -- two ranks sharing one GPU over gloo (RCCL refuses
two ranks on one device; no multi-GPU node was available to this project, so RCCL has not carried this path yet).

1. the refreshed neighbours are exactly what the oracle finds on the same embeddings (bit-exact ids), on both ranks;
2. `python -m textreact_amd.main --live_every 1 ...` (train_RetroSyn_tf.sh's options: --mlm, --use_gold_neighbor,
   --random_neighbor_ratio 0.2, --num_neighbors 3) trains, validates on both loaders, tests and writes its files, with
   one rank and with two."""
import json
import os
import socket
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLS, SEP, PAD, MASK = 101, 102, 0, 103


def _corpus_dict(P, Lp, seed, vocab):
    g = torch.Generator().manual_seed(seed)
    plen = torch.randint(4, Lp + 1, (P,), generator=g)
    pids = torch.randint(200, vocab, (P, Lp), generator=g)
    for i in range(7, P, 7):
        pids[i], plen[i] = pids[i - 2], plen[i - 2]            # duplicated texts
    return {"passage_ids": pids, "passage_len": plen, "marker_ids": torch.tensor([[40, 50 + j, 41] for j in range(3)]),
            "cls_id": CLS, "sep_id": SEP, "pad_id": PAD, "mask_id": MASK}


def _worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from textreact_amd import dense, live
    from textreact_amd.predictor.model import Config
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    corpus = live.LiveCorpus(_corpus_dict(3001, 40, 1, 5000), dev)
    torch.manual_seed(0)
    enc = dense.DenseEncoder(Config(vocab_size=5000, num_hidden_layers=2, max_position_embeddings=64)).to(dev).eval()
    g = torch.Generator().manual_seed(2)
    qlen = torch.randint(3, 30, (777,), generator=g)
    qids = torch.randint(200, 5000, (777, 30), generator=g)
    nn, emb_q, emb_p = live.refresh_neighbors(enc, enc, corpus, qids, qlen, 10, rank, world, return_embeddings=True)
    ret[rank] = (nn.cpu().numpy(), emb_q.float().cpu().numpy(), emb_p.float().cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [1, 2])
def test_refreshed_neighbours_equal_the_oracle_on_the_same_embeddings(world):
    import torch.multiprocessing as mp
    from oracle import flat_knn as oracle
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0)); port = s_.getsockname()[1]
    mgr = mp.Manager(); ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    emb_p = np.concatenate([ret[r][2] for r in range(world)])              # the shards, in rank order = row order
    assert emb_p.shape == (3001, 768)
    for r in range(world):
        assert np.array_equal(ret[r][1], ret[0][1])                        # one all-gather: the same query matrix everywhere
        _, want = oracle.knn_canonical(0, ret[r][1], emb_p, 10)
        assert np.array_equal(ret[r][0], want), "rank %d" % r


def _toy(tmp_path, n=26):
    g = torch.Generator().manual_seed(0)
    enc = dict(vocab_size=5000, hidden_size=256, num_hidden_layers=1, num_attention_heads=4, intermediate_size=512,
               max_position_embeddings=128)
    dec = dict(vocab_size=20, hidden_size=256, num_hidden_layers=1, num_attention_heads=4, intermediate_size=512,
               max_position_embeddings=32, type_vocab_size=1, layer_norm_eps=1e-5)
    (tmp_path / "enc.json").write_text(json.dumps(enc))
    (tmp_path / "dec.json").write_text(json.dumps(dec))
    torch.save(_corpus_dict(300, 20, 3, 5000), tmp_path / "corpus.pt")
    for name, m in (("train", n), ("val", 6), ("test", 5)):
        dids = torch.randint(3, 20, (m, 6), generator=g)
        dids[:, 0] = 1
        dids[:, -1] = 2
        torch.save({"indices": list(range(100, 100 + m)), "decoder_input_ids": dids, "decoder_attention_mask": torch.ones_like(dids),
                    "query_ids": torch.randint(200, 5000, (m, 12), generator=g), "query_len": torch.randint(3, 13, (m,), generator=g),
                    "gold_passage": torch.randint(-1, 300, (m,), generator=g)}, tmp_path / (name + ".pt"))
    return ["--task", "retro", "--encoder", "allenai/scibert_scivocab_uncased", "--arch_encoder", str(tmp_path / "enc.json"),
            "--decoder", str(tmp_path / "dec.json"), "--save_path", str(tmp_path / "out"),
            "--tensors_train", str(tmp_path / "train.pt"), "--tensors_valid", str(tmp_path / "val.pt"),
            "--tensors_test", str(tmp_path / "test.pt"), "--live_every", "1", "--live_corpus", str(tmp_path / "corpus.pt"),
            "--num_neighbors", "3", "--use_gold_neighbor", "--random_neighbor_ratio", "0.2", "--max_length", "96",
            "--mlm", "--mlm_ratio", "0.15", "--mlm_layer", "mlp", "--mlm_lambda", "0.1",
            "--batch_size", "6", "--lr", "1e-3", "--warmup", "0.5", "--num_beams", "3", "--max_dec_length", "8",
            "--test_batch_size", "2", "--val_metric", "val_loss", "--print_freq", "1", "--precision", "bf16-mixed",
            "--epochs", "2", "--do_train", "--do_valid", "--do_test", "--overwrite"]


def test_trainer_with_on_the_fly_retrieval_one_rank(tmp_path, capsys):
    from textreact_amd import main as M
    assert M.main(_toy(tmp_path)) == 0
    out = tmp_path / "out"
    printed = capsys.readouterr().out
    assert printed.count("neighbours refreshed") == 2 and '"val_loss/1"' in printed       # every epoch; both loaders
    ck = torch.load(out / "best.ckpt", weights_only=False)
    assert all(torch.isfinite(v).all() for v in ck["state_dict"].values() if v.is_floating_point())
    for di in (0, 1):                                         # as retrieved / gold text removed
        pred = json.loads((out / ("prediction_test_%d.json" % di)).read_text())
        assert sorted(pred) == ["100", "101", "102", "103", "104"] and len(pred["100"]["prediction"]) == 3


def _rank_main(rank, world, port, argv):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      TRX_DIST_BACKEND="gloo", TRX_DEVICE="0")
    sys.path.insert(0, ROOT)
    from textreact_amd import main as M
    assert M.main(argv) == 0


def test_trainer_with_on_the_fly_retrieval_two_ranks_on_one_gpu(tmp_path):
    import torch.multiprocessing as mp
    argv = _toy(tmp_path, n=25) + ["--gpus", "2"]              # 25 samples on 2 ranks: uneven shards, padded
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0)); port = s_.getsockname()[1]
    mp.spawn(_rank_main, args=(2, port, argv), nprocs=2, join=True)
    out = tmp_path / "out"
    ck = torch.load(out / "last.ckpt", weights_only=False)
    assert ck["epoch"] == 1 and ck["global_step"] == 2 * 3     # ceil(25 / 12) = 3 steps per epoch
    assert all(torch.isfinite(v).all() for v in ck["state_dict"].values() if v.is_floating_point())
    pred = json.loads((out / "prediction_test_1.json").read_text())
    assert sorted(pred) == ["100", "101", "102", "103", "104"]


# ---- BASELINE.json configs[4] at the size of scripts/train_RetroSyn_tf.sh (bench_predictor.py --live) -------------------------
# BERT-base retriever, 204,800 passages of 16-128 tokens, 40,000 queries (a USPTO-50K-sized training split), k = 10: one refresh,
# stage by stage, on one rank and as a 2-rank one-device rehearsal; the refreshed ids of 64 sampled queries must be exactly what
# the oracle finds over the FULL passage-embedding matrix.
LIVE_P, LIVE_N = 204_800, 40_000


def _check_against_the_oracle(nn, emb_q, emb_p, k=10):
    from oracle import flat_knn as oracle
    rows = np.random.default_rng(3).choice(emb_q.shape[0], 64, replace=False)
    _, want = oracle.knn_canonical(0, emb_q[rows], emb_p, k)
    assert np.array_equal(nn[rows], want)


def test_refresh_at_the_scripts_size_one_rank():
    sys.path.insert(0, ROOT)
    import bench_predictor
    dev = torch.device("cuda", 0)
    row, nn, emb_q, emb_p = bench_predictor.live_bench(dev, LIVE_P, LIVE_N, return_tensors=True)
    assert nn.shape == (LIVE_N, 10) and emb_p.shape == (LIVE_P, 768) and int(nn.min()) >= 0 and int(nn.max()) < LIVE_P
    assert set(row["ms"]) >= {"encode_passages", "index_add", "encode_queries", "search", "assemble_epoch_inputs"}
    assert row["assembled_width"] <= 512
    _check_against_the_oracle(nn.cpu().numpy(), emb_q.float().cpu().numpy(), emb_p.float().cpu().numpy())
    print(json.dumps(row))


def _live_full_worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench_predictor
    from textreact_amd.sharded import shard_bounds
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    row, nn, emb_q, emb_p = bench_predictor.live_bench(dev, LIVE_P, LIVE_N, rank=rank, world=world, return_tensors=True)
    lo, hi = shard_bounds(LIVE_P, world, rank)
    assert emb_p.shape == (hi - lo, 768)
    per = -(-LIVE_P // world)
    mine = torch.zeros((per, 768), dtype=torch.bfloat16)
    mine[:hi - lo] = emb_p.cpu()
    parts = [torch.empty_like(mine).view(torch.uint8) for _ in range(world)]      # gloo gathers bytes
    dist.all_gather(parts, mine.view(torch.uint8))
    full = torch.cat([parts[r].view(torch.bfloat16)[:shard_bounds(LIVE_P, world, r)[1] - shard_bounds(LIVE_P, world, r)[0]] for r in range(world)])
    h = torch.tensor([int(nn.sum()), int((nn * torch.arange(1, 11, device=nn.device)).sum())])
    hs = [torch.empty_like(h) for _ in range(world)]
    dist.all_gather(hs, h)
    assert all(torch.equal(x, hs[0]) for x in hs), "ranks disagree on the neighbours"
    if rank == 0:
        _check_against_the_oracle(nn.cpu().numpy(), emb_q.float().cpu().numpy(), full.float().numpy())
        ret["row"] = json.dumps(row)
    dist.barrier()
    dist.destroy_process_group()


def test_refresh_at_the_scripts_size_two_ranks_on_one_gpu():
    import torch.multiprocessing as mp
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0)); port = s_.getsockname()[1]
    ret = mp.Manager().dict()
    mp.spawn(_live_full_worker, args=(2, port, ret), nprocs=2, join=True)
    row = json.loads(ret["row"])
    assert row["ranks"] == 2 and "REHEARSAL" in row["transport"]
    print(json.dumps(row))
