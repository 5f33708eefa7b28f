"""`python -m textreact_amd.main`: the reference's flag surface (main.py:26-97) and the train / validate / test /
resume cycle with its files (best.ckpt, last.ckpt, prediction_{split}_{i}.json), on CPU with toy tensors (the
`reference_ops` fixture puts the PyTorch statement of the ops, oracle/nn_ref.py, under the module tree: the CLI itself has
one implementation and refuses to start without a GPU) and on the GPU through the HIP ops."""
import glob
import json
import os
import re
import shlex
import sys

import pytest
import torch

from textreact_amd import main as M

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

REF_SCRIPTS = "/root/reference/scripts"


def _script_argv(path):
    """the argument list a scripts/train_*.sh hands to main.py, shell variables expanded the way bash would"""
    text = open(path).read()
    env = {}
    for m in re.finditer(r"^([A-Z_]+)=(.*)$", text, re.M):
        env[m.group(1)] = m.group(2).strip()
    cmd = text[text.index("python main.py") + len("python main.py"):]
    cmd = cmd.replace("\\\n", " ")

    def arith(m):
        expr = re.sub(r"[A-Z_]+", lambda v: env[v.group(0)], m.group(1))
        return str(int(eval(expr.replace("/", "//"))))
    cmd = re.sub(r"\$\(\((.*?)\)\)", arith, cmd)
    cmd = re.sub(r"\$\{([A-Z_]+)\}", lambda m: env[m.group(1)], cmd)
    cmd = re.sub(r"\$([A-Z_]+)", lambda m: env[m.group(1)], cmd)
    return shlex.split(cmd, comments=True)


@pytest.mark.skipif(not os.path.isdir(REF_SCRIPTS), reason="the reference tree is not on this box")
def test_every_training_script_parses():
    scripts = sorted(glob.glob(os.path.join(REF_SCRIPTS, "train_*.sh")))
    assert len(scripts) == 6
    parser = M.get_parser()
    for s in scripts:
        argv = _script_argv(s)
        args = parser.parse_args(argv)          # SystemExit on an unknown or ambiguous flag
        assert args.do_train and args.save_path.startswith("output/")
        assert args.template_based or args.decoder.endswith(".json")
        if s.endswith("train_RCR.sh"):          # scripts/train_RCR.sh:37 relies on prefix matching: --warmup -> --warmup_ratio
            assert args.warmup_ratio == 0.02 and args.num_beams == 15 and args.precision == "16-mixed"
            assert args.mlm and args.mlm_layer == "mlp" and args.mlm_lambda == 0.1 and args.batch_size == 32


def test_flag_surface_is_the_references():
    # every option string of main.py:26-97; parsed out of the reference when it is here, else the count is pinned
    ours = {a.option_strings[0] for a in M.get_parser()._actions if a.option_strings and a.option_strings[0] != "-h"}
    if os.path.isfile("/root/reference/main.py"):
        ref = set(re.findall(r"add_argument\('(--[a-z_]+)'", open("/root/reference/main.py").read()))
        assert len(ref) == 62 and ref <= ours, sorted(ref - ours)
    assert len(ours) >= 62


def _toy(tmp_path, n=12, seed=0):
    g = torch.Generator().manual_seed(seed)
    enc = dict(vocab_size=60, hidden_size=64, num_hidden_layers=1, num_attention_heads=1, intermediate_size=64,
               max_position_embeddings=32, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    dec = dict(vocab_size=20, hidden_size=64, num_hidden_layers=1, num_attention_heads=1, intermediate_size=64,
               max_position_embeddings=32, type_vocab_size=1, layer_norm_eps=1e-5, hidden_dropout_prob=0.0,
               attention_probs_dropout_prob=0.0)
    (tmp_path / "enc.json").write_text(json.dumps(enc))
    (tmp_path / "dec.json").write_text(json.dumps(dec))
    for name, m in (("train", n), ("val", 5), ("val_nogold", 5), ("test", 4)):
        ids = torch.randint(3, 60, (m, 9), generator=g)
        dids = torch.randint(3, 20, (m, 6), generator=g)
        dids[:, 0] = 1
        dids[:, -1] = 2
        torch.save({"indices": list(range(100, 100 + m)), "input_ids": ids, "attention_mask": torch.ones_like(ids),
                    "decoder_input_ids": dids, "decoder_attention_mask": torch.ones_like(dids),
                    "mlm_labels": torch.randint(0, 60, (m, 3), generator=g)}, tmp_path / (name + ".pt"))
    return ["--task", "condition", "--encoder", "allenai/scibert_scivocab_uncased", "--arch_encoder", str(tmp_path / "enc.json"),
            "--decoder", str(tmp_path / "dec.json"), "--save_path", str(tmp_path / "out"),
            "--tensors_train", str(tmp_path / "train.pt"), "--tensors_valid", "%s,%s" % (tmp_path / "val.pt", tmp_path / "val_nogold.pt"),
            "--tensors_test", str(tmp_path / "test.pt"), "--batch_size", "6", "--lr", "1e-3", "--mlm", "--mlm_layer", "mlp",
            "--mlm_lambda", "0.1", "--warmup", "0.5", "--num_beams", "3", "--max_dec_length", "8", "--test_batch_size", "2",
            "--val_metric", "val_loss", "--print_freq", "1"]


def test_train_validate_test_resume(tmp_path, capsys, reference_ops):
    argv = _toy(tmp_path)
    out = tmp_path / "out"
    # two optimiser steps (12 samples / batch 6), one epoch, then validate + test from best.ckpt
    assert M.main(argv + ["--epochs", "1", "--do_train", "--do_valid", "--do_test", "--overwrite"]) == 0
    assert (out / "best.ckpt").is_file() and (out / "last.ckpt").is_file()
    ck = torch.load(out / "best.ckpt", weights_only=False)
    assert ck["global_step"] == 2 and ck["epoch"] == 0 and ck["pytorch-lightning_version"].startswith("2.")
    assert all(k.startswith("model.") or k.startswith("mlm_head.") for k in ck["state_dict"])
    assert ck["callbacks"]["ModelCheckpoint"]["monitor"] == "val_loss"
    pred = json.loads((out / "prediction_test_0.json").read_text())
    assert sorted(pred) == ["100", "101", "102", "103"] and len(pred["100"]["prediction"]) == 3 and len(pred["100"]["score"]) == 3
    printed = capsys.readouterr().out
    assert "Num training steps: 2" in printed and '"val_loss/1"' in printed         # second dataloader = gold-removed set
    # resume: --load_ckpt last.ckpt without --overwrite continues at epoch 1 (main.py:389-391)
    assert M.main(argv + ["--epochs", "2", "--do_train", "--load_ckpt", "last.ckpt"]) == 0
    assert "Resumed from" in capsys.readouterr().out
    ck2 = torch.load(out / "last.ckpt", weights_only=False)
    assert ck2["epoch"] == 1 and ck2["global_step"] == 4
    # the checkpoint reloads into a fresh module
    from textreact_amd.predictor import train as T
    args = M.get_args(argv)
    enc_cfg, dec_cfg = M._configs(args)
    fresh = T.Predictor(enc_cfg, dec_cfg, mlm=True, mlm_layer="mlp")
    _, missing, unexpected = T.load_checkpoint(str(out / "last.ckpt"), fresh)
    assert not missing and not unexpected


def _rank_main(rank, world, port, argv):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from oracle import nn_ref
    nn_ref.install()                 # this child process: the module tree on the PyTorch statement of the ops
    assert M.main(argv) == 0


def test_the_cli_refuses_to_run_without_a_gpu(tmp_path):
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from textreact_amd.predictor import ops
    with pytest.raises(ops.TrxNNError):
        M.main(_toy(tmp_path) + ["--epochs", "1", "--do_train", "--overwrite"])


def test_epoch_shards_are_equal_and_follow_the_distributed_sampler():
    # torch's DistributedSampler is what Lightning puts under the reference's loader: same permutation, same padding
    from torch.utils.data.distributed import DistributedSampler
    for n, world in ((13, 2), (12, 2), (7, 4), (5, 8)):
        shards = [M.epoch_shard(n, 42, 3, r, world) for r in range(world)]
        assert len({len(s) for s in shards}) == 1 and len(shards[0]) == -(-n // world)
        assert set(sum(shards, [])) == set(range(n))
        for r in range(world):
            ds = DistributedSampler(range(n), num_replicas=world, rank=r, shuffle=True, seed=42)
            ds.set_epoch(3)
            assert list(ds) == shards[r]
    assert M.epoch_shard(13, 42, 0, 0, 2) != M.epoch_shard(13, 42, 1, 0, 2)     # a resumed run does not replay epoch 0


def test_two_ranks_gloo_uneven_shards(tmp_path):
    # 13 samples on 2 ranks, batch 6: without padding rank 0 would run 2 micro-batches and rank 1 only 1, and rank 0's
    # gradient all-reduce would pair with rank 1's barrier (a hang on RCCL)
    import socket
    import torch.multiprocessing as mp
    argv = _toy(tmp_path, n=13) + ["--epochs", "2", "--do_train", "--do_test", "--overwrite", "--gpus", "2"]
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0)); port = s_.getsockname()[1]
    mp.spawn(_rank_main, args=(2, port, argv), nprocs=2, join=True)
    ck = torch.load(tmp_path / "out" / "last.ckpt", weights_only=False)
    assert ck["global_step"] == 4 and ck["epoch"] == 1          # ceil(13 / 12) = 2 steps per epoch (7 samples per rank)


def test_two_ranks_gloo(tmp_path):
    # one process per "GPU" (gloo on CPU): shards of the epoch per rank, gradients averaged, files written by rank 0,
    # evaluation outputs of both ranks merged (main.py:259-268)
    import socket
    import torch.multiprocessing as mp
    argv = _toy(tmp_path) + ["--epochs", "1", "--do_train", "--do_test", "--overwrite", "--gpus", "2"]
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0)); port = s_.getsockname()[1]
    mp.spawn(_rank_main, args=(2, port, argv), nprocs=2, join=True)
    out = tmp_path / "out"
    ck = torch.load(out / "best.ckpt", weights_only=False)
    assert ck["global_step"] == 1            # 12 samples / (batch 6 x 2 ranks)
    pred = json.loads((out / "prediction_test_0.json").read_text())
    assert sorted(pred) == ["100", "101", "102", "103"]


@pytest.mark.gpu
def test_train_and_test_through_the_hip_ops(tmp_path, capsys):
    """the same cycle on a GPU through the HIP ops (widths of 256 so that every fused path is taken), bf16 autocast as
    `--precision bf16-mixed`, then fp16 autocast + GradScaler as the scripts' `--precision 16-mixed`"""
    argv = _toy(tmp_path)
    for name in ("enc.json", "dec.json"):
        cfg = json.loads((tmp_path / name).read_text())
        cfg.update(hidden_size=256, num_attention_heads=4, intermediate_size=512)
        (tmp_path / name).write_text(json.dumps(cfg))
    out = tmp_path / "out"
    for prec in ("bf16-mixed", "16-mixed"):
        assert M.main(argv + ["--epochs", "1", "--do_train", "--do_valid", "--do_test", "--overwrite", "--precision", prec]) == 0
        ck = torch.load(out / "best.ckpt", weights_only=False)
        assert ck["global_step"] == 2
        assert all(torch.isfinite(v).all() for v in ck["state_dict"].values() if v.is_floating_point())
        pred = json.loads((out / "prediction_test_0.json").read_text())
        assert sorted(pred) == ["100", "101", "102", "103"] and len(pred["100"]["prediction"]) == 3
    assert '"val_loss/1"' in capsys.readouterr().out


@pytest.mark.gpu
def test_train_with_the_step_replayed_from_a_hip_graph(tmp_path):
    """--hip_graph_step: three epochs of two full batches each; the first three steps run eagerly, the fourth is captured,
    the rest replay it; validation, checkpoints and the test step as always.  A child process, as a user starts the trainer:
    main() selects the runtime's graph path (train.prepare_graph_runtime) before anything touches the GPU"""
    import subprocess
    argv = _toy(tmp_path)
    for name in ("enc.json", "dec.json"):
        cfg = json.loads((tmp_path / name).read_text())
        cfg.update(hidden_size=256, num_attention_heads=4, intermediate_size=512, hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1)
        (tmp_path / name).write_text(json.dumps(cfg))
    out = tmp_path / "out"
    env = {k: v for k, v in os.environ.items() if k != "DEBUG_CLR_GRAPH_PACKET_CAPTURE"}
    r = subprocess.run([sys.executable, "-m", "textreact_amd.main"] + argv +
                       ["--epochs", "3", "--do_train", "--do_valid", "--do_test", "--overwrite", "--precision", "bf16-mixed",
                        "--hip_graph_step", "--print_freq", "4"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    ck = torch.load(out / "last.ckpt", weights_only=False)
    assert ck["global_step"] == 6 and ck["epoch"] == 2
    assert all(torch.isfinite(v).all() for v in ck["state_dict"].values() if v.is_floating_point())
    assert len(re.findall(r"train_loss ([0-9.]+)", r.stdout)) >= 1
    pred = json.loads((out / "prediction_test_0.json").read_text())
    assert sorted(pred) == ["100", "101", "102", "103"]
    # the checkpoint of a graphed run is interchangeable with an eager run's (and the reference's): a float learning
    # rate, capturable off -- and resumes either way: eagerly (epoch 3), then graphed again from that eager checkpoint
    # (epochs 4-6: enough steps for a capture on top of the restored AdamW state)
    grp = ck["optimizer_states"][0]["param_groups"][0]
    assert isinstance(grp["lr"], float) and grp.get("capturable", False) is False
    assert all(isinstance(v, float) for v in ck["lr_schedulers"][0]["_last_lr"] + ck["lr_schedulers"][0]["base_lrs"])
    for extra, epochs, steps in (([], 4, 8), (["--hip_graph_step"], 7, 14)):
        r = subprocess.run([sys.executable, "-m", "textreact_amd.main"] + argv +
                           ["--epochs", str(epochs), "--do_train", "--precision", "bf16-mixed", "--load_ckpt", "last.ckpt",
                            "--print_freq", "1"] + extra, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
        assert "Resumed from" in r.stdout
        ck = torch.load(out / "last.ckpt", weights_only=False)
        assert ck["global_step"] == steps and ck["epoch"] == epochs - 1
        assert all(torch.isfinite(v).all() for v in ck["state_dict"].values() if v.is_floating_point())
        grp = ck["optimizer_states"][0]["param_groups"][0]
        assert isinstance(grp["lr"], float) and grp.get("capturable", False) is False


def _toy_template(tmp_path, seed=0):
    """tensor files of the --template_based branch: ragged atom / bond template labels per sample"""
    g = torch.Generator().manual_seed(seed)
    enc = dict(vocab_size=60, hidden_size=64, num_hidden_layers=1, num_attention_heads=1, intermediate_size=64,
               max_position_embeddings=32, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    (tmp_path / "enc.json").write_text(json.dumps(enc))
    na_t, nb_t = 7, 5
    for name, m in (("train", 12), ("val", 5), ("test", 4)):
        ids = torch.randint(3, 60, (m, 14), generator=g)
        atoms, alab, blab, bonds, raw = [], [], [], [], []
        for i in range(m):
            n = 3 + int(torch.randint(0, 4, (1,), generator=g))
            atoms.append(torch.arange(1, 1 + n))
            alab.append(torch.randint(0, na_t, (n,), generator=g))
            bl = torch.full((n, n), -100, dtype=torch.long)
            bs = [(a, a + 1) for a in range(n - 1)]
            for a, b in bs:
                bl[a, b] = int(torch.randint(0, nb_t, (1,), generator=g))
            blab.append(bl); bonds.append([list(b) for b in bs]); raw.append([["a", 0, 1]])
        torch.save({"indices": list(range(200, 200 + m)), "input_ids": ids, "attention_mask": torch.ones_like(ids),
                    "atom_indices": atoms, "decoder_atom_template_labels": alab, "decoder_bond_template_labels": blab,
                    "bonds": bonds, "decoder_raw_template_labels": raw}, tmp_path / (name + ".pt"))
    return ["--task", "retro", "--template_based", "--encoder", "allenai/scibert_scivocab_uncased", "--arch_encoder", str(tmp_path / "enc.json"),
            "--tok_atom_templates", str(na_t), "--tok_bond_templates", str(nb_t), "--save_path", str(tmp_path / "out"),
            "--tensors_train", str(tmp_path / "train.pt"), "--tensors_valid", str(tmp_path / "val.pt"),
            "--tensors_test", str(tmp_path / "test.pt"), "--batch_size", "6", "--lr", "1e-3", "--test_batch_size", "2",
            "--val_metric", "val_acc", "--print_freq", "1"]


def test_template_based_branch(tmp_path, capsys, reference_ops):
    """scripts/train_RetroSyn_tb.sh's branch (main.py:112-123, 138-150, 201-216): encoder + template heads, monitored by the
    greedy edit accuracy, test step writing the ranked edits"""
    argv = _toy_template(tmp_path)
    out = tmp_path / "out"
    assert M.main(argv + ["--epochs", "2", "--do_train", "--do_valid", "--do_test", "--overwrite"]) == 0
    ck = torch.load(out / "best.ckpt", weights_only=False)
    assert ck["callbacks"]["ModelCheckpoint"]["monitor"] == "val_acc"
    assert any(k.startswith("model.template_head.") for k in ck["state_dict"]) and all(k.startswith("model.") for k in ck["state_dict"])
    pred = json.loads((out / "prediction_test_0.json").read_text())
    assert sorted(pred) == ["200", "201", "202", "203"]
    one = pred["200"]
    assert set(one) == {"prediction", "score", "raw_template_labels", "top1_template_match"}
    assert len(one["prediction"]) == len(one["score"]) > 0 and one["prediction"][0][0] in ("a", "b")
    assert one["score"] == sorted(one["score"], reverse=True)
    assert '"val_acc"' in capsys.readouterr().out


@pytest.mark.gpu
def test_template_based_branch_through_the_hip_ops(tmp_path):
    argv = _toy_template(tmp_path)
    cfg = json.loads((tmp_path / "enc.json").read_text())
    cfg.update(hidden_size=256, num_attention_heads=4, intermediate_size=512)
    (tmp_path / "enc.json").write_text(json.dumps(cfg))
    assert M.main(argv + ["--epochs", "1", "--do_train", "--do_test", "--overwrite", "--precision", "bf16-mixed"]) == 0
    pred = json.loads((tmp_path / "out" / "prediction_test_0.json").read_text())
    assert sorted(pred) == ["200", "201", "202", "203"] and pred["200"]["score"] == sorted(pred["200"]["score"], reverse=True)
