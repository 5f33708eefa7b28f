"""Trainer entry point: `python -m textreact_amd.main` takes the flags of the reference's main.py
(/root/reference main.py:26-97, all 62 of them, prefix abbreviations included: scripts/train_RCR.sh:37 passes
`--warmup 0.02` for `--warmup_ratio`) and drives predictor/train.py -- the step, the losses, AdamW + schedule, DDP,
`best.ckpt` / `last.ckpt` with the monitor logic of main.py:358-360, resume (:390-397), validate / test with
`prediction_{split}_{0,1}.json` (:243-245).  SURVEY.md section 8b, "Training CLI + ckpt".

Out of scope: tokenizers and datasets.  The data side of the reference (CSV -> tokenizer ->
collator, textreact/dataset.py) is replaced by PRE-TOKENISED tensor files, one per split, which is what its collator
hands the model anyway:

    --tensors_train / --tensors_valid / --tensors_test  FILE[,FILE2]
        torch.save'd dict: indices [N] (sample ids as they appear in the prediction files), input_ids, attention_mask,
        decoder_input_ids, decoder_attention_mask [N, L] / [N, T]; optional mlm_labels [N, trunc] (the masked tokens
        moved to the front, dataset.py:109-122).  A second file of a split is the gold-neighbour-removed variant
        (dataloader index 1, main.py:336-340).
    --arch_encoder JSON   encoder architecture when `--encoder` names a hub model (no network here); default: BERT-base
                            with the SciBERT vocabulary size
    --tok_vocab_size N, --tok_bos_id / --tok_eos_id / --tok_pad_id   what the decoder tokenizer would have said
    --tok_atom_templates N, --tok_bond_templates N   (--template_based) the sizes of the two template vocabularies
        (model.py:19: len(dec_tokenizer[0]), len(dec_tokenizer[1])); the tensor files of that branch carry, per sample and
        ragged, atom_indices [n_atoms], decoder_atom_template_labels [n_atoms], decoder_bond_template_labels
        [n_atoms, n_atoms] (-100 = no bond), bonds [[i, j], ...] and decoder_raw_template_labels -- what
        dataset.py's collator pads into a batch (labels with -100)

    --live_every N, --live_corpus FILE [--live_k K] [--live_retriever FILE]   on-the-fly retrieval (BASELINE.json
        configs[4]; textreact_amd/live.py): every N epochs the corpus passages (token ids in FILE) are re-embedded with the
        retriever's current weights, row-sharded over the ranks, and every training query's neighbours are searched
        again on the GPUs; each epoch the encoder inputs are assembled on the device from `query_ids` / `query_len` /
        `gold_passage` of the tensor files following the reference's dataset options (--num_neighbors,
        --use_gold_neighbor, --max_num_neighbors, --random_neighbor_ratio, --max_length, --mlm / --mlm_ratio); validation
        and test inputs likewise (second loader: gold text removed).  The retriever is the predictor's own encoder
        ([CLS] embedding) unless --live_retriever names a state dict (`lm_q.*` / `lm_p.*` as Tevatron saves its
        bi-encoder, or `encoder.*` for a tied one).  The reference itself reads static neighbor files (main.py:311-323).

The data flags of the reference are accepted and ignored with a note.  One process per GPU: launch with
`python -m torch.distributed.run --nproc-per-node G -m textreact_amd.main ...` (backend nccl = RCCL); `--gpus` is
checked against WORLD_SIZE.
"""
import argparse
import json
import math
import os
import sys

import torch


def get_parser():
    p = argparse.ArgumentParser(prog="textreact_amd.main")     # allow_abbrev is argparse's default
    # ---- main.py:28-36
    p.add_argument('--task', type=str, default='condition')
    p.add_argument('--do_train', action='store_true')
    p.add_argument('--do_valid', action='store_true')
    p.add_argument('--do_test', action='store_true')
    p.add_argument('--precision', type=str, default='32')
    p.add_argument('--seed', type=int, default=42)
    p.add_argument('--gpus', type=int, default=1)
    p.add_argument('--print_freq', type=int, default=200)
    p.add_argument('--debug', action='store_true')
    # ---- Model, main.py:38-45
    p.add_argument('--template_based', action='store_true')
    p.add_argument('--unattend_nonbonds', action='store_true')
    p.add_argument('--encoder', type=str, default=None)
    p.add_argument('--decoder', type=str, default=None)
    p.add_argument('--encoder_pretrained', action='store_true')
    p.add_argument('--decoder_pretrained', action='store_true')
    p.add_argument('--share_embedding', action='store_true')
    p.add_argument('--encoder_tokenizer', type=str, default='text')
    # ---- Data, main.py:47-72
    p.add_argument('--data_path', type=str, default=None)
    p.add_argument('--template_path', type=str, default=None)
    p.add_argument('--train_file', type=str, default=None)
    p.add_argument('--valid_file', type=str, default=None)
    p.add_argument('--test_file', type=str, default=None)
    p.add_argument('--vocab_file', type=str, default=None)
    p.add_argument('--corpus_file', type=str, default=None)
    p.add_argument('--train_label_corpus', action='store_true')
    p.add_argument('--cache_path', type=str, default=None)
    p.add_argument('--nn_path', type=str, default=None)
    p.add_argument('--train_nn_file', type=str, default=None)
    p.add_argument('--valid_nn_file', type=str, default=None)
    p.add_argument('--test_nn_file', type=str, default=None)
    p.add_argument('--max_length', type=int, default=128)
    p.add_argument('--max_dec_length', type=int, default=128)
    p.add_argument('--num_workers', type=int, default=8)
    p.add_argument('--shuffle_smiles', action='store_true')
    p.add_argument('--no_smiles', action='store_true')
    p.add_argument('--num_neighbors', type=int, default=-1)
    p.add_argument('--use_gold_neighbor', action='store_true')
    p.add_argument('--max_num_neighbors', type=int, default=10)
    p.add_argument('--random_neighbor_ratio', type=float, default=0.8)
    p.add_argument('--mlm', action='store_true')
    p.add_argument('--mlm_ratio', type=float, default=0.15)
    p.add_argument('--mlm_layer', type=str, default='linear')
    p.add_argument('--mlm_lambda', type=float, default=1)
    # ---- Training, main.py:74-89
    p.add_argument('--epochs', type=int, default=8)
    p.add_argument('--batch_size', type=int, default=256)
    p.add_argument('--lr', type=float, default=1e-4)
    p.add_argument('--weight_decay', type=float, default=0.01)
    p.add_argument('--max_grad_norm', type=float, default=5.)
    p.add_argument('--scheduler', type=str, choices=['cosine', 'constant'], default='cosine')
    p.add_argument('--warmup_ratio', type=float, default=0)
    p.add_argument('--gradient_accumulation_steps', type=int, default=1)
    p.add_argument('--load_ckpt', type=str, default='best.ckpt')
    p.add_argument('--eval_per_epoch', type=int, default=1)
    p.add_argument('--val_metric', type=str, default='val_acc')
    p.add_argument('--save_path', type=str, default='output/')
    p.add_argument('--overwrite', action='store_true')
    p.add_argument('--num_train_example', type=int, default=None)
    p.add_argument('--label_smoothing', type=float, default=0.0)
    # ---- Inference, main.py:91-94
    p.add_argument('--test_batch_size', type=int, default=64)
    p.add_argument('--num_beams', type=int, default=1)
    p.add_argument('--test_each_neighbor', action='store_true')
    p.add_argument('--test_num_neighbors', type=int, default=1)
    # ---- this repo: pre-tokenised inputs in place of the dataset / tokenizer stack (see the module docstring).  The
    # names are chosen so that no prefix that identifies a reference flag uniquely (argparse abbreviations, which the
    # reference's scripts use) becomes ambiguous: none of them shares its first three letters with a reference flag.
    p.add_argument('--tensors_train', type=str, default=None)
    p.add_argument('--tensors_valid', type=str, default=None)
    p.add_argument('--tensors_test', type=str, default=None)
    p.add_argument('--arch_encoder', type=str, default=None)
    p.add_argument('--tok_vocab_size', type=int, default=None, help="decoder vocabulary size (the decoder tokenizer's)")
    p.add_argument('--tok_bos_id', type=int, default=1)
    p.add_argument('--tok_eos_id', type=int, default=2)
    p.add_argument('--tok_pad_id', type=int, default=0)
    p.add_argument('--tok_atom_templates', type=int, default=None)
    p.add_argument('--tok_bond_templates', type=int, default=None)
    p.add_argument('--hip_graph_step', action='store_true',
                   help="replay the optimisation step from a HIP graph captured once per batch shape (predictor/train.py: "
                        "GraphedStep): single process, no fp16 loss scaling, no gradient accumulation")
    p.add_argument('--live_every', type=int, default=0, help="re-embed the corpus and re-retrieve every N epochs (0 = off)")
    p.add_argument('--live_corpus', type=str, default=None, help="pre-tokenised corpus passages (textreact_amd/live.py)")
    p.add_argument('--live_k', type=int, default=None, help="neighbours retrieved per query (default 2 x --max_num_neighbors)")
    p.add_argument('--live_retriever', type=str, default=None, help="state dict of a separate bi-encoder retriever")
    return p


def get_args(argv=None):
    return get_parser().parse_args(argv)


METRIC_TO_MODE = {'val_loss': 'min', 'val_acc': 'max'}      # textreact/utils.py:12-15


def _configs(args):
    from .predictor.model import Config
    if args.arch_encoder:
        enc = json.load(open(args.arch_encoder))
    elif args.encoder and os.path.isfile(args.encoder):
        enc = json.load(open(args.encoder))
    else:       # BERT-base with the SciBERT vocabulary (allenai/scibert_scivocab_uncased; the hub is unreachable here)
        enc = dict(vocab_size=31090)
    if args.template_based:
        dec = {}                                        # model.py:12: no decoder in this branch
    elif not args.decoder or not os.path.isfile(args.decoder):
        raise SystemExit("--decoder must be a decoder config JSON (the reference passes textreact/configs/bert_l6.json)")
    else:
        dec = json.load(open(args.decoder))
    if args.tok_vocab_size:
        dec["vocab_size"] = args.tok_vocab_size         # model.py:26: the decoder's vocabulary is the tokenizer's
    keep = ("vocab_size", "hidden_size", "num_hidden_layers", "num_attention_heads", "intermediate_size",
            "max_position_embeddings", "type_vocab_size", "layer_norm_eps", "hidden_dropout_prob",
            "attention_probs_dropout_prob", "pad_token_id")
    import inspect
    ok = set(inspect.signature(Config.__init__).parameters)
    enc = {k: v for k, v in enc.items() if k in keep and k in ok}
    dec = {k: v for k, v in dec.items() if k in keep and k in ok}
    return Config(**enc), (Config(is_decoder=True, **dec) if dec else None)


class TensorSplit:
    """One pre-tokenised split: what the reference's DataLoader + collator yield, batch by batch
    (`indices, batch_in, batch_out`, main.py:165)."""
    IN_KEYS = ("input_ids", "attention_mask", "decoder_input_ids", "decoder_attention_mask")
    PADDED = ("decoder_atom_template_labels", "decoder_bond_template_labels")     # per-sample tensors, padded with -100
    LISTS = ("atom_indices", "bonds", "decoder_raw_template_labels")              # stay per-sample lists in the batch

    def __init__(self, path, name):
        d = torch.load(path, map_location="cpu", weights_only=False)
        self.name = name
        self.indices = [int(i) for i in (d["indices"].tolist() if torch.is_tensor(d["indices"]) else d["indices"])]
        self.tensors = {k: d[k] for k in self.IN_KEYS if k in d}
        self.ragged = {k: d[k] for k in self.PADDED + self.LISTS if k in d}
        self.mlm_labels = d.get("mlm_labels")
        # on-the-fly retrieval: the query tokens alone, and the row of the sample's own passage in the corpus
        self.live = {k: d[k] for k in ("query_ids", "query_len", "gold_passage") if k in d}
        first = self.tensors.get("input_ids", self.live.get("query_ids"))
        assert first is not None and len(self.indices) == first.shape[0], path
        assert all(len(v) == len(self.indices) for v in self.ragged.values()), path

    def __len__(self):
        return len(self.indices)

    def collate(self, sel, device):
        """the batch of samples `sel` (a list of positions): (batch_in, batch_out) on `device`"""
        st = torch.tensor(sel, dtype=torch.long)
        batch_in = {k: v[st].to(device) for k, v in self.tensors.items()}
        for k in self.PADDED:
            if k in self.ragged:
                items = [torch.as_tensor(self.ragged[k][i]) for i in sel]
                shape = [max(t.shape[dim] for t in items) for dim in range(items[0].dim())]
                out = torch.full([len(items)] + shape, -100, dtype=torch.long)
                for j, t in enumerate(items):
                    out[(j,) + tuple(slice(0, n) for n in t.shape)] = t
                batch_in[k] = out.to(device)
        for k in self.LISTS:
            if k in self.ragged:
                vals = [self.ragged[k][i] for i in sel]
                if k == "atom_indices":
                    vals = [torch.as_tensor(v, dtype=torch.long).to(device) for v in vals]
                elif k == "bonds":
                    vals = [[tuple(b) for b in v] for v in vals]                  # combined_edit tests `idx in bonds`
                else:
                    vals = [[tuple(t) for t in v] for v in vals]
                batch_in[k] = vals
        batch_out = {"mlm_labels": self.mlm_labels[st].to(device)} if self.mlm_labels is not None else {}
        return batch_in, batch_out

    def with_inputs(self, input_ids, attention_mask):
        """this split with other encoder inputs (assembled by the on-the-fly retrieval)"""
        import copy
        other = copy.copy(self)
        other.tensors = dict(self.tensors, input_ids=input_ids, attention_mask=attention_mask)
        return other

    def batches(self, batch_size, rank=0, world=1, device="cpu", limit=None):
        n = len(self) if limit is None else min(limit, len(self))
        order = list(range(rank, n, world))             # DistributedSampler(shuffle=False) of Lightning's eval loaders
        for b0 in range(0, len(order), batch_size):
            sel = order[b0:b0 + batch_size]
            batch_in, batch_out = self.collate(sel, device)
            yield [self.indices[i] for i in sel], batch_in, batch_out


def epoch_shard(n, seed, epoch, rank, world):
    """the sample positions rank `rank` trains on in `epoch`: torch's DistributedSampler(shuffle=True), which Lightning
    puts under the reference's train loader (main.py:372) -- one permutation per epoch from a generator seeded with
    seed + epoch (so a resumed run continues the sequence of an uninterrupted one), padded by wrapping around to a
    multiple of the world size, dealt rank-strided.  Every rank gets ceil(n / world) samples: the same number of
    micro-batches and optimiser steps everywhere, which the gradient all-reduce of the step relies on."""
    g = torch.Generator().manual_seed(seed + epoch)
    perm = torch.randperm(n, generator=g).tolist()
    total = -(-n // world) * world
    while len(perm) < total:
        perm += perm[:total - len(perm)]
    return perm[rank:total:world]


def _load_splits(spec, name):
    return [TensorSplit(f, name) for f in spec.split(",")] if spec else []


def load_live_retriever(path, enc_cfg, device):
    """--live_retriever: (query encoder, passage encoder) from a bi-encoder state dict -- Tevatron's `lm_q.*` / `lm_p.*`
    (README.md:44-47), or one `encoder.*` tree for both.  strict=False tolerates what such a file may lack or carry (the
    pooler Tevatron does not use; the position_ids buffer of transformers 4.27.3) -- never the encoder weights themselves:
    a file with other key names would otherwise leave a randomly initialised retriever behind, silently."""
    from . import dense
    sd = torch.load(path, map_location="cpu", weights_only=False)
    sd = sd.get("state_dict", sd)
    encs = []
    for prefix in ("lm_q.", "lm_p."):
        part = {"encoder." + k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
        part = part or {k: v for k, v in sd.items() if k.startswith("encoder.")}
        e = dense.DenseEncoder(enc_cfg).to(device).eval()
        missing, unexpected = e.load_state_dict(part, strict=False)
        missing = [k for k in missing if ".pooler." not in k]
        unexpected = [k for k in unexpected if not k.endswith("position_ids") and ".pooler." not in k]
        if missing or unexpected or not part:
            raise SystemExit("--live_retriever %s: the %s encoder did not load (%d weights missing, e.g. %s; %d unexpected, e.g. %s): "
                             "expected Tevatron's lm_q.* / lm_p.* or encoder.* names" % (
                                 path, prefix.rstrip("."), len(missing), missing[:2], len(unexpected), unexpected[:2]))
        encs.append(e)
    return encs


class LiveData:
    """--live_every: the retriever, the neighbour ids of every split, and the per-epoch assembly of encoder inputs on the
    device (textreact_amd/live.py restates the reference's dataset logic on tensors)."""

    def __init__(self, args, module, enc_cfg, device, rank, world):
        from . import dense, live
        self.args, self.live, self.device, self.rank, self.world = args, live, device, rank, world
        if not args.live_corpus:
            raise SystemExit("--live_every needs --live_corpus (pre-tokenised passages, see textreact_amd/live.py)")
        self.corpus = live.LiveCorpus(args.live_corpus, device)
        own = dense.DenseEncoder.wrap(module.model.encoder)       # the predictor's own encoder, current weights
        self.q_enc = self.p_enc = own
        if args.live_retriever:
            self.q_enc, self.p_enc = load_live_retriever(args.live_retriever, enc_cfg, device)
        self.k = args.live_k or 2 * max(args.max_num_neighbors, args.num_neighbors)
        self.retriever = live.LiveRetriever(self.corpus, rank, world, batch_size=max(64, args.test_batch_size))
        self.nn = {}                                                # split name -> [N, k] neighbour rows, on the device
        self.gen = torch.Generator().manual_seed(args.seed + 7919 * (rank + 1))

    def refresh(self, splits):
        """re-embed the passages with the retriever's current weights and search every query of `splits` again"""
        self.retriever.refresh_index(self.p_enc)
        for ds in splits:
            assert "query_ids" in ds.live, "--live_every: %s has no query_ids / query_len" % ds.name
            self.nn[id(ds)] = self.retriever.neighbors(self.q_enc, ds.live["query_ids"], ds.live["query_len"], self.k)

    def inputs(self, ds, rows, train, skip_gold=False):
        """encoder inputs of the samples `rows` of `ds` for one pass: (input_ids, attention_mask, lengths, position_ids,
        mlm_labels); the last two only for a training pass with --mlm.  In blocks of 32,768 samples."""
        a, live, dev = self.args, self.live, self.device
        rows_t = torch.as_tensor(rows, dtype=torch.long)
        outs = []
        for b0 in range(0, len(rows), 32768):
            r = rows_t[b0:b0 + 32768]
            gold = ds.live["gold_passage"][r].to(dev) if "gold_passage" in ds.live else None
            sel = live.select_neighbors(self.nn[id(ds)][r.to(dev)], gold, self.corpus, train, a.use_gold_neighbor, a.max_num_neighbors,
                                        a.num_neighbors, a.random_neighbor_ratio, skip_gold, self.gen)
            ids, mask, lens = live.assemble_inputs(ds.live["query_ids"][r], ds.live["query_len"][r], sel, self.corpus, a.max_length,
                                                   with_neighbors=a.num_neighbors > 0)
            pos = labels = None
            if train and a.mlm:
                ids, pos, labels = live.apply_mlm(ids, lens, a.mlm_ratio, self.corpus.mask_id, self.gen)
            outs.append((ids, mask, lens, pos, labels))
        width = max(o[0].shape[1] for o in outs)
        trunc = max((o[4].shape[1] for o in outs if o[4] is not None), default=0)

        def cat(i, fill, w):
            if outs[0][i] is None:
                return None
            return torch.cat([torch.nn.functional.pad(o[i], (0, w - o[i].shape[1]), value=fill) for o in outs])
        return (cat(0, self.corpus.pad_id, width), cat(1, 0, width), torch.cat([o[2] for o in outs]), cat(3, 0, width), cat(4, -100, trunc))

    def eval_views(self, sets):
        """the reference's two evaluation loaders (main.py:336-340): neighbours as retrieved, and with the gold text removed"""
        out = []
        for ds in sets[:1]:
            for skip in (False, True):
                ids, mask, _, _, _ = self.inputs(ds, list(range(len(ds))), train=False, skip_gold=skip)
                out.append(ds.with_inputs(ids, mask))
        return out


def _autocast(args, device):
    prec = str(args.precision)
    if prec.startswith("16"):
        return torch.autocast(device_type=device.type, dtype=torch.float16 if device.type == "cuda" else torch.bfloat16)
    if prec.startswith("bf16"):
        return torch.autocast(device_type=device.type, dtype=torch.bfloat16)
    import contextlib
    return contextlib.nullcontext()


def main(argv=None):
    args = get_args(argv)
    from .predictor import ops, train as T
    import torch.distributed as dist

    if args.hip_graph_step:
        T.prepare_graph_runtime()     # an environment switch the HIP runtime reads when it starts: before anything touches the GPU
    from . import _dist
    # `--gpus N` (main.py:75, `devices=args.gpus` at :372-374) is what runs or nothing does: a launcher that set another WORLD_SIZE
    # is refused (exit code 2) before anything touches the GPU, as bench.py refuses it
    _dist.refuse_mismatch(args.gpus, "main.py")
    cuda = torch.cuda.is_available()
    device = torch.device("cuda", _dist.device_ordinal()) if cuda else torch.device("cpu")
    ops.require_device(device)        # the attention / add+LayerNorm ops exist as HIP kernels only: no GPU, no trainer
    rank, world, device = _dist.setup()      # set_device, then init_process_group("nccl", device_id=...): textreact_amd/_dist.py
    local_rank = _dist.device_ordinal()
    if args.gpus != world and rank == 0:     # (no launcher: WORLD_SIZE unset, one process, one GPU)
        print("note: --gpus %d without a launcher runs on ONE GPU; start N ranks with `python -m torch.distributed.run "
              "--nproc-per-node %d -m textreact_amd.main ...`" % (args.gpus, args.gpus), file=sys.stderr)
    ignored = [k for k in ("data_path", "train_file", "valid_file", "test_file", "vocab_file", "corpus_file", "nn_path")
               if getattr(args, k)]
    if ignored and rank == 0:
        print("note: dataset / tokenizer flags are accepted and not used here (%s): inputs are the --tensors_* files"
              % ", ".join("--" + k for k in ignored), file=sys.stderr)
    torch.manual_seed(args.seed)                      # pl.seed_everything (main.py:351)
    enc_cfg, dec_cfg = _configs(args)
    if args.template_based:                           # model.py:11-19: encoder + atom / bond template heads, no decoder
        if not (args.tok_atom_templates and args.tok_bond_templates):
            raise SystemExit("--template_based needs --tok_atom_templates and --tok_bond_templates (the template vocabularies' sizes)")
        from .predictor.template import TemplatePredictor
        module = TemplatePredictor(enc_cfg, args.tok_atom_templates, args.tok_bond_templates).to(device)
    else:
        module = T.Predictor(enc_cfg, dec_cfg, mlm=args.mlm, mlm_layer=args.mlm_layer, mlm_lambda=args.mlm_lambda,
                             pad_token_id=args.tok_pad_id).to(device)
    train_sets = _load_splits(args.tensors_train, "train") if args.do_train else []
    val_sets = _load_splits(args.tensors_valid, "val") if (args.do_train or args.do_valid) else []
    test_sets = _load_splits(args.tensors_test, "test") if args.do_test else []
    os.makedirs(args.save_path, exist_ok=True)
    best_path, last_path = os.path.join(args.save_path, "best.ckpt"), os.path.join(args.save_path, "last.ckpt")
    live = None
    if args.live_every > 0:
        if args.template_based:
            raise SystemExit("--live_every drives the template-free predictor (train_RetroSyn_tf.sh / train_RCR.sh)")
        live = LiveData(args, module, enc_cfg, device, rank, world)

    def validate(mod):
        """main.py:177-196: per-sample scores of every dataloader, gathered, averaged; the first one is the monitor"""
        mod.eval()
        out = {}
        for di, ds in enumerate(live.eval_views(val_sets) if live else val_sets):
            scores = {}
            for indices, batch_in, _ in ds.batches(args.batch_size, rank, world, device):
                with _autocast(args, device):
                    scores.update(mod.validation_step(indices, batch_in, args.val_metric))
            scores = T.gather_outputs(scores)
            name = args.val_metric if di == 0 else "%s/%d" % (args.val_metric, di)
            out[name] = float(sum(scores.values()) / max(len(scores), 1))
        return out

    best_model_path = os.path.join(args.save_path, args.load_ckpt)
    if args.do_train:
        assert train_sets, "--do_train needs --tensors_train"
        train = train_sets[0]
        n_train = len(train) if args.num_train_example is None else min(args.num_train_example, len(train))
        steps_per_epoch = math.ceil(n_train / (args.batch_size * world * args.gradient_accumulation_steps))
        num_training_steps = steps_per_epoch * args.epochs                                  # main.py:383-384
        if rank == 0:
            print("Num training steps: %d" % num_training_steps)
        graphed = bool(args.hip_graph_step)
        # --live_every slices every batch to its own width (and its own mlm_labels width): hundreds of distinct shapes, each
        # of which GraphedStep would capture into a graph with a private activation pool -- the run would end out of memory
        if graphed and (world > 1 or str(args.precision).startswith("16") or args.gradient_accumulation_steps != 1 or args.template_based
                        or args.live_every > 0):
            if rank == 0:
                print("note: --hip_graph_step needs one process, bf16 / fp32 precision, no gradient accumulation, the "
                      "template-free model and static inputs (no --live_every): running the step eagerly", file=sys.stderr)
            graphed = False
        opt, sched = T.configure_optimizer(module, args.lr, args.weight_decay, num_training_steps, args.warmup_ratio,
                                           scheduler=args.scheduler, capturable=graphed)
        gstep = None
        if graphed:
            ac = torch.bfloat16 if str(args.precision).startswith("bf16") else None
            gstep = T.GraphedStep(module, opt, max_grad_norm=args.max_grad_norm, autocast_dtype=ac)
        start_epoch, global_step = 0, 0
        mode = METRIC_TO_MODE[args.val_metric]
        best = None
        if args.overwrite:                                                                 # main.py:386-388
            if rank == 0:
                T.clear_checkpoints(args.save_path)
        else:                                                                              # main.py:389-391
            ckpt_path = os.path.join(args.save_path, args.load_ckpt)
            if os.path.isfile(ckpt_path):
                ck, _, _ = T.load_checkpoint(ckpt_path, module, opt, sched)
                start_epoch, global_step = int(ck["epoch"]) + 1, int(ck["global_step"])
                best = (ck.get("callbacks") or {}).get("ModelCheckpoint", {}).get("best_model_score")
                if rank == 0:
                    print("Resumed from %s (epoch %d, step %d)" % (ckpt_path, start_epoch, global_step))
        if world > 1:
            dist.barrier()
        # data parallelism (main.py:372, DDPStrategy): every rank steps its shard of the shuffled epoch and the gradients
        # are averaged by ONE flat all-reduce per optimiser step (RCCL over xGMI; a 770 MB fp32 payload is bandwidth-
        # bound, so one collective instead of DDP's 25 MB buckets costs nothing and needs no forward() wrapper)
        scaler = torch.amp.GradScaler(enabled=cuda and str(args.precision).startswith("16"))
        for epoch in range(start_epoch, args.epochs):
            mine = epoch_shard(n_train, args.seed, epoch, rank, world)
            if live:
                if epoch == start_epoch or (epoch - start_epoch) % args.live_every == 0:
                    live.refresh([train] + val_sets[:1])          # passages re-embedded, every query searched again
                    if rank == 0:
                        print("epoch %d: neighbours refreshed (%d passages, k = %d)" % (epoch, len(live.corpus), live.k))
                ep_ids, ep_mask, ep_len, ep_pos, ep_mlm = live.inputs(train, mine, train=True)
                ep_len_h = ep_len.cpu()
                ep_cnt_h = (ep_mlm != -100).sum(dim=1).cpu() if ep_mlm is not None else None
            module.train()
            micro = 0
            opt.zero_grad(set_to_none=True)
            for b0 in range(0, len(mine), args.batch_size):
                batch_in, batch_out = train.collate(mine[b0:b0 + args.batch_size], device)
                if live:        # this epoch's encoder inputs, assembled on the device from the refreshed neighbours
                    sl = slice(b0, b0 + args.batch_size)
                    w = int(ep_len_h[sl].max())
                    batch_in["input_ids"], batch_in["attention_mask"] = ep_ids[sl, :w], ep_mask[sl, :w]
                    if ep_mlm is not None:
                        batch_in["position_ids"] = ep_pos[sl, :w]
                        batch_out = {"mlm_labels": ep_mlm[sl, :max(1, int(ep_cnt_h[sl].max()))]}
                if gstep is not None:       # forward, backward, clipping and the update: one graph replay
                    total, logs = gstep.step(batch_in, batch_out)
                    sched.step()
                    global_step += 1
                    if rank == 0 and global_step % max(1, args.print_freq) == 0:
                        print("epoch %d step %d train_loss %.4f lr %.3g" % (epoch, global_step, float(logs["train_loss"]),
                                                                            float(sched.get_last_lr()[0])))
                    continue
                with _autocast(args, device):
                    total, logs = module.training_step(batch_in, batch_out)
                ops.backward(scaler.scale(total / args.gradient_accumulation_steps))      # the Linear layers' weight gradients: one grouped launch
                micro += 1
                if micro % args.gradient_accumulation_steps == 0 or b0 + args.batch_size >= len(mine):
                    if world > 1:
                        for p_ in module.parameters():      # a parameter unused on this rank still takes part (find_unused_parameters)
                            if p_.grad is None and p_.requires_grad:
                                p_.grad = torch.zeros_like(p_)
                        grads = [p.grad for p in module.parameters() if p.grad is not None]
                        flat = torch.cat([gr.reshape(-1) for gr in grads])
                        if flat.is_cuda and dist.get_backend() == "gloo":      # one-GPU rehearsal (TRX_DIST_BACKEND): through the host
                            host = flat.cpu(); dist.all_reduce(host); flat.copy_(host)
                        else:
                            dist.all_reduce(flat)
                        flat /= world
                        off = 0
                        for gr in grads:
                            gr.copy_(flat[off:off + gr.numel()].view_as(gr)); off += gr.numel()
                    scaler.unscale_(opt)
                    torch.nn.utils.clip_grad_norm_(module.parameters(), args.max_grad_norm)      # gradient_clip_val
                    scaler.step(opt); scaler.update(); sched.step()
                    T.mark_parameters_updated(module)
                    opt.zero_grad(set_to_none=True)
                    global_step += 1
                    if rank == 0 and global_step % max(1, args.print_freq) == 0:
                        print("epoch %d step %d train_loss %.4f lr %.3g" % (epoch, global_step, float(logs["train_loss"]),
                                                                            sched.get_last_lr()[0]))
            if (epoch + 1) % args.eval_per_epoch == 0 and val_sets:                       # check_val_every_n_epoch
                metrics = validate(module)
                score = metrics[args.val_metric]
                improved = best is None or (score < best if mode == "min" else score > best)
                if rank == 0:
                    print("epoch %d %s" % (epoch, json.dumps(metrics)))
                    if improved:                                                         # save_top_k=1, filename='best'
                        T.save_checkpoint(best_path, module, opt, sched, epoch, global_step, monitor=args.val_metric)
                        _stamp_best(best_path, score)
                    T.save_checkpoint(last_path, module, opt, sched, epoch, global_step, monitor=args.val_metric)  # save_last
                    _stamp_best(last_path, score if improved else best)
                if improved:
                    best = score
            elif rank == 0:
                T.save_checkpoint(last_path, module, opt, sched, epoch, global_step, monitor=args.val_metric)
            if world > 1:
                dist.barrier()
        if gstep is not None:
            gstep.close()
        best_model_path = best_path if os.path.isfile(best_path) else last_path

    if args.do_valid or args.do_test:
        if rank == 0:
            print("Load model checkpoint:", best_model_path)
        T.load_checkpoint(best_model_path, module, strict=False)                          # main.py:404
    if live and (args.do_valid or args.do_test):      # the checkpoint just loaded is the retriever now
        live.refresh((val_sets[:1] if args.do_valid else []) + (test_sets[:1] if args.do_test else []))
    if args.do_valid and val_sets:
        metrics = validate(module)
        if rank == 0:
            print(json.dumps(metrics))
    if args.do_test:
        module.eval()
        for di, ds in enumerate(live.eval_views(test_sets) if live else test_sets):
            outputs = {}
            for indices, batch_in, _ in ds.batches(args.test_batch_size, rank, world, device):
                with _autocast(args, device), torch.no_grad():
                    if args.template_based:
                        outputs.update(module.test_step(indices, batch_in))               # main.py:201-216, top 500 edits
                    else:
                        outputs.update(T.test_step(module, indices, batch_in, args.num_beams, args.max_dec_length,
                                                   args.tok_bos_id, args.tok_eos_id, args.tok_pad_id))
            outputs = T.gather_outputs(outputs)
            if args.test_each_neighbor:                                                  # main.py:239-240
                outputs = T.merge_predictions_per_neighbor(outputs, args.test_num_neighbors)
            if rank == 0:       # main.py:243-245; json keys become strings exactly as json.dump of the reference's dict does
                with open(os.path.join(args.save_path, "prediction_%s_%d.json" % (ds.name, di)), "w") as f:
                    json.dump(outputs, f)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def _stamp_best(path, score):
    """record the monitored score in the checkpoint's ModelCheckpoint state (what a resume compares against)"""
    ck = torch.load(path, map_location="cpu", weights_only=False)
    ck["callbacks"]["ModelCheckpoint"]["best_model_score"] = None if score is None else float(score)
    torch.save(ck, path)


if __name__ == "__main__":
    sys.exit(main())
