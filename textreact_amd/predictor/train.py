"""Training / evaluation step of the predictor and its checkpoint layout (SURVEY.md section 8a rows
P3, P4; section 5.4).  Mirrors main.py's ReactionConditionRecommender for the template-free
models: same attribute names (`model`, `mlm_head` -> state-dict prefixes `model.` / `mlm_head.`,
main.py:106-108), same losses (main.py:112-134, :158-162), same step (:164-175), same gather of
evaluation outputs (:259-268), same optimiser (:270-276); Lightning itself is not needed.

Data parallelism is plain `torch.nn.parallel.DistributedDataParallel(find_unused_parameters=True)`
(main.py:372) over `torch.distributed` -- backend "nccl" is RCCL on ROCm -- one process per GPU.
"""
import collections
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .model import Config, LayerNormParams, TextReactModel


class MLMTransform(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.dense = nn.Linear(cfg.hidden_size, cfg.hidden_size)
        self.LayerNorm = LayerNormParams(cfg.hidden_size)


class MLMHead(nn.Module):
    """BertLMPredictionHead (model.py:40-47, --mlm_layer mlp): transform.{dense,LayerNorm} -> decoder"""

    def __init__(self, cfg):
        super().__init__()
        self.transform = MLMTransform(cfg)
        self.decoder = nn.Linear(cfg.hidden_size, cfg.vocab_size)
        self.bias = nn.Parameter(torch.zeros(cfg.vocab_size))
        self.decoder.bias = self.bias
        self.eps = cfg.layer_norm_eps

    def forward(self, h):
        x = F.gelu(self.transform.dense(h))
        x = ops.add_layernorm(x, None, self.transform.LayerNorm.weight, self.transform.LayerNorm.bias, self.eps)
        return self.decoder(x)


class Predictor(nn.Module):
    """the LightningModule's parameter tree without Lightning"""

    def __init__(self, enc_cfg, dec_cfg, mlm=False, mlm_layer="mlp", mlm_lambda=1.0, pad_token_id=0):
        super().__init__()
        self.model = TextReactModel(enc_cfg, dec_cfg)
        if mlm:
            self.mlm_head = MLMHead(enc_cfg) if mlm_layer == "mlp" else nn.Linear(enc_cfg.hidden_size, enc_cfg.vocab_size)
        self.mlm, self.mlm_lambda, self.pad = mlm, mlm_lambda, pad_token_id

    # main.py:129-134
    def compute_loss(self, logits, batch_in, reduction="mean"):
        b, _, vocab = logits.shape
        labels = batch_in["decoder_input_ids"][:, 1:]
        loss = F.cross_entropy(logits[:, :-1].reshape(-1, vocab), labels.reshape(-1), ignore_index=self.pad,
                               reduction=reduction)
        return loss.view(b, -1).mean(dim=1) if reduction == "none" else loss

    # main.py:151-156: greedy-search accuracy (per sample with reduction="none", as validation_step uses it)
    def compute_acc(self, logits, batch_in, reduction="mean"):
        preds = logits.argmax(dim=-1)[:, :-1]
        labels = batch_in["decoder_input_ids"][:, 1:]
        acc = torch.logical_or(preds.eq(labels), labels.eq(self.pad)).all(dim=-1).float()
        return acc.mean() if reduction == "mean" else acc

    # main.py:158-162: masked tokens were moved to the front, so only the first trunc_len positions count
    def compute_mlm_loss(self, encoder_last_hidden_state, labels):
        b, trunc = labels.shape
        logits = self.mlm_head(encoder_last_hidden_state[:, :trunc].contiguous())
        return F.cross_entropy(logits.view(b * trunc, -1), labels.reshape(-1))

    # main.py:164-175
    def training_step(self, batch_in, batch_out=None):
        logits, enc = self.model(**batch_in)
        loss = self.compute_loss(logits, batch_in)
        logs = {"train_loss": loss.detach()}
        total = loss
        if self.mlm:
            mlm_loss = self.compute_mlm_loss(enc, batch_out["mlm_labels"])
            total = total + mlm_loss * self.mlm_lambda
            logs.update(mlm_loss=mlm_loss.detach(), total_loss=total.detach())
        return total, logs

    # main.py:177-188: per-sample scores of the validation step
    @torch.no_grad()
    def validation_step(self, indices, batch_in, val_metric="val_loss"):
        logits, _ = self.model(**batch_in)
        if val_metric == "val_loss":
            scores = self.compute_loss(logits, batch_in, reduction="none")
        elif val_metric == "val_acc":
            scores = self.compute_acc(logits, batch_in, reduction="none")
        else:
            raise ValueError(val_metric)       # main.py:185
        return {int(i): float(s) for i, s in zip(indices, scores)}


# main.py:198-233 (template-free branch): beam search, num_return_sequences = num_beams
def test_step(predictor, indices, batch_in, num_beams, max_dec_length, bos_token_id, eos_token_id, pad_token_id=0,
              decode=None, reference_scores=False):
    """{idx: {'prediction': [...num_beams items...], 'score': [...]}} as test_step stores it.  `decode` maps a
    [n, T] tensor of token ids to n strings (the reference's dec_tokenizer.batch_decode(...,
    skip_special_tokens=True)); without it the predictions are the token-id lists with the special tokens
    removed, which is what that call strips.

    `score`: the beam scores (sequences_scores).  The reference means to write them too, but tests
    `'sequences_scores' in predictions` on the LIST OF DECODED STRINGS (main.py:228-231), which is never true, so its
    files carry zeros; `reference_scores=True` reproduces that byte for byte."""
    from .generate import generate
    seqs, scores = generate(predictor.model, batch_in["input_ids"], batch_in.get("attention_mask"), num_beams=num_beams,
                            num_return_sequences=num_beams, max_length=max_dec_length, length_penalty=0,
                            bos_token_id=bos_token_id, eos_token_id=eos_token_id, pad_token_id=pad_token_id)
    if decode is not None:
        preds = decode(seqs)
    else:
        special = {bos_token_id, eos_token_id, pad_token_id}
        preds = [[int(t) for t in row if int(t) not in special] for row in seqs.cpu().numpy()]
    score_list = scores.tolist() if (scores is not None and not reference_scores) else [0] * len(preds)
    return {int(idx): {"prediction": preds[i * num_beams:(i + 1) * num_beams],
                       "score": score_list[i * num_beams:(i + 1) * num_beams]} for i, idx in enumerate(indices)}


def gather_outputs(outputs, group=None):
    """main.py:259-268: merge the per-rank {idx: value} dicts on every rank"""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return outputs
    gathered = [None] * dist.get_world_size(group)
    dist.all_gather_object(gathered, outputs, group=group)
    merged = {}
    for o in gathered:
        merged.update(o)
    return merged


def merge_predictions_per_neighbor(outputs, num_neighbors):
    """`--test_each_neighbor` (main.py:239-240, utils.py:55-64): the test set holds every reaction num_neighbors times,
    sample i = reaction i // num_neighbors with its (i % num_neighbors)-th neighbour alone; the per-sample outputs of a
    reaction are concatenated, key by key, in neighbour order"""
    merged = {}
    for i in sorted(outputs):
        tgt = merged.setdefault(i // num_neighbors, {})
        for key, val in outputs[i].items():
            tgt[key] = (tgt[key] + val) if key in tgt else (list(val) if isinstance(val, (list, tuple)) else val)
    return merged


def configure_optimizer(module, lr, weight_decay, num_training_steps, warmup_ratio, scheduler="linear", capturable=False):
    """main.py:270-276: AdamW + warm-up/decay schedule stepped per optimiser step.  capturable=True (GraphedStep): the
    step counter and the learning rate live in device tensors, so the update can sit inside a captured HIP graph and
    the schedule still moves the rate between replays."""
    params = list(module.parameters())
    # same update rule; `fused` runs it as one multi-tensor kernel when the parameters live on the GPU
    fused = bool(params) and all(p.is_cuda for p in params)
    if capturable and fused:
        opt = torch.optim.AdamW(params, lr=torch.tensor(float(lr), device=params[0].device), weight_decay=weight_decay, fused=True,
                                capturable=True)
    else:
        opt = torch.optim.AdamW(params, lr=lr, weight_decay=weight_decay, fused=fused)
    warm = int(num_training_steps * warmup_ratio)

    def lr_lambda(step):
        if scheduler == "constant":     # transformers' "constant" has no warm-up
            return 1.0
        if step < warm:
            return float(step) / float(max(1, warm))
        if scheduler == "constant":
            return 1.0
        if scheduler == "cosine":       # transformers get_cosine_schedule_with_warmup, num_cycles = 0.5 (main.py:275)
            import math
            progress = float(step - warm) / float(max(1, num_training_steps - warm))
            return max(0.0, 0.5 * (1.0 + math.cos(math.pi * progress)))
        return max(0.0, float(num_training_steps - step) / float(max(1, num_training_steps - warm)))
    return opt, torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda)


# ROCm 7.0 on MI355X: a captured step that holds the backward is replayed wrongly by the runtime's pre-built-packet path
# ("graph packet capture", the default): after the host has waited on the stream once, replays queued behind one another
# end in HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION.  The host pattern alone decides it (tools/graph_step_probe.py:
# ".....t...." faults, "....." never does); no copy, allocation or host pointer is captured (rocprofv3 --hip-runtime-trace:
# 834 kernel launches, 10 memsets, nothing else), and with the packets built at launch time every pattern -- 59 replays
# under random waits -- is clean and the replay costs the same 25.4 ms.  The flag is read when the runtime starts.
GRAPH_RUNTIME_ENV = ("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")


def prepare_graph_runtime():
    """call before the first HIP call of the process (main.py does, on --hip_graph_step): selects the runtime's
    launch-time packet path for HIP graphs; raises when it is too late to choose"""
    name, want = GRAPH_RUNTIME_ENV
    have = os.environ.get(name)
    if have == want:
        return
    if torch.cuda.is_initialized():
        raise ops.TrxNNError("GraphedStep needs %s=%s in the environment BEFORE the process first touches the GPU (it is %r): the "
                             "default graph path of this ROCm faults on replays of a captured backward, see train.GRAPH_RUNTIME_ENV"
                             % (name, want, have))
    os.environ[name] = want


class GraphedStep:
    """One optimisation step -- the dropout seed's bump, forward, backward, gradient clipping, the AdamW update -- captured
    ONCE per batch shape into a HIP graph and replayed.

    What it buys: the step is ~830 kernel launches; at the scripts' per-GPU shapes (B 32 x L 512) their GPU time is the
    step -- 25.4 ms replayed against 25.9 ms eager at T = 160 (bench_predictor.py): the GPU is the limit
    there and the graph only frees the host.  It pays on small batches, where the launches are not hidden (the last,
    ragged batches of an epoch are not captured: see below).  What made the step capturable: dropout seeds
    that live on the device (ops.set_seed_device: a launch's `seed` only numbers its site, the kernels mix it with a
    device counter this step bumps), the attention backward's scratch as a caller's tensor, AdamW with a device-side
    step count and learning rate (configure_optimizer(capturable=True)), no host read inside the step, and the runtime
    switch GRAPH_RUNTIME_ENV (prepare_graph_runtime).

    The first `warmup` steps of a shape run eagerly (same code, same device seeds: real steps), the next one is
    captured, later ones replay it.  A shape seen fewer times than that (the last, shorter batch of an epoch) simply
    stays eager.  Single process only: with several ranks the gradient all-reduce would have to be captured too, and
    RCCL capture could not be tested here (no multi-GPU node)."""

    def __init__(self, module, optimizer, max_grad_norm=None, autocast_dtype=None, warmup=3):
        from . import ops as _ops
        self.module, self.opt, self.max_grad_norm, self.autocast_dtype, self.warmup = module, optimizer, max_grad_norm, autocast_dtype, warmup
        dev = next(module.parameters()).device
        if dev.type != "cuda":
            raise _ops.TrxNNError("GraphedStep needs the module on a GPU")
        if not all(g.get("capturable", False) for g in optimizer.param_groups):
            raise ValueError("GraphedStep: build the optimizer with configure_optimizer(..., capturable=True)")
        if os.environ.get(GRAPH_RUNTIME_ENV[0]) != GRAPH_RUNTIME_ENV[1]:
            prepare_graph_runtime()           # raises: the module is on the GPU, the runtime has started
        self.seed = torch.zeros(1, dtype=torch.int64, device=dev)
        self.seed.fill_(int(torch.randint(0, 2 ** 40, (1,)).item()))       # torch.manual_seed still fixes the run
        _ops.set_seed_device(self.seed)
        self._ops = _ops
        self.shapes = {}          # shape key -> {"n": eager steps done, "graph", "inputs", "outputs"}
        self.replays = 0
        # every step of this object -- eager or captured -- runs on ONE side stream: autograd's per-parameter gradient
        # accumulators remember the stream they were created on, and a capture breaks on one that belongs to another stream
        self.stream = torch.cuda.Stream(device=dev)

    def close(self):
        self._ops.set_seed_device(None)
        self.shapes.clear()

    def _run(self, batch_in, batch_out, capturing):
        self.seed.add_(1)
        if self.autocast_dtype is not None:
            ctx = torch.autocast("cuda", dtype=self.autocast_dtype, cache_enabled=not capturing)
        else:
            import contextlib
            ctx = contextlib.nullcontext()
        with ctx:
            total, logs = self.module.training_step(batch_in, batch_out)
        ops.backward(total)       # the Linear layers' weight gradients: one grouped launch (also inside the capture)
        if self.max_grad_norm is not None:
            torch.nn.utils.clip_grad_norm_(self.module.parameters(), self.max_grad_norm)
        self.opt.step()
        return total.detach(), {k: v.detach() for k, v in logs.items()}

    @staticmethod
    def _key(batch_in, batch_out):
        items = []
        for name, d in (("in", batch_in), ("out", batch_out or {})):
            for k in sorted(d):
                v = d[k]
                items.append((name, k, tuple(v.shape), str(v.dtype)) if torch.is_tensor(v) else (name, k, "object"))
        return tuple(items)

    def step(self, batch_in, batch_out=None):
        """-> (total loss, logs) as tensors on the device (read them with .item() only when they are printed)"""
        batch_out = batch_out or {}
        key = self._key(batch_in, batch_out)
        st = self.shapes.setdefault(key, {"n": 0, "graph": None})
        if any(not torch.is_tensor(v) for v in list(batch_in.values()) + list(batch_out.values())):
            st["n"] = -(1 << 30)                  # per-sample python lists (the template branch): never captured
        if st["graph"] is not None:
            for k, v in batch_in.items():
                st["inputs"][0][k].copy_(v, non_blocking=True)
            for k, v in batch_out.items():
                st["inputs"][1][k].copy_(v, non_blocking=True)
            st["graph"].replay()
            self.replays += 1
            mark_parameters_updated(self.module)
            return st["outputs"]
        cur = torch.cuda.current_stream()
        if 0 <= st["n"] < self.warmup or st["n"] < 0:
            self.opt.zero_grad(set_to_none=True)
            self.stream.wait_stream(cur)
            with torch.cuda.stream(self.stream):
                out = self._run(batch_in, batch_out, capturing=False)
            cur.wait_stream(self.stream)
            st["n"] += 1
            mark_parameters_updated(self.module)
            return out
        # capture: static copies of the inputs; gradients must not exist yet (they are born inside the graph's pool)
        s_in = {k: v.clone() for k, v in batch_in.items()}
        s_out = {k: v.clone() for k, v in batch_out.items()}
        self.opt.zero_grad(set_to_none=True)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=self.stream):
            outputs = self._run(s_in, s_out, capturing=True)
        st.update(graph=g, inputs=(s_in, s_out), outputs=outputs)
        g.replay()                                 # the capture recorded the step, it did not run it
        self.replays += 1
        mark_parameters_updated(self.module)
        return outputs


# ---- checkpoint layout of pytorch-lightning 2.0 (main.py:358-360, :390-405) --------
CKPT_KEYS = ("epoch", "global_step", "pytorch-lightning_version", "state_dict", "loops", "callbacks",
             "optimizer_states", "lr_schedulers")


def save_checkpoint(path, module, optimizer=None, lr_scheduler=None, epoch=0, global_step=0, monitor=None):
    """`best.ckpt` / `last.ckpt`: a torch.save dict with Lightning's keys; `state_dict` carries the
    module's own names (`model.…`, `mlm_head.…`), so the reference's
    `load_from_checkpoint(best, strict=False, args=args)` (main.py:404) reads it."""
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    ckpt = collections.OrderedDict()
    ckpt["epoch"] = int(epoch)
    ckpt["global_step"] = int(global_step)
    ckpt["pytorch-lightning_version"] = "2.0.0"
    ckpt["state_dict"] = collections.OrderedDict((k, v.detach().cpu()) for k, v in module.state_dict().items())
    ckpt["loops"] = None      # Lightning's restore_loops skips a None entry (an empty dict would KeyError on 'fit_loop')
    ckpt["callbacks"] = {"ModelCheckpoint": {"monitor": monitor, "best_model_path": os.path.abspath(path)}}
    ckpt["optimizer_states"] = [_portable_optimizer_state(optimizer.state_dict())] if optimizer is not None else []
    ckpt["lr_schedulers"] = [_portable_scheduler_state(lr_scheduler.state_dict())] if lr_scheduler is not None else []
    torch.save(ckpt, path)
    return path


def _as_float(v):
    return float(v.item()) if torch.is_tensor(v) else v


def _portable_optimizer_state(sd):
    """what an eager run (and Lightning's own AdamW) writes, whatever this run was: the learning rate as a Python float
    and capturable = False.  A --hip_graph_step run keeps both on the device (configure_optimizer(capturable=True));
    written as they are they would make the checkpoint unreadable for an eager resume (it would inherit capturable = True)
    and differ from the reference's checkpoint contents."""
    sd = dict(sd)
    groups = []
    for g in sd["param_groups"]:
        g = dict(g)
        g["lr"] = _as_float(g["lr"])
        if "initial_lr" in g:
            g["initial_lr"] = _as_float(g["initial_lr"])
        if "capturable" in g:
            g["capturable"] = False
        groups.append(g)
    sd["param_groups"] = groups
    return sd


def _portable_scheduler_state(sd):
    sd = dict(sd)
    for k in ("base_lrs", "_last_lr"):
        if k in sd:
            sd[k] = [_as_float(v) for v in sd[k]]
    return sd


# every registration of a submodule anywhere in the process (add_module, register_module, attribute assignment) moves this
# counter: the module list mark_parameters_updated caches is rebuilt when it has moved, so a submodule added or replaced
# after the first step (an encoder swap, adapters, a load that re-creates layers) is marked like the others
_module_tree_epoch = [0]


def _on_module_registered(module, name, submodule):
    _module_tree_epoch[0] += 1


try:
    torch.nn.modules.module.register_module_module_registration_hook(_on_module_registered)
except AttributeError:      # a torch without the global hook: no caching (see mark_parameters_updated)
    _module_tree_epoch = None


def mark_parameters_updated(module):
    """call after an optimizer step or a load_state_dict: the bf16 weight copies `ops.linear` shares between
    forwards (ops.WeightShadows) belong to a new generation; a backward of an older forward raises instead of reading them"""
    # (the module list is made once: walking the tree of ~400 modules through the generator at every step was 0.5 ms of a
    # step's ~13 ms of host time, tools/host_profile.py; a registry itself is made lazily by the first forward)
    epoch = _module_tree_epoch[0] if _module_tree_epoch is not None else None
    cached = module.__dict__.get("_shadow_holders")
    if cached is None or epoch is None or cached[0] != epoch:
        # (without the module itself: a list that holds its owner is a reference cycle, and the model would wait for the
        # cycle collector instead of being freed with its last reference)
        cached = module.__dict__["_shadow_holders"] = (epoch, [m for m in module.modules() if m is not module])
    holders = cached[1]
    for m in [module] + holders:
        reg = m.__dict__.get("_weight_shadows")
        if reg is not None:
            reg.mark_stale()


def load_checkpoint(path, module, optimizer=None, lr_scheduler=None, strict=False):
    ckpt = torch.load(path, map_location="cpu", weights_only=False)
    sd = {k: v for k, v in ckpt["state_dict"].items() if not k.endswith("position_ids")}   # 4.27.3 buffer
    missing, unexpected = module.load_state_dict(sd, strict=strict)
    mark_parameters_updated(module)
    if optimizer is not None and ckpt.get("optimizer_states"):
        # load_state_dict replaces the groups' hyper-parameters with the checkpoint's: a float lr and capturable = False
        # from an eager run or from the reference.  An optimizer built for GraphedStep keeps ITS form: the rate stays
        # the device tensor the captured update reads (the loaded value is copied into it), capturable stays on, and the
        # step counters move to the device (torch casts `step` by the flags of the CURRENT groups only when they are set
        # in the loaded ones, so it is done here).
        was = [(g.get("capturable", False), g["lr"]) for g in optimizer.param_groups]
        optimizer.load_state_dict(ckpt["optimizer_states"][0])
        for g, (capturable, lr_t) in zip(optimizer.param_groups, was):
            if capturable:
                g["capturable"] = True
            if torch.is_tensor(lr_t):
                lr_t.fill_(_as_float(g["lr"]))
                g["lr"] = lr_t
            if capturable or g.get("fused", False):
                for p_ in g["params"]:
                    st = optimizer.state.get(p_)
                    if st and torch.is_tensor(st.get("step")):
                        st["step"] = st["step"].to(device=p_.device, dtype=torch.float32)
    if lr_scheduler is not None and ckpt.get("lr_schedulers"):
        lr_scheduler.load_state_dict(ckpt["lr_schedulers"][0])
        if optimizer is not None:      # the schedule's view of the rates is the groups' (tensors stay tensors)
            lr_scheduler._last_lr = [g["lr"] for g in optimizer.param_groups]
    return ckpt, missing, unexpected


def clear_checkpoints(save_path):
    """--overwrite: delete every *.ckpt first (textreact/utils.py:47-52)"""
    if os.path.isdir(save_path):
        for f in os.listdir(save_path):
            if f.endswith(".ckpt"):
                os.remove(os.path.join(save_path, f))
