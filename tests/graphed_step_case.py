"""child process of test_predictor_gpu.py::test_graphed_step_equals_the_eager_steps (started with train.GRAPH_RUNTIME_ENV set)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_predictor_gpu import _graph_case

curves, finals = [], []
for warmup in (3, 10 ** 9):                      # 10**9: never captures
    train, p, batches = _graph_case()
    opt, sched = train.configure_optimizer(p, 1e-3, 0.01, 100, 0.1, capturable=True)
    torch.manual_seed(5)                         # the seed counter's starting value
    gs = train.GraphedStep(p, opt, max_grad_norm=1.0, autocast_dtype=torch.bfloat16, warmup=warmup)
    losses = []
    try:
        for b_in, b_out in batches:              # the loss read after every step
            total, logs = gs.step(b_in, b_out)
            sched.step()
            losses.append(float(total))
        for i, (b_in, b_out) in enumerate(batches[:8]):      # replays queued behind one another, the host waiting once in between
            total, logs = gs.step(b_in, b_out)
            sched.step()
            if i == 2:
                torch.cuda.current_stream().synchronize()
        losses.append(float(total))
        assert gs.replays == (14 if warmup == 3 else 0), gs.replays
    finally:
        gs.close()
    curves.append(losses)
    finals.append({n: t.detach().float().clone() for n, t in p.named_parameters()})
assert all(np.isfinite(curves[0])) and curves[0][-1] < curves[0][0], curves
for a, b in zip(*curves):
    assert abs(a - b) <= 2e-3 * max(1.0, abs(b)), curves
for n in finals[0]:
    assert float((finals[0][n] - finals[1][n]).abs().max()) <= 2e-3 * max(1.0, float(finals[1][n].abs().max())), n
print("graphed step: ok")
