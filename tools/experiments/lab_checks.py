"""Lab checks of the kernels that live in tools/experiments/ (not part of the product library, not collected by `pytest tests/`):
    make -C textreact_amd/csrc lab && TRX_NN_LIB=libtrxnn_lab.so python -m pytest tools/experiments/lab_checks.py -q
They were tests/test_predictor_gpu.py::test_ping_pong_forward_kernel / ::test_persistent_forward_kernel until round 6 moved
the kernels out of libtrxnn.so."""
import os

import pytest

pytestmark = pytest.mark.skipif(os.environ.get("TRX_NN_LIB") != "libtrxnn_lab.so", reason="needs the lab build: TRX_NN_LIB=libtrxnn_lab.so")


def test_ping_pong_forward_kernel():
    """attn_fwd_pp.h (two wave groups in ping-pong; off by default: it measured slower) stays correct: the randomised
    attention cases with TRX_NN_ATTN_PP=1, in a process of their own (the switch is read once)"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "attn_fuzz.py"), "60", "5"], capture_output=True, text=True,
                       timeout=600, env=dict(os.environ, TRX_NN_ATTN_PP="1"))
    assert r.returncode == 0 and "0 failures" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_persistent_forward_kernel():
    """attn_fwd_persist.h (the encoder's shape class on 768 resident workgroups that take several 128-query items each; off by
    default: it measured no faster, profiles/r05_attention_dropout_ab.json) returns the bits of the default kernel: eight cases
    with one or two items per workgroup, mask and dropout on and off, hashed in a process with TRX_NN_ATTN_PERSIST=1 and in one without"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    outs = []
    for switch in ("1", "0"):
        r = subprocess.run([sys.executable, os.path.join(root, "tools", "r05", "persist_check.py")], capture_output=True, text=True,
                           timeout=600, env=dict(os.environ, TRX_NN_ATTN_PERSIST=switch))
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        outs.append(json.loads(r.stdout.strip().splitlines()[-1]))
    assert len(outs[0]) == 8 and outs[0] == outs[1]
