"""bench.py through its launch contract: N = 1 directly, N = 2 as the driver launches it (torch.distributed.run, one rank
per "GPU") -- rehearsed on one device with the gloo backend (RCCL refuses two ranks on one GPU), small sizes."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--steps", "2", "--warmup", "1", "--n-corpus", "30000", "--n-queries", "1024"]


def _line(out):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


def test_single_gpu_line_has_the_contract_fields():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + SMALL, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    j = _line(r.stdout)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in j, key
    assert j["n_gpus"] == 1 and j["steps"] == 2 and j["vs_baseline"] is None and j["value"] > 0
    assert j["roofline"]["bound"] == "mfma" and 0 < j["roofline"]["frac"] < 1
    assert j["cpu_baseline"]["kind"] in ("port", "reference") and j["cpu_baseline"]["index_agreement_with_gpu"] == 1.0
    assert j["cpu_baseline"]["tflops"] > 0 and j["cpu_baseline"]["cores"] >= 1
    # round 6: the second roofline (L2 -> LDS fill), the drop-in call on host arrays beside `value` (same ids), and both CPU candidates' rates
    assert j["roofline"]["fill"]["peak"] == 14.0 and 0 < j["roofline"]["fill"]["frac"] < 1.2
    assert j["value_host_api"] > 0 and j["host_api"]["same_ids_as_device_resident"] is True
    assert j["cpu_baseline"]["kind"] == "reference" or len(j["cpu_baseline"]["candidates_on_the_probe"]) == 2


def test_fingerprint_workload_line_runs_the_int8_form():
    """bench.py --workload fingerprint (the reference's own FAISS call: L2, k = 20, 2048-d counts searching themselves), small:
    the line says which arithmetic ran, prices it against that arithmetic's peak, and every row finds itself first"""
    for env_extra, want in (({}, 1), ({"TRX_NO_I8": "1"}, 0)):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "fingerprint", "--steps", "1", "--warmup", "1",
                            "--n-corpus", "20000", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, env=dict(os.environ, **env_extra))
        assert r.returncode == 0, r.stderr[-2000:]
        j = _line(r.stdout)
        assert j["config"]["int8_scan"] == want and j["dtype"] == ("i8" if want else "bf16"), j
        assert j["roofline"]["peak"] == (5000.0 if want else 2500.0) and j["config"]["self_is_first"] and j["config"]["exact_class"] == 1
        assert j["roofline"]["fill"]["achieved"] > 0 and j["roofline"]["traffic"] is None      # (traffic: measured at the full size only)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "fingerprint", "--steps", "1", "--warmup", "1", "--n-corpus", "20000",
                        "--no-cpu-baseline", "--host-api"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    j = _line(r.stdout)      # --host-api: the same self-search through host int64 arrays, same neighbours
    assert j["value_host_api"] > 0 and j["host_api"]["same_ids_as_device_resident"] is True and j["host_api"]["int8_scan"] == 1


def test_morgan_workload_line_runs_the_fp4_form():
    """bench.py --workload morgan (retrieve/retro.sh's call: 1024-component Morgan bit vectors searching themselves), small: the
    scan runs on fp4 operands and the line prices it against that peak"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "morgan", "--steps", "1", "--warmup", "1",
                        "--n-corpus", "20000", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    j = _line(r.stdout)
    assert j["config"]["int8_scan"] == 2 and j["dtype"] == "fp4" and j["roofline"]["peak"] == 10000.0, j
    assert j["config"]["self_is_first"] and j["config"]["exact_class"] == 1 and j["config"]["dim"] == 1024


@pytest.mark.parametrize("mode", [[], ["--replicas"]])
def test_two_ranks_as_the_driver_launches_it(mode):
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0)); port = s_.getsockname()[1]
    env = dict(os.environ, TRX_BENCH_BACKEND="gloo", TRX_BENCH_DEVICE="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2"] + SMALL + mode
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    j = _line(r.stdout)
    assert j["n_gpus"] == 2 and j["value"] > 0 and j["scaling"] == "strong"
    assert ("replicas" in j["config"]["parallelism"]) == bool(mode)
    assert bool(mode) or j["config"]["selfcheck"].startswith("passed")      # the row-sharded line checks itself before it times


def test_gpus_2_without_a_launcher_measures_two_ranks():
    """`python bench.py --gpus 2` with no launcher around it (the shape of the driver's N = 1 command) must not print an
    n_gpus = 1 line: bench.py starts the two ranks itself as a child process and relays rank 0's line"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(TRX_BENCH_BACKEND="gloo", TRX_BENCH_DEVICE="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + SMALL, capture_output=True, text=True,
                       timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    j = _line(r.stdout)
    assert j["n_gpus"] == 2 and j["config"]["selfcheck"].startswith("passed")


def test_gpus_that_disagree_with_world_size_exit_non_zero():
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0", TRX_BENCH_BACKEND="gloo", TRX_BENCH_DEVICE="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + SMALL, capture_output=True, text=True,
                       timeout=300, env=env)
    assert r.returncode != 0 and not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_selfcheck_turns_a_wrong_exchange_into_a_failing_exit_code():
    """a rank that reports its rows under the wrong ids (TRX_BENCH_INJECT_FAULT=offset) must not produce a bench line"""
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0)); port = s_.getsockname()[1]
    env = dict(os.environ, TRX_BENCH_BACKEND="gloo", TRX_BENCH_DEVICE="0", TRX_BENCH_INJECT_FAULT="offset")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2"] + SMALL
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode != 0 and "selfcheck FAILED" in r.stderr, r.stderr[-2000:]
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_single_gpu_selfcheck_flag():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--selfcheck", "--no-cpu-baseline"] + SMALL,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert _line(r.stdout)["config"]["selfcheck"].startswith("passed")
