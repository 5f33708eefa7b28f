"""One group setup and one refusal rule for every N > 1 entry point (retrieve_faiss.py, main.py, tanimoto.py, bench.py).

The reference starts its ranks through Lightning's DDP strategy (main.py:372-374: `strategy=DDPStrategy(...)`, `devices=args.gpus`);
here a launcher (`python -m torch.distributed.run --nproc-per-node N ...`) starts one process per GPU and each of them calls
`setup()` ONCE, before anything else touches the GPU or the process group:

  1. `torch.cuda.set_device(ordinal)` -- this rank's GPU is the current device BEFORE the group exists (with the nccl backend the
     object collectives put their tensors on `torch.cuda.current_device()`, which would be cuda:0 on every rank; RCCL refuses two
     ranks on one device);
  2. `init_process_group("nccl", device_id=cuda:ordinal)` -- the communicator is bound to that device eagerly, so a wrong mapping
     fails here, in one place, and not inside the first collective of whichever CLI ran.

`refuse_mismatch(gpus, what)`: a `--gpus N` that is not the WORLD_SIZE the launcher set is a different job than the command names:
exit code 2, for every CLI alike (bench.py's rule since round 5).  A process that has to START ranks does so as a child
process before its own first GPU call (`launch_children`), never by re-executing itself.

TRX_DIST_BACKEND=gloo TRX_DEVICE=0 lets several ranks share one GPU to rehearse the N > 1 paths on a one-GPU box; the real run is
nccl (= RCCL on ROCm), one GPU per rank."""
import os
import sys


def world_size():
    return int(os.environ.get("WORLD_SIZE", "1"))


def device_ordinal():
    """this rank's GPU: TRX_DEVICE (rehearsals: several ranks on one GPU), else LOCAL_RANK, else 0"""
    return int(os.environ.get("TRX_DEVICE", os.environ.get("LOCAL_RANK", "0")))


def refuse_mismatch(gpus, what):
    """exit(2) when a launcher set WORLD_SIZE and it is not the --gpus the command names"""
    ws = os.environ.get("WORLD_SIZE")
    if ws is not None and gpus is not None and int(ws) != int(gpus):
        sys.stderr.write("%s: --gpus %d but the launcher set WORLD_SIZE=%s; refusing to run a different job than the command "
                         "names (launch with `python -m torch.distributed.run --nproc-per-node %d ...`)\n" % (what, gpus, ws, gpus))
        raise SystemExit(2)


def setup(backend=None):
    """-> (rank, world, device).  world == 1 (no launcher): (0, 1, cuda:ordinal or cpu), no group.  Otherwise the group is
    initialised as described above unless the caller already did.  backend: None = TRX_DIST_BACKEND, else "nccl" with a GPU,
    "gloo" without."""
    import torch
    cuda = torch.cuda.is_available()
    ordinal = device_ordinal()
    device = torch.device("cuda", ordinal) if cuda else torch.device("cpu")
    if cuda:
        torch.cuda.set_device(ordinal)
    world = world_size()
    if world <= 1:
        return 0, 1, device
    import torch.distributed as dist
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend is None:
            backend = os.environ.get("TRX_DIST_BACKEND", "nccl" if cuda else "gloo")
        if backend == "nccl":
            ngpu = torch.cuda.device_count() if cuda else 0
            if ordinal >= ngpu:      # refused here, with the reason, instead of inside the first collective
                sys.stderr.write("rank %s of %d wants cuda:%d and this process sees %d GPU(s): the nccl (RCCL) backend needs one GPU per "
                                 "rank (TRX_DIST_BACKEND=gloo TRX_DEVICE=0 rehearses several ranks on one device)\n"
                                 % (os.environ.get("RANK", "?"), world, ordinal, ngpu))
                raise SystemExit(2)
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
    return dist.get_rank(), dist.get_world_size(), device


def launch_children(gpus, script, argv, port_env="TRX_MASTER_PORT"):
    """Start `python -m torch.distributed.run --nnodes=1 --nproc-per-node gpus ... script argv` as a CHILD process and return its
    exit code.  For a process that was asked for N > 1 GPUs without a launcher: it must not have touched the GPU yet (a parent that
    has initialised HIP and then replaces or forks itself takes the machine down on this pool), and it never execs."""
    import socket
    import subprocess
    port = os.environ.get(port_env)
    if port is None:
        with socket.socket() as s_:
            s_.bind(("127.0.0.1", 0))
            port = s_.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(script)] + list(argv)
    sys.stderr.write("%s: --gpus %d without a launcher: starting %s\n" % (os.path.basename(script), gpus, " ".join(cmd)))
    sys.stderr.flush()
    return subprocess.run(cmd).returncode
