"""`python bench.py --gpus N` without a rank environment: N ranks as a child process tree."""
from __future__ import annotations

import json
import os
import sys

import torch

from .common import BENCH_PY


def rank_launch_command(n_gpus: int, argv, port: int):
    """The command line `python bench.py --gpus N` runs as a child: one rank per GPU under torch.distributed.run, rendezvous on
    127.0.0.1 (the container hostname may not resolve) — the same form the driver uses for its own N > 1 launches."""
    child_args = [a for a in argv if a != "--dry-launch"]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), BENCH_PY] + child_args


def launch_ranks(n_gpus: int, argv, dry: bool) -> int:
    """`python bench.py --gpus N` without a rank environment: start N ranks as a CHILD process tree and forward rank 0's JSON line.
    Nothing here touches the GPU (torch.cuda.device_count() does not initialise HIP on this image; a process that has must never
    exec or be replaced), so the children are the first to do so; the parent only waits and passes the exit code on."""
    import socket
    import subprocess
    if n_gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = rank_launch_command(n_gpus, argv, port)
    env = {**os.environ, "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")}
    if dry:
        print(json.dumps({"cmd": cmd, "n_ranks": n_gpus, "env": {"HSA_ENABLE_IPC_MODE_LEGACY": env["HSA_ENABLE_IPC_MODE_LEGACY"]}}), flush=True)
        return 0
    have = torch.cuda.device_count()
    if have < n_gpus:
        sys.stderr.write(f"bench.py: --gpus {n_gpus} but only {have} device(s) visible\n")
        return 3
    if n_gpus == 1:
        raise AssertionError("launch_ranks is for N > 1")
    return subprocess.call(cmd, env=env)
