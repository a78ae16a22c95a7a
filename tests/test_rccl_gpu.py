"""-m gpu: the N > 1 code path of bench.py (RCCL process group, StepPipeline with an asynchronous all-gather per lane) executed
on hardware with ONE rank.  It claims nothing about scaling — the driver's SCALE run owns that — but the collective path, its
stream ordering and the lane reuse logic run on a real MI355X in the driver's pytest."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_rccl_one_rank_step_pipeline(dev):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_rccl_worker.py")
    env = {**os.environ, "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": "0"}
    r = subprocess.run([sys.executable, worker, str(port)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RCCL_SMOKE_OK" in r.stdout, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])
