"""`python bench.py --gpus N` must start N ranks itself (the driver runs it that way): the launcher's command line, that the
command really yields N ranks that can rendezvous (gloo, CPU), and the refusals — all without a GPU."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env():
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    return e


def test_dry_launch_builds_one_rank_per_gpu():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-launch"], capture_output=True, text=True,
                       env=_env(), timeout=120)
    assert r.returncode == 0, r.stderr
    d = json.loads(r.stdout.strip().splitlines()[-1])
    cmd = d["cmd"]
    assert d["n_ranks"] == 2 and cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nproc-per-node=2" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    i = cmd.index(BENCH)
    assert cmd[i + 1:] == ["--gpus", "2", "--steps", "3", "--warmup", "1"]          # the child sees the same arguments, minus --dry-launch
    assert d["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_dry_launch_for_every_sharded_workload():
    """BASELINE configs 4 and 5 at N GPUs go through the same launcher as the headline: `bench.py --workload c4|c5 --gpus N` = N ranks
    under torch.distributed.run with the workload's arguments passed on (c4: StepPipeline + per-step all-gather of the logits; c5: rank-
    sharded extraction + sharded retrieval — both covered at world 2 on gloo in tests/test_distributed_cpu.py)."""
    for wl, extra in (("c4", ["--batch", "8"]), ("c5", ["--inflight", "1"]), ("c3", [])):
        r = subprocess.run([sys.executable, BENCH, "--workload", wl, "--gpus", "8", "--dry-launch"] + extra, capture_output=True, text=True,
                           env=_env(), timeout=120)
        assert r.returncode == 0, r.stderr
        d = json.loads(r.stdout.strip().splitlines()[-1])
        cmd = d["cmd"]
        assert d["n_ranks"] == 8 and "--nproc-per-node=8" in cmd and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
        assert cmd[cmd.index(BENCH) + 1:] == ["--workload", wl, "--gpus", "8"] + extra


def test_launch_command_starts_n_ranks(tmp_path):
    """The launcher's command with a stand-in script: two processes, RANK 0 and 1, WORLD_SIZE 2, a gloo all-reduce between them."""
    sys.path.insert(0, ROOT)
    import bench
    stub = tmp_path / "stub.py"
    stub.write_text(
        "import os, torch, torch.distributed as dist\n"
        "dist.init_process_group('gloo')\n"
        "t = torch.tensor([float(os.environ['RANK']) + 1]); dist.all_reduce(t)\n"
        "import sys\n"
        "sys.stdout.write('RANK %s WORLD %s LOCAL %s SUM %d\\n' % (os.environ['RANK'], os.environ['WORLD_SIZE'], os.environ['LOCAL_RANK'], int(t.item())))\n"
        "sys.stdout.flush()\n"           # ONE write per rank: print() writes its arguments piecewise and the two ranks share the pipe
        "dist.destroy_process_group()\n")
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = bench.rank_launch_command(2, ["--gpus", "2"], port)
    cmd[cmd.index(BENCH)] = str(stub)
    r = subprocess.run(cmd, capture_output=True, text=True, env=_env(), timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = sorted(l for l in r.stdout.splitlines() if l.startswith("RANK"))
    assert lines == ["RANK 0 WORLD 2 LOCAL 0 SUM 3", "RANK 1 WORLD 2 LOCAL 1 SUM 3"]


def test_refuses_more_ranks_than_devices_and_mismatched_world():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "64"], capture_output=True, text=True, env=_env(), timeout=120)
    assert r.returncode != 0 and "device(s) visible" in r.stderr
    e = _env()
    e["WORLD_SIZE"] = "2"
    r = subprocess.run([sys.executable, BENCH, "--gpus", "4"], capture_output=True, text=True, env=e, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr
