"""Developer: the batch-1 evaluation loop as its callers run it (forward, then instance predict, image after image) with the host
side of the predict broken into segments (perf_counter around the engine calls of one predict).
   gpurun -- python tools/b1_joint.py"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
dev = torch.device("cuda:0")
net, x, g, thr, H, W = bench.c3_model(dev, "exact")
eng = net._get_engine()
inst = lambda o: net.predict(o, mask_type="instance", threshold=thr, size=(H, W), image_ids=[7], nms_type="hard")


def joint(n=60):
    for _ in range(5):
        inst(net(x))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        inst(net(x))
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for rep in range(3):
    print(f"{joint():.3f} ms per image (forward + instance predict, joint loop)", flush=True)

# host segments of one joint iteration
seg = {}
def wrap(obj, name):
    f = getattr(obj, name)
    def w(*a, **k):
        t0 = time.perf_counter()
        r = f(*a, **k)
        seg[name] = seg.get(name, 0.0) + time.perf_counter() - t0
        return r
    setattr(obj, name, w)
from zutis_amd import rle
wrap(eng, "instance_candidates"); wrap(eng, "instance_nms_encode"); wrap(rle, "rles_from_transitions"); wrap(eng, "forward_graphed")
# the serial host stretch between the predict's synchronisation and the next forward's graph launch (the GPU idles through it)
import zutis_amd.engine as E
marks = {}
_th = E._to_host
def to_host(t, *a):
    r = _th(t, *a); marks["sync"] = time.perf_counter(); return r
E._to_host = to_host
_rp = torch.cuda.CUDAGraph.replay
def replay(self):
    marks["replay"] = time.perf_counter(); return _rp(self)
torch.cuda.CUDAGraph.replay = replay
n = 60
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    t_in = time.perf_counter()
    o = net(x)
    if "ret" in marks:
        seg["host: predict returned -> graph launch"] = seg.get("host: predict returned -> graph launch", 0.0) + marks["replay"] - marks["ret"]
    t1 = time.perf_counter()
    inst(o)
    marks["ret"] = time.perf_counter()
    seg["host: synchronised -> predict returns"] = seg.get("host: synchronised -> predict returns", 0.0) + marks["ret"] - marks["sync"]
    seg["predict (whole)"] = seg.get("predict (whole)", 0.0) + marks["ret"] - t1
torch.cuda.synchronize()
tot = (time.perf_counter() - t0) / n * 1e3
print(f"joint {tot:.3f} ms per image; host time per image inside:")
for k, v in seg.items():
    print(f"   {k:28s} {v / n * 1e3:.3f} ms")

# the body of trainer.evaluate's loop (trainer.py:327-348): + semantic predict, the metric meter's update and get_scores()
import numpy as np
from utils.running_score import RunningScore
meter = RunningScore(81, device=dev)
gt = np.random.default_rng(3).integers(0, 81, (1, H, W)).astype(np.int64)
seg2 = {}
def timed(name, fn):
    t = time.perf_counter(); r = fn(); seg2[name] = seg2.get(name, 0.0) + time.perf_counter() - t; return r
for it in range(n + 5):
    if it == 5:
        seg2.clear(); torch.cuda.synchronize(); t0 = time.perf_counter()
    o = timed("forward", lambda: net(x))
    sem = timed("predict semantic", lambda: net.predict(o, mask_type="semantic", size=(H, W)))
    timed("predict instance", lambda: inst(o))
    timed("meter.update", lambda: meter.update(gt, sem))
    timed("meter.get_scores", lambda: meter.get_scores())
torch.cuda.synchronize()
print(f"trainer.evaluate body {(time.perf_counter() - t0) / n * 1e3:.3f} ms per image; host time per image inside:")
for k, v in seg2.items():
    print(f"   {k:28s} {v / n * 1e3:.3f} ms")
