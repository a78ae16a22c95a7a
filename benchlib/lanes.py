"""The lanes the headline / config-4 runs TIME (tests/test_timed_path_gpu.py builds its lanes through these functions)."""
from __future__ import annotations

import contextlib

import torch


def build_lanes(eng, x, text, S, n, n_lanes, world=1, dist_on=False, h2d=False, d2h=False):
    """The lanes bench.py times: lane i = a fork of `eng` (own activation buffers, shared packed weights), ONE step recorded into
    a native launch plan (zutis_amd/plan.py) — forward + low-res class logits + fused upsample/argmax to [B, S, S] int64 labels — a HIP
    stream and (N > 1) a gather buffer.  With one lane the step runs eagerly on the current stream.  tests/test_timed_path_gpu.py
    builds its lanes through this function, so what the test checks is what the bench times."""
    from zutis_amd import distributed as zd
    from zutis_amd import ops
    from zutis_amd import plan as zplan
    B, dev = x.shape[0], x.device
    lanes = []
    for li in range(n_lanes):
        e = eng if li == 0 else eng.fork()       # own activation buffers, shared packed weights
        e.forward(x)                             # eager warm-up: packs weights, sizes the buffer cache
        plan = None
        xin = x.clone() if h2d else x            # h2d: the lane's own input buffer, refilled from the host every step

        def one_step(e=e, xin=xin):
            out = e.forward(xin)
            lo = e.semantic_logits_lowres(out["patch_tokens"], text)
            labels = torch.empty((B, S, S), dtype=torch.int64, device=dev)
            ops.upsample_argmax(lo, labels, B, n, lo.shape[2], lo.shape[3], S, S)
            return lo, labels
        if n_lanes > 1:
            with zplan.Recorder() as rec:
                lo, labels = one_step()
            plan = rec.build()
        else:
            lo, labels = one_step()
        hw2 = lo.shape[2] * lo.shape[3]
        lanes.append(zd.Lane(lo.view(B, n, hw2), gathered=torch.empty((world * B, n, hw2), dtype=torch.float32, device=dev) if dist_on else None,
                             stream=torch.cuda.Stream(device=dev) if n_lanes > 1 else None,
                             state={"eng": e, "plan": plan, "labels": labels, "step": one_step, "xin": xin, "lo_shape": tuple(lo.shape),
                                    "host_labels": torch.empty((B, S, S), dtype=torch.int64).pin_memory() if d2h else None}))
    return lanes


def make_launch(n_lanes, host_x=None, h2d=False, d2h=False):
    """The `launch(group, step_ids)` callback of zutis_amd.distributed.StepPipeline for lanes from build_lanes()."""
    from zutis_amd import plan as zplan

    def launch(grp, ids):
        if h2d:              # the step's batch crosses PCIe first, in stream order before the step's kernels
            for ln in grp:
                with torch.cuda.stream(ln.stream) if ln.stream is not None else contextlib.nullcontext():
                    ln.state["xin"].copy_(host_x, non_blocking=True)
        if n_lanes > 1:      # consecutive steps replayed interleaved, one stream each, from one C loop
            zplan.run_many([ln.state["plan"] for ln in grp], [ln.stream.cuda_stream for ln in grp])
        else:                # plain eager loop on the current stream (payload tensor is re-bound: eager steps allocate)
            for ln in grp:
                lo, ln.state["labels"] = ln.state["step"]()
                ln.payload = lo.view(ln.payload.shape)
        if d2h:              # networks/zutis.py:372 `.cpu().numpy()`: the label maps leave the device, in stream order
            for ln in grp:
                with torch.cuda.stream(ln.stream) if ln.stream is not None else contextlib.nullcontext():
                    ln.state["host_labels"].copy_(ln.state["labels"], non_blocking=True)
    return launch


def check_timed_outputs(lanes):
    """What the timed region produced, checked AFTER it (round-4 review: the replayed plans' own outputs were never looked at):
    every lane's label maps and low-res logits — as the last replay of its plan left them — against ONE eager step of that lane's
    engine on the same input, bitwise.  Returns (ok, lo, labels): lane 0's timed outputs (clones) for the oracle parity leg."""
    torch.cuda.synchronize()
    kept = [(ln.payload.clone(), ln.state["labels"].clone()) for ln in lanes]
    ok = True
    for ln, (lo_t, lab_t) in zip(lanes, kept):
        lo_e, lab_e = ln.state["step"]()             # eager launches on the current stream, the lane's own engine and input
        torch.cuda.synchronize()
        ok = ok and bool(torch.equal(lo_e.reshape(lo_t.shape), lo_t)) and bool(torch.equal(lab_e, lab_t))
    lo0, lab0 = kept[0]
    return ok, lo0.view(lanes[0].state["lo_shape"]), lab0
