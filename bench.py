#!/usr/bin/env python3
"""bench.py — headline metric of BASELINE.json on MI355X: images/sec of ZUTIS ViT-B/16 dense semantic
segmentation @336px (forward + semantic predict), synthetic data, random-init weights of the real architecture.

    python bench.py --gpus 1 --steps 50 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path over one batch per GPU: ZutisEngine.forward (encoder, x2 upsample, ffn1,
6-layer decoder, ffn2, mask einsum, text-space projection) + predict_semantic (class-logit GEMM + fused
bilinear-upsample/argmax to 336x336 labels).  Inputs are resident in HBM before the timed region.
For N > 1 images are sharded by rank (weak scaling, 32 per GPU) and each step all-gathers the low-res class
logits over RCCL/xGMI (north_star: "RCCL all-gather of logits for evaluation"), overlapped with the next step.

Steps are independent batches (an evaluation loop), so `--inflight 3` (default) keeps three of them in flight: each step
is recorded once into a native launch plan (zutis_amd/plan.py) and consecutive steps are replayed interleaved on three HIP
streams by one C loop (zh_plan_run_multi), so one batch's kernel tails, launch gaps and HBM-bound epilogues overlap the
others' MFMA phases (measured: 1 -> 2 in flight +15 %, 2 -> 3 +2.7 %, 4 loses).  Every step still does all of its work inside the timed region; `--inflight 1` is the plain
one-stream eager loop.

The headline runs at `--precision exact`: every contraction in the f16x3 mode (fp16 split pairs, three MFMA products per
accumulator, fp32-class), i.e. the reference's own fp32 arithmetic class; the narrower `fast` precision (north-star tolerance
1e-3) is timed as `second_precision`.

Prints ONE JSON line on rank 0 (contract in the task brief) incl. `roofline` (dominant kernel = the f16x3 MFMA
GEMM, measured with HIP events around every launch of an instrumented step) and `cpu_baseline` (the oracle = CPU
port of the reference path, timed on the host cores on a bounded sample, rank 0, N=1 only).
"""
from __future__ import annotations

import argparse
import contextlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# arithmetic type of the contractions: "f16x3" = fp16 split pairs (hi + lo, 22 significand bits), three MFMA products per fp32
# accumulator — the reference's fp32 arithmetic class; "f16" = fp16 MFMA operands (narrower than the reference)
PRECISION_DTYPE = {"fast": "f16", "exact": "f16x3 (fp32-class)", "f16": "f16"}
PRECISION_TEXT = {
    "fast": "fp16 MFMA operands / fp32 accumulate in the transformer bodies, fp32 residual stream + LayerNorm + softmax; the "
            "output-facing contractions (ffn1, ffn2, mask einsum, text-space projection, class logits) in the f16x3 mode",
    "exact": "every contraction in the reference-equivalent f16x3 mode: operands as fp16 split pairs (hi + lo, 22 bits), three MFMA "
             "products per accumulator in fp32, split-pair attention scores; fp32 residual stream + LayerNorm + softmax",
    "f16": "fp16 MFMA operands / fp32 accumulate everywhere (round-1 behaviour)",
}
MFMA_F16_DENSE_PEAK_TFLOPS = 2500.0   # /opt/skills/guides/MI355X_MICROARCH.md: BF16/FP16 MFMA ~2.5 PF dense
FLOPS_PER_IMAGE_C2 = 124.4e9 + 0.146e9  # forward + semantic predict


def live_pmc_traffic(extra_args, split: int, timeout_s=240):
    """HBM bytes per launch of the dominant GEMM kernel family (split = 1: gemm_f16_kernel<..., SPLIT=1>, the f16x3 kernel;
    0: the plain fp16-operand kernel), measured NOW: two child runs of this script under
    `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes: the TCC block cannot hold both counters; --kernel-trace
    only, as /opt/skills/guides/MI355X_MICROARCH.md prescribes), corrected for gfx950 (FETCH_SIZE tallies 128-B requests at
    64 B: read bytes = 2 * FETCH_SIZE; WRITE_SIZE exact; both in KiB).  Returns (bytes_per_launch | None, note)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None, "this run is itself being profiled: no nested rocprofv3 passes"
    tot, cnt, seq = {}, {}, {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="zh_pmc_", dir="/tmp")
        cmd = [exe, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable, os.path.abspath(__file__),
               "--inflight", "1", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-torch-gpu-baseline", "--no-second-precision",
               "--no-live-traffic", "--no-batch1", "--no-configs"] + list(extra_args)
        try:
            subprocess.run(cmd, cwd="/tmp", env={**os.environ, "TMPDIR": "/tmp"}, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                           timeout=timeout_s, check=True)
            rows = []
            for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
                for r in csv.DictReader(open(f)):
                    k = r["Kernel_Name"]
                    if "gemm_f16_kernel" in k and k.rstrip().endswith(", %d>(GemmArgs)" % split) and r["Counter_Name"] == counter:
                        rows.append((int(r.get("Dispatch_Id", len(rows))), float(r["Counter_Value"])))
            rows.sort()
            seq[counter] = [v for _, v in rows]
            tot[counter] = sum(seq[counter])
            cnt[counter] = len(rows)
        except Exception as e:                                   # profiler unavailable / refused: report, never fail the bench
            shutil.rmtree(d, ignore_errors=True)
            return None, f"live rocprofv3 pass failed ({type(e).__name__})", None
        shutil.rmtree(d, ignore_errors=True)
    if not cnt.get("FETCH_SIZE") or not cnt.get("WRITE_SIZE"):
        return None, "no GEMM dispatches in the counter output", None
    fetch = tot["FETCH_SIZE"] / cnt["FETCH_SIZE"] * 1024.0
    write = tot["WRITE_SIZE"] / cnt["WRITE_SIZE"] * 1024.0
    # per dispatch, in dispatch order (both passes run the same launch sequence): bytes = 2 * FETCH_SIZE + WRITE_SIZE
    per = None
    if cnt["FETCH_SIZE"] == cnt["WRITE_SIZE"]:
        per = [(2.0 * a + b) * 1024.0 for a, b in zip(seq["FETCH_SIZE"], seq["WRITE_SIZE"])]
    return round(2.0 * fetch + write), (f"live: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE passes of `bench.py --inflight 1 --steps 2` run by this "
                                        f"bench ({cnt['FETCH_SIZE']} launches of gemm_f16_kernel<..., SPLIT={split}>): 2*FETCH_SIZE ({2 * fetch / 1e6:.1f} MB) + WRITE_SIZE "
                                        f"({write / 1e6:.1f} MB) per launch, gfx950 correction"), per


def gemm_roofline(ops, run_once, step_seconds):
    """HIP events around every GEMM / attention launch of `run_once()` (eager, torch's current stream == launch stream):
    roofline object for the GEMM family with the larger GPU time; FLOPs are ALGORITHMIC (2*M*N*K per launch)."""
    prof = {}

    def profiler(name, work, launch):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = launch()
        e1.record()
        prof.setdefault(name, []).append((work, e0, e1))
        return r
    ops.PROFILER = profiler
    try:
        for _ in range(2):
            prof.clear()
            run_once()
        torch.cuda.synchronize()
    finally:
        ops.PROFILER = None
    fl = lambda w: w[0] if isinstance(w, tuple) else w
    stats = {k: (len(v), sum(fl(w) for w, _, _ in v), sum(a.elapsed_time(b) for _, a, b in v) * 1e-3) for k, v in prof.items()}
    algo_bytes = {k: sum(w[1] for w, _, _ in v if isinstance(w, tuple)) / max(1, len(v)) for k, v in prof.items()}
    fams = {"gemm_f16": 1, "gemm_f16x2": 2, "gemm_f16x3": 3}           # family -> fp16 MFMA products per algorithmic product
    g3 = stats.get("gemm_f16x3", (0, 0.0, 0.0))
    dom = max(fams, key=lambda k: stats.get(k, (0, 0.0, 0.0))[2])      # the kernel family with the largest GPU time
    nl, flops_dom, tt = stats[dom]
    ach = flops_dom / tt / 1e12
    # f16x3: every algorithmic product is three fp16 MFMAs (hi*hi + lo*hi + hi*lo), so the ceiling for ALGORITHMIC flops is a
    # third of the dense fp16 MFMA peak; achieved / peak then equals (MFMA flops issued per second) / 2.5 PF.  f16x2 (fp16-valued
    # weights: the W lo plane is zero and its product is skipped): two MFMAs per product, ceiling = half the peak.
    npr = fams[dom]
    x3 = npr > 1
    peak = MFMA_F16_DENSE_PEAK_TFLOPS / float(npr)
    kname = {1: " (zh_gemm_f16)", 2: "<SPLIT=2> (zh_gemm_f16x3 with planeW = 0: fp16-valued weights, two products)", 3: "<SPLIT=1> (zh_gemm_f16x3)"}[npr]
    roof = {"bound": "mfma", "kernel": "gemm_f16_kernel" + kname,
            "achieved": round(ach, 1), "peak": round(peak, 1), "unit": "TFLOP/s", "frac": round(ach / peak, 4),
            "peak_note": (("algorithmic-flop ceiling of the f16x%d mode = 2500 TFLOP/s dense fp16 MFMA peak / %d MFMA products per fp32-class "
                           "product; frac == MFMA flops issued per second / 2500" % (npr, npr)) if x3 else "dense fp16 MFMA peak (MI355X_MICROARCH.md)"),
            "mfma_issue_tflops": round(ach * npr, 1),
            "traffic": None, "algorithmic_bytes_per_launch": round(algo_bytes.get(dom, 0.0)), "flops_per_launch": round(flops_dom / nl), "launches_per_step": nl, "avg_launch_us": round(tt / nl * 1e6, 1),
            "measured_on": "HIP events around every GEMM launch of an instrumented eager step on one stream (kernels not overlapped)",
            "gemm_share_of_step": round(tt / step_seconds, 3)}
    # the same launches grouped by problem shape (M x N x K [x batch]), largest GPU time first: which GEMMs set the average
    by = {}
    for w, a, b in prof[dom]:
        if isinstance(w, tuple) and len(w) > 2:
            e = by.setdefault(w[2], [0, 0.0, 0.0])
            e[0] += 1; e[1] += w[0]; e[2] += a.elapsed_time(b) * 1e-3
    gemm_roofline.last_launch_shapes = [(w[2], w[1]) for w, _, _ in prof[dom] if isinstance(w, tuple) and len(w) > 2]   # (shape, algorithmic bytes), launch order
    roof["by_shape"] = [{"MxNxK": "x".join(str(d) for d in (k[:3] if k[3] == 1 else k)), "launches": v[0], "avg_us": round(v[2] / v[0] * 1e6, 1),
                         "tflops": round(v[1] / v[2] / 1e12, 1), "frac": round(v[1] / v[2] / 1e12 / peak, 3),
                         "share_of_kernel_time": round(v[2] / tt, 3)}
                        for k, v in sorted(by.items(), key=lambda kv: -kv[1][2])[:8]]
    others = [k for k in fams if k != dom and k in stats]
    if others:
        og = [{"kernel": k, "launches_per_step": stats[k][0], "algorithmic_tflops": round(stats[k][1] / stats[k][2] / 1e12, 1),
               "share_of_step": round(stats[k][2] / step_seconds, 3)} for k in others]
        roof["other_gemm"] = og[0] if len(og) == 1 else og
    if g3[0]:
        roof["x3_note"] = "zh_gemm_f16x3 issues three MFMAs per algorithmic product: its MFMA-pipe rate is 3x its algorithmic TFLOP/s"
    # flops the engine EXECUTES per step (sum of 2*M*N*K / 4*Tq*Tk*dh over the launches; the pack-time compositions of DESIGN 2a
    # remove work the reference's 124.5 GFLOP / image counts)
    roof["executed_algorithmic_flops_per_step"] = round(sum(v[1] for k, v in stats.items() if k.startswith(("gemm", "attention"))))
    for an in ("attention_f16", "attention_f16x3"):
        if an in stats:
            na, fa, ta = stats[an]
            roof[an + "_tflops"] = round(fa / ta / 1e12, 1)
            roof[an + "_share_of_step"] = round(ta / step_seconds, 3)
    return roof


def bench_c5(args):
    """Config 5: CLIP ViT-L/14@336 `encode_image` over synthetic batches generated on the device, images sharded by
    rank, no communication until one final all-gather of the last step's embeddings (per-rank shards are the product)."""
    world = int(os.environ.get("WORLD_SIZE", "1")); rank = int(os.environ.get("RANK", "0")); local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    dist_on = world > 1 or args.force_dist
    if dist_on:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29532")
        dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)
    from zutis_amd import detgen, ops
    from zutis_amd.engine import ClipImageEncoder
    D, L, p, g, E = 1024, args.c5_layers, 14, 24, 768
    B = 256 if args.batch == 32 else args.batch

    def w(name, shape, std, mean=0.0):
        return torch.from_numpy(detgen.det_normal("c5." + name, shape, std, mean, 5)).to(dev)
    P = {"visual.class_embedding": w("cls", (D,), D ** -0.5), "visual.positional_embedding": w("pos", (g * g + 1, D), D ** -0.5),
         "visual.proj": w("proj", (D, E), D ** -0.5), "visual.conv1.weight": w("conv", (D, 3, p, p), (3 * p * p) ** -0.5)}
    for ln in ("ln_pre", "ln_post"):
        P[f"visual.{ln}.weight"], P[f"visual.{ln}.bias"] = w(ln + "w", (D,), 0.1, 1.0), w(ln + "b", (D,), 0.1)
    for i in range(L):
        q = f"visual.transformer.resblocks.{i}."
        P[q + "attn.in_proj_weight"], P[q + "attn.in_proj_bias"] = w(q + "a", (3 * D, D), D ** -0.5), w(q + "ab", (3 * D,), 0.02)
        P[q + "attn.out_proj.weight"], P[q + "attn.out_proj.bias"] = w(q + "o", (D, D), D ** -0.5 * (2 * L) ** -0.5), w(q + "ob", (D,), 0.02)
        P[q + "mlp.c_fc.weight"], P[q + "mlp.c_fc.bias"] = w(q + "f", (4 * D, D), (2 * D) ** -0.5), w(q + "fb", (4 * D,), 0.02)
        P[q + "mlp.c_proj.weight"], P[q + "mlp.c_proj.bias"] = w(q + "p", (D, 4 * D), D ** -0.5 * (2 * L) ** -0.5), w(q + "pb", (D,), 0.02)
        for ln in ("ln_1", "ln_2"):
            P[q + ln + ".weight"], P[q + ln + ".bias"] = w(q + ln + "w", (D,), 0.1, 1.0), w(q + ln + "b", (D,), 0.1)
    # The reference builds this tower with clip.load() (utils/extract_image_embeddings.py:43) = build_model(): convert_weights rounds
    # every conv / Linear weight and bias, the attention in_proj tensors and `proj` to fp16 (clip_arch.py:566-587,625) — the released
    # checkpoints hold fp16 values anyway.  Random weights "of that architecture" therefore carry fp16 VALUES in those tensors (stored
    # as fp32 here, the oracle reads the same numbers); LayerNorm / embedding parameters stay generic fp32.  --c5-fp32-weights keeps
    # generic fp32 values everywhere (a fine-tuned tower: the three-product kernel).
    P_generic = dict(P)
    if not args.c5_fp32_weights:
        for k in list(P):
            if k.endswith(("conv1.weight", "in_proj_weight", "in_proj_bias", "out_proj.weight", "out_proj.bias", "c_fc.weight", "c_fc.bias",
                           "c_proj.weight", "c_proj.bias")) or k == "visual.proj":
                P[k] = P[k].to(torch.float16).to(torch.float32)
    enc = ClipImageEncoder(P, p, prefix="visual.", precision=args.precision)
    x = torch.randn((B, 3, 336, 336), generator=torch.Generator(device="cpu").manual_seed(2000 + rank)).to(dev)
    # Steps are independent batches (the extraction loop, extract_image_embeddings.py:70-80): `--inflight N` keeps N of them in flight
    # on N HIP streams, each on its own fork of the engine (shared packed weights, own activation buffers).  Measured, same box:
    # 1225 / 1234 / 1205 images/s for 1 / 2 / 3 in flight — a step here is 200 ms of 0.7 - 2.2-ms GEMMs that own the chip, there
    # are no launch gaps or short tails for a second batch to fill — so the c5 default is ONE (the plain loop).
    n_lanes = max(1, args.inflight if args.inflight_given else 1)
    lanes = [enc] + [enc.fork() for _ in range(n_lanes - 1)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(n_lanes)]
    embs = [None] * n_lanes
    torch.cuda.synchronize()

    def step(i):
        l = i % n_lanes
        with torch.cuda.stream(streams[l]):
            embs[l] = lanes[l].encode_image(x)
        return l
    for i in range(max(n_lanes, args.warmup)):
        step(i)
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        last = step(i)
    torch.cuda.synchronize()
    emb = embs[last]
    if dist_on:
        allemb = torch.empty((world * B, E), dtype=torch.float32, device=dev)
        dist.all_gather_into_tensor(allemb, emb)
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist_on:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # ---- the pipeline's second half (datasets/index_dataset.py:158-167): per-category top-500 retrieval over the extracted embeddings.
    # Rank r holds the embeddings of images [r*B, (r+1)*B) of the last step; every rank takes the exact top-k of ITS shard, the [C, k]
    # candidates are all-gathered (the only collective of this config besides the embeddings gather) and merged identically everywhere.
    # Outside the timed region (the metric is extraction rate); timed on its own and checked against the unsharded form on rank 0.
    from zutis_amd import retrieval as zr
    Ccat, ktop = 919, 500
    tcat = torch.nn.functional.normalize(torch.randn((Ccat, E), generator=torch.Generator(device="cpu").manual_seed(77)), dim=1).to(dev)
    retr = None
    if dist_on:
        zr.retrieve_topk_sharded(tcat, emb, rank * B, ktop)
        torch.cuda.synchronize(); dist.barrier()
        t1 = time.perf_counter()
        ridx, rval = zr.retrieve_topk_sharded(tcat, emb, rank * B, ktop)
        torch.cuda.synchronize(); dist.barrier()
        dtr = time.perf_counter() - t1
        same = None
        if rank == 0:
            fidx, fval = zr.retrieve_topk(tcat, allemb, ktop)             # the gathered embeddings, unsharded
            same = bool(torch.equal(fidx, ridx) and torch.equal(fval, rval))
        retr = {"form": "sharded: local exact top-k + all-gather of [C, k] candidates + merge", "ms": round(dtr * 1e3, 3),
                "equals_unsharded_on_rank0": same}
    else:
        zr.retrieve_topk(tcat, emb, ktop)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        ridx, rval = zr.retrieve_topk(tcat, emb, ktop)
        torch.cuda.synchronize()
        retr = {"form": "one rank: similarity GEMM (f16x3) + exact radix top-k", "ms": round((time.perf_counter() - t1) * 1e3, 3)}
    retr.update({"categories": Ccat, "k": min(ktop, world * B), "images": world * B,
                 "what": "top-k image indices per category over the last step's embeddings (datasets/index_dataset.py:158-167), outside the timed region"})
    T = g * g + 1
    flop = L * (2 * T * (D * 3 * D + D * D + 2 * D * 4 * D) + 4 * T * T * D) + 2 * g * g * 3 * p * p * D + 2 * D * E
    roof = cpu = parity = None
    if rank == 0:
        roof = gemm_roofline(ops, lambda: enc.encode_image(x), elapsed / args.steps)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import zutis_ref as O
        torch.set_num_threads(max(1, min(args.cpu_threads, os.cpu_count() or 1)))
        Pc = {k.replace("visual.", "encoder."): v.cpu() for k, v in P.items()}
        ns = max(1, min(4, B))
        xs = x[:ns].cpu()
        with torch.no_grad():
            O.clip_encode_image(Pc, xs[:1], p)                            # warm-up
            times = []
            for _ in range(3):
                t1 = time.perf_counter()
                ref = O.clip_encode_image(Pc, xs, p)
                times.append(time.perf_counter() - t1)
        dt = sorted(times)[1]
        cpu = {"value": round(ns / dt, 3), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
               "sample": f"{ns} of the {B} step images, oracle encode_image (24-layer ViT-L/14@336), median of 3 passes "
                         f"({', '.join('%.1f' % t for t in times)} s); host has {os.cpu_count()} hardware threads"}
        got = enc.encode_image(x[:ns]).cpu()
        parity = {"embedding_max_abs_err": float((got - ref).abs().max()), "tolerance": 1e-3,
                  "against": "fp32 oracle on the same %d images (unit-norm embeddings)" % ns}
    generic = None
    if rank == 0 and world == 1 and not args.c5_fp32_weights and not args.no_second_precision and args.precision == "exact":
        # the same tower with generic fp32 VALUES in the GEMM weights (a fine-tuned tower): every weight keeps its lo plane, the
        # three-product kernel runs — reported next to the headline so that both cases are on the line
        del lanes, embs
        enc3 = ClipImageEncoder(P_generic, p, prefix="visual.", precision=args.precision)
        enc3.encode_image(x)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        n3 = max(2, min(5, args.steps))
        for _ in range(n3):
            e3 = enc3.encode_image(x)
        torch.cuda.synchronize()
        dt3 = (time.perf_counter() - t1) / n3
        generic = {"value": round(B / dt3, 1), "unit": "images/s", "ms_per_step": round(dt3 * 1e3, 3), "steps": n3,
                   "what": "generic fp32 values in every GEMM weight (--c5-fp32-weights): zh_gemm_f16x3 with both weight planes, three MFMA "
                           "products per accumulator"}
        del enc3, e3
    second = None
    if rank == 0 and world == 1 and not args.no_second_precision and args.precision in ("exact", "fast"):
        # the other precision on the same line.  The reference itself runs THIS config in half precision on a GPU (clip.load leaves the
        # model in fp16 unless the device is the CPU; extract_image_embeddings.py:76 converts the fp16 embeddings back): `fast` (fp16 MFMA
        # operands in the transformer body, fp32 accumulate / residual stream / LayerNorm / softmax) is its arithmetic class and the
        # headline; `exact` (fp32-class split pairs) is MORE precise than the reference here
        oprec = "fast" if args.precision == "exact" else "exact"
        try:
            del lanes, embs
        except NameError:
            pass
        encf = ClipImageEncoder(P, p, prefix="visual.", precision=oprec)
        ef = encf.encode_image(x)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        nf = max(2, min(5, args.steps))
        for _ in range(nf):
            ef = encf.encode_image(x)
        torch.cuda.synchronize()
        dtf = (time.perf_counter() - t1) / nf
        second = {"mode": oprec, "dtype": PRECISION_DTYPE[oprec], "value": round(B / dtf, 1), "unit": "images/s", "ms_per_step": round(dtf * 1e3, 3),
                  "steps": nf, "embedding_max_abs_diff_vs_headline": float((ef - emb).abs().max()),
                  "note": "the reference runs config 5 in fp16 on a GPU (third-party clip.load; extract_image_embeddings.py:76): fast is its "
                          "arithmetic class, exact is fp32-class"}
        del encf, ef
    if dist_on:
        dist.barrier()                    # rank 0 measured the roofline after the timed region: leave together
        dist.destroy_process_group()
    if rank == 0:
        total = world * B * args.steps
        print(json.dumps({
            "metric": "images/sec, CLIP ViT-L/14@336 image-embedding extraction (BASELINE config 5)" +
                      (f", {n_lanes} independent batches in flight" if n_lanes > 1 else ""), "value": round(total / elapsed, 1),
            "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": PRECISION_DTYPE[args.precision], "data": "synthetic",
            "precision": {"mode": args.precision, "what": PRECISION_TEXT[args.precision]},
            "config": {"workload": ("" if L == 24 else f"NOT CONFIG 5 ({L} layers, --c5-layers): ") +
                                   f"C5: CLIP ViT-L/14@336 encode_image, {B}x3x336x336 per GPU per step, embeddings fp32 [{B},{E}], "
                                   "one all-gather of the last step's embeddings", "global_batch": world * B, "parallelism": f"dp{world}",
                       "flops_per_image": flop,
                       "weights": ("generic fp32 values in every tensor (--c5-fp32-weights)" if args.c5_fp32_weights else
                                   "fp16-VALUED conv / Linear / attention / proj tensors, as the reference's build_model -> convert_weights "
                                   "leaves them (clip_arch.py:566-587,625); the engine detects it per weight at pack time and skips the "
                                   "product with the all-zero lo plane (f16x2: bit-identical to f16x3)")},
            "model_tflops": round(total * flop / elapsed / 1e12 / world, 1), "roofline": roof, "cpu_baseline": cpu, "parity": parity,
            "generic_fp32_weights": generic, "second_precision": second, "retrieval": retr,
            "embedding_norm": round(float(emb.norm(dim=1).mean().item()), 6)}), flush=True)


def c3_model(dev, precision):
    """The config-3 fixture as the drop-in module: weights / text rows / threshold of tests/golden/c3_vitb16.npz (generated from the
    reference: 100 candidates, 9 categories, 17 hard-NMS survivors at 480x640), one 480x640 image.  Returns (net, x, golden, thr, H, W)."""
    root = os.path.dirname(os.path.abspath(__file__))
    dp = os.path.join(root, "zutis_amd", "dropin")
    if dp not in sys.path:
        sys.path.insert(0, dp)
    from zutis_amd import detgen
    from networks.zutis import ZUTIS
    cfg = detgen.VIT_B16
    g = np.load(os.path.join(root, "tests", "golden", "c3_vitb16.npz"))
    H, W, thr = 480, 640, detgen.C3_THRESHOLD
    with contextlib.redirect_stdout(sys.stderr):      # the constructor prints "clip is loaded." like the reference's (zutis.py:105): keep stdout to the one JSON line
        net = ZUTIS(categories=[f"c{i}" for i in range(81)], clip_arch="ViT-B/16", device=dev, text_embeddings=torch.from_numpy(g["text"]))
    net.load_state_dict({k: torch.from_numpy(v) for k, v in detgen.c3_state_dict(cfg).items()}, strict=True)
    net = net.to(dev).eval().requires_grad_(False)
    net.precision = precision
    x = torch.from_numpy(detgen.images(1, H, W, seed=21)).to(dev)
    return net, x, g, thr, H, W


def c3_parity(preds, g, tag):
    """The step's predictions against the reference's own for this image (fixture generated by oracle/gen_golden.py from /root/reference)."""
    from zutis_amd import rle
    ref_cat, ref_score, ref_area = g[f"{tag}_cat"], g[f"{tag}_score"], g[f"{tag}_area"]
    cats = [p["category_id"] for p in preds]
    areas = [int(rle.decode(p["segmentation"]).sum()) for p in preds]
    same_list = cats == list(ref_cat)
    return {"predictions": len(preds), "reference_predictions": int(len(ref_cat)), "category_list_identical": bool(same_list),
            "score_max_abs_err": (float(np.abs(np.array([p["score"] for p in preds]) - ref_score).max()) if same_list else None),
            "mask_area_max_abs_diff_sorted_per_category": (int(max(abs(a - b) for c in set(cats) for a, b in zip(
                sorted(a for a, cc in zip(areas, cats) if cc == c), sorted(int(a) for a, cc in zip(ref_area, ref_cat) if cc == c)))) if same_list else None),
            "against": "tests/golden/c3_vitb16.npz: the reference's ZUTIS.forward + predict(instance, hard NMS) on the same image and weights "
                       "(tests/test_configs_gpu.py::test_c3_native_resolution_instance_predict holds scores to 5e-4, areas to 8 px)"}


def batch1_object(precision, dev, steps=40):
    """What every unchanged caller of the reference runs — one image per call at its native resolution (configs/*.yaml val batch_size 1;
    trainer.py:328-345 and coco20k_eval.py:258-267: forward, then predict per image) — through the drop-in module, bounded to a fraction
    of a second: ms per image of forward + instance predict (hard NMS, RLE dicts), of the forward alone and of the semantic predict,
    launches per image, parity with the reference's predictions for this image."""
    from zutis_amd import _lib
    net, x, g, thr, H, W = c3_model(dev, precision)
    inst = lambda o: net.predict(o, mask_type="instance", threshold=thr, size=(H, W), image_ids=[7], nms_type="hard")

    def timed(fn, n):
        for _ in range(3):
            r = fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            r = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3, r
    ms_step, preds = timed(lambda: inst(net(x)), steps)
    ms_fwd, out = timed(lambda: net(x), steps)
    ms_sem, _ = timed(lambda: net.predict(out, mask_type="semantic", size=(H, W)), steps)
    # the body of trainer.evaluate's loop for coco2017 / voc2012 (trainer.py:327-348): forward, semantic predict, instance predict, the metric
    # meter's update with the ground truth (a host int64 array, as the loader hands it over) and get_scores(), every image
    from utils.running_score import RunningScore
    meter = RunningScore(81, device=dev)
    gt = np.random.default_rng(3).integers(0, 81, (1, H, W)).astype(np.int64)

    def trainer_body():
        o = net(x)
        sem = net.predict(o, mask_type="semantic", size=(H, W))
        r = inst(o)
        meter.update(gt, sem)
        meter.get_scores()
        return r
    ms_loop, _ = timed(trainer_body, steps)
    graph = bool(net.use_hip_graph)
    counts = {}
    net.use_hip_graph = False                      # count the C-ABI launches of one eager forward / predict (a graph replays the same ones)
    _lib.COUNTER = counts
    try:
        o = net(x)
        n_fwd = sum(counts.values())
        inst(o)
        n_all = sum(counts.values())
    finally:
        _lib.COUNTER = None
        net.use_hip_graph = graph
    return {"what": "ONE 480x640 image per call through the drop-in networks.zutis.ZUTIS (the reference's evaluation regime: val batch_size 1, "
                    "trainer.py:328-345, coco20k_eval.py:258-267): forward + predict(instance, hard NMS) to COCO RLE dicts",
            "ms_per_image": round(ms_step, 3), "images_per_s": round(1e3 / ms_step, 1), "forward_ms": round(ms_fwd, 3),
            "instance_predict_ms": round(ms_step - ms_fwd, 3), "semantic_predict_ms": round(ms_sem, 3),
            "trainer_evaluate_body_ms": round(ms_loop, 3),
            "trainer_evaluate_body_what": "trainer.py:327-348 per image: forward + predict(semantic) + predict(instance, hard NMS) + RunningScore.update(host "
                                          "ground truth, predictions) + get_scores()",
            "steps": steps,
            "hip_graph_replay": graph, "precision": precision, "library_calls_forward": n_fwd, "library_calls_instance_predict": n_all - n_fwd,
            "library_calls_note": "C-ABI entry-point calls (zh_*) of one eager forward / predict; a few launch two kernels (split attention + "
                                  "merge, global LayerNorm, IoU pack + counts, run extraction): profiles/r04_c3_launch_list.txt lists the kernels",
            "parity": c3_parity(preds, g, f"{H}x{W}")}


def bench_c3(args):
    """Config 3: COCO-20K-style instance segmentation at its own shape and batch — coco20k_eval.py:241-268 evaluates image
    by image — through the drop-in `networks.zutis.ZUTIS`: a step = ONE 480x640 image, forward + predict(mask_type="instance",
    nms_type="hard", size=(H, W)) down to the list of COCO prediction dicts (RLE strings, boxes: host objects, as in the
    reference).  Weights / text rows / threshold are the config-3 fixture's (tests/golden/c3_vitb16.npz: 100 candidates, 9 categories,
    17 hard-NMS survivors — generated from the reference), so the step's predictions are checked against the reference's own.
    The second half of the config, the bilateral-solver refinement (utils/bilateral_solver.py; 512x683 as in the pseudo-label
    pipeline), is timed as its own object with an HBM roofline.  Ranks evaluate independent images: no collective in the path."""
    world = int(os.environ.get("WORLD_SIZE", "1")); rank = int(os.environ.get("RANK", "0")); local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)
    from zutis_amd import detgen, ops, rle
    cfg = detgen.VIT_B16
    net, x, g, thr, H, W = c3_model(dev, args.precision)
    tag = f"{H}x{W}"

    def step():
        out = net(x)
        return net.predict(out, mask_type="instance", threshold=thr, size=(H, W), image_ids=[7], nms_type="hard")
    for _ in range(max(3, args.warmup)):
        preds = step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        preds = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    roof = cpu = parity = solver = None
    if rank == 0:
        graph = net.use_hip_graph
        net.use_hip_graph = False                  # the per-launch events need the eager launches (a graph replay bypasses ops.PROFILER)
        try:
            roof = gemm_roofline(ops, step, elapsed / args.steps)
        finally:
            net.use_hip_graph = graph
        roof["measured_on"] += "; the timed steps replay the forward from a hipGraph (drop-in default for batches <= 4)" if graph else ""
        parity = c3_parity(preds, g, tag)
    if rank == 0 and world == 1:
        # ---- bilateral solver at the pseudo-label size: one image per call and 8 per call (zh_bilateral_solve_batch)
        solver = solver_object(dev)
        yy, xx = np.mgrid[:512, :683]
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import torch.nn.functional as F
        from oracle import zutis_ref as O
        from oracle import bilateral_ref as OB
        torch.set_num_threads(max(1, min(args.cpu_threads, os.cpu_count() or 1)))
        Pc = O.to_torch_params(detgen.c3_state_dict(cfg))
        textc = torch.from_numpy(g["text"])
        xc = x.cpu()

        def cpu_step():
            o = O.zutis_forward(Pc, xc, cfg.patch, cfg.dec_heads)
            mp = o["mask_proposals"][:, -1]
            _, cat, score = O.instance_scores(o["mask_proposals"], o["patch_tokens"], textc, threshold=thr)
            masks = (F.interpolate(mp, size=(H, W), mode="bilinear") > thr).numpy()
            kept = O.mask_nms(masks[0], score[0], cat[0], "hard")
            return [rle.encode(np.asfortranarray(masks[0][m]).astype(np.uint8)) for _, m, _ in kept]
        with torch.no_grad():
            cpu_step()
            times = []
            for _ in range(3):
                t1 = time.perf_counter(); kept = cpu_step(); times.append(time.perf_counter() - t1)
        dt = sorted(times)[1]
        cpu = {"value": round(1.0 / dt, 3), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
               "sample": f"the step's image, oracle forward + instance predict + hard NMS + RLE ({len(kept)} kept), median of 3 passes "
                         f"({', '.join('%.1f' % t for t in times)} s); host has {os.cpu_count()} hardware threads"}
        if solver is not None:
            rgb1 = detgen.selfmask_like_rgb(512, 683, seed=3)
            tg1 = (((yy - 250) ** 2 + (xx - 300) ** 2) < 150 ** 2).astype(np.uint8)
            t1 = time.perf_counter(); OB.bilateral_solver_output(rgb1, tg1); dts = time.perf_counter() - t1
            solver["cpu_baseline"] = {"ms_per_image": round(dts * 1e3, 1), "kind": "port", "cores": 1,
                                      "sample": "oracle/bilateral_ref.py (numpy / scipy.sparse restatement of utils/bilateral_solver.py), one 512x683 image"}
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        total = world * args.steps
        print(json.dumps({
            "metric": "images/sec, COCO-20K-shaped instance segmentation, ViT-B/16, one 480x640 image per step: ZUTIS forward + instance predict "
                      "with hard mask NMS to COCO RLE dicts (BASELINE config 3)", "value": round(total / elapsed, 1),
            "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": PRECISION_DTYPE[args.precision], "data": "synthetic",
            "precision": {"mode": args.precision, "what": PRECISION_TEXT[args.precision]},
            "config": {"workload": f"C3: batch 1, {H}x{W}, 81 categories, 100 queries, threshold {thr}, hard NMS; drop-in networks.zutis.ZUTIS "
                                   "(coco20k_eval.py:241-268 evaluates image by image)", "global_batch": world, "parallelism": f"dp{world}"},
            "roofline": roof, "cpu_baseline": cpu, "parity": parity, "bilateral_solver": solver}), flush=True)


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


def solver_object(dev, reps=20):
    """Bilateral-solver refinement (utils/bilateral_solver.py; BASELINE config 3's second half) at the pseudo-label size 512x683, natural-image
    colour statistics: ms per image at 1 and 8 images per call, HBM roofline on SURVEY 8d's algorithmic bytes."""
    from zutis_amd import detgen, ops
    Hs, Ws = 512, 683
    yy, xx = np.mgrid[:Hs, :Ws]
    solver = {"size": [Hs, Ws], "unit": "ms per image", "bound": "hbm", "peak_TBps": 8.0,
              "algorithmic_bytes_note": "N*(3+1+8+16) + V*250*(25 CG + 11 bistochastisation iterations) per image (SURVEY 8d)"}
    for Bs in (1, 8):
        rgb = torch.from_numpy(np.stack([detgen.selfmask_like_rgb(Hs, Ws, seed=3 + i) for i in range(Bs)])).to(dev)
        tg = torch.from_numpy(np.stack([(((yy - 250) ** 2 + (xx - 300 - 3 * i) ** 2) < 150 ** 2).astype(np.uint8) for i in range(Bs)])).to(dev)
        soft, stats = ops.bilateral_solve(rgb, tg)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(reps):
            ops.bilateral_solve(rgb, tg)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t1) / reps
        V = float(stats[:, 0].float().mean().item())
        byts = Hs * Ws * (3 + 1 + 8 + 16) + V * 250 * 36
        solver[f"batch{Bs}"] = {"ms_per_image": round(dt / Bs * 1e3, 4), "vertices": round(V), "achieved_TBps": round(byts * Bs / dt / 1e12, 3),
                                "frac": round(byts * Bs / dt / 8e12, 3), "cg_iterations": [int(v) for v in stats[:, 1].tolist()[:2]]}
    return solver


def pseudo_label_object(dev, precision="exact", B=4, reps=5):
    """The pseudo-label path north_star names (SelfMask, networks/selfmask + utils/bilateral_solver.py, as datasets/*.py
    generate_pseudo_masks drives them): DINO ViT-S/8 SelfMask at its working shape 512x683 (T = 5505 tokens) -> query selection ->
    bilateral solver -> > 0.5 -> nearest resize to 480x640, `B` images per call, device side (the RLE JSON files are host work).
    Parity of this path is held by tests/test_e2e_gpu.py::test_selfmask_* (reference goldens + the oracle at 512x683) and
    tests/test_bilateral_gpu.py; synthetic noise images give the solver one lattice vertex per pixel (17x a natural image's)."""
    from zutis_amd import detgen, pseudo_masks
    from zutis_amd.engine import SelfMaskEngine
    H, W = 512, 683
    eng = SelfMaskEngine({k: torch.from_numpy(v).to(dev) for k, v in detgen.selfmask_state_dict().items()}, precision=precision)
    x = torch.from_numpy(detgen.images(B, H, W, seed=7)).to(dev)
    for _ in range(2):
        pseudo_masks.pseudo_masks_batch(eng, x, [(480, 640)] * B, True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        pseudo_masks.pseudo_masks_batch(eng, x, [(480, 640)] * B, True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    eng._bufs.clear()
    return {"what": f"SelfMask (DINO ViT-S/8 @{H}x{W}, T = 5505) + bilateral solver + threshold + nearest resize, {B} images per call, device side",
            "value": round(B / dt, 1), "unit": "images/s", "ms_per_call": round(dt * 1e3, 2), "batch": B, "precision": precision, "calls": reps}


def c4_object(P, cfg, dev, precision, steps=12, warmup=4, n_lanes=3, cpu_images=1, cpu_threads=16):
    """BASELINE config 4 on one GPU, bounded: ViT-B/16 @518 px, 920 classes, 8 images per step (the reference's own batch for this
    config, configs/imagenet_s919_*.yaml), three launch plans in flight exactly as the headline — value, roofline fraction of the
    dominant GEMM, timed outputs bitwise an eager step, parity of the TIMED outputs against the oracle on `cpu_images` image(s)."""
    from zutis_amd import detgen, ops
    from zutis_amd import distributed as zd
    from zutis_amd.engine import ZutisEngine
    B, S, n = 8, 518, 920
    text = torch.from_numpy(detgen.text_embeddings(n, cfg.embed_dim)).to(dev)
    x = torch.randn((B, 3, S, S), generator=torch.Generator(device="cpu").manual_seed(4000)).to(dev)
    eng = ZutisEngine(P, cfg.patch, cfg.dec_heads, precision=precision)
    lanes = build_lanes(eng, x, text, S, n, n_lanes)
    torch.cuda.synchronize()
    pipe = zd.StepPipeline(lanes, make_launch(n_lanes), gather=False)
    pipe.run(max(warmup, n_lanes))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pipe.run(steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ok, lo_t, lab_t = check_timed_outputs(lanes)

    def one_eager_step():
        out = eng.forward(x)
        eng.predict_semantic(out["patch_tokens"], text, (S, S))
    roof = gemm_roofline(ops, one_eager_step, dt / steps)
    obj = {"what": f"C4: ViT-B/16 @{S}px, {n} classes, {B} images per step, {n_lanes} launch plans in flight (`bench.py --workload c4` is the full line)",
           "value": round(B * steps / dt, 1), "unit": "images/s", "ms_per_step": round(dt / steps * 1e3, 3), "steps": steps, "precision": precision,
           "dtype": PRECISION_DTYPE[precision], "timed_outputs_bitwise_equal_eager": bool(ok),
           "roofline": {k: roof[k] for k in ("kernel", "achieved", "peak", "unit", "frac", "avg_launch_us", "launches_per_step", "gemm_share_of_step")}}
    for an in ("attention_f16x3_tflops", "attention_f16x3_share_of_step", "attention_f16_tflops", "attention_f16_share_of_step"):
        if an in roof:
            obj["roofline"][an] = roof[an]
    if cpu_images:
        from oracle import zutis_ref as O
        from oracle.parity import unexplained_label_mismatches
        from oracle import resample as R
        torch.set_num_threads(max(1, min(cpu_threads, os.cpu_count() or 1)))
        Pc = O.to_torch_params(detgen.zutis_state_dict(cfg))
        with torch.no_grad():
            t1 = time.perf_counter()
            o = O.zutis_forward(Pc, x[:cpu_images].cpu(), cfg.patch, cfg.dec_heads)
            lo_ref = O.semantic_logits_lowres(o["patch_tokens"], text.cpu()).numpy()
            lab_ref = R.bilinear_argmax_nchw(lo_ref, S, S)
            dtc = time.perf_counter() - t1
        lo = lo_t[:cpu_images].cpu().numpy()
        lab = lab_t[:cpu_images].cpu().numpy()
        err = float(np.abs(lo - lo_ref).max())
        n_mis, n_bad, worst = unexplained_label_mismatches(lab, lab_ref, lo_ref, err, (S, S))
        obj["parity"] = {"logit_max_abs_err": err, "label_mismatches": n_mis, "unexplained_label_mismatches": n_bad, "tolerance": 1e-3,
                         "against": f"fp32 oracle on the first {cpu_images} image(s) of the timed batch, outputs of the timed plans (lane 0)"}
        obj["cpu_baseline"] = {"value": round(cpu_images / dtc, 3), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
                               "sample": f"{cpu_images} image(s), one pass (oracle forward + 920-class semantic predict)"}
    for ln in lanes:
        ln.state["eng"]._bufs.clear()
    return obj


def c5_object(dev, precision="fast", steps=3, cpu_images=1, cpu_threads=16):
    """BASELINE config 5 on one GPU, bounded: CLIP ViT-L/14@336 `encode_image`, ONE 256-image batch per step, at the reference's own
    arithmetic class for this config (fp16 on a GPU: utils/extract_image_embeddings.py:43,72-76 -> `fast`).  Weights are random values of
    the architecture drawn on the device (fp16-valued conv / Linear / attention / proj tensors as convert_weights leaves them); the full
    line with the deterministic host-generated weights is `bench.py --workload c5`."""
    from zutis_amd import ops
    from zutis_amd.engine import ClipImageEncoder
    D, L, p, g, E, B = 1024, 24, 14, 24, 768, 256
    gen = torch.Generator(device=dev).manual_seed(5005)

    def w(shape, std, mean=0.0, f16v=True):
        t = torch.randn(shape, generator=gen, device=dev, dtype=torch.float32) * std + mean
        return t.half().float() if f16v else t
    P = {"visual.class_embedding": w((D,), D ** -0.5, f16v=False), "visual.positional_embedding": w((g * g + 1, D), D ** -0.5, f16v=False),
         "visual.proj": w((D, E), D ** -0.5), "visual.conv1.weight": w((D, 3, p, p), (3 * p * p) ** -0.5)}
    for ln in ("ln_pre", "ln_post"):
        P[f"visual.{ln}.weight"], P[f"visual.{ln}.bias"] = w((D,), 0.1, 1.0, False), w((D,), 0.1, 0.0, False)
    for i in range(L):
        q = f"visual.transformer.resblocks.{i}."
        P[q + "attn.in_proj_weight"], P[q + "attn.in_proj_bias"] = w((3 * D, D), D ** -0.5), w((3 * D,), 0.02)
        P[q + "attn.out_proj.weight"], P[q + "attn.out_proj.bias"] = w((D, D), D ** -0.5 * (2 * L) ** -0.5), w((D,), 0.02)
        P[q + "mlp.c_fc.weight"], P[q + "mlp.c_fc.bias"] = w((4 * D, D), (2 * D) ** -0.5), w((4 * D,), 0.02)
        P[q + "mlp.c_proj.weight"], P[q + "mlp.c_proj.bias"] = w((D, 4 * D), D ** -0.5 * (2 * L) ** -0.5), w((D,), 0.02)
        for ln in ("ln_1", "ln_2"):
            P[q + ln + ".weight"], P[q + ln + ".bias"] = w((D,), 0.1, 1.0, False), w((D,), 0.1, 0.0, False)
    enc = ClipImageEncoder(P, p, prefix="visual.", precision=precision)
    x = torch.randn((B, 3, 336, 336), generator=gen, device=dev, dtype=torch.float32)
    emb = enc.encode_image(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        emb = enc.encode_image(x)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    roof = gemm_roofline(ops, lambda: enc.encode_image(x), dt)
    T = g * g + 1
    flop = L * (2 * T * (D * 3 * D + D * D + 2 * D * 4 * D) + 4 * T * T * D) + 2 * g * g * 3 * p * p * D + 2 * D * E
    obj = {"what": f"C5: CLIP ViT-L/14@336 encode_image, one {B}-image batch per step, one stream (`bench.py --workload c5` is the full line)",
           "value": round(B / dt, 1), "unit": "images/s", "ms_per_step": round(dt * 1e3, 2), "steps": steps, "precision": precision,
           "dtype": PRECISION_DTYPE[precision], "model_tflops": round(B * flop / dt / 1e12, 1),
           "precision_note": "the reference runs this config in fp16 on a GPU (clip.load; extract_image_embeddings.py:43,72-76): `fast` is its arithmetic class",
           "roofline": {k: roof[k] for k in ("kernel", "achieved", "peak", "unit", "frac", "avg_launch_us", "launches_per_step", "gemm_share_of_step")}}
    for an in ("attention_f16x3_tflops", "attention_f16_tflops"):
        if an in roof:
            obj["roofline"][an] = roof[an]
    if cpu_images:
        from oracle import zutis_ref as O
        torch.set_num_threads(max(1, min(cpu_threads, os.cpu_count() or 1)))
        Pc = {k.replace("visual.", "encoder."): v.cpu() for k, v in P.items()}
        with torch.no_grad():
            t1 = time.perf_counter()
            ref = O.clip_encode_image(Pc, x[:cpu_images].cpu(), p)
            dtc = time.perf_counter() - t1
        got = emb[:cpu_images].cpu()             # rows of the TIMED step's output (the last of the timed encode_image calls)
        obj["parity"] = {"embedding_max_abs_err": float((got - ref).abs().max()), "tolerance": 1e-3,
                         "against": f"fp32 oracle on the first {cpu_images} image(s) of the timed batch (unit-norm embeddings); the compared rows are the timed step's output"}
        obj["cpu_baseline"] = {"value": round(cpu_images / dtc, 3), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
                               "sample": f"{cpu_images} image(s), one pass (oracle encode_image, 24-layer ViT-L/14@336)"}
    enc._bufs.clear()
    return obj


def rank_launch_command(n_gpus: int, argv, port: int):
    """The command line `python bench.py --gpus N` runs as a child: one rank per GPU under torch.distributed.run, rendezvous on
    127.0.0.1 (the container hostname may not resolve) — the same form the driver uses for its own N > 1 launches."""
    child_args = [a for a in argv if a != "--dry-launch"]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + child_args


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50, help="timed steps (default 50: SURVEY 8d asks for >= 50; 0.54 s of `exact` steps)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32, help="images per GPU per step (BASELINE config[1]: batch 32)")
    ap.add_argument("--size", type=int, default=336)
    ap.add_argument("--classes", type=int, default=81)
    ap.add_argument("--workload", default="c2", choices=["c2", "c3", "c4", "c5"],
                    help="c2 (default, the headline): ViT-B/16 @336, 81 classes, 32 / GPU.  c4: @518, 920 classes, 8 / GPU "
                         "(BASELINE config 4).  c5: CLIP ViT-L/14@336 image-embedding extraction, 256 / GPU / step (config 5).  "
                         "c3: instance segmentation image by image at 480x640 + the bilateral solver at 512x683 (config 3)")
    ap.add_argument("--c5-fp32-weights", action="store_true", help="c5: generic fp32 values in the GEMM weights instead of the fp16 values the "
                    "reference's build_model -> convert_weights leaves there (forces the three-product kernel)")
    ap.add_argument("--c5-layers", type=int, default=24, help="developer (tests): depth of the c5 tower; anything but 24 is not config 5")
    ap.add_argument("--inflight", type=int, default=3, help="independent steps in flight (HIP streams); 1 = eager, one stream")
    ap.add_argument("--force-dist", action="store_true", help="developer: run the N>1 code path (RCCL group + per-step all-gather) on one rank")
    ap.add_argument("--precision", default=None, choices=["fast", "exact", "f16"],
                    help="engine precision (zutis_amd/engine.py): exact (default, the headline) = every contraction in the f16x3 mode, the "
                         "reference's fp32 arithmetic class; fast = fp16 MFMA operands in the transformer bodies + x3 on the output-facing "
                         "contractions (passes tests/test_precision_gpu.py at the north-star 1e-3; reported as second_precision)")
    ap.add_argument("--no-live-traffic", action="store_true", help="do not spawn the two rocprofv3 --pmc child passes that measure roofline.traffic")
    ap.add_argument("--no-second-precision", action="store_true", help="skip the secondary timed run at the other precision (N = 1)")
    ap.add_argument("--h2d", action="store_true", help="developer: every step first copies its batch from pinned host memory (async, on the "
                    "step's stream) — the PCIe-inclusive rate quoted in DESIGN.md; the headline keeps inputs resident in HBM")
    ap.add_argument("--d2h", action="store_true", help="developer: every step ends with its int64 label maps copied to pinned host memory on the "
                    "step's stream (the reference's predict ends in .cpu().numpy(), networks/zutis.py:372)")
    ap.add_argument("--no-io-rates", action="store_true", help="skip the short extra runs that report the PCIe-inclusive rates (N = 1)")
    ap.add_argument("--no-batch1", action="store_true", help="skip the bounded batch-1 object (one 480x640 image per call through the drop-in module)")
    ap.add_argument("--no-configs", action="store_true", help="skip the bounded objects for BASELINE configs 4 / 5 and the bilateral solver (N = 1, default workload)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cpu-all-cores", action="store_true", help="skip the extra CPU-baseline pass with one thread per physical core")
    ap.add_argument("--no-torch-gpu-baseline", action="store_true")
    ap.add_argument("--torch-gpu-baseline", action="store_true", default=True,
                    help="also time the oracle (= the reference's op sequence) with stock PyTorch-ROCm fp32 eager ops on this GPU "
                         "(SURVEY 8d: the 'reference single-GPU PyTorch' the north-star's >= 10x target is quoted against)")
    ap.add_argument("--cpu-sample", type=int, default=16, help="images in the CPU-baseline sample (timed three times)")
    ap.add_argument("--cpu-threads", type=int, default=16,
                    help="torch intra-op threads of the CPU baseline (16 was the fastest of 8..128 on the 2x64-core GPU box)")
    ap.add_argument("--dry-launch", action="store_true", help="print the rank launcher's command line (JSON) for --gpus N and exit; no GPU is touched")
    args = ap.parse_args()
    args.inflight_given = any(a == "--inflight" or a.startswith("--inflight=") for a in sys.argv[1:])
    if args.precision is None:
        # the reference's own arithmetic class per config: fp32 for the ZUTIS network (zutis.py:55 casts the encoder back to fp32) -> exact;
        # config 5 runs third-party clip's HALF-precision tower on a GPU (utils/extract_image_embeddings.py:43,72-76) -> fast (fp16 MFMA operands)
        args.precision = "fast" if args.workload == "c5" else "exact"
    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or args.dry_launch):
        return launch_ranks(args.gpus, sys.argv[1:], args.dry_launch)
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus and not args.force_dist:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={os.environ.get('WORLD_SIZE', '1')}: launch with "
                         f"--nproc-per-node {args.gpus} (or run plain `python bench.py --gpus {args.gpus}`, which starts the ranks itself)")
    if args.workload == "c4":
        args.size, args.classes = 518, 920
        if args.batch == 32:
            args.batch = 8
    if args.workload == "c5":
        return bench_c5(args)
    if args.workload == "c3":
        return bench_c3(args)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    dist_on = world > 1 or args.force_dist
    if dist_on:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)   # nccl == RCCL on ROCm

    from zutis_amd import detgen, ops
    from zutis_amd import plan as zplan
    from zutis_amd import distributed as zd
    from zutis_amd.engine import ZutisEngine

    cfg = detgen.VIT_B16
    B, S, n = args.batch, args.size, args.classes
    P = {k: torch.from_numpy(v).to(dev) for k, v in detgen.zutis_state_dict(cfg).items()}
    text = torch.from_numpy(detgen.text_embeddings(n, cfg.embed_dim)).to(dev)
    # rank r owns global images [r*B, (r+1)*B): contiguous shards so a gather reproduces reference order
    g = torch.Generator(device="cpu").manual_seed(1000 + rank)
    x = torch.randn((B, 3, S, S), generator=g).to(dev)
    hw2 = (2 * ((S - cfg.patch) // cfg.patch + 1)) ** 2
    n_lanes = max(1, args.inflight)

    engines = {}

    def timed_run(precision: str, steps: int, warmup: int, h2d: bool = False, d2h: bool = False, check: bool = False):
        """Builds the engine for `precision`, one lane (engine fork + launch plan + stream + gather buffer) per step in
        flight, and times `steps` steps through zutis_amd.distributed.StepPipeline.  Returns (engine, seconds).
        h2d: every step first copies its batch from pinned host memory; d2h: every step ends with its label maps copied to
        pinned host memory (both asynchronous, in stream order on the step's own stream)."""
        eng = engines.get(precision)
        if eng is None:
            eng = engines[precision] = ZutisEngine(P, cfg.patch, cfg.dec_heads, precision=precision)
        host_x = x.cpu().pin_memory() if h2d else None
        lanes = build_lanes(eng, x, text, S, n, n_lanes, world=world, dist_on=dist_on, h2d=h2d, d2h=d2h)
        torch.cuda.synchronize()
        launch = make_launch(n_lanes, host_x, h2d, d2h)
        pipe = zd.StepPipeline(lanes, launch, gather=dist_on)
        pipe.run(max(warmup, n_lanes))
        pipe.drain()
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pipe.run(steps)
        pipe.drain()
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
        dt = time.perf_counter() - t0
        if dist_on:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        # the outputs of the plans that were just timed (outside the timed region): bitwise an eager step, lane by lane
        checked = check_timed_outputs(lanes) if check else None
        return eng, dt, checked

    eng, elapsed, timed_out = timed_run(args.precision, args.steps, args.warmup, h2d=args.h2d, d2h=args.d2h, check=True)
    # second line (N = 1 only, bounded): the same workload at the other precision (default: "fast", narrower than the reference
    # in the transformer bodies — reported, not the headline)
    other = None
    if world == 1 and not args.no_second_precision:
        oprec = "exact" if args.precision != "exact" else "fast"
        osteps = args.steps
        oeng, odt, otimed = timed_run(oprec, osteps, max(1, args.warmup // 2), h2d=args.h2d, d2h=args.d2h, check=True)
        other = {"precision": oprec, "eng": oeng, "timed": otimed, "value": round(B * osteps / odt, 2), "ms_per_step": round(odt / osteps * 1e3, 3), "steps": osteps}
    # PCIe-inclusive rates of the headline precision (N = 1, same number of steps): labels out, and batch in + labels out
    io_rates = None
    if world == 1 and not args.no_io_rates and not (args.h2d or args.d2h):
        io_rates = {}
        for key, kw in (("d2h", dict(d2h=True)), ("h2d_d2h", dict(h2d=True, d2h=True))):
            _, idt, _ = timed_run(args.precision, args.steps, max(1, args.warmup // 2), **kw)
            io_rates[key] = {"value": round(B * args.steps / idt, 2), "ms_per_step": round(idt / args.steps * 1e3, 3)}
        io_rates["what"] = ("same run with, per step, d2h: the int64 label maps [%d,%d,%d] (%.1f MB) copied to pinned host memory on the step's "
                            "stream (networks/zutis.py:372 ends in .cpu().numpy()); h2d_d2h: additionally the fp32 batch (%.1f MB) copied in "
                            "from pinned host memory first (trainer.py:328 image.to(device)); `value` of this line keeps both resident"
                            % (B, S, S, B * S * S * 8 / 1e6, B * 3 * S * S * 4 / 1e6))

    # ---- roofline of the dominant kernel: HIP events (torch current stream == launch stream) around every launch
    roof = None
    if rank == 0:
        def one_eager_step():
            out = eng.forward(x)
            eng.predict_semantic(out["patch_tokens"], text, (S, S))
        roof = gemm_roofline(ops, one_eager_step, elapsed / args.steps)
        if world == 1 and not args.no_live_traffic:
            extra = ["--precision", args.precision, "--batch", str(B), "--size", str(S), "--classes", str(n), "--no-io-rates"]
            roof["traffic"], roof["traffic_source"], per = live_pmc_traffic(extra, 1 if "SPLIT=1" in roof["kernel"] else (2 if "SPLIT=2" in roof["kernel"] else 0))
            shapes = getattr(gemm_roofline, "last_launch_shapes", None) or []
            nls = len(shapes)
            whole = (len(per) // nls - 1) if (per and nls) else 0
            if whole >= 1:
                # the counter passes ran whole steps of the same launch sequence and END with one: the last `whole` x n dispatches are
                # aligned steps (the first forward also launches the once-per-weights decoder prefix: it is dropped with the remainder)
                per = per[-whole * nls:]
                acc = {}
                for i, b in enumerate(per):
                    k, algo = shapes[i % len(shapes)]
                    e = acc.setdefault(k, [0, 0.0, algo])
                    e[0] += 1; e[1] += b
                for row in roof["by_shape"]:
                    for k, (c, b, algo) in acc.items():
                        if row["MxNxK"] == "x".join(str(d) for d in (k[:3] if k[3] == 1 else k)):
                            row["traffic"] = round(b / c)
                            row["traffic_over_algorithmic"] = round(b / c / algo, 2) if algo else None
        if roof.get("traffic") and roof.get("algorithmic_bytes_per_launch"):
            roof["traffic_over_algorithmic"] = round(roof["traffic"] / roof["algorithmic_bytes_per_launch"], 2)
        roof["measured_on"] += ("; rocprofv3 --kernel-trace --stats of `bench.py --inflight 1` (this precision) = profiles/r05_bench_%s_kernel_stats.csv"
                                % args.precision)

    # ---- CPU baseline: the oracle (CPU port of the reference path) on a bounded sample, rank 0 at N=1 only
    cpu = None
    parity = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import zutis_ref as O
        torch.set_num_threads(max(1, min(args.cpu_threads, os.cpu_count() or 1)))
        Pc = O.to_torch_params(detgen.zutis_state_dict(cfg))
        ns = max(1, min(args.cpu_sample if S <= 336 else 4, B))          # 518 px / 920 classes: ~3 s per image on the host
        xs = x[:ns].cpu()
        tc = text.cpu()

        def cpu_pass(xi):
            with torch.no_grad():
                o = O.zutis_forward(Pc, xi, cfg.patch, cfg.dec_heads)
                return o, O.predict_semantic(o["patch_tokens"], tc, size=(S, S))
        cpu_pass(xs[:1])                                  # warm-up
        times, chunks = [], None
        for _ in range(3):                                # median of three passes over the sample
            t1 = time.perf_counter()
            chunks = [cpu_pass(xs[i:i + 8]) for i in range(0, ns, 8)]
            times.append(time.perf_counter() - t1)
        dt = sorted(times)[1]
        o_ref = {"patch_tokens": torch.cat([c[0]["patch_tokens"] for c in chunks])}
        lab_ref = np.concatenate([c[1] for c in chunks])
        cpu = {"value": round(ns / dt, 3), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
               "sample": f"{ns} of the {B} step images in chunks of 8, oracle forward + semantic predict, median of 3 passes "
                         f"({', '.join('%.1f' % t for t in times)} s) after a 1-image warm-up; host has {os.cpu_count()} hardware "
                         f"threads, {torch.get_num_threads()} torch threads was the fastest setting of 8..128 on this host class"}
        lo_ref = O.semantic_logits_lowres(o_ref["patch_tokens"], tc).numpy()
        # BASELINE.md promised the host's physical cores: one more pass of a smaller sample with one torch thread per physical core, next to
        # the 16-thread figure above (which is the faster one on this host class and stays `value`)
        phys = max(1, (os.cpu_count() or 2) // 2)
        if phys != torch.get_num_threads() and not args.no_cpu_all_cores:
            torch.set_num_threads(phys)
            n2 = min(4, ns)
            cpu_pass(xs[:1])
            t1 = time.perf_counter()
            cpu_pass(xs[:n2])
            cpu["all_physical_cores"] = {"cores": phys, "value": round(n2 / (time.perf_counter() - t1), 3), "unit": "images/s",
                                         "sample": f"{n2} images, one pass after a 1-image warm-up"}
            torch.set_num_threads(max(1, min(args.cpu_threads, os.cpu_count() or 1)))

        def parity_of(timed):
            # the oracle against what the TIMED launch plans left behind (lane 0's last replay: the first `ns` images of the very batch
            # the timed steps ran), not a separate eager forward at another batch size
            ok, lo_t, lab_t = timed
            lo = lo_t[:ns].cpu().numpy()
            lab = lab_t[:ns].cpu().numpy()
            hist = O.confusion_hist(lab_ref, lab, n)
            from oracle.parity import unexplained_label_mismatches
            err = float(np.abs(lo - lo_ref).max())
            n_mis, n_bad, worst = unexplained_label_mismatches(lab, lab_ref, lo_ref, err, (S, S))
            return {"logit_max_abs_err": err, "label_agreement": float((lab == lab_ref).mean()),
                    "label_mismatches": n_mis, "unexplained_label_mismatches": n_bad,
                    "label_note": "a differing pixel is explained when the oracle's own full-resolution logits separate the two labels by "
                                  "<= 2 x logit_max_abs_err (largest such margin: %.2e); the argmax kernel is bit-exact on equal logits" % worst,
                    "miou_vs_oracle_labels": float(O.scores_from_hist(hist)[0]["Mean IoU"]), "tolerance": 1e-3,
                    "timed_outputs_bitwise_equal_eager": bool(ok),
                    "against": "fp32 oracle (CPU restatement of the reference path) on the first %d images of the timed batch; the compared "
                               "logits / labels are the outputs of the timed launch plans themselves (lane 0, last replay)" % ns}
        parity = parity_of(timed_out)
        if other is not None:
            other["parity"] = parity_of(other["timed"])

    torch_gpu = None
    if rank == 0 and world == 1 and args.torch_gpu_baseline and not args.no_torch_gpu_baseline:
        import torch.nn.functional as F
        from oracle import zutis_ref as O
        Pg = {k: v for k, v in P.items()}                 # fp32 parameters already on the device

        def gpu_pass(xi):
            with torch.no_grad():
                o = O.zutis_forward(Pg, xi, cfg.patch, cfg.dec_heads)
                lo = O.semantic_logits_lowres(o["patch_tokens"], text)
                return F.interpolate(lo, size=(S, S), mode="bilinear", align_corners=False).argmax(dim=1)   # zutis.py:366-372

        def time_leg(sdpa: bool):
            O.ENCODER_SDPA = sdpa
            try:
                for _ in range(2):
                    gpu_pass(x)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(5):
                    gpu_pass(x)
                torch.cuda.synchronize()
                return (time.perf_counter() - t1) / 5
            finally:
                O.ENCODER_SDPA = False
        # faithful leg: the encoder's nn.MultiheadAttention(need_weights=False) (clip_arch.py:314-316) reaches torch's fused
        # F.scaled_dot_product_attention; the decoder's calls (transformer.py:272-286, need_weights left True) the explicit
        # matmul-softmax-matmul.  Second leg: explicit attention everywhere (what rounds 1-2 timed).
        try:
            dt_sdpa = time_leg(True)
        except Exception as e:                            # SDPA unavailable for fp32 on this build: say so, keep the explicit leg
            dt_sdpa, sdpa_err = None, f"{type(e).__name__}: {e}"
        dt_expl = time_leg(False)
        dt = dt_sdpa if dt_sdpa is not None else dt_expl
        torch_gpu = {"value": round(B / dt, 1), "unit": "images/s", "kind": "port",
                     "encoder_attention": "F.scaled_dot_product_attention (fused)" if dt_sdpa is not None else "explicit (SDPA failed: %s)" % sdpa_err,
                     "explicit_attention_everywhere": round(B / dt_expl, 1),
                     "what": "the reference's op sequence (F.conv2d / F.linear / SDPA in the encoder as nn.MultiheadAttention(need_weights=False) "
                             "dispatches, explicit softmax attention in the decoder / F.interpolate / einsum) in stock PyTorch-ROCm fp32 eager "
                             "on the same MI355X, batch %d, 5 timed passes after 2 warm-ups per leg" % B}

    batch1 = None
    if rank == 0 and world == 1 and not args.no_batch1:
        for e in engines.values():                    # the headline's engines are done: free their buffers first
            e._bufs.clear()
        batch1 = batch1_object(args.precision, dev)
    # ---- the other BASELINE configs, bounded, in the driver-run line (N = 1, default workload only): c4 (518 px / 920 classes / 8 per step),
    # c5 (ViT-L/14@336 embedding extraction, one 256-image step, at the reference's fp16 arithmetic class) and the bilateral solver
    c4o = c5o = solvero = pseudoo = None
    if rank == 0 and world == 1 and not args.no_configs and args.workload == "c2" and (S, n) == (336, 81):
        for e in engines.values():
            e._bufs.clear()
        engines.clear()
        torch.cuda.empty_cache()
        ncpu = 0 if args.no_cpu_baseline else 1
        c4o = c4_object(P, cfg, dev, args.precision, cpu_images=ncpu, cpu_threads=args.cpu_threads)
        torch.cuda.empty_cache()
        c5o = c5_object(dev, "fast", cpu_images=ncpu, cpu_threads=args.cpu_threads)
        torch.cuda.empty_cache()
        solvero = solver_object(dev)
        pseudoo = pseudo_label_object(dev, args.precision)
    if rank == 0:
        total_images = world * B * args.steps
        line = {
            "metric": f"images/sec, COCO2017-val-shaped ViT-B/16 dense semantic segmentation @{S}px (ZUTIS forward + semantic predict), "
                      f"batches of {B} per GPU, {n_lanes} independent batch{'es' if n_lanes > 1 else ''} in flight",
            "value": round(total_images / elapsed, 2), "unit": "images/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": PRECISION_DTYPE[args.precision], "data": "synthetic",
            "precision": {"mode": args.precision, "what": PRECISION_TEXT[args.precision],
                          "stress_test": "tests/test_precision_gpu.py::test_stress_model_c2 (x100 outlier residual channels, sharpened "
                                         "attention, generic fp32 weights): fast <= 2.5e-4 logits / 1e-3 masks, exact <= 2e-5 / 2e-4 vs the fp32 oracle"},
            "config": {"workload": f"{'C2' if (S, n) == (336, 81) else 'C4' if (S, n) == (518, 920) else 'custom'}: ViT-B/16 CLIP encoder + ZUTIS head, {B}x3x{S}x{S} per GPU, {n} classes, "
                                   f"predict(semantic,size=({S},{S}))", "global_batch": world * B, "image_size": S,
                       "n_classes": n, "parallelism": f"dp{world}", "accumulate": "f32", "residual_stream": "f32",
                       "collective": "all_gather(low-res logits) per step, overlapped" if dist_on else "none",
                       "steps_in_flight": n_lanes, "cross_attention_key_split": ZutisEngine.cross_ksplit, **({"input": "pinned host batch copied in every step (--h2d)"} if args.h2d else {}),
                       "launch": ("native launch plans, interleaved on %d HIP streams" % n_lanes) if n_lanes > 1 else "eager, 1 stream",
                       "flops_per_image": FLOPS_PER_IMAGE_C2 if (S, n) == (336, 81) else None},
            "model_tflops": round(total_images * FLOPS_PER_IMAGE_C2 / elapsed / 1e12 / world, 1) if (S, n) == (336, 81) else None,
            "model_tflops_note": "images/s x the REFERENCE model's 124.5 GFLOP per image (SURVEY 8d; MFU convention) per GPU — not executed flops: "
                                 "the engine executes fewer (roofline.executed_algorithmic_flops_per_step, DESIGN 2a)",
            "timed_outputs_checked": bool(timed_out is not None and timed_out[0]),
            "timed_outputs_note": "after the timed region every lane's label maps and low-res logits (as the last replay of its launch plan "
                                  "left them) were compared bitwise with one eager step of the lane's engine on the same batch; `parity` "
                                  "compares lane 0's timed outputs with the oracle",
            "roofline": roof, "cpu_baseline": cpu, "parity": parity,
            **({"batch1": batch1} if batch1 else {}),
            **({"c4": c4o} if c4o else {}), **({"c5": c5o} if c5o else {}), **({"bilateral_solver": solvero} if solvero else {}),
            **({"pseudo_labels": pseudoo} if pseudoo else {}),
            **({"torch_gpu_baseline": torch_gpu} if torch_gpu else {}),
            **({"io_inclusive": io_rates} if io_rates else {}),
        }
        if other is not None:       # the same workload at the other precision (same steps-in-flight setup, fewer steps)
            line["second_precision"] = {"mode": other["precision"], "dtype": PRECISION_DTYPE[other["precision"]],
                                        "what": PRECISION_TEXT[other["precision"]], "value": other["value"], "unit": "images/s",
                                        "ms_per_step": other["ms_per_step"], "steps": other["steps"], "parity": other.get("parity")}
        if torch_gpu:
            line["vs_torch_gpu_fp32_eager"] = round(line["value"] / torch_gpu["value"], 2)
            if other is not None:
                line["second_precision"]["vs_torch_gpu_fp32_eager"] = round(other["value"] / torch_gpu["value"], 2)
        # LAST key, compact (the driver keeps the last ~2000 characters of stdout): the numbers of every object above, no prose
        sm = {"c2_" + args.precision: line["value"], "frac": roof["frac"] if roof else None, "checked": line["timed_outputs_checked"]}
        if parity:
            sm["c2_err"] = float("%.2g" % parity["logit_max_abs_err"]); sm["c2_bad_labels"] = parity["unexplained_label_mismatches"]
        if other is not None:
            sm["c2_" + other["precision"]] = other["value"]
            if other.get("parity"):
                sm["c2_" + other["precision"] + "_err"] = float("%.2g" % other["parity"]["logit_max_abs_err"])
                sm["c2_" + other["precision"] + "_checked"] = other["parity"]["timed_outputs_bitwise_equal_eager"]
        if torch_gpu:
            sm["torch_eager"] = torch_gpu["value"]
            sm["x_torch"] = [line["vs_torch_gpu_fp32_eager"]] + ([line["second_precision"]["vs_torch_gpu_fp32_eager"]] if other is not None else [])
        if cpu:
            sm["cpu"] = cpu["value"]
        if batch1:
            sm["b1_ms"] = [batch1["ms_per_image"], batch1["forward_ms"], batch1["instance_predict_ms"]]
            sm["b1_calls"] = batch1["library_calls_forward"]
        if c4o:
            sm["c4"] = {"v": c4o["value"], "frac": c4o["roofline"]["frac"], "ok": c4o["timed_outputs_bitwise_equal_eager"],
                        **({"err": float("%.2g" % c4o["parity"]["logit_max_abs_err"]), "bad": c4o["parity"]["unexplained_label_mismatches"]} if "parity" in c4o else {})}
        if c5o:
            sm["c5_" + c5o["precision"]] = {"v": c5o["value"], "frac": c5o["roofline"]["frac"], "tf": c5o["model_tflops"],
                                            **({"err": float("%.2g" % c5o["parity"]["embedding_max_abs_err"])} if "parity" in c5o else {})}
        if solvero:
            sm["solver_ms"] = [solvero["batch1"]["ms_per_image"], solvero["batch8"]["ms_per_image"]]
            sm["solver_frac"] = [solvero["batch1"]["frac"], solvero["batch8"]["frac"]]
        if pseudoo:
            sm["selfmask_solver_ips"] = pseudoo["value"]
        if io_rates:
            sm["io"] = [io_rates["d2h"]["value"], io_rates["h2d_d2h"]["value"]]
        line["summary"] = sm
    if dist_on:
        dist.barrier()                    # rank 0 measured the roofline / baselines after the timed region: leave together
        dist.destroy_process_group()      # first: RCCL prints its version banner on stdout when the communicator goes away
    if rank == 0:
        import ctypes
        ctypes.CDLL(None).fflush(None)    # RCCL's banner sits in libc's stdout buffer: push it out before the result line
        sys.stdout.flush()
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    sys.exit(main() or 0)
