"""Roofline of the dominant kernel: HIP events around every launch of an instrumented step, and HBM traffic from live rocprofv3
counter passes (two child runs of bench.py)."""
from __future__ import annotations

import os
import sys

import torch

from .common import BENCH_PY, MFMA_F16_DENSE_PEAK_TFLOPS

ROCPROFV3_FALLBACK = "/opt/rocm/bin/rocprofv3"


def live_pmc_traffic(extra_args, split: int, timeout_s=240):
    """HBM bytes per launch of the dominant GEMM kernel family (split = 1: gemm_f16_kernel<..., SPLIT=1>, the f16x3 kernel;
    0: the plain fp16-operand kernel), measured NOW: two child runs of this script under
    `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes: the TCC block cannot hold both counters; --kernel-trace
    only, as /opt/skills/guides/MI355X_MICROARCH.md prescribes), corrected for gfx950 (FETCH_SIZE tallies 128-B requests at
    64 B: read bytes = 2 * FETCH_SIZE; WRITE_SIZE exact; both in KiB).  Returns (bytes_per_launch | None, note, per-dispatch bytes in dispatch order | None) — three values on EVERY path."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or ROCPROFV3_FALLBACK
    if not os.path.exists(exe):
        return None, "rocprofv3 not found", None
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None, "this run is itself being profiled: no nested rocprofv3 passes", None
    tot, cnt, seq = {}, {}, {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="zh_pmc_", dir="/tmp")
        cmd = [exe, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable, BENCH_PY,
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


def launch_profile(ops, run_once, passes=2):
    """HIP events around every profiled launch (ops.PROFILER) of `run_once()`, eager on torch's current stream (== the launch stream).
    Returns {family: (launches, algorithmic flops, seconds)} of the last pass."""
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
        for _ in range(passes):
            prof.clear()
            run_once()
        torch.cuda.synchronize()
    finally:
        ops.PROFILER = None
    fl = lambda w: w[0] if isinstance(w, tuple) else w
    return {k: (len(v), sum(fl(w) for w, _, _ in v), sum(a.elapsed_time(b) for _, a, b in v) * 1e-3) for k, v in prof.items()}


def attention_roofline(ops, run_once, step_seconds):
    """Roofline object with the flash-attention kernel as the dominant kernel (the SelfMask pseudo-label path at T = 5505: attention is
    half of its GPU time).  Algorithmic flops = 4 * Tq * Tk * dh per (image, head); the split-pair form issues three MFMA products per
    algorithmic product in QK^T and in P.V, so its ceiling is a third of the dense fp16 MFMA peak."""
    stats = launch_profile(ops, run_once)
    fam = max((k for k in stats if k.startswith("attention")), key=lambda k: stats[k][2])
    nl, fl, tt = stats[fam]
    npr = 3 if fam.endswith("x3") else 1
    peak = MFMA_F16_DENSE_PEAK_TFLOPS / npr
    ach = fl / tt / 1e12
    gem = {k: v for k, v in stats.items() if k.startswith("gemm")}
    gt = sum(v[2] for v in gem.values())
    return {"bound": "mfma", "kernel": "attn_f16_kernel (zh_attention_f16" + (", split-pair operands: three MFMA products per product)" if npr == 3 else ")"),
            "achieved": round(ach, 1), "peak": round(peak, 1), "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": None,
            "flops_per_launch": round(fl / nl), "launches_per_step": nl, "avg_launch_us": round(tt / nl * 1e6, 1),
            "attention_share_of_step": round(tt / step_seconds, 3), "gemm_share_of_step": round(gt / step_seconds, 3),
            "gemm_algorithmic_tflops": round(sum(v[1] for v in gem.values()) / gt / 1e12, 1) if gt else None,
            "executed_algorithmic_flops_per_step": round(sum(v[1] for k, v in stats.items() if k.startswith(("gemm", "attention")))),
            "measured_on": "HIP events around every attention / GEMM launch of an instrumented eager call on one stream"}
