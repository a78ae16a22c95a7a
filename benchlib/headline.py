"""The headline line: BASELINE config 2 (and `--workload c4`, the same path at 518 px / 920 classes) — ViT-B/16 ZUTIS forward + semantic
predict, batches sharded by rank, `--inflight` launch plans in flight, one all-gather of the low-res logits per step at N > 1."""
from __future__ import annotations

import json
import os
import sys
import time

import torch

from .baselines import cpu_baseline_c2, parity_of, torch_gpu_baseline
from .c3 import batch1_object
from .common import FLOPS_PER_IMAGE_C2, PRECISION_DTYPE, PRECISION_TEXT, rank_env
from .lanes import build_lanes, check_timed_outputs, make_launch
from .objects import c4_object, c5_object, pseudo_label_object, solver_object
from .roofline import gemm_roofline, live_pmc_traffic


def run(args):
    world, rank, local = rank_env()
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

    coll_box = {}

    def timed_run(precision: str, steps: int, warmup: int, h2d: bool = False, d2h: bool = False, check: bool = False, gather=None):
        """Builds the engine for `precision`, one lane (engine fork + launch plan + stream + gather buffer) per step in
        flight, and times `steps` steps through zutis_amd.distributed.StepPipeline.  Returns (engine, seconds).
        h2d: every step first copies its batch from pinned host memory; d2h: every step ends with its label maps copied to
        pinned host memory (both asynchronous, in stream order on the step's own stream).  gather: None = as the run is configured
        (the all-gather per step whenever there is a process group); False = the same steps without it (the exposed-gather A/B)."""
        gather = dist_on if gather is None else gather
        eng = engines.get(precision)
        if eng is None:
            eng = engines[precision] = ZutisEngine(P, cfg.patch, cfg.dec_heads, precision=precision)
            if B * cfg.dec_heads < 256 and args.cross_ksplit_auto:
                eng.cross_ksplit = "auto"      # fewer cross-attention workgroups than CUs (config 4: 8 images): keys split by the batch, engine_base._decoder
        host_x = x.cpu().pin_memory() if h2d else None
        lanes = build_lanes(eng, x, text, S, n, n_lanes, world=world, dist_on=dist_on, h2d=h2d, d2h=d2h)
        torch.cuda.synchronize()
        launch = make_launch(n_lanes, host_x, h2d, d2h)
        pipe = zd.StepPipeline(lanes, launch, gather=gather)
        pipe.run(max(warmup, n_lanes))
        pipe.drain()
        retired0 = pipe.retired
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
            own = torch.tensor([dt], dtype=torch.float64, device=dev)
            every = torch.empty((world,), dtype=torch.float64, device=dev)
            dist.all_gather_into_tensor(every, own)
            dt = float(every.max().item())                     # the contract's time: MAX over ranks
            if gather and check:
                # did the gathers of the timed region see every rank?  (outside the timed region; every rank takes part)
                c = zd.verify_gather(lanes[0])
                c.update(backend="rccl (torch.distributed 'nccl')", what="all_gather_into_tensor of the low-res class logits per step, async on the step's stream",
                         gathers_retired_in_timed_region=pipe.retired - retired0, steps=steps,
                         per_rank_images_per_s={"min": round(B * steps / float(every.max().item()), 2), "max": round(B * steps / float(every.min().item()), 2)})
                coll_box[precision] = c
        # the outputs of the plans that were just timed (outside the timed region): bitwise an eager step, lane by lane
        checked = check_timed_outputs(lanes) if check else None
        return eng, dt, checked

    eng, elapsed, timed_out = timed_run(args.precision, args.steps, args.warmup, h2d=args.h2d, d2h=args.d2h, check=True)
    collective = coll_box.get(args.precision)
    if collective is not None:
        # what the per-step gather costs on the critical path: the same steps, same lanes, without it (max over ranks both times)
        _, dt_nog, _ = timed_run(args.precision, args.steps, max(1, args.warmup // 2), h2d=args.h2d, d2h=args.d2h, gather=False)
        collective["gather_ms_exposed"] = round((elapsed - dt_nog) / args.steps * 1e3, 4)
        collective["ms_per_step_without_gather"] = round(dt_nog / args.steps * 1e3, 3)
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
        roof["measured_on"] += ("; rocprofv3 --kernel-trace --stats of `bench.py --inflight 1` (this precision) = profiles/r06_bench_%s_kernel_stats.csv"
                                % args.precision)

    # ---- CPU baseline: the oracle (CPU port of the reference path) on a bounded sample, rank 0 at N=1 only
    cpu = None
    parity = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu, refs = cpu_baseline_c2(args, cfg, x, text, S, n, B)
        parity = parity_of(timed_out, refs, n, S)
        if other is not None:
            other["parity"] = parity_of(other["timed"], refs, n, S)

    torch_gpu = None
    if rank == 0 and world == 1 and args.torch_gpu_baseline and not args.no_torch_gpu_baseline:
        torch_gpu = torch_gpu_baseline(P, cfg, x, text, S, B)

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
        pseudoo = pseudo_label_object(dev, args.precision, cpu=not args.no_cpu_baseline, cpu_threads=args.cpu_threads)
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
                       "steps_in_flight": n_lanes, "cross_attention_key_split": eng.cross_ksplit, **({"input": "pinned host batch copied in every step (--h2d)"} if args.h2d else {}),
                       "launch": ("native launch plans, interleaved on %d HIP streams" % n_lanes) if n_lanes > 1 else "eager, 1 stream",
                       "flops_per_image": FLOPS_PER_IMAGE_C2 if (S, n) == (336, 81) else None},
            "model_tflops": round(total_images * FLOPS_PER_IMAGE_C2 / elapsed / 1e12 / world, 1) if (S, n) == (336, 81) else None,
            "model_tflops_note": "images/s x the REFERENCE model's 124.5 GFLOP per image (SURVEY 8d; MFU convention) per GPU — not executed flops: "
                                 "the engine executes fewer (roofline.executed_algorithmic_flops_per_step, DESIGN 2a)",
            "timed_outputs_checked": bool(timed_out is not None and timed_out[0]),
            "timed_outputs_note": "after the timed region every lane's label maps and low-res logits (as the last replay of its launch plan "
                                  "left them) were compared bitwise with one eager step of the lane's engine on the same batch; `parity` "
                                  "compares lane 0's timed outputs with the oracle",
            **({"collective": collective} if collective is not None else {}),
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
        if collective is not None:   # N > 1 (or --force-dist): the proof that the gather saw every rank, in the part of the line the driver keeps
            sm["coll"] = {"ranks": collective["ranks_in_gather"], "ok": bool(collective["verified"] and collective["gathers_retired_in_timed_region"] == args.steps),
                          "distinct": collective["slices_distinct"], "exposed_ms": collective["gather_ms_exposed"], "rank_ips": collective["per_rank_images_per_s"]}
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
            sm["selfmask_solver_ips"] = {"b4": pseudoo["value"], "b8": pseudoo["value_batch8"], "b1": pseudoo["value_batch1"]}
            sm["pseudo"] = {"frac": pseudoo["roofline"]["frac"], **({"bad_px": pseudoo["parity"]["differing_pixels_off_the_oracle_contour"],
                                                                    "diff_px": pseudoo["parity"]["differing_pixels"]} if "parity" in pseudoo else {})}
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
    if collective is not None and not (collective["verified"] and collective["gathers_retired_in_timed_region"] == args.steps
                                       and (world == 1 or collective["slices_distinct"])):
        sys.stderr.write("bench.py: the all-gather did NOT deliver every rank's logits: %r\n" % (collective,))
        return 4


