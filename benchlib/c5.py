"""`bench.py --workload c5`: BASELINE config 5, the full line."""
from __future__ import annotations

import json
import os
import time

import torch

from .common import PRECISION_DTYPE, PRECISION_TEXT
from .roofline import gemm_roofline


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
