"""Bounded objects of the default line: the other BASELINE configs (c4, c5), the bilateral solver and the pseudo-label path."""
from __future__ import annotations

import os
import time

import numpy as np
import torch

from .common import PRECISION_DTYPE
from .lanes import build_lanes, make_launch, check_timed_outputs
from .roofline import gemm_roofline


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


def natural_images(B, H, W, dev, seed=7):
    """B normalised images [B,3,H,W] with natural-image colour statistics (detgen.selfmask_like_rgb: gradients, blobs, 5 % noise —
    the bilateral solver's lattice then has ~20 k vertices as on photographs; pure noise images give it one vertex per pixel)."""
    from zutis_amd import detgen
    mean = np.array([0.485, 0.456, 0.406], np.float32)
    std = np.array([0.229, 0.224, 0.225], np.float32)
    xs = [((detgen.selfmask_like_rgb(H, W, seed=seed + i).astype(np.float32) / 255.0 - mean) / std).transpose(2, 0, 1) for i in range(B)]
    return torch.from_numpy(np.ascontiguousarray(np.stack(xs))).to(dev)


SELFMASK_FLOPS_PER_IMAGE = 792e9     # DINO ViT-S/8 @512x683 (T = 5505) + 6-layer decoder


def pseudo_label_object(dev, precision="exact", B=4, reps=5, cpu=True, cpu_threads=16):
    """The pseudo-label path north_star names (SelfMask, networks/selfmask + utils/bilateral_solver.py, as datasets/index_dataset.py:177-226
    generate_pseudo_masks drives them): DINO ViT-S/8 SelfMask at its working shape 512x683 (T = 5505 tokens) -> query selection ->
    bilateral solver -> > 0.5 -> nearest resize to 480x640, `B` images per call, device side (the RLE JSON files are host work).
    Measured like the headline: value on images with natural colour statistics, `roofline` with the flash-attention kernel as the
    dominant kernel, `parity` of image 0's final mask against the oracle chain, a bounded `cpu_baseline` (the oracle chain, one image)."""
    from zutis_amd import detgen, ops, pseudo_masks
    from zutis_amd.engine import SelfMaskEngine
    from .roofline import attention_roofline
    H, W, out_size = 512, 683, (480, 640)
    sd = detgen.selfmask_state_dict()
    eng = SelfMaskEngine({k: torch.from_numpy(v).to(dev) for k, v in sd.items()}, precision=precision)

    def timed(x):
        n = x.shape[0]
        for _ in range(2):
            pseudo_masks.pseudo_masks_batch(eng, x, [out_size] * n, True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            masks = pseudo_masks.pseudo_masks_batch(eng, x, [out_size] * n, True)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps, masks
    x = natural_images(B, H, W, dev)
    dt, masks = timed(x)
    roof = attention_roofline(ops, lambda: pseudo_masks.pseudo_masks_batch(eng, x, [out_size] * B, True), dt)
    dt_noise, _ = timed(torch.from_numpy(detgen.images(B, H, W, seed=7)).to(dev))
    dt8, _ = timed(natural_images(8, H, W, dev))             # the dataset driver's own default group size (pseudo_masks.dataset_generate_pseudo_masks batch_size=8)
    dt1, _ = timed(x[:1].contiguous())                       # ONE image per call: the reference's own loop (datasets/index_dataset.py:187-204, DataLoader batch_size=1)
    obj = {"what": f"SelfMask (DINO ViT-S/8 @{H}x{W}, T = 5505) + bilateral solver + threshold + nearest resize to {out_size[0]}x{out_size[1]}, "
                   f"{B} images per call, device side (datasets/index_dataset.py:177-226)",
           "value": round(B / dt, 1), "unit": "images/s", "ms_per_call": round(dt * 1e3, 2), "batch": B, "precision": precision, "calls": reps,
           "images": "natural colour statistics (detgen.selfmask_like_rgb, normalised): ~20 k lattice vertices per image",
           "value_batch1": round(1 / dt1, 1),
           "value_batch1_note": "ONE image per call, as the reference's generate_pseudo_masks loops (DataLoader batch_size=1, datasets/index_dataset.py:187-204)",
           "value_batch8": round(8 / dt8, 1),
           "value_batch8_note": "8 images per call: the default group size of the dataset-signature driver (pseudo_masks.dataset_generate_pseudo_masks)",
           "value_noise_images": round(B / dt_noise, 1),
           "value_noise_images_note": "the workload rounds 2-5 quoted: N(0,1) images give the solver one lattice vertex per pixel (17x a photograph's)",
           "model_tflops": round(B * SELFMASK_FLOPS_PER_IMAGE / dt / 1e12, 1), "flops_per_image": SELFMASK_FLOPS_PER_IMAGE, "roofline": roof}
    if cpu:
        from oracle import zutis_ref as O
        from oracle.parity import contour_mismatches, pseudo_mask_chain
        torch.set_num_threads(max(1, min(cpu_threads, os.cpu_count() or 1)))
        Pc = O.to_torch_params(sd)
        t1 = time.perf_counter()
        ref = pseudo_mask_chain(Pc, x[:1].cpu(), out_size)
        dtc = time.perf_counter() - t1
        got = masks[0].cpu().numpy().astype(bool)                 # image 0 of the LAST timed call
        inf = eng.forward(x[:1].contiguous(), inference=True)
        n_sm, bad_sm = contour_mismatches(inf["dts"][0].cpu().numpy(), ref["selfmask"])
        n_diff, n_bad = contour_mismatches(got, ref["mask"], native=ref["mask_native"])
        obj["parity"] = {"mask_pixels": int(got.size), "differing_pixels": n_diff, "differing_pixels_off_the_oracle_contour": n_bad,
                         "selfmask_differing_pixels": n_sm, "selfmask_differing_pixels_off_the_oracle_contour": bad_sm,
                         "selected_query_identical": bool(int(inf["index"][0]) == ref["query"]) if "index" in inf else None,
                         "oracle_objectness_margin": round(ref["objectness_margin"], 5),
                         "against": "the oracle chain on image 0 of the timed batch (oracle/parity.py::pseudo_mask_chain: SelfMask inference -> "
                                    "scipy bilateral solver -> > 0.5 -> nearest resize); a differing pixel is explained when it lies within one "
                                    "pixel of the oracle's own mask contour"}
        obj["cpu_baseline"] = {"value": round(1.0 / dtc, 4), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
                               "sample": "ONE 512x683 image, one pass of the oracle chain (torch fp32 SelfMask on %d threads + numpy / scipy.sparse "
                                         "bilateral solver on one)" % torch.get_num_threads()}
    eng._bufs.clear()
    return obj


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
    eng.cross_ksplit = "auto"        # 8 images x 8 heads = 64 cross-attention workgroups on 256 CUs: split the 5476 keys by the batch (engine_base._decoder)
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
    obj = {"what": f"C4: ViT-B/16 @{S}px, {n} classes, {B} images per step, {n_lanes} launch plans in flight, cross-attention keys split by the batch (`bench.py --workload c4` is the full line)",
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
