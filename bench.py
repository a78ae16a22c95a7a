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
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# The measurement code lives in benchlib/ (round 6: one file per workload); this file is the entry the driver calls and re-exports the
# names tests and tools use (`bench.build_lanes`, `bench.c3_model`, ...).  torch is imported by these modules, not before --dry-launch needs it.
from benchlib.common import FLOPS_PER_IMAGE_C2, MFMA_F16_DENSE_PEAK_TFLOPS, PRECISION_DTYPE, PRECISION_TEXT   # noqa: E402,F401
from benchlib.lanes import build_lanes, check_timed_outputs, make_launch                                        # noqa: E402,F401
from benchlib.launch import launch_ranks, rank_launch_command                                                  # noqa: E402,F401
from benchlib.roofline import gemm_roofline, live_pmc_traffic                                                  # noqa: E402,F401
from benchlib.c3 import batch1_object, c3_model, c3_parity                                                     # noqa: E402,F401
from benchlib.objects import c4_object, c5_object, natural_images, pseudo_label_object, solver_object                          # noqa: E402,F401


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
    ap.add_argument("--no-cross-ksplit-auto", dest="cross_ksplit_auto", action="store_false",
                    help="batches with fewer (image, head) pairs than CUs (config 4's 8 images): do NOT split the cross-attention keys by the batch")
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
        from benchlib.c5 import bench_c5
        return bench_c5(args)
    if args.workload == "c3":
        from benchlib.c3 import bench_c3
        return bench_c3(args)
    from benchlib.headline import run
    return run(args)


if __name__ == "__main__":
    sys.exit(main() or 0)
