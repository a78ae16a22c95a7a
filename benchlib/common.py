"""Shared constants of the bench line (bench.py is the entry the driver calls; this package holds its parts)."""
from __future__ import annotations

import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH_PY = os.path.join(ROOT, "bench.py")            # the script the driver (and the live counter passes) run

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
HBM_PEAK_TBPS = 8.0                   # same guide: HBM3E ~8 TB/s
FLOPS_PER_IMAGE_C2 = 124.4e9 + 0.146e9  # forward + semantic predict


def rank_env():
    """(world, rank, local_rank) from the launcher's environment (torch.distributed.run); 1, 0, 0 for a plain run."""
    return int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
