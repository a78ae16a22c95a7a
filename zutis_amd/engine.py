"""ZutisEngine: ZUTIS forward / predict as a sequence of libzutis_hip kernels (networks/zutis.py:472-532, :340-470).
Layout, precision sites and the shared encoder / decoder sequences: zutis_amd/engine_base.py.  The other engines are re-exported
here (ClipImageEncoder, ClipTextEncoder: engine_clip.py; SelfMaskEngine: engine_selfmask.py) so that
`from zutis_amd.engine import ...` keeps working."""
from __future__ import annotations

import math
import weakref
from typing import Dict, Optional, Tuple

import numpy as np
import torch

from . import _lib, compose, ops
from ._lib import ZutisHipError
from .engine_base import (ALL_SITES, DECODER_SITES, ENCODER_SITES, HEAD_SITES, PRECISIONS, P_shape0, _EngineBase, _rup, f16, f32,
                          resolve_precision)
from .ops import Act


def _to_host(t: torch.Tensor, cache: Dict[int, torch.Tensor]) -> np.ndarray:
    """A 1-D uint8 device tensor on the host: one asynchronous copy into a cached PINNED buffer + one stream synchronisation (`.cpu()`
    goes through pageable memory: an allocation, a staged copy and its own synchronisation).  `cache` belongs to ONE engine instance
    (forks — one per stream / thread — have their own: a shared buffer would be overwritten by a sibling's predict).  The array is a
    view of the cached buffer: valid until that engine's next call with the same size (callers take what they need out of it before
    they return); a buffer that falls out of the cache stays alive as long as a view of it does."""
    n = t.numel()
    if cache is None:                                 # called without an engine instance (tests drive the predict's pieces that way)
        cache = {}
    buf = cache.get(n)
    if buf is None:
        while len(cache) >= 8:
            cache.pop(next(iter(cache)))
        buf = cache[n] = torch.empty((n,), dtype=torch.uint8, pin_memory=True)
    buf.copy_(t, non_blocking=True)
    torch.cuda.current_stream(t.device).synchronize()
    return buf.numpy()


class ZutisEngine(_EngineBase):
    """Inference engine for one ZUTIS network.  `params` maps reference state_dict keys to fp32 CUDA tensors
    (typically the nn.Parameters of the drop-in module, so load_state_dict() is picked up via version counters)."""

    def __init__(self, params: Dict[str, torch.Tensor], patch: int, dec_heads: int = 8, precision="exact"):
        self.params = params
        self.patch = patch
        self.D = params["encoder.class_embedding"].shape[0]
        self.heads = self.D // 64                                              # clip_arch.py:606
        self.layers = 1 + max(int(k.split(".")[3]) for k in params if k.startswith("encoder.transformer.resblocks."))
        self.dec_layers = 1 + max(int(k.split(".")[2]) for k in params if k.startswith("decoder.layers."))
        self.dec_heads = dec_heads
        self.dec_dh = self.D // dec_heads
        self.Q = params["query_embed"].shape[0]
        self.E = params["encoder.proj"].shape[1]
        self.grid = int(math.isqrt(params["encoder.positional_embedding"].shape[0] - 1))
        if self.dec_dh not in (64, 96):
            raise ZutisHipError(f"decoder head_dim {self.dec_dh} unsupported by zh_attention_f16 (64 or 96)")
        self._init_base(precision)

    # ------------------------------------------------------------------ packing
    def _pack(self):
        key = self._version_key()
        if key == self._packed_key:
            return
        P, D, w = self.params, self.D, {}
        c32 = self._c32
        self._pack_clip_visual(w, P, "encoder.", D, self.layers, self.patch)
        for ffn in ("ffn1", "ffn2"):
            for j in range(3):
                if (ffn, j) == ("ffn1", 2):
                    continue                              # composed into its consumers below: decoder_input is never formed
                w[f"{ffn}.{j}.w"], w[f"{ffn}.{j}.b"] = self._hw(P[f"{ffn}.layers.{j}.weight"], ffn), c32(P[f"{ffn}.layers.{j}.bias"])
        # decoder_input = f @ W2^T + b2 (ffn1's last Linear, zutis.py:500-503; f = its 256-wide hidden layer) has two consumers,
        # both linear in it: the decoder's K / V projections (_pack_decoder composes W2 into them) and the mask einsum
        # (zutis.py:196-198)  q . decoder_input[m] = (W2^T q) . f[m] + q . b2.  With f stored with a constant ones column
        # (FX columns: f | 1 | 0...), the einsum contracts [W2^T q | q.b2 | 0] with it over FX = 320 instead of D = 768, and
        # the [B*M, 768] decoder_input tensor and its GEMM disappear.  "mask_q.w" maps a query to that FX-vector.
        W2, b2 = P["ffn1.layers.2.weight"].detach(), P["ffn1.layers.2.bias"].detach()
        self.Fh = W2.shape[1]
        self.FX = _rup(self.Fh + 1, 64)
        wq = compose.mask_query_weight(W2, b2, self.FX)
        w["mask_q.w"] = self._hw(wq, "mask")
        self._pack_decoder(w, P, D, self.dec_layers, memory_linear=(P["ffn1.layers.2.weight"], P["ffn1.layers.2.bias"]))
        self._w, self._packed_key = w, key
        self._geo.clear()
        self._graphs.clear()

    def _geometry(self, h: int, w: int):
        """Input-independent tables per token grid: bicubic pos-embed (clip_arch.py:356-374) and sine PE
        (positional_embedding.py:29-52) — computed once on device, cached."""
        g = self._geo.get((h, w))
        if g is None:
            D, dev = self.D, self._device()
            pos = torch.empty((1 + h * w, D), dtype=f32, device=dev)
            sh = np.float32(1.0 / ((h + 0.1) / self.grid))
            sw = np.float32(1.0 / ((w + 0.1) / self.grid))
            ops.posembed_bicubic(self._w["encoder.positional_embedding"], pos, self.grid, h, w, D, sh, sw, True)
            pe = torch.empty((4 * h * w, D), dtype=f32, device=dev)
            ops.sine_pe(pe, 2 * h, 2 * w, D)
            # `pos @ Wk^T` (transformer.py:281 through :283's key projection, all layers): the sine PE is [py(y) | px(x)]
            # (positional_embedding.py:47-52), so the term is Ty[y] + Tx[x] with two small tables (fp64 products, stored fp32)
            # that the K GEMM's accumulators start from
            tdt = f32 if self._x3("dec_kv") else f16      # fp16 K: fp16 tables (the tile's slice is staged through LDS)
            Ty, Tx = compose.separable_pos_tables(pe, self._w["ca_k_pos_w"], 2 * h, 2 * w, tdt)   # [2h, L*D], [2w, L*D]
            g = {"pos": pos, "pe": pe, "k_pos": (Ty, Tx)}
            self._geo_put((h, w), g)
        return g

    # ------------------------------------------------------------------ encoder
    def encode(self, x: torch.Tensor):
        """clip_arch.py:377-411 -> (patch tokens f32 [B,hw,D] (ln_post applied, cls dropped), h, w)."""
        tok, _, h, w = self._encode(x, False)
        return tok, h, w

    def _encode(self, x: torch.Tensor, want16: bool, want32: bool = True):
        """encode() plus, on request, the fp16 / split-pair copy of the tokens the head's GEMMs read (same LayerNorm launch)."""
        self._pack()
        if not (x.is_cuda and x.dtype == f32 and x.dim() == 4 and x.shape[1] == 3):
            raise ZutisHipError("encode: expected float32 CUDA tensor [B,3,H,W]")
        x = x.contiguous()
        W_, D, p = self._w, self.D, self.patch
        B, _, H, Wd = x.shape
        h, w = (H - p) // p + 1, (Wd - p) // p + 1
        T = 1 + h * w
        X = self._clip_trunk(x, self._geometry(h, w)["pos"], h, w)
        tok = self._buf("tok", (B, h * w, D), f32) if want32 else None
        tok16 = self._abuf("tok16", (B * h * w, D), self._x3("ffn1", "textproj")) if want16 else None
        ops.layernorm(X, W_["encoder.ln_post.weight"], W_["encoder.ln_post.bias"], 1e-5, B * h * w, D, out_f32=tok, out_f16=tok16,
                      in_group_rows=h * w, in_group_stride=T, in_offset=1, status=self.status_word())   # :403-404
        return tok, tok16, h, w

    # ------------------------------------------------------------------ full forward
    def forward(self, x: torch.Tensor) -> Dict[str, torch.Tensor]:
        """networks/zutis.py:472-532."""
        _, tok16, h, w = self._encode(x, True, want32=False)
        W_, D, B, Q, L = self._w, self.D, x.shape[0], self.Q, self.dec_layers
        h2, w2 = 2 * h, 2 * w
        M = h2 * w2
        geo = self._geometry(h, w)
        # zutis.py:491-503 upsamples the tokens x2 and then applies ffn1; zutis.py:319 projects the upsampled tokens.  Both first
        # steps are LINEAR maps of the tokens and bilinear interpolation is a convex combination (weights 0.25 / 0.75, sum 1),
        # so W.up(t) + b == up(W.t + b): the first ffn1 layer and the text-space projection run on the h*w tokens (4x fewer rows)
        # and their outputs are upsampled (ReLU after the interpolation, where the reference has it).  Same function, different
        # rounding order (fp32-class in the x3 mode); the [B, 4hw, 768] upsampled token tensor is never formed.
        Fh, FX = self.Fh, self.FX
        h1 = self._buf("ffn_h1_lo", (B * h * w, Fh), f32)
        self._gemm("ffn1", tok16, W_["ffn1.0.w"], h1, bias=W_["ffn1.0.b"])                  # :500-503 (layer 0, pre-ReLU)
        f1 = self._abuf("ffn_h1", (B * M, Fh), self._x3("ffn1"))
        ops.upsample2x_cl(h1, B, h, w, Fh, out_f16=f1, relu=True)                           # :491-495 + ReLU
        # ffn1's hidden layer 2 with the constant columns [1, 0, ...] behind it (see _pack): rows are FX wide
        f2x = self._abuf("ffn_h2x", (B * M, FX), self._x3("ffn1"))
        if self._buf_const.get("ffn_h2x") is not f2x.t:              # once per (re)allocation of the cached buffer
            f2x.t[:, :, Fh:] = 0
            f2x.t[0, :, Fh] = 1
            self._buf_const["ffn_h2x"] = f2x.t
        f2 = f2x.view(f2x.hi[:, :Fh])
        self._gemm("ffn1", f1, W_["ffn1.1.w"], f2, bias=W_["ffn1.1.b"], act=ops.ACT_RELU)
        # ffn1's last Linear (-> decoder_input) is composed into its consumers at pack time.  The decoder's K / V projections
        # of decoder_input (+ pos) contract over the hidden width (256) instead of D (768) — 3x fewer flops on what was 22 % of
        # the model's GEMM work — and `memory + pos` (transformer.py:281) is never materialised
        KALL, VALL = self._decoder_kv(f2, f2, B, M, D, L, k_pos=geo["k_pos"])
        inter16 = self._decoder(KALL, VALL, B, M, D, Q, L, self.dec_heads, stack_all=True)  # transformer.py:114-152
        RQ = B * L * Q
        g1 = self._abuf("ffn2_h1", (RQ, Fh), self._x3("ffn2"))
        g2 = self._abuf("ffn2_h2", (RQ, Fh), self._x3("ffn2"))
        q32 = self._buf("q32", (RQ, D), f32)
        q16 = self._abuf("q16", (RQ, D), self._x3("mask"), unit_norm=True)
        self._gemm("ffn2", inter16, W_["ffn2.0.w"], g1, bias=W_["ffn2.0.b"], act=ops.ACT_RELU)        # zutis.py:514
        self._gemm("ffn2", g1, W_["ffn2.1.w"], g2, bias=W_["ffn2.1.b"], act=ops.ACT_RELU)
        self._gemm("ffn2", g2, W_["ffn2.2.w"], q32, bias=W_["ffn2.2.b"])
        ops.l2norm_rows(q32, RQ, D, out_f16=q16)                                            # :515
        masks = torch.empty((B, L, Q, h2, w2), dtype=f32, device=x.device)
        qw = self._abuf("mask_q", (RQ, FX), self._x3("mask"))
        self._gemm("mask", q16, W_["mask_q.w"], qw)                                         # [W2^T q | q.b2 | 0]
        self._gemm("mask", qw, f2x, masks, act=ops.ACT_SIGMOID, M=L * Q, N=M, K=FX, lda=FX, ldw=FX, ldc=M,
                   batch=B, strideA=L * Q * FX, strideW=M * FX, strideC=L * Q * M)          # :196-198,209
        tsl = self._buf("textspace_lo", (B * h * w, self.E), f32)
        self._gemm("textproj", tok16, W_["projT"], tsl)                                     # :319 on the h*w tokens
        ts = self._buf("textspace", (B * M, self.E), f32)
        ops.upsample2x_cl(tsl, B, h, w, self.E, out_f32=ts)                                 # :491-495
        pt = torch.empty((B, h2, w2, self.E), dtype=f32, device=x.device)
        ws = self._buf("gln_ws", (max(1, ops.global_ln_l2_workspace_size(B, M, self.E)),), torch.uint8)
        pt16 = self._abuf("pt16", (B * M, self.E), self._x3("logits"), unit_norm=True)   # the copy predict_semantic's class-logit GEMM consumes
        ops.global_ln_l2(ts, B, M, self.E, out_f32=pt, out_f16=pt16, eps=1e-5, l2_eps=1e-7, workspace=ws, status=self.status_word())  # :320-322
        self._pt16_of = (weakref.ref(pt), pt._version)           # identity, not address: a freed tensor's address can be reused
        return {"mask_proposals": masks, "patch_tokens": pt}

    # ------------------------------------------------------------------ hipGraph replay (latency path)
    def forward_graphed(self, x: torch.Tensor) -> Dict[str, torch.Tensor]:
        """Same result as forward(), replayed from a hipGraph captured once per input shape.  At batch 1-4 the eager path is
        host-bound (~230 launches x ~11 us of Python/ctypes = 2.7 ms per forward regardless of B); COCO-20K evaluation
        (coco20k_eval.py:241-268) runs batch 1.  Outputs are fresh tensors (copied out of the graph's static buffers)."""
        key = tuple(x.shape)
        graphs = self._graphs
        g = graphs.get(key)
        if g is not None and g["weights"] is self._packed_key:
            graphs.move_to_end(key)                  # LRU: the shapes most images share stay
            if g["graph"] is None:                   # capture failed for this shape before: eager from then on
                return self.forward(x)
            # Launch first, check the parameters' versions behind the launch: walking the 275 parameters (_version_key) is 55 - 150 us of
            # Python, and in a batch-1 evaluation loop the GPU idles through everything the host does between one image's predict and the
            # next image's launch.  A parameter that changed since the capture (rare: load_state_dict after the first forward) makes
            # this replay a wasted one: its outputs are dropped and the forward runs again on re-packed weights.
            g["x"].copy_(x)
            g["graph"].replay()
            out = {k: v.clone() for k, v in g["out"].items()}
            if self._version_key() == self._packed_key:
                return out
            self.status_word().zero_()               # whatever the stale weights raised
        self._pack()                                 # (drops every graph when the parameters changed)
        g = graphs.get(key)
        if g is not None and g["weights"] is not self._packed_key:
            g = None
        if g is None:
            static_x = x.clone()
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):              # warm-up on a side stream: fills the buffer / geometry caches
                for _ in range(2):
                    self.forward(static_x)
            torch.cuda.current_stream().wait_stream(s)
            graph = torch.cuda.CUDAGraph()
            try:
                # "thread_local": only THIS thread's calls are checked against the capture.  The reference's validation loaders run with
                # pin_memory=True (configs/*.yaml val_dataloader_kwargs): their pin-memory thread calls hipHostMalloc / records events
                # while we capture, which the default "global" mode turns into hipErrorStreamCaptureUnsupported in THAT thread (it kills
                # the loader) or invalidates the capture.
                with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                    out = self.forward(static_x)
            except Exception as e:                   # a capture that cannot be made: this shape runs eagerly from now on, loudly once
                import warnings
                warnings.warn(f"hipGraph capture failed for input shape {key} ({type(e).__name__}: {e}); this shape runs eagerly")
                torch.cuda.synchronize()
                g = {"x": None, "graph": None, "out": None, "weights": self._packed_key, "keep": None}
                self._graph_put(key, g)
                return self.forward(x)
            # The graph bakes in raw pointers to this shape's scratch buffers, geometry tables and packed weights.  _buf()
            # drops a name's buffers when another shape arrives and _geo is a bounded cache, so the graph keeps its own
            # references: shape A, then B, then A again replays A into memory that is still A's.
            g = {"x": static_x, "graph": graph, "out": out, "weights": self._packed_key,
                 "keep": (dict(self._bufs), dict(self._geo), self._w)}
            self._graph_put(key, g)
        g["x"].copy_(x)
        g["graph"].replay()
        return {k: v.clone() for k, v in g["out"].items()}

    # ------------------------------------------------------------------ native launch plans
    def build_plan(self, x_shape, text: Optional[torch.Tensor] = None, size: Optional[Tuple[int, int]] = None):
        """Record forward() (and predict_semantic() when `text` is given) for one input shape into a native launch plan
        (zutis_amd/plan.py).  Returns a dict with the static input `x`, the static outputs and the plan.  The engine must
        not be used with other shapes between build and replay (its buffer cache backs the recorded pointers)."""
        from . import plan as zplan
        self._pack()
        dev = self._device()
        static_x = torch.zeros(tuple(x_shape), dtype=f32, device=dev)
        text32 = None if text is None else text.detach().to(device=dev, dtype=f32).contiguous()
        self.forward(static_x)                                     # eager warm-up: packs weights, fills caches
        if text32 is not None:
            self.predict_semantic(self.forward(static_x)["patch_tokens"], text32, size)
        gen = self._buf_gen
        with zplan.Recorder() as rec:
            out = self.forward(static_x)
            labels = None if text32 is None else self.predict_semantic(out["patch_tokens"], text32, size)
        if self._buf_gen != gen:
            raise ZutisHipError("build_plan: buffers were re-allocated while recording")
        return {"x": static_x, "out": out, "labels": labels, "text": text32, "plan": rec.build(), "gen": gen,
                "weights": self._packed_key}

    def run_plan(self, p, x: Optional[torch.Tensor] = None):
        """Replay a plan from build_plan() on the current stream.  Outputs are the plan's static tensors (overwritten by
        the next replay: clone what must survive)."""
        if self._buf_gen != p["gen"]:
            raise ZutisHipError("run_plan: the engine's buffers changed since the plan was built")
        if self._version_key() != p["weights"]:
            raise ZutisHipError("run_plan: parameters changed since the plan was built (it holds the old packed weights): rebuild it")
        if x is not None:
            p["x"].copy_(x)
        p["plan"].run(torch.cuda.current_stream().cuda_stream)
        return p["out"], p["labels"]

    # ------------------------------------------------------------------ predict (semantic)
    def semantic_logits_lowres(self, patch_tokens: torch.Tensor, text: torch.Tensor) -> torch.Tensor:
        """einsum("nc,bchw->bnhw") zutis.py:361-365 -> f32 [B,n,h,w]."""
        B, h, w, E = patch_tokens.shape
        n = text.shape[0]
        xl = self._x3("logits")
        pt16 = self._abuf("pt16", (B * h * w, E), xl, unit_norm=True)
        src = self._pt16_of
        if not (src is not None and src[0]() is patch_tokens and src[1] == patch_tokens._version):
            # tokens not produced by the last forward(): a caller's own tensor.  The 2^10 storage scale of unit-norm rows would push
            # |x| >= 64 past the fp16 range (inf -> NaN logits -> a silent argmax), so rows that are not unit-norm-like keep scale 1
            # (one max-abs read on this rare path; forward()'s own tokens are unit-norm by construction and never come here)
            if pt16.out_scale != 1.0 and not self._unit_scale_ok(patch_tokens):
                pt16 = Act(pt16.t)
            ops.cast_f16(patch_tokens.contiguous(), pt16, B * h * w, E)
            self._pt16_of = None
        t32 = text.detach().to(device=patch_tokens.device, dtype=f32).contiguous()
        t16 = self._abuf("text16", (n, E), xl, unit_norm=True)
        recording = _lib.RECORDER is not None                              # a launch plan always contains the cast
        src = self._text16_of
        same = src is not None and src[0]() is text and src[1] == text._version and src[2] == self._buf_gen
        # caller-supplied category embeddings: checked ONCE per (tensor, version) — un-normalised text features take scale 1
        scaled = src[3] if same else (t16.out_scale == 1.0 or self._unit_scale_ok(t32))
        if not scaled:
            t16 = Act(t16.t)
        if recording or not same:                                            # eager: the category embeddings rarely change
            ops.cast_f16(t32, t16, n, E)
            self._text16_of = None if recording else (weakref.ref(text), text._version, self._buf_gen, scaled)
        lo = torch.empty((B, n, h, w), dtype=f32, device=patch_tokens.device)
        self._gemm("logits", t16, pt16, lo, M=n, N=h * w, K=E, lda=E, ldw=E, ldc=h * w, batch=B, strideA=0, strideW=h * w * E,
                   strideC=n * h * w)
        return lo

    def predict_semantic(self, patch_tokens: torch.Tensor, text: torch.Tensor, size: Optional[Tuple[int, int]],
                         return_logits: bool = False):
        """zutis.py:355-372.  Labels come from the fused upsample+argmax kernel; [B,n,H,W] is never materialised."""
        lo = self.semantic_logits_lowres(patch_tokens, text)
        B, n, h, w = lo.shape
        if return_logits:
            if size is None:
                return lo
            out = torch.empty((B, n, size[0], size[1]), dtype=f32, device=lo.device)
            ops.upsample_bilinear_nchw(lo, B * n, h, w, size[0], size[1], out=out)
            return out
        H, Wd = (h, w) if size is None else (int(size[0]), int(size[1]))
        labels = torch.empty((B, H, Wd), dtype=torch.int64, device=lo.device)
        ops.upsample_argmax(lo, labels, B, n, h, w, H, Wd)
        return labels

    # ------------------------------------------------------------------ predict (instance)
    def instance_candidates(self, mask_proposals_last: torch.Tensor, patch_tokens: torch.Tensor, text: torch.Tensor,
                            threshold: float = 0.5, temperature: float = 5.0, size: Optional[Tuple[int, int]] = None,
                            range_flag: Optional[torch.Tensor] = None):
        """zutis.py:376-423 on device: binary masks, sizes, confidence, masked-mean tokens, class + score, and the
        full-resolution thresholded masks.  Returns (masks u8 [B,Q,H,W], scores f32 [B,Q], category int64 [B,Q]).
        range_flag (int32 [1], zeroed): bit 0 is set when a proposal lies outside [0, 1] (the asserts of zutis.py:385-386)."""
        self._pack()
        mp = mask_proposals_last.contiguous()
        B, Q, h, w = mp.shape
        M, E, dev = h * w, patch_tokens.shape[-1], mp.device
        pt = patch_tokens.contiguous()
        t32 = text.detach().to(device=dev, dtype=f32).contiguous()
        sizes = torch.empty((B * Q,), dtype=f32, device=dev)
        conf = torch.empty((B * Q,), dtype=f32, device=dev)
        binary = torch.empty((B, Q, h, w), dtype=torch.uint8, device=dev)
        ops.instance_mask_stats(mp, Q * M, threshold, B, Q, M, sizes, conf, binary, range_flag)
        avg = torch.empty((B * Q, E), dtype=f32, device=dev)
        ops.masked_mean_tokens(pt, binary, sizes, avg, B, Q, M, E)
        cat = torch.empty((B, Q), dtype=torch.int64, device=dev)
        score = torch.empty((B, Q), dtype=f32, device=dev)
        ops.instance_classify(avg, t32, conf, temperature, B * Q, t32.shape[0], E, cat, score)
        if size is not None:
            masks = torch.empty((B, Q, size[0], size[1]), dtype=torch.uint8, device=dev)
            ops.upsample_bilinear_nchw(mp, B * Q, h, w, size[0], size[1], mask_u8=masks, threshold=threshold)
        else:
            masks = binary
        return masks, score, cat

    def mask_iou_matrix(self, masks_u8: torch.Tensor, return_areas: bool = False):
        """Pairwise IoU of one image's [Q,H,W] u8 masks: exact popcounts on device, float64 divide on the host
        (= utils/iou.py:30-32 on boolean masks).  The diagonal of the intersection counts is each mask's area."""
        n = masks_u8.shape[0]
        px = masks_u8[0].numel()
        inter = torch.empty((n, n), dtype=torch.int32, device=masks_u8.device)
        uni = torch.empty((n, n), dtype=torch.int32, device=masks_u8.device)
        ops.mask_iou_counts(masks_u8.contiguous(), n, px, inter, uni)
        ih = inter.cpu().numpy()
        iou = ih / (uni.cpu().numpy() + 1e-7)
        return (iou, np.diag(ih).copy()) if return_areas else iou

    def instance_nms(self, masks_u8: torch.Tensor, scores: torch.Tensor, category_ids: torch.Tensor, nms_type: str = "hard",
                     nms_threshold: float = 0.3, sigma: float = 0.5, threshold: float = 0.001):
        """zutis.py:211-299 for a batch, entirely on the device: masks u8 [B,Q,H,W], scores f32 [B,Q], category_ids int64 [B,Q]
        -> list of (batch index, category, query index, score) in the reference's emission order.  Popcount IoU counts per
        image (zh_mask_iou_counts), then one launch of the greedy per-category loop (zh_mask_nms, one workgroup per image);
        only the kept (index, score, category) triples and their count cross PCIe."""
        B, Q, H, W = masks_u8.shape
        dev = masks_u8.device
        inter = torch.empty((B, Q, Q), dtype=torch.int32, device=dev)
        uni = torch.empty((B, Q, Q), dtype=torch.int32, device=dev)
        m = masks_u8.contiguous()
        for b in range(B):
            ops.mask_iou_counts(m[b], Q, H * W, inter[b], uni[b])
        idx, sc, cat, cnt = ops.mask_nms(inter, uni, scores.contiguous(), category_ids.contiguous(), nms_type, nms_threshold, sigma,
                                         threshold)
        # ONE device -> host copy for the five small results (every copy synchronises the stream): indices, categories and counts are
        # small integers, exact in float64 next to the float64 scores
        f64 = torch.float64
        packed = torch.cat([idx.to(f64), sc.to(f64), cat.to(f64), category_ids.to(f64), cnt.to(f64).view(B, 1)], dim=1).cpu().numpy()
        idx_h, sc_h, cat_h = packed[:, :Q], packed[:, Q:2 * Q], packed[:, 2 * Q:3 * Q]    # entries past cnt are uninitialised: read per element
        all_cat, cnt_h = packed[:, 3 * Q:4 * Q].astype(np.int64), packed[:, 4 * Q].astype(np.int64)
        # The kernel walks the categories in ascending id; the reference walks `set(category_ids_per_image)` (zutis.py:237-238), i.e.
        # CPython's iteration order of a set of numpy int64 scalars — ascending only while every id is below the hash table's
        # size.  Re-create that very set on the host (Q ids per image) and order the per-category groups by it (stable: the
        # selection order inside a category is the kernel's, which is the reference's).
        out = []
        for b in range(B):
            rank = {int(c): i for i, c in enumerate(set(all_cat[b]))}
            rows = [(b, int(cat_h[b, j]), int(idx_h[b, j]), float(sc_h[b, j])) for j in range(int(cnt_h[b]))]
            rows.sort(key=lambda r: rank[r[1]])
            out += rows
        return out

    def instance_nms_encode(self, masks_u8: torch.Tensor, scores: torch.Tensor, category_ids: torch.Tensor, nms_type: str = "hard",
                            nms_threshold: float = 0.3, sigma: float = 0.5, threshold: float = 0.001,
                            range_flag: Optional[torch.Tensor] = None, max_runs: int = 8192, pack_head: Optional[int] = None,
                            fused: Optional[bool] = None):
        """instance_nms + encode_masks chained on the device (zutis.py:211-299,423-469): popcount IoU counts, the greedy per-category
        loop, then the run extraction of the kept masks straight from the loop's device outputs (zh_mask_runs_kept) — the NMS result
        does not visit the host in between.  ONE device -> host copy brings the kept triples, every query's category, the counts, the
        range flag, the run counts, the boxes AND the kept masks' COCO RLE strings, encoded on the device (zh_mask_rle_kept) from the packed
        run list; the host encodes only when that list outgrows its PACK_HEAD ints per image (a second copy) or a mask its max_runs.
        Returns (kept [(batch index, category, query index, score)] in the reference's emission order, rles, boxes, areas, status) — status =
        the word behind `range_flag` as the NMS kernel read it (bit ops.STATUS_RANGE: a proposal outside [0, 1]; the engine's own
        status_word() also carries ops.STATUS_NONFINITE from the forward).  fused (None = where supported): runs, boxes, areas and strings
        from ONE launch (zh_mask_rle_fused_kept) instead of run extraction (two launches) + string kernel; same results."""
        from . import rle
        B, Q, H, W = masks_u8.shape
        dev = masks_u8.device
        inter = torch.empty((B, Q, Q), dtype=torch.int32, device=dev)
        uni = torch.empty((B, Q, Q), dtype=torch.int32, device=dev)
        m = masks_u8.contiguous()
        bits = torch.empty((B, Q, (H * W + 63) // 64), dtype=torch.int64, device=dev)     # the IoU step's bit-packed masks, read again below
        for b in range(B):
            ops.mask_iou_counts(m[b], Q, H * W, inter[b], uni[b], workspace=bits[b])
        per_image = (ZutisEngine.PACK_HEAD if B <= 4 else ZutisEngine.PACK_HEAD // 4) if pack_head is None else pack_head
        if fused is None:
            fused = ops.mask_rle_fused_supported(H, W, max_runs)     # masks up to 1024 columns whose bits + tables fit the LDS; else three launches
        if fused:
            # ONE launch behind the NMS loop does runs, boxes, areas and strings (zh_mask_rle_fused_kept: one workgroup per kept mask);
            # ONE buffer = one copy for everything the host needs: [kept triples + categories + count + status (f64) | info | cursor | strings]
            n1, n2 = B * (4 * Q + 2) * 8, B * Q * 8 * 4
            cap = int(max(64, 4 * B * per_image))                    # bytes of strings that ride along (a string is ~2.2 B per transition)
            small = torch.empty((n1 + n2 + 8 + cap,), dtype=torch.uint8, device=dev)
            packed = small[:n1].view(torch.float64).view(B, 4 * Q + 2)
            info = small[n1:n1 + n2].view(torch.int32).view(B * Q, 8)
            cursor = small[n1 + n2:n1 + n2 + 8].view(torch.int32)    # zeroed by the NMS kernel (zero_word), used by the launch behind it
            idx, _, _, cnt = ops.mask_nms(inter, uni, scores.contiguous(), category_ids.contiguous(), nms_type, nms_threshold, sigma, threshold,
                                          packed=packed, range_flag=range_flag, zero_word=cursor)
            ops.mask_rle_fused_kept(m, idx, cnt, max_runs, small[n1 + n2 + 8:], cursor, info, bits=bits)
            host = _to_host(small, getattr(self, "_pinned", None))                                   # the one synchronisation of the predict
            pk = host[:n1].view(np.float64).reshape(B, 4 * Q + 2)
            info_h = host[n1:n1 + n2].view(np.int32).reshape(B, Q, 8)
            chars_h = host[n1 + n2 + 8:]
            cnt_l = pk[:, 4 * Q].astype(np.int64).tolist()
            range_bad = int(pk[:, 4 * Q + 1].max()) if range_flag is not None else 0  # the status word as the NMS kernel read it (ops.STATUS_*)
            kept, rles, boxes, areas = [], [], [], []
            size = [int(H), int(W)]
            redo = []                                                # (position in the output lists, flat mask index): strings the device did not write
            for b in range(B):
                n = cnt_l[b]
                if n == 0:
                    continue
                row, inf = pk[b].tolist(), info_h[b, :n].tolist()
                # the kernel walks the categories in ascending id; the reference walks `set(category_ids_per_image)` (zutis.py:237-238):
                # order the per-category groups by that very set (stable inside a category: the kernel's = the reference's selection order)
                rank = {int(c): i for i, c in enumerate(set(pk[b, 3 * Q:4 * Q].astype(np.int64)))}
                for j in sorted(range(n), key=lambda j: rank[int(row[2 * Q + j])]):
                    q, (c0, ln, x0, y0, x1, y1, ar, _) = int(row[j]), inf[j]
                    if ln < 0:                                       # over max_runs transitions, or the strings outgrew `cap`
                        redo.append((len(kept), b * Q + q))
                    kept.append((b, int(row[2 * Q + j]), q, float(row[Q + j])))
                    rles.append({"size": size, "counts": chars_h[c0:c0 + ln].tobytes()} if ln >= 0 else None)
                    boxes.append([float(x0), float(y0), float(x1), float(y1)])
                    areas.append(int(ar))
            if redo:
                r2, _, _ = ZutisEngine.encode_masks(self, masks_u8.view(B * Q, H, W), np.array([f for _, f in redo], dtype=np.int32))
                for (at, _), r in zip(redo, r2):
                    rles[at] = r
            return kept, rles, boxes, areas, range_bad
        # ONE buffer for everything the host needs: [kept triples + categories + count + status (f64) | run counts | boxes + areas | string
        # lengths | the RLE strings of the kept masks, written by the device (zh_mask_rle_kept) from the packed transition list].  The list
        # itself (PACK_HEAD ints per image) stays on the device.
        n1, n2, n3, n4 = B * (4 * Q + 2) * 8, B * Q * 2 * 4, B * Q * 5 * 4, B * Q * 4
        head = int(min(B * Q * max_runs, B * per_image))
        small = torch.empty((n1 + n2 + n3 + n4 + 5 * head + 16 * B * Q,), dtype=torch.uint8, device=dev)
        packed = small[:n1].view(torch.float64).view(B, 4 * Q + 2)
        nr = small[n1:n1 + n2].view(torch.int32).view(B * Q, 2)
        ba = small[n1 + n2:n1 + n2 + n3].view(torch.int32).view(B * Q, 5)
        slen = small[n1 + n2 + n3:n1 + n2 + n3 + n4].view(torch.int32)
        chars = small[n1 + n2 + n3 + n4:]
        pos_head = torch.empty((head,), dtype=torch.int32, device=dev)
        idx, _, _, cnt = ops.mask_nms(inter, uni, scores.contiguous(), category_ids.contiguous(), nms_type, nms_threshold, sigma, threshold,
                                      packed=packed, range_flag=range_flag)
        ops.mask_runs_kept(m, idx, cnt, max_runs, pos_head, nr, ba, packed=True)
        ops.mask_rle_kept(pos_head, nr, cnt, B, Q, max_runs, H * W, chars, slen)
        host = _to_host(small, getattr(self, "_pinned", None))                                       # the one synchronisation of the predict
        pk = host[:n1].view(np.float64).reshape(B, 4 * Q + 2)
        nr_h = host[n1:n1 + n2].view(np.int32).reshape(B, Q, 2)
        ba_h = host[n1 + n2:n1 + n2 + n3].view(np.int32).reshape(B, Q, 5)
        slen_h = host[n1 + n2 + n3:n1 + n2 + n3 + n4].view(np.int32).reshape(B, Q)
        chars_h = host[n1 + n2 + n3 + n4:]
        cnt_l = pk[:, 4 * Q].astype(np.int64).tolist()
        range_bad = int(pk[:, 4 * Q + 1].max()) if range_flag is not None else 0      # the status word as the NMS kernel read it (ops.STATUS_*)
        kept, rles, boxes, areas = [], [], [], []
        if B and max(cnt_l) > 0:
            lens = [np.minimum(nr_h[b, :cnt_l[b], 0], max_runs).tolist() for b in range(B)]          # list length of every kept mask
            total = sum(sum(l) for l in lens)
            flat = None
            if total > head:                                         # the lists outgrew the head: the whole packed list in a second copy,
                big = torch.empty((total,), dtype=torch.int32, device=dev)                          # strings built on the host
                ops.mask_runs_kept(m, idx, cnt, max_runs, big, nr, ba, packed=True)
                flat = big.cpu().numpy()
            size = [int(H), int(W)]
            at = rank_all = 0
            for b in range(B):
                n = cnt_l[b]
                if n == 0:
                    continue
                if flat is not None:
                    r = rle.rles_from_transitions(flat[at:], nr_h[b, :n], H, W, packed_max_runs=max_runs)     # image b's lists start at `at`
                    at += sum(lens[b])
                else:
                    r, sl = [], slen_h[b, :n].tolist()
                    for j in range(n):                               # mask j's string: 5 * (start of its list) + 16 * (kept masks before it)
                        c0 = 5 * at + 16 * rank_all
                        r.append({"size": size, "counts": chars_h[c0:c0 + sl[j]].tobytes()} if sl[j] >= 0 else None)
                        at += lens[b][j]
                        rank_all += 1
                # the kernel walks the categories in ascending id; the reference walks `set(category_ids_per_image)` (zutis.py:237-238):
                # order the per-category groups by that very set (stable inside a category: the kernel's = the reference's selection order)
                row = pk[b].tolist()
                rank = {int(c): i for i, c in enumerate(set(pk[b, 3 * Q:4 * Q].astype(np.int64)))}
                order = sorted(range(n), key=lambda j: rank[int(row[2 * Q + j])])
                ba_l = ba_h[b, :n].tolist()
                for j in order:
                    q = int(row[j])
                    if r[j] is None:                                 # pathological mask (> max_runs transitions): the host encoder
                        r[j] = rle.encode(masks_u8[b, q].cpu().numpy())
                    kept.append((b, int(row[2 * Q + j]), q, float(row[Q + j])))
                    rles.append(r[j])
                    boxes.append([float(v) for v in ba_l[j][:4]])
                    areas.append(int(ba_l[j][4]))
        return kept, rles, boxes, areas, range_bad

    # ints of the packed transition list that ride along with the small tables, per image (256 KB; a quarter of it per image in batches
    # above 4): the 17 kept masks of the config-3 fixture (480x640, noisy: ~1750 transitions each) are 29.8 k
    PACK_HEAD = 65536

    def encode_masks(self, masks_u8: torch.Tensor, sel: np.ndarray, max_runs: int = 8192):
        """COCO RLE dicts, xyxy boxes and areas of the masks `sel` (flat indices into [n,H,W]) without moving the masks
        to the host: zh_mask_runs extracts the column-major run boundaries on the device; only those cross PCIe."""
        from . import rle
        n, H, W = masks_u8.shape
        if len(sel) == 0:
            return [], [], []
        sel_dev = torch.from_numpy(np.ascontiguousarray(sel, dtype=np.int32)).to(masks_u8.device)
        pos, nr, ba = ops.mask_runs(masks_u8.contiguous(), sel_dev, max_runs)
        nb_h = torch.cat([nr, ba], dim=1).cpu().numpy()           # one copy (= one stream synchronisation) for both small tables
        nr_h, ba_h = nb_h[:, :2], nb_h[:, 2:]
        keep = int(min(max_runs, max(1, nr_h[:, 0].max())))
        pos_h = pos[:, :keep].cpu().numpy()
        rles = rle.rles_from_transitions(pos_h, nr_h, H, W)      # all strings in one C call (was one numpy diff + ctypes call per mask)
        for j, q in enumerate(sel):
            if rles[j] is None:                                  # pathological mask (> max_runs transitions): the host encoder
                rles[j] = rle.encode(masks_u8[int(q)].cpu().numpy())
        boxes = [[float(v) for v in row[:4]] for row in ba_h]
        areas = [int(row[4]) for row in ba_h]
        return rles, boxes, areas


from .engine_clip import ClipImageEncoder, ClipTextEncoder      # noqa: E402,F401  (re-exports)
from .engine_selfmask import SelfMaskEngine                     # noqa: E402,F401
