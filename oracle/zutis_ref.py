"""ORACLE — test infrastructure, NOT the product path.

CPU restatement (plain PyTorch fp32 functional ops + NumPy) of the reference's
ZUTIS dense-prediction hot path.  Only tests/, __graft_entry__.smoke() and
bench.py's baseline legs (cpu_baseline; the optional --torch-gpu-baseline, which
runs these same functions on the GPU as the "stock PyTorch" reference) may import this package; the product path
(zutis_amd/) never does.

Pinned: every function here is checked against outputs of the real reference
(/root/reference imported in the authoring container by oracle/gen_golden.py)
through the committed fixtures in tests/golden/ — see tests/test_oracle_golden.py.

Layout is this build's own (batch-first, channels-last tokens [B,T,D]); the
reference's sequence-first layout is a layout choice only.  Parameters are a
flat dict with the reference state_dict keys.

Each function cites the reference file:line it restates (paths relative to
/root/reference).
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from . import resample as R

Tensor = torch.Tensor


def to_torch_params(sd) -> Dict[str, Tensor]:
    return {k: (v if isinstance(v, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(v))).float()
            for k, v in sd.items()}


# --------------------------------------------------------------------------- encoder
def layer_norm(x: Tensor, w: Optional[Tensor], b: Optional[Tensor], eps: float = 1e-5) -> Tensor:
    """networks/clip_arch.py:286-292 (fp32 LayerNorm over the last dim, biased variance)."""
    return F.layer_norm(x, (x.shape[-1],), w, b, eps)


# bench.py's GPU-eager baseline leg only: route the encoder's attention (need_weights=False call sites) through
# F.scaled_dot_product_attention — what nn.MultiheadAttention(batch_first=False, need_weights=False) dispatches to via
# F.multi_head_attention_forward in this PyTorch (the fused SDPA kernel) — instead of the explicit matmul-softmax-matmul.
ENCODER_SDPA = False


def mha(q_in: Tensor, k_in: Tensor, v_in: Tensor, in_w: Tensor, in_b: Tensor,
        out_w: Tensor, out_b: Tensor, n_heads: int, attn_mask: Optional[Tensor] = None, need_weights: bool = True) -> Tensor:
    """nn.MultiheadAttention forward (packed in_proj rows = [Wq;Wk;Wv]), batch-first.

    Reference call sites: networks/clip_arch.py:314-316 (self-attention, need_weights=False)
    and networks/transformer.py:272-286 (decoder self/cross attention, need_weights left at its default True).
    q_in [B,Tq,D], k_in/v_in [B,Tk,D] -> [B,Tq,D].  Scale 1/sqrt(dh), softmax over keys.
    need_weights=False + ENCODER_SDPA: the same function through torch's fused SDPA (baseline timing leg only).
    """
    B, Tq, D = q_in.shape
    Tk = k_in.shape[1]
    dh = D // n_heads
    q = F.linear(q_in, in_w[:D], in_b[:D]).view(B, Tq, n_heads, dh).transpose(1, 2)
    k = F.linear(k_in, in_w[D:2 * D], in_b[D:2 * D]).view(B, Tk, n_heads, dh).transpose(1, 2)
    v = F.linear(v_in, in_w[2 * D:], in_b[2 * D:]).view(B, Tk, n_heads, dh).transpose(1, 2)
    if ENCODER_SDPA and not need_weights:
        o = F.scaled_dot_product_attention(q, k, v, attn_mask=attn_mask).transpose(1, 2).reshape(B, Tq, D)
        return F.linear(o, out_w, out_b)
    s = torch.matmul(q * (1.0 / math.sqrt(dh)), k.transpose(-1, -2))
    if attn_mask is not None:                    # additive float mask [Tq,Tk] (clip_arch.py:525-531: -inf above the diagonal)
        s = s + attn_mask
    p = torch.softmax(s, dim=-1)
    o = torch.matmul(p, v).transpose(1, 2).reshape(B, Tq, D)
    return F.linear(o, out_w, out_b)


def interpolate_positional_embedding(pos: Tensor, h: int, w: int) -> Tensor:
    """networks/clip_arch.py:356-374: bicubic with scale_factor=((h+.1)/g,(w+.1)/g) [L,D] -> [1+h*w, D]."""
    g = int(math.isqrt(pos.shape[0] - 1))
    patch = pos[1:].detach().cpu().numpy().reshape(g, g, -1)
    out = R.bicubic_cl(patch, h, w, scale_factor_h=(h + 0.1) / g, scale_factor_w=(w + 0.1) / g)
    return torch.cat([pos[:1], torch.from_numpy(out.reshape(h * w, -1)).to(pos.device)], dim=0)


def clip_vit_forward(P: Dict[str, Tensor], x: Tensor, patch: int, prefix: str = "encoder.") -> Tuple[Tensor, int, int]:
    """networks/clip_arch.py:377-411 VisionTransformer.forward -> (patch_tokens [B,hw,D], h, w).  No @proj."""
    B = x.shape[0]
    D = P[prefix + "class_embedding"].shape[0]
    heads = D // 64
    t = F.conv2d(x, P[prefix + "conv1.weight"], None, stride=patch)           # :378
    h, w = t.shape[-2:]
    t = t.reshape(B, D, h * w).permute(0, 2, 1)                                # :381-382
    cls = P[prefix + "class_embedding"][None, None].expand(B, 1, D)
    t = torch.cat([cls, t], dim=1)                                             # :384-390
    t = t + interpolate_positional_embedding(P[prefix + "positional_embedding"], h, w)[None]  # :392-396
    t = layer_norm(t, P[prefix + "ln_pre.weight"], P[prefix + "ln_pre.bias"])  # :397
    n_layers = 1 + max(int(k.split(".")[3]) for k in P if k.startswith(prefix + "transformer.resblocks."))
    for i in range(n_layers):                                                  # :400, block = :318-321
        p = f"{prefix}transformer.resblocks.{i}."
        y = layer_norm(t, P[p + "ln_1.weight"], P[p + "ln_1.bias"])
        t = t + mha(y, y, y, P[p + "attn.in_proj_weight"], P[p + "attn.in_proj_bias"],
                    P[p + "attn.out_proj.weight"], P[p + "attn.out_proj.bias"], heads, need_weights=False)
        y = layer_norm(t, P[p + "ln_2.weight"], P[p + "ln_2.bias"])
        y = F.linear(y, P[p + "mlp.c_fc.weight"], P[p + "mlp.c_fc.bias"])
        y = y * torch.sigmoid(1.702 * y)                                       # QuickGELU :295-297
        t = t + F.linear(y, P[p + "mlp.c_proj.weight"], P[p + "mlp.c_proj.bias"])
    t = layer_norm(t[:, 1:], P[prefix + "ln_post.weight"], P[prefix + "ln_post.bias"])  # :403-404
    return t, h, w


def clip_encode_image(P: Dict[str, Tensor], x: Tensor, patch: int, prefix: str = "encoder.") -> Tensor:
    """Original CLIP CLS embedding (clip_arch.py:413-431 comments, :531-532): ln_post(x[:,0]) @ proj, fixed pos-embed,
    then L2 normalise (utils/extract_image_embeddings.py:72-73).  Pinned: tests/golden/encode_image.npz holds the embeddings of
    the reference's own VisionTransformer submodules run in the order of that original forward (gen_golden.py::gen_encode_image)."""
    B = x.shape[0]
    D = P[prefix + "class_embedding"].shape[0]
    heads = D // 64
    t = F.conv2d(x, P[prefix + "conv1.weight"], None, stride=patch)
    t = t.reshape(B, D, -1).permute(0, 2, 1)
    t = torch.cat([P[prefix + "class_embedding"][None, None].expand(B, 1, D), t], dim=1)
    t = t + P[prefix + "positional_embedding"][None]
    t = layer_norm(t, P[prefix + "ln_pre.weight"], P[prefix + "ln_pre.bias"])
    n_layers = 1 + max(int(k.split(".")[3]) for k in P if k.startswith(prefix + "transformer.resblocks."))
    for i in range(n_layers):
        p = f"{prefix}transformer.resblocks.{i}."
        y = layer_norm(t, P[p + "ln_1.weight"], P[p + "ln_1.bias"])
        t = t + mha(y, y, y, P[p + "attn.in_proj_weight"], P[p + "attn.in_proj_bias"],
                    P[p + "attn.out_proj.weight"], P[p + "attn.out_proj.bias"], heads, need_weights=False)
        y = layer_norm(t, P[p + "ln_2.weight"], P[p + "ln_2.bias"])
        y = F.linear(y, P[p + "mlp.c_fc.weight"], P[p + "mlp.c_fc.bias"])
        y = y * torch.sigmoid(1.702 * y)
        t = t + F.linear(y, P[p + "mlp.c_proj.weight"], P[p + "mlp.c_proj.bias"])
    e = layer_norm(t[:, 0], P[prefix + "ln_post.weight"], P[prefix + "ln_post.bias"]) @ P[prefix + "proj"]
    return e / e.norm(dim=-1, keepdim=True)


def clip_encode_text(P: Dict[str, Tensor], tokens: Tensor, prefix: str = "") -> Tensor:
    """networks/clip_arch.py:534-547 CLIP.encode_text: token embedding + positional embedding, pre-LN resblocks (:318-321)
    under the causal mask of build_attention_mask (:525-531), ln_final, the EOT row (argmax of the token ids) @
    text_projection.  tokens int64 [n, ctx] -> f32 [n, embed] (NOT normalised).  heads = width // 64 (:606)."""
    n, ctx = tokens.shape
    D = P[prefix + "positional_embedding"].shape[1]
    heads = D // 64
    t = P[prefix + "token_embedding.weight"][tokens] + P[prefix + "positional_embedding"][None]
    mask = torch.full((ctx, ctx), float("-inf")).triu_(1)
    k0 = len((prefix + "transformer.resblocks.").split(".")) - 1
    n_layers = 1 + max(int(k.split(".")[k0]) for k in P if k.startswith(prefix + "transformer.resblocks."))
    for i in range(n_layers):
        p = f"{prefix}transformer.resblocks.{i}."
        y = layer_norm(t, P[p + "ln_1.weight"], P[p + "ln_1.bias"])
        t = t + mha(y, y, y, P[p + "attn.in_proj_weight"], P[p + "attn.in_proj_bias"],
                    P[p + "attn.out_proj.weight"], P[p + "attn.out_proj.bias"], heads, attn_mask=mask)
        y = layer_norm(t, P[p + "ln_2.weight"], P[p + "ln_2.bias"])
        y = F.linear(y, P[p + "mlp.c_fc.weight"], P[p + "mlp.c_fc.bias"])
        y = y * torch.sigmoid(1.702 * y)
        t = t + F.linear(y, P[p + "mlp.c_proj.weight"], P[p + "mlp.c_proj.bias"])
    t = layer_norm(t, P[prefix + "ln_final.weight"], P[prefix + "ln_final.bias"])
    return t[torch.arange(n), tokens.argmax(dim=-1)] @ P[prefix + "text_projection"]


def prompt_ensemble(P: Dict[str, Tensor], tokens: Tensor, prefix: str = "") -> Tensor:
    """utils/extract_text_embeddings.py:98-115: per category, encode the T prompts, L2-normalise each, average, L2-normalise.
    tokens int64 [C, T, ctx] -> f32 [C, embed]."""
    out = []
    for c in range(tokens.shape[0]):
        e = clip_encode_text(P, tokens[c], prefix)
        e = e / e.norm(dim=-1, keepdim=True)
        m = e.mean(dim=0)
        out.append(m / m.norm())
    return torch.stack(out)


# --------------------------------------------------------------------------- head
def mlp3(P: Dict[str, Tensor], name: str, x: Tensor) -> Tensor:
    """networks/zutis.py:535-549 MLP(num_layers=3): Linear-ReLU-Linear-ReLU-Linear."""
    x = F.relu(F.linear(x, P[f"{name}.layers.0.weight"], P[f"{name}.layers.0.bias"]))
    x = F.relu(F.linear(x, P[f"{name}.layers.1.weight"], P[f"{name}.layers.1.bias"]))
    return F.linear(x, P[f"{name}.layers.2.weight"], P[f"{name}.layers.2.bias"])


def sine_pe(h: int, w: int, n_dims: int, temperature: float = 10000.0) -> Tensor:
    """networks/positional_embedding.py:29-52 (normalize=True, scale=2π, num_pos_feats=n_dims/2) -> [h*w, n_dims]
    channels-last; channels [0,n/2) = y part, [n/2,n) = x part; even i sin, odd i cos."""
    npf = n_dims // 2
    y = torch.arange(1, h + 1, dtype=torch.float32)
    x = torch.arange(1, w + 1, dtype=torch.float32)
    y = y / (y[-1:] + 1e-6) * (2 * math.pi)
    x = x / (x[-1:] + 1e-6) * (2 * math.pi)
    dim_t = torch.arange(npf, dtype=torch.float32)
    dim_t = temperature ** (2 * (dim_t // 2) / npf)
    px = x[:, None] / dim_t
    py = y[:, None] / dim_t
    px = torch.stack((px[:, 0::2].sin(), px[:, 1::2].cos()), dim=2).flatten(1)   # [w, npf]
    py = torch.stack((py[:, 0::2].sin(), py[:, 1::2].cos()), dim=2).flatten(1)   # [h, npf]
    pos = torch.cat([py[:, None, :].expand(h, w, npf), px[None, :, :].expand(h, w, npf)], dim=2)
    return pos.reshape(h * w, n_dims).contiguous()


def decoder_forward(P: Dict[str, Tensor], memory: Tensor, pos: Tensor, query_pos: Tensor,
                    n_heads: int, prefix: str = "decoder.", return_intermediate: bool = True) -> Tensor:
    """networks/transformer.py:114-152 (TransformerDecoder) over :262-291 (forward_post), batch-first.

    memory [B,M,D], pos [M,D] or None, query_pos [Q,D]; tgt = zeros.  Returns [B,L,Q,D] (norm applied to every
    layer's output, stacked in layer order) or [B,Q,D] when return_intermediate=False.
    """
    B, M, D = memory.shape
    qp = query_pos[None].expand(B, -1, -1)
    tgt = torch.zeros_like(qp)
    key_in = memory if pos is None else memory + pos[None]
    n_layers = 1 + max(int(k.split(".")[2]) for k in P if k.startswith(prefix + "layers."))
    outs = []
    for i in range(n_layers):
        p = f"{prefix}layers.{i}."
        q = tgt + qp
        t2 = mha(q, q, tgt, P[p + "self_attn.in_proj_weight"], P[p + "self_attn.in_proj_bias"],
                 P[p + "self_attn.out_proj.weight"], P[p + "self_attn.out_proj.bias"], n_heads)
        tgt = layer_norm(tgt + t2, P[p + "norm1.weight"], P[p + "norm1.bias"])
        t2 = mha(tgt + qp, key_in, memory, P[p + "multihead_attn.in_proj_weight"], P[p + "multihead_attn.in_proj_bias"],
                 P[p + "multihead_attn.out_proj.weight"], P[p + "multihead_attn.out_proj.bias"], n_heads)
        tgt = layer_norm(tgt + t2, P[p + "norm2.weight"], P[p + "norm2.bias"])
        t2 = F.linear(F.relu(F.linear(tgt, P[p + "linear1.weight"], P[p + "linear1.bias"])),
                      P[p + "linear2.weight"], P[p + "linear2.bias"])
        tgt = layer_norm(tgt + t2, P[p + "norm3.weight"], P[p + "norm3.bias"])
        outs.append(layer_norm(tgt, P[prefix + "norm.weight"], P[prefix + "norm.bias"]))
    if return_intermediate:
        return torch.stack(outs, dim=1)
    return outs[-1]


def image_to_text_space(tokens: Tensor, proj: Tensor) -> Tensor:
    """networks/zutis.py:318-322 (ViT, channel_last): x@proj; layer_norm over (h,w,c) no affine; / (||x||_c + 1e-7)."""
    t = tokens @ proj
    t = F.layer_norm(t, t.shape[1:])
    return t / (t.norm(dim=-1, keepdim=True) + 1e-7)


def zutis_forward(P: Dict[str, Tensor], x: Tensor, patch: int, dec_heads: int = 8) -> Dict[str, Tensor]:
    """networks/zutis.py:472-532 ZUTIS.forward (CLIP-ViT branch)."""
    B = x.shape[0]
    tok, h, w = clip_vit_forward(P, x, patch)                                    # :479
    D = tok.shape[-1]
    if tok.is_cuda:      # GPU-eager baseline leg of bench.py: the reference's own call (ATen CUDA kernel), :491-495
        tok = F.interpolate(tok.reshape(B, h, w, D).permute(0, 3, 1, 2), scale_factor=2, mode="bilinear").permute(0, 2, 3, 1)
    else:
        tok = torch.from_numpy(R.bilinear_up2_cl(tok.numpy().reshape(B, h, w, D)))   # :491-495
    h, w = 2 * h, 2 * w
    tok = tok.reshape(B, h * w, D)
    dec_in = mlp3(P, "ffn1", tok)                                                # :500-503
    pos = sine_pe(h, w, D).to(x.device)                                          # :507
    q = decoder_forward(P, dec_in, pos, P["query_embed"], dec_heads)             # :510-513
    q = mlp3(P, "ffn2", q)                                                       # :514
    q = q / q.norm(dim=-1, keepdim=True)                                         # :515
    masks = torch.sigmoid(torch.einsum("bdqc,bnc->bdqn", q, dec_in))             # :196-198,209
    masks = masks.reshape(B, q.shape[1], q.shape[2], h, w)
    pt = image_to_text_space(tok.reshape(B, h, w, D), P["encoder.proj"])         # :528-530
    return {"mask_proposals": masks, "patch_tokens": pt}


# --------------------------------------------------------------------------- predict
def semantic_logits_lowres(patch_tokens: Tensor, text: Tensor) -> Tensor:
    """networks/zutis.py:361-365 einsum("nc,bchw->bnhw") -> [B,n,h,w]."""
    return torch.einsum("nc,bhwc->bnhw", text, patch_tokens).contiguous()


def predict_semantic(patch_tokens: Tensor, text: Tensor, size: Optional[Tuple[int, int]] = None,
                     return_logits: bool = False):
    """networks/zutis.py:355-372: low-res cosine logits -> bilinear to `size` -> argmax (first max on ties) int64."""
    lo = semantic_logits_lowres(patch_tokens, text).numpy()
    if return_logits:
        return torch.from_numpy(lo if size is None else R.bilinear_nchw(lo, size[0], size[1]))
    if size is None:
        return np.argmax(lo, axis=1).astype(np.int64)
    return R.bilinear_argmax_nchw(lo, size[0], size[1])


def instance_scores(mask_proposals: Tensor, patch_tokens: Tensor, text: Tensor,
                    threshold: float = 0.5, temperature: float = 5.0):
    """networks/zutis.py:376-420: per-query confidence, category id and score from last-layer proposals.

    Returns (binary_lowres bool[B,Q,h,w], category_ids int64[B,Q], scores float32[B,Q]).
    """
    mp = mask_proposals[:, -1] if mask_proposals.dim() == 5 else mask_proposals
    binary = mp > threshold
    sizes = binary.sum(dim=(-2, -1))
    conf = (mp * binary).sum(dim=(-2, -1)) / (sizes + 1e-7)
    bf = binary.flatten(2).float()                                               # [B,Q,hw]
    avg = torch.einsum("bqn,bnc->bqc", bf, patch_tokens.flatten(1, 2)) / (sizes.unsqueeze(-1) + 1e-7)
    sem = torch.sigmoid(torch.einsum("nc,bqc->bqn", text, avg / (avg.norm(dim=-1, keepdim=True) + 1e-7)) * temperature)
    cat = torch.argmax(sem, dim=-1)
    score = conf * sem.max(dim=-1).values
    return binary, cat.numpy(), score.numpy()


def compute_iou(pred: np.ndarray, gt: np.ndarray, eps: float = 1e-7):
    """utils/iou.py:6-37 for boolean masks (valid-mask is all-true for {0,1} inputs)."""
    inter = np.logical_and(pred, gt).sum()
    union = np.logical_or(pred, gt).sum()
    return inter / (union + eps)


def mask_nms(masks: np.ndarray, scores: np.ndarray, cats: np.ndarray, nms_type: str = "hard",
             nms_threshold: float = 0.3, sigma: float = 0.5, threshold: float = 0.001):
    """networks/zutis.py:211-299 greedy per-category mask NMS.  Returns list of (category_id, mask_index, score)
    in the reference's emission order: categories in the iteration order of `set(category_ids_per_image)` (:237-238, a set of
    numpy int64 scalars: CPython's hash-slot order, ascending only while the ids are below the table size), then selection
    order.  Empty masks and category 0 dropped."""
    assert nms_type in ("hard", "linear", "gaussian")
    out = []
    for c in set(np.asarray(cats, dtype=np.int64)):
        c = int(c)
        if c == 0:
            continue
        idx = np.nonzero(cats == c)[0]
        cand = list(idx)
        cs = scores[idx].astype(scores.dtype).copy()
        sel = []
        while len(cand) > 0:
            order = np.argsort(cs)
            cand = [cand[i] for i in order]
            cs = cs[order]
            best, best_s = cand[-1], cs[-1]
            sel.append((best, best_s))
            nc, ns = [], []
            for m, s in zip(cand[:-1], cs[:-1]):
                iou = compute_iou(masks[m], masks[best])
                if nms_type == "hard":
                    wgt = 0 if iou > nms_threshold else 1
                elif nms_type == "linear":
                    wgt = (1 - iou) if iou > nms_threshold else 1
                else:
                    wgt = np.exp(-(iou * iou) / sigma)
                s = s * wgt
                if s > threshold:
                    nc.append(m)
                    ns.append(s)
            cand, cs = nc, np.array(ns)
        for m, s in sel:
            if masks[m].sum() == 0:
                continue
            out.append((c, int(m), float(s)))
    return out


def confusion_hist(label_true: np.ndarray, label_pred: np.ndarray, n_class: int) -> np.ndarray:
    """utils/running_score.py:11-16 _fast_hist."""
    lt, lp = label_true.reshape(-1), label_pred.reshape(-1)
    m = (lt >= 0) & (lt < n_class)
    return np.bincount(n_class * lt[m].astype(int) + lp[m], minlength=n_class ** 2).reshape(n_class, n_class)


def scores_from_hist(hist: np.ndarray):
    """utils/running_score.py:22-49 get_scores."""
    hist = hist.astype(np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        acc = np.diag(hist).sum() / hist.sum()
        acc_cls = np.nanmean(np.diag(hist) / hist.sum(axis=1))
        iu = np.diag(hist) / (hist.sum(axis=1) + hist.sum(axis=0) - np.diag(hist))
        mean_iu = np.nanmean(iu)
        freq = hist.sum(axis=1) / hist.sum()
        fwavacc = (freq[freq > 0] * iu[freq > 0]).sum()
    return {"Pixel Acc": acc, "Mean Acc": acc_cls, "FreqW Acc": fwavacc, "Mean IoU": mean_iu}, dict(enumerate(iu))


def retrieve_topk(text: np.ndarray, image: np.ndarray, k: int):
    """datasets/index_dataset.py:163-167: similarities = text @ image.T; per category argsort(descending)[:k].
    Tie order is unspecified in the reference (torch.argsort is not stable); the restatement fixes it to ascending index."""
    sim = text.astype(np.float32) @ image.astype(np.float32).T
    idx = np.stack([np.lexsort((np.arange(sim.shape[1]), -row))[:k] for row in sim])
    return idx, np.take_along_axis(sim, idx, axis=1)


def merge_topk(cand_idx: np.ndarray, cand_val: np.ndarray, k: int):
    """Merge of per-chunk / per-rank candidate lists (new code: the reference is single-GPU and argsorts everything at once,
    datasets/index_dataset.py:163-167): the k best of each row by (score descending, column ascending); returns the
    candidates' global indices and scores.  Padding columns carry index -1 and score -inf."""
    order = np.stack([np.lexsort((np.arange(cand_val.shape[1]), -row))[:k] for row in cand_val])
    return np.take_along_axis(cand_idx, order, axis=1), np.take_along_axis(cand_val, order, axis=1)
