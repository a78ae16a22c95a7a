"""CPU study (no GPU needed): which contractions of the ZUTIS forward tolerate fp16 MFMA operands?

Re-runs the oracle's op sequence with the operand rounding of the HIP engine emulated site by site (fp32 accumulate
throughout), on the stress model of detgen.stress_state_dict (x100 outlier residual channels, true-fp32 weights), and
prints the error of the class logits / mask proposals against the plain fp32 oracle for a set of precision policies.

    python tools/precision_study.py [--batch 1] [--size 336] [--stress 1]

A policy maps a site name to "f16" (operands rounded to fp16), "x3" (hi+lo fp16 split, the dropped lo*lo term emulated)
or "f32".  Sites: conv, qkv, attn_qk, attn_pv, out, fc, proj (encoder); up_ffn1, dec_kv, dec_attn, dec_lin, ffn2, mask,
textproj, logits (head).
"""
from __future__ import annotations

import argparse
import math
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import zutis_ref as O          # noqa: E402
from oracle import resample as R           # noqa: E402
from zutis_amd import detgen               # noqa: E402


def q16(x):
    return x.half().float()


def split(x):
    hi = x.half().float()
    lo = (x - hi).half().float()
    return hi, lo


def lin(x, w, b, mode, out16=False):
    """x @ w.T + b with the operand rounding of `mode`; out16 = result stored as fp16 (consumed by an fp16 kernel)."""
    if mode == "f16":
        y = F.linear(q16(x), q16(w))
    elif mode == "x3":
        xh, xl = split(x)
        wh, wl = split(w)
        y = F.linear(xh, wh) + (F.linear(xh, wl) + F.linear(xl, wh))
    else:
        y = F.linear(x, w)
    if b is not None:
        y = y + b
    return y


def store(x, mode):
    """An activation handed to the next kernel: fp16, split pair (22 bits), or fp32."""
    if mode == "f16":
        return q16(x)
    if mode == "x3":
        hi, lo = split(x)
        return hi + lo
    return x


def attention(q, k, v, heads, pol_qk, pol_pv):
    B, Tq, D = q.shape
    Tk = k.shape[1]
    dh = D // heads
    q = q.view(B, Tq, heads, dh).transpose(1, 2)
    k = k.view(B, Tk, heads, dh).transpose(1, 2)
    v = v.view(B, Tk, heads, dh).transpose(1, 2)
    if pol_qk == "f16":
        s = torch.matmul(q16(q), q16(k).transpose(-1, -2))
    elif pol_qk == "x3":
        qh, ql = split(q)
        kh, kl = split(k)
        s = torch.matmul(qh, kh.transpose(-1, -2)) + (torch.matmul(qh, kl.transpose(-1, -2)) + torch.matmul(ql, kh.transpose(-1, -2)))
    else:
        s = torch.matmul(q, k.transpose(-1, -2))
    s = s * (1.0 / math.sqrt(dh))
    m = s.max(dim=-1, keepdim=True).values
    e = torch.exp(s - m)
    l = e.sum(dim=-1, keepdim=True)
    if pol_pv == "f16":
        o = torch.matmul(q16(e), q16(v)) / l
    else:
        o = torch.matmul(e, v) / l
    return o.transpose(1, 2).reshape(B, Tq, D)


def forward(P, x, patch, dec_heads, pol):
    g = lambda k: pol.get(k, pol.get("default", "f16"))
    B = x.shape[0]
    D = P["encoder.class_embedding"].shape[0]
    heads = D // 64
    h, w = (x.shape[2] - patch) // patch + 1, (x.shape[3] - patch) // patch + 1
    cols = F.unfold(x[:, :, :h * patch, :w * patch], patch, stride=patch).transpose(1, 2)       # [B, hw, 3pp]
    t = lin(cols, P["encoder.conv1.weight"].reshape(D, -1), None, g("conv"))
    t = torch.cat([P["encoder.class_embedding"][None, None].expand(B, 1, D), t], 1)
    t = t + O.interpolate_positional_embedding(P["encoder.positional_embedding"], h, w)[None]
    t = O.layer_norm(t, P["encoder.ln_pre.weight"], P["encoder.ln_pre.bias"])
    L = 1 + max(int(k.split(".")[3]) for k in P if k.startswith("encoder.transformer.resblocks."))
    for i in range(L):
        p = f"encoder.transformer.resblocks.{i}."
        y = O.layer_norm(t, P[p + "ln_1.weight"], P[p + "ln_1.bias"])
        qkv = lin(y, P[p + "attn.in_proj_weight"], P[p + "attn.in_proj_bias"], g("qkv"))
        q, k, v = qkv.split(D, dim=-1)
        o = attention(store(q, g("attn_qk")), store(k, g("attn_qk")), store(v, g("attn_pv")), heads, g("attn_qk"), g("attn_pv"))
        t = t + lin(store(o, g("out")), P[p + "attn.out_proj.weight"], P[p + "attn.out_proj.bias"], g("out"))
        y = O.layer_norm(t, P[p + "ln_2.weight"], P[p + "ln_2.bias"])
        y = lin(y, P[p + "mlp.c_fc.weight"], P[p + "mlp.c_fc.bias"], g("fc"))
        y = y * torch.sigmoid(1.702 * y)
        t = t + lin(store(y, g("proj")), P[p + "mlp.c_proj.weight"], P[p + "mlp.c_proj.bias"], g("proj"))
    tok = O.layer_norm(t[:, 1:], P["encoder.ln_post.weight"], P["encoder.ln_post.bias"])
    tok = torch.from_numpy(R.bilinear_up2_cl(tok.numpy().reshape(B, h, w, D)))
    h, w = 2 * h, 2 * w
    tok = tok.reshape(B, h * w, D)
    m1 = g("up_ffn1")
    a = torch.relu(lin(tok, P["ffn1.layers.0.weight"], P["ffn1.layers.0.bias"], m1))
    a = torch.relu(lin(store(a, m1), P["ffn1.layers.1.weight"], P["ffn1.layers.1.bias"], m1))
    dec_in = lin(store(a, m1), P["ffn1.layers.2.weight"], P["ffn1.layers.2.bias"], m1)
    mem = store(dec_in, g("dec_kv"))
    pos = O.sine_pe(h, w, D)
    key_in = store(mem + pos[None], g("dec_kv"))
    qp = P["query_embed"][None].expand(B, -1, -1)
    tgt = torch.zeros_like(qp)
    Ld = 1 + max(int(k.split(".")[2]) for k in P if k.startswith("decoder.layers."))
    outs = []
    ma, ml = g("dec_attn"), g("dec_lin")
    for i in range(Ld):
        p = f"decoder.layers.{i}."
        sw, sb = P[p + "self_attn.in_proj_weight"], P[p + "self_attn.in_proj_bias"]
        qi = tgt + qp
        q = lin(qi, sw[:D], sb[:D], ml); k = lin(qi, sw[D:2 * D], sb[D:2 * D], ml); v = lin(tgt, sw[2 * D:], sb[2 * D:], ml)
        o = attention(store(q, ma), store(k, ma), store(v, "f16" if ma != "f32" else "f32"), dec_heads, ma, "f16" if ma != "f32" else "f32")
        tgt = O.layer_norm(tgt + lin(store(o, ml), P[p + "self_attn.out_proj.weight"], P[p + "self_attn.out_proj.bias"], ml),
                           P[p + "norm1.weight"], P[p + "norm1.bias"])
        cw, cb = P[p + "multihead_attn.in_proj_weight"], P[p + "multihead_attn.in_proj_bias"]
        q = lin(tgt + qp, cw[:D], cb[:D], ml)
        k = lin(key_in, cw[D:2 * D], cb[D:2 * D], g("dec_kv"))
        v = lin(mem, cw[2 * D:], cb[2 * D:], g("dec_kv"))
        o = attention(store(q, ma), store(k, ma), store(v, "f16" if ma != "f32" else "f32"), dec_heads, ma, "f16" if ma != "f32" else "f32")
        tgt = O.layer_norm(tgt + lin(store(o, ml), P[p + "multihead_attn.out_proj.weight"], P[p + "multihead_attn.out_proj.bias"], ml),
                           P[p + "norm2.weight"], P[p + "norm2.bias"])
        f = torch.relu(lin(tgt, P[p + "linear1.weight"], P[p + "linear1.bias"], ml))
        tgt = O.layer_norm(tgt + lin(store(f, ml), P[p + "linear2.weight"], P[p + "linear2.bias"], ml), P[p + "norm3.weight"], P[p + "norm3.bias"])
        outs.append(O.layer_norm(tgt, P["decoder.norm.weight"], P["decoder.norm.bias"]))
    qd = torch.stack(outs, 1)
    m2 = g("ffn2")
    a = torch.relu(lin(qd, P["ffn2.layers.0.weight"], P["ffn2.layers.0.bias"], m2))
    a = torch.relu(lin(store(a, m2), P["ffn2.layers.1.weight"], P["ffn2.layers.1.bias"], m2))
    qd = lin(store(a, m2), P["ffn2.layers.2.weight"], P["ffn2.layers.2.bias"], m2)
    qd = qd / qd.norm(dim=-1, keepdim=True)
    mm = g("mask")
    if mm == "f16":
        masks = torch.sigmoid(torch.einsum("bdqc,bnc->bdqn", q16(qd), q16(dec_in)))
    elif mm == "x3":
        ah, al = split(qd); bh, bl = split(dec_in)
        masks = torch.sigmoid(torch.einsum("bdqc,bnc->bdqn", ah, bh) + torch.einsum("bdqc,bnc->bdqn", ah, bl) + torch.einsum("bdqc,bnc->bdqn", al, bh))
    else:
        masks = torch.sigmoid(torch.einsum("bdqc,bnc->bdqn", qd, dec_in))
    ts = lin(tok, P["encoder.proj"].t(), None, g("textproj"))
    ts = F.layer_norm(ts.reshape(B, h, w, -1), (h, w, ts.shape[-1]))
    pt = ts / (ts.norm(dim=-1, keepdim=True) + 1e-7)
    return {"mask_proposals": masks.reshape(B, Ld, -1, h, w), "patch_tokens": pt}


def logits(pt, text, mode):
    B, h, w, E = pt.shape
    return lin(pt.reshape(B, h * w, E), text, None, mode).transpose(1, 2).reshape(B, -1, h, w)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--size", type=int, default=336)
    ap.add_argument("--stress", type=float, default=100.0, help="outlier magnitude (0 = the plain detgen weights)")
    ap.add_argument("--tiny", action="store_true")
    ap.add_argument("--sharp", type=float, default=1.0, help="scale of the encoder's q/k projection rows (sharper attention)")
    args = ap.parse_args()
    cfg = detgen.TINY if args.tiny else detgen.VIT_B16
    sd = detgen.stress_state_dict(cfg, args.stress) if args.stress > 0 else detgen.zutis_state_dict(cfg)
    P = O.to_torch_params(sd)
    if args.sharp != 1.0:
        D = cfg.width
        for i in range(cfg.layers):
            P[f"encoder.transformer.resblocks.{i}.attn.in_proj_weight"][:2 * D] *= args.sharp
    x = torch.from_numpy(detgen.images(args.batch, args.size, args.size))
    text = torch.from_numpy(detgen.text_embeddings(81, cfg.embed_dim))
    with torch.no_grad():
        ref = O.zutis_forward(P, x, cfg.patch, cfg.dec_heads)
        lo_ref = O.semantic_logits_lowres(ref["patch_tokens"], text)
        lab_ref = O.predict_semantic(ref["patch_tokens"], text, size=(args.size, args.size))
        # residual-stream statistics of the stress model
        tok, h, w = O.clip_vit_forward(P, x, cfg.patch)
        print(f"stress={args.stress}: |logit| mean {lo_ref.abs().mean():.3f} max {lo_ref.abs().max():.3f}")
        pols = {
            "all f32 (sanity)": {"default": "f32"},
            "all f16 (round-1 engine)": {"default": "f16"},
            "all x3": {"default": "x3", "attn_pv": "f16"},
            "all x3, f16 attention (enc+dec)": {"default": "x3", "attn_pv": "f16", "attn_qk": "f16", "dec_attn": "f16"},
            "fast: f16, x3 for mask/textproj/logits/ffn2": {"default": "f16", "mask": "x3", "textproj": "x3", "logits": "x3", "ffn2": "x3"},
            "fast2: + up_ffn1 x3": {"default": "f16", "mask": "x3", "textproj": "x3", "logits": "x3", "ffn2": "x3", "up_ffn1": "x3"},
            "x3: qkv": {"default": "f16", "qkv": "x3"},
            "x3: qkv+attn_qk": {"default": "f16", "qkv": "x3", "attn_qk": "x3"},
            "x3: out+proj (residual writers)": {"default": "f16", "out": "x3", "proj": "x3"},
            "x3: fc+proj": {"default": "f16", "fc": "x3", "proj": "x3"},
            "x3: encoder gemms, f16 attn": {"default": "f16", "conv": "x3", "qkv": "x3", "out": "x3", "fc": "x3", "proj": "x3"},
            "x3: encoder all": {"default": "f16", "conv": "x3", "qkv": "x3", "attn_qk": "x3", "out": "x3", "fc": "x3", "proj": "x3"},
            "x3: encoder all + textproj+logits": {"default": "f16", "conv": "x3", "qkv": "x3", "attn_qk": "x3", "out": "x3", "fc": "x3", "proj": "x3", "textproj": "x3", "logits": "x3"},
            "f16 encoder, x3 head": {"default": "x3", "conv": "f16", "qkv": "f16", "attn_qk": "f16", "attn_pv": "f16", "out": "f16", "fc": "f16", "proj": "f16"},
        }
        for name, pol in pols.items():
            out = forward(P, x, cfg.patch, cfg.dec_heads, pol)
            lo = logits(out["patch_tokens"], text, pol.get("logits", pol.get("default")))
            lab = R.bilinear_argmax_nchw(lo.numpy(), args.size, args.size)
            e_lo = (lo - lo_ref).abs().max().item()
            e_m = (out["mask_proposals"] - ref["mask_proposals"]).abs().max().item()
            e_pt = (out["patch_tokens"] - ref["patch_tokens"]).abs().max().item()
            print(f"{name:40s} logits {e_lo:.2e}  masks {e_m:.2e}  patch_tokens {e_pt:.2e}  labels {float((lab == lab_ref).mean()):.5f}", flush=True)


if __name__ == "__main__":
    main()
