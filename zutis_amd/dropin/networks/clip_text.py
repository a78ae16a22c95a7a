"""MI355X text tower behind the call surface the reference uses for text embeddings.

The reference never replaces its text tower: `utils/extract_text_embeddings.py:98-141` (`extract_text_embeddings(model,
categories, templates)`, `prompt_engineering(categories, model_name=<str | model>)`) and `networks/zutis.py:35-38` call
`model.encode_text(clip.tokenize(texts))` on a third-party `clip` model.  `HipClipText` is that `model`: hand it to the
reference's unchanged functions (`prompt_engineering(categories, model_name=HipClipText(sd))` takes the `else: model =
model_name` branch, :128-129) and `encode_text` runs on zutis_amd.engine.ClipTextEncoder (fp16 MFMA operands, fp32
accumulate / residual; the reference's GPU path runs CLIP in fp16 end to end).

`prompt_engineering_batched` is the fast path: every (category, template) pair in one batch and the ensembling
(normalise, mean over templates, normalise) on the device — the reference loops category by category.
The tokenizer stays the caller's (`clip.tokenize`, a BPE over a vocabulary file this repository does not ship).
"""
import pickle as pkl
from typing import Callable, Dict, List, Optional, Sequence

import torch

from zutis_amd.engine import ClipTextEncoder


class HipClipText:
    """Duck-types the text half of `clip.model.CLIP` (clip_arch.py:534-547): `.encode_text(tokens) -> [n, embed]`."""

    def __init__(self, state_dict: dict, device: torch.device = torch.device("cuda:0"), prefix: str = "", precision: str = "exact"):
        keys = ("token_embedding.weight", "positional_embedding", "ln_final.weight", "ln_final.bias", "text_projection")
        text = {k: v.detach().float().to(device) for k, v in state_dict.items()
                if k.startswith(prefix + "transformer.resblocks.") or k in tuple(prefix + s for s in keys)}
        self.engine = ClipTextEncoder(text, prefix=prefix, precision=precision)
        self.context_length = self.engine.ctx
        self.device = device

    @torch.no_grad()
    def encode_text(self, text: torch.Tensor) -> torch.Tensor:
        return self.engine.encode_text(text)

    def eval(self):
        return self

    def requires_grad_(self, flag: bool = False):
        return self


@torch.no_grad()
def prompt_engineering_batched(model: HipClipText, tokenize: Callable[[List[str]], torch.Tensor], categories: Sequence[str],
                               templates: Sequence[str] = ("{}",), fp: Optional[str] = None) -> Dict[str, torch.Tensor]:
    """Same result dictionary (category -> float32 [embed] on the model's device) and pickle as
    extract_text_embeddings / prompt_engineering (utils/extract_text_embeddings.py:98-141)."""
    C, T = len(categories), len(templates)
    texts = [t.format(c) for c in categories for t in templates]
    tokens = tokenize(texts).view(C, T, -1)
    emb = model.engine.prompt_ensemble(tokens)
    out = {c: emb[i] for i, c in enumerate(categories)}
    if fp is not None:
        pkl.dump(out, open(fp, "wb"))
    return out
