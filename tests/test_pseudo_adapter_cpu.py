"""Host logic of the dataset-signature adapter (zutis_amd.pseudo_masks.dataset_generate_pseudo_masks): grouping by shape, the
paths it asks `self` for, and that anything but the MI355X SelfMask is refused (no torch / CPU fallback).  No GPU."""
import numpy as np
import pytest
import torch

from zutis_amd import detgen, pseudo_masks
from zutis_amd.engine import SelfMaskEngine


class _DS(torch.utils.data.Dataset):
    def __init__(self, p_images):
        self.p_images = p_images

    def __len__(self):
        return len(self.p_images)

    def __getitem__(self, i):
        h, w = {"a": (8, 12), "b": (8, 12), "c": (6, 6), "d": (8, 12), "e": (8, 12), "f": (8, 12), "g": (8, 12)}[self.p_images[i]]
        return {"image": torch.full((3, h, w), float(i)), "p_image": self.p_images[i]}


class _Owner:
    device = torch.device("cpu")

    def _convert_p_image_to_p_pseudo_mask(self, p_image):
        return f"/out/{p_image}.json"


def test_groups_consecutive_equal_shapes_up_to_batch_size(monkeypatch):
    eng = SelfMaskEngine({k: torch.from_numpy(v) for k, v in detgen.selfmask_state_dict().items()})
    eng.to = lambda d: eng
    eng.eval = lambda: None
    calls = []
    monkeypatch.setattr(pseudo_masks, "generate_pseudo_masks_batched",
                        lambda e, imgs, sizes, paths, bilateral_solver, batch_size: calls.append(([tuple(i.shape) for i in imgs], list(sizes), list(paths), bilateral_solver, batch_size)))
    pseudo_masks.dataset_generate_pseudo_masks(_Owner(), list("abcdefg"), "/data", 0, False, batch_size=3, network=eng, mask_dataset_cls=_DS,
                                               image_size_fn=lambda p: (ord(p), 7))
    groups = [c[2] for c in calls]
    assert groups == [["/out/a.json", "/out/b.json"], ["/out/c.json"], ["/out/d.json", "/out/e.json", "/out/f.json"], ["/out/g.json"]]
    assert calls[0][1] == [(ord("a"), 7), (ord("b"), 7)] and calls[0][3] is False and all(c[4] == 3 for c in calls)
    assert calls[1][0] == [(3, 6, 6)]


def test_refuses_a_network_that_is_not_the_hip_selfmask():
    with pytest.raises(TypeError, match="no torch / CPU fallback"):
        pseudo_masks.dataset_generate_pseudo_masks(_Owner(), ["a"], "/data", 0, True, network=torch.nn.Linear(2, 2), mask_dataset_cls=_DS)
