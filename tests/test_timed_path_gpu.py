"""-m gpu: what bench.py TIMES is what is checked here (round-4 review: the replayed launch plans' outputs at the headline's shape were
never compared with anything).  Lanes are built through bench.build_lanes / bench.make_launch / StepPipeline — the bench's own code —
at the headline's shape: ViT-B/16 @336, B = 32 (M = 14144 token rows: the two-slot 256 x 256 / 192 x 256 split-pair tiles, the tail
peel, 3 interleaved lanes), then compared bitwise with an eager step and against the fp32 oracle on two images of the batch."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

TOLS = {"fast": 2.5e-4, "exact": 2e-5}       # low-res class logits against the fp32 oracle (north-star: 1e-3)


def _oracle_rows(cfg, x_rows, text, S):
    from zutis_amd import detgen
    from oracle import zutis_ref as O
    from oracle import resample as R
    Pc = O.to_torch_params(detgen.zutis_state_dict(cfg))
    with torch.no_grad():
        o = O.zutis_forward(Pc, x_rows.cpu(), cfg.patch, cfg.dec_heads)
        lo = O.semantic_logits_lowres(o["patch_tokens"], text.cpu()).numpy()
    return lo, R.bilinear_argmax_nchw(lo, S, S)


@pytest.mark.parametrize("precision", ["exact", "fast"])
def test_headline_step_b32_three_lanes_plan_replay(dev, precision):
    import bench
    from zutis_amd import detgen
    from zutis_amd import distributed as zd
    from zutis_amd.engine import ZutisEngine
    from oracle.parity import unexplained_label_mismatches
    cfg = detgen.VIT_B16
    B, S, n, n_lanes = 32, 336, 81, 3
    P = {k: torch.from_numpy(v).to(dev) for k, v in detgen.zutis_state_dict(cfg).items()}
    text = torch.from_numpy(detgen.text_embeddings(n, cfg.embed_dim)).to(dev)
    x = torch.randn((B, 3, S, S), generator=torch.Generator(device="cpu").manual_seed(1000)).to(dev)     # bench.py's rank-0 batch
    eng = ZutisEngine(P, cfg.patch, cfg.dec_heads, precision=precision)
    lanes = bench.build_lanes(eng, x, text, S, n, n_lanes)
    assert all(ln.state["plan"] is not None and ln.state["plan"].n > 100 for ln in lanes)
    pipe = zd.StepPipeline(lanes, bench.make_launch(n_lanes), gather=False)
    pipe.run(7)                                   # two full groups and a ragged one (7 = 3 + 3 + 1)
    ok, lo_t, lab_t = bench.check_timed_outputs(lanes)
    assert ok, "a lane's replayed launch plan and an eager step of the same engine disagree bitwise"
    assert lo_t.shape == (B, n, 42, 42) and lab_t.shape == (B, S, S)
    # the three lanes ran the same batch: identical outputs, bit for bit (independent buffers, shared packed weights)
    torch.cuda.synchronize()
    for ln in lanes[1:]:
        assert torch.equal(ln.state["labels"], lanes[0].state["labels"]) and torch.equal(ln.payload, lanes[0].payload)
    # ... and the oracle on two images OF THAT BATCH (first and last), compared with the timed outputs themselves
    rows = [0, B - 1]
    lo_ref, lab_ref = _oracle_rows(cfg, x[rows], text, S)
    lo = lo_t[rows].cpu().numpy()
    lab = lab_t[rows].cpu().numpy()
    err = float(np.abs(lo - lo_ref).max())
    assert err < TOLS[precision], err
    n_mis, n_bad, worst = unexplained_label_mismatches(lab, lab_ref, lo_ref, err, (S, S))
    print(f"headline step [{precision}]: timed outputs == eager bitwise; vs oracle logits {err:.2e}, {n_mis} labels differ, {n_bad} unexplained")
    assert n_bad == 0, (n_mis, n_bad, worst, err)


def test_c4_batch8_timed_path_vs_oracle(dev):
    """BASELINE config 4 at its own batch (8 x 518 x 518, 920 classes: 8200 token rows) through the bench's lanes, exact precision:
    replay == eager bitwise, logits / 920-class labels of two images against the oracle."""
    import bench
    from zutis_amd import detgen
    from zutis_amd import distributed as zd
    from zutis_amd.engine import ZutisEngine
    from oracle.parity import unexplained_label_mismatches
    cfg = detgen.VIT_B16
    B, S, n, n_lanes = 8, 518, 920, 3
    P = {k: torch.from_numpy(v).to(dev) for k, v in detgen.zutis_state_dict(cfg).items()}
    text = torch.from_numpy(detgen.text_embeddings(n, cfg.embed_dim)).to(dev)
    x = torch.randn((B, 3, S, S), generator=torch.Generator(device="cpu").manual_seed(4000)).to(dev)
    eng = ZutisEngine(P, cfg.patch, cfg.dec_heads, precision="exact")
    lanes = bench.build_lanes(eng, x, text, S, n, n_lanes)
    pipe = zd.StepPipeline(lanes, bench.make_launch(n_lanes), gather=False)
    pipe.run(4)
    ok, lo_t, lab_t = bench.check_timed_outputs(lanes)
    assert ok and lo_t.shape == (B, n, 64, 64)
    rows = [0, B - 1]
    lo_ref, lab_ref = _oracle_rows(cfg, x[rows], text, S)
    err = float(np.abs(lo_t[rows].cpu().numpy() - lo_ref).max())
    assert err < TOLS["exact"], err
    n_mis, n_bad, worst = unexplained_label_mismatches(lab_t[rows].cpu().numpy(), lab_ref, lo_ref, err, (S, S))
    print(f"c4 b=8 [exact]: logits {err:.2e}, {n_mis} of {2 * S * S} labels differ, {n_bad} unexplained")
    assert n_bad == 0, (n_mis, n_bad, worst, err)
