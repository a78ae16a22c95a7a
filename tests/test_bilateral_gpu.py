"""-m gpu: float64 bilateral-solver kernels against the reference's golden outputs and the NumPy/SciPy oracle."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_yuv_binning_all_colours_bit_exact(dev):
    """Integer output: lattice coordinates of ALL 2^24 RGB colours equal np.tensordot's (bilateral_solver.py:21-22,46-49),
    incl. the bin-edge cases (gray 16 -> Y = 15.999999999999998 -> bin 0)."""
    from zutis_amd import ops
    from oracle import bilateral_ref as B
    r, g, b = np.meshgrid(np.arange(256, dtype=np.uint8), np.arange(256, dtype=np.uint8), np.arange(256, dtype=np.uint8), indexing="ij")
    rgb = np.stack([r, g, b], -1).reshape(4096, 4096, 3)
    got = ops.bgrid_coords(torch.from_numpy(rgb).to(dev)).cpu().numpy()
    ref = B.grid_coords(rgb)
    assert np.array_equal(got, ref)


def test_denormalize_bit_exact(dev, golden_dir):
    from zutis_amd import ops, detgen
    g = np.load(f"{golden_dir}/bilateral.npz")
    x = detgen.det_normal("denorm", (3, 40, 56))
    x[0, 0, :16] = ((np.arange(16) * 16 / 255.0 - 0.485) / 0.229).astype(np.float32)
    got = ops.denormalize_u8(torch.from_numpy(x).to(dev)).cpu().numpy()
    assert np.array_equal(got, g["denorm_u8"])                      # the reference's convert_tensor_to_pil_image output


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_solver_vs_reference_golden(dev, golden_dir, tag):
    from zutis_amd import ops, detgen
    g = np.load(f"{golden_dir}/bilateral.npz")
    h, w, seed = (int(v) for v in g[f"{tag}_hw"])
    rgb = detgen.selfmask_like_rgb(h, w, seed=seed)
    soft, stats, n, m = ops.bilateral_solve(torch.from_numpy(rgb).to(dev), torch.from_numpy(g[f"{tag}_target"]).to(dev), debug=True)
    V, its = (int(v) for v in stats.cpu().numpy())
    assert V == int(g[f"{tag}_nvertices"])                          # integer: exact
    assert its == int(g[f"{tag}_cg_iters"])
    assert np.abs(n.cpu().numpy()[:V] - g[f"{tag}_n"]).max() < 1e-13      # bistochastisation (sqrt/div are IEEE)
    assert np.abs(m.cpu().numpy()[:V] - g[f"{tag}_m"]).max() < 1e-11
    err = np.abs(soft.cpu().numpy() - g[f"{tag}_soft"]).max()
    assert err < 1e-9, err                                          # output_solver float64, tol 1e-9
    assert np.array_equal(soft.cpu().numpy() > 0.5, g[f"{tag}_soft"] > 0.5)


def test_dropin_bilateral_solver_output(dev, golden_dir):
    from zutis_amd import detgen
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "zutis_amd", "dropin")
    if d not in sys.path:
        sys.path.insert(0, d)
    from PIL import Image
    from utils.bilateral_solver import bilateral_solver_output
    g = np.load(f"{golden_dir}/bilateral.npz")
    h, w, seed = (int(v) for v in g["c_hw"])
    soft, binary = bilateral_solver_output(Image.fromarray(detgen.selfmask_like_rgb(h, w, seed=seed)), g["c_target"])
    assert soft.dtype == np.float64 and binary.dtype == np.bool_
    assert np.abs(soft - g["c_soft"]).max() < 1e-9
    assert np.array_equal(binary, g["c_binary"])


def test_solver_empty_target_is_zeros_like_the_reference(dev, golden_dir):
    """An all-zero target (no foreground from the pseudo-labeller): b = 0, the reference's scipy cg returns zeros without iterating and its
    post-processing falls back to the all-True mask (golden `z_*`, generated from the real reference).  The device loop must not form
    0 / 0: soft output exactly zero, 0 iterations, alone and as one image of a batch whose other images solve normally (bitwise)."""
    from zutis_amd import ops, detgen
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "zutis_amd", "dropin")
    if d not in sys.path:
        sys.path.insert(0, d)
    from PIL import Image
    from utils.bilateral_solver import bilateral_solver_output
    g = np.load(f"{golden_dir}/bilateral.npz")
    rgb = detgen.selfmask_like_rgb(96, 128, seed=3)
    zero = np.zeros((96, 128), np.uint8)
    soft, stats = ops.bilateral_solve(torch.from_numpy(rgb).to(dev), torch.from_numpy(zero).to(dev))
    assert int(stats.cpu().numpy().reshape(-1)[1]) == 0 and np.array_equal(soft.cpu().numpy(), g["z_soft"])
    s2, b2 = bilateral_solver_output(Image.fromarray(rgb), zero)
    assert np.array_equal(s2, g["z_soft"]) and np.array_equal(b2, g["z_binary"]) and b2.all()
    # in a batch: image 0 empty, image 1 the golden "a" case (same picture)
    tg = np.stack([zero, g["a_target"]])
    sb, st = ops.bilateral_solve(torch.from_numpy(np.stack([rgb, rgb])).to(dev), torch.from_numpy(tg).to(dev))
    sb = sb.cpu().numpy()
    assert not sb[0].any() and np.abs(sb[1] - g["a_soft"]).max() < 1e-9 and [int(v) for v in st.cpu().numpy()[:, 1]] == [0, 25]


def test_solver_cg_tol_zero_iterates_to_maxiter_like_scipy(dev):
    """cg_tol = 0 makes atol = rtol * ||b|| zero although b is not: scipy (and the oracle's restated loop) then run all `cg_maxiter`
    iterations — only ||b|| == 0 itself returns early (round-5 advisor: the device loop used atol == 0 as that flag and stopped at x0)."""
    from zutis_amd import ops, detgen
    from oracle import bilateral_ref as B
    rgb = detgen.selfmask_like_rgb(96, 128, seed=3)
    yy, xx = np.mgrid[:96, :128]
    target = (((yy - 50) ** 2 + (xx - 60) ** 2) < 30 ** 2).astype(np.uint8)
    grid = B.Grid(rgb)
    ref, its, _, _ = B.solve(grid, target.reshape(-1).astype(np.float64), np.full(target.size, 0.999), cg_tol=0.0, cg_maxiter=9)
    soft, stats = ops.bilateral_solve(torch.from_numpy(rgb).to(dev), torch.from_numpy(target).to(dev), cg_tol=0.0, cg_maxiter=9)
    assert its == 9 and int(stats.cpu().numpy().reshape(-1)[1]) == 9
    assert np.abs(soft.cpu().numpy().reshape(-1) - ref).max() < 1e-9
    # and the empty target still returns at iteration 0 with cg_tol = 0
    soft0, st0 = ops.bilateral_solve(torch.from_numpy(rgb).to(dev), torch.from_numpy(np.zeros_like(target)).to(dev), cg_tol=0.0, cg_maxiter=9)
    assert int(st0.cpu().numpy().reshape(-1)[1]) == 0 and not soft0.cpu().numpy().any()


def test_solver_vs_oracle_selfmask_size(dev):
    """Full SelfMask-style size (512x683, V ~ 10^4-10^5): round trip against the oracle + float target path."""
    from zutis_amd import ops, detgen
    from oracle import bilateral_ref as B
    h, w = 512, 683
    rgb = detgen.selfmask_like_rgb(h, w, seed=3)
    yy, xx = np.mgrid[:h, :w]
    target = (((yy - 250) ** 2 + (xx - 300) ** 2) < 150 ** 2).astype(np.uint8)
    ref_soft, _ = B.bilateral_solver_output(rgb, target)
    soft, stats = ops.bilateral_solve(torch.from_numpy(rgb).to(dev), torch.from_numpy(target).to(dev))
    grid = B.Grid(rgb)
    assert int(stats[0]) == grid.nvertices
    assert np.abs(soft.cpu().numpy() - ref_soft).max() < 1e-9
    soft64, _ = ops.bilateral_solve(torch.from_numpy(rgb).to(dev), torch.from_numpy(target.astype(np.float64)).to(dev))
    assert np.abs(soft64.cpu().numpy() - ref_soft).max() < 1e-9


def test_resize_nearest_bit_exact(dev):
    from zutis_amd import ops, detgen
    import torch.nn.functional as F
    m = (detgen.det_normal("nearest", (61, 83)) > 0).astype(np.uint8)
    for (H, W) in [(427, 640), (61, 83), (30, 200), (512, 683)]:
        got = ops.resize_nearest_u8(torch.from_numpy(m).to(dev), H, W).cpu().numpy()
        ref = F.interpolate(torch.from_numpy(m)[None, None], size=(H, W), mode="nearest")[0, 0].numpy()
        assert np.array_equal(got, ref)


def test_pseudo_mask_driver_end_to_end(dev, tmp_path):
    """SelfMask -> device bilateral solver -> nearest resize -> RLE JSON, against the oracle chain on the same inputs, at the engine's
    default precision (`exact`: fp32-class contractions).  The masks may differ only where a soft value sits within rounding of a 0.5
    threshold: every differing pixel lies within one (source) pixel of the oracle's own mask contour, and there are few of them."""
    from zutis_amd import detgen, pseudo_masks, rle
    from zutis_amd.engine import SelfMaskEngine
    from oracle import zutis_ref as O
    from oracle.parity import contour_mismatches, pseudo_mask_chain
    import json
    eng = SelfMaskEngine({k: torch.from_numpy(v).to(dev) for k, v in detgen.selfmask_state_dict().items()})
    assert eng.precision == "exact"
    x = torch.from_numpy(detgen.images(1, 72, 100, seed=11))[0]
    p = str(tmp_path / "m" / "a.json")
    pseudo_masks.generate_pseudo_masks(eng, [x.to(dev)], [(144, 200)], [p])
    got = rle.decode(json.load(open(p))).astype(bool)
    assert got.shape == (144, 200)
    ref = pseudo_mask_chain(O.to_torch_params(detgen.selfmask_state_dict()), x[None], (144, 200))
    n_diff, n_bad = contour_mismatches(got, ref["mask"], native=ref["mask_native"])
    print(f"pseudo-mask chain 72x100 -> 144x200 [exact]: {n_diff} of {got.size} pixels differ, {n_bad} off the oracle's contour")
    assert n_bad == 0 and n_diff <= 64, (n_diff, n_bad)          # <= 16 source pixels (each is a 2 x 2 block after the resize)


def test_pseudo_mask_chain_at_the_working_shape_512x683(dev, tmp_path):
    """The whole pseudo-label chain at the shape the pipeline runs it (datasets/index_dataset.py:177-226: 512 on the short side, T = 5505
    tokens; natural colour statistics so the solver's lattice is photograph-sized): SelfMask -> solver -> > 0.5 -> nearest resize to the
    file's 480x640 -> RLE JSON -> decode, against the oracle chain; and the batched driver writes the same file."""
    from zutis_amd import detgen, pseudo_masks, rle
    from zutis_amd.engine import SelfMaskEngine
    from oracle import zutis_ref as O
    from oracle.parity import contour_mismatches, pseudo_mask_chain
    import json
    import bench
    H, W, out = 512, 683, (480, 640)
    sd = detgen.selfmask_state_dict()
    eng = SelfMaskEngine({k: torch.from_numpy(v).to(dev) for k, v in sd.items()})
    x = bench.natural_images(2, H, W, dev, seed=7)
    pa, pb = str(tmp_path / "one" / "a.json"), [str(tmp_path / "batch" / f"{i}.json") for i in range(2)]
    pseudo_masks.generate_pseudo_masks(eng, [x[0]], [out], [pa], n_streams=1)
    pseudo_masks.generate_pseudo_masks_batched(eng, [x[0], x[1]], [out, out], pb, batch_size=2)
    assert open(pa).read() == open(pb[0]).read()
    got = rle.decode(json.load(open(pa))).astype(bool)
    ref = pseudo_mask_chain(O.to_torch_params(sd), x[:1].cpu(), out)
    assert ref["objectness_margin"] > 1e-3                       # the selected query is unambiguous for this image
    n_diff, n_bad = contour_mismatches(got, ref["mask"], native=ref["mask_native"])
    print(f"pseudo-mask chain 512x683 -> 480x640 [exact]: {n_diff} of {got.size} pixels differ, {n_bad} off the oracle's contour; "
          f"mask covers {got.mean():.3f}")
    assert 0.02 < ref["mask"].mean() < 0.98 and n_bad == 0 and n_diff <= 200, (n_diff, n_bad)


def test_pseudo_mask_pipeline_matches_sequential(dev, tmp_path):
    """The multi-stream driver (forked engines, device-side query selection, pinned async D2H) writes byte-identical RLE JSON
    to the one-stream loop, for images of different sizes."""
    from zutis_amd import detgen, pseudo_masks
    from zutis_amd.engine import SelfMaskEngine
    eng = SelfMaskEngine({k: torch.from_numpy(v).to(dev) for k, v in detgen.selfmask_state_dict().items()})
    sizes_in = [(72, 100), (64, 64), (72, 100), (80, 56), (64, 64)]
    imgs = [torch.from_numpy(detgen.images(1, h, w, seed=20 + i))[0].to(dev) for i, (h, w) in enumerate(sizes_in)]
    sizes_out = [(2 * h, 2 * w + 1) for h, w in sizes_in]
    pa = [str(tmp_path / "seq" / f"{i}.json") for i in range(len(imgs))]
    pb = [str(tmp_path / "pipe" / f"{i}.json") for i in range(len(imgs))]
    pseudo_masks.generate_pseudo_masks(eng, imgs, sizes_out, pa, n_streams=1)
    pseudo_masks.generate_pseudo_masks(eng, imgs, sizes_out, pb, n_streams=3)
    for a, b in zip(pa, pb):
        assert open(a).read() == open(b).read()


def test_select_upsample_mask_kernel(dev):
    """SelfMask inference tail on the device == torch: argmax(objectness) -> x4 bilinear of that plane -> crop -> > 0.5."""
    from zutis_amd import ops
    import torch.nn.functional as F
    B, Q, h, w, H, W = 3, 20, 18, 26, 70, 101
    g = torch.Generator().manual_seed(5)
    obj = torch.randn((B, Q), generator=g)
    obj[1, 7] = obj[1, 3] = obj[1].max() + 1.0                       # tie: the first maximum wins
    masks = torch.rand((B, Q, h, w), generator=g)
    out = torch.empty((B, H, W), dtype=torch.uint8, device=dev)
    idx = torch.empty((B,), dtype=torch.int64, device=dev)
    ops.select_upsample_mask(obj.to(dev), masks.to(dev), out, idx, B, Q, h, w, H, W, 0.25, 0.25, 0.5)
    sel = obj.argmax(dim=1)
    assert torch.equal(idx.cpu(), sel) and int(sel[1]) == 3
    up = F.interpolate(masks[torch.arange(B), sel][:, None], scale_factor=4, mode="bilinear", align_corners=False)[:, 0, :H, :W]
    ref = (up > 0.5).to(torch.uint8)
    assert (out.cpu() != ref).float().mean().item() < 2e-4          # fp32 rounding at the 0.5 threshold only


def test_solver_batched_equals_per_image_bitwise(dev):
    """zh_bilateral_solve_batch (blockIdx.y = image, per-image workspace slices) == B separate solves, bit for bit; images
    with different vertex counts share the launches.  Also: two runs of the same batch are bitwise identical (the splat is
    integer atomics + ordered float sums — nothing depends on atomic order)."""
    from zutis_amd import ops, detgen
    h, w = 120, 168
    rgbs = np.stack([detgen.selfmask_like_rgb(h, w, seed=s) for s in (3, 5, 9, 11)])
    rgbs[3] = (detgen.det_normal("noise_rgb", (h, w, 3)) * 60 + 128).clip(0, 255).astype(np.uint8)      # many more vertices
    yy, xx = np.mgrid[:h, :w]
    tg = np.stack([(((yy - 60 - 5 * i) ** 2 + (xx - 80) ** 2) < (30 + 4 * i) ** 2).astype(np.uint8) for i in range(4)])
    R, T = torch.from_numpy(rgbs).to(dev), torch.from_numpy(tg).to(dev)
    soft_b, stats_b = ops.bilateral_solve(R, T)
    soft_b2, _ = ops.bilateral_solve(R, T)
    assert torch.equal(soft_b, soft_b2)
    vs = []
    for i in range(4):
        s1, st1 = ops.bilateral_solve(R[i].contiguous(), T[i].contiguous())
        assert torch.equal(s1, soft_b[i]) and torch.equal(st1, stats_b[i])
        vs.append(int(st1[0]))
    assert vs[3] > 2 * max(vs[:3])


def test_solver_nonbinary_and_float_targets_ordered_splat(dev):
    """Targets that are not {0,1}: the splat S.(t*w) is summed per vertex in ascending pixel order (SciPy's CSR row order),
    for uint8 0..255 and for float64 targets, against the NumPy/SciPy oracle."""
    from zutis_amd import ops, detgen
    from oracle import bilateral_ref as B
    h, w = 96, 128
    rgb = detgen.selfmask_like_rgb(h, w, seed=3)
    t8 = (np.abs(detgen.det_normal("t8", (h, w))) * 90).clip(0, 255).astype(np.uint8)
    tf = np.abs(detgen.det_normal("tf", (h, w))).astype(np.float64) * 0.7
    for t in (t8, tf):
        ref_soft, _ = B.bilateral_solver_output(rgb, t)
        soft, _ = ops.bilateral_solve(torch.from_numpy(rgb).to(dev), torch.from_numpy(t).to(dev))
        assert np.abs(soft.cpu().numpy() - ref_soft).max() < 1e-9 * max(1.0, float(np.abs(ref_soft).max()))


def test_pseudo_mask_batched_driver_matches_batch1(dev, tmp_path):
    """Batched SelfMask + batched solver (generate_pseudo_masks_batched) write the same RLE JSON files as the batch-1 loop."""
    from zutis_amd import detgen, pseudo_masks
    from zutis_amd.engine import SelfMaskEngine
    eng = SelfMaskEngine({k: torch.from_numpy(v).to(dev) for k, v in detgen.selfmask_state_dict().items()})
    sizes_in = [(72, 100), (72, 100), (72, 100), (64, 64), (64, 64), (72, 100)]
    imgs = [torch.from_numpy(detgen.images(1, h, w, seed=40 + i))[0].to(dev) for i, (h, w) in enumerate(sizes_in)]
    sizes_out = [(2 * h + 1, 2 * w) for h, w in sizes_in]
    pa = [str(tmp_path / "seq" / f"{i}.json") for i in range(len(imgs))]
    pb = [str(tmp_path / "bat" / f"{i}.json") for i in range(len(imgs))]
    pseudo_masks.generate_pseudo_masks(eng, imgs, sizes_out, pa, n_streams=1)
    pseudo_masks.generate_pseudo_masks_batched(eng, imgs, sizes_out, pb, batch_size=4)
    for a, b in zip(pa, pb):
        assert open(a).read() == open(b).read()


def test_dataset_method_adapter_writes_the_same_files(dev, tmp_path):
    """`generate_pseudo_masks(self, p_images, dir_dataset, n_workers, bilateral_solver)` with the dataset method's own signature
    (datasets/index_dataset.py:177-226) over the batched device path: same paths, byte-identical JSON to the batch-1 driver."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "zutis_amd", "dropin"))
    from networks.selfmask.selfmask import SelfMask
    from zutis_amd import detgen, pseudo_masks
    net = SelfMask()
    net.load_state_dict({k: torch.from_numpy(v) for k, v in detgen.selfmask_state_dict().items()}, strict=True)
    sizes_in = [(72, 100), (72, 100), (64, 64), (72, 100), (72, 100), (72, 100)]
    root = tmp_path / "pass" / "images"
    root.mkdir(parents=True)
    p_images = []
    for i, (h, w) in enumerate(sizes_in):
        p = str(root / f"im{i}.npy")
        np.save(p, detgen.images(1, h, w, seed=60 + i)[0])
        p_images.append(p)

    class MaskDataset(torch.utils.data.Dataset):                  # stands in for the dataset module's own loader (PIL / torchvision absent here)
        def __init__(self, p_images):
            self.p_images = p_images

        def __len__(self):
            return len(self.p_images)

        def __getitem__(self, i):
            return {"image": torch.from_numpy(np.load(self.p_images[i])), "p_image": self.p_images[i]}

    class FakeIndexDataset:
        device = dev

        def _convert_p_image_to_p_pseudo_mask(self, p_image):       # the "/pass/" branch of index_dataset.py:250-255
            d = "/".join(p_image.split("/")[:-1]).replace("/images", "") + "/pseudo_masks_selfmask"
            return f"{d}/{p_image.split('/')[-1].replace('npy', 'json')}"
    FakeIndexDataset.generate_pseudo_masks = pseudo_masks.dataset_generate_pseudo_masks
    ds = FakeIndexDataset()
    size_of = lambda p: tuple(2 * s + 1 for s in np.load(p).shape[1:])
    ds.generate_pseudo_masks(p_images, str(tmp_path / "pass"), 0, True, batch_size=4, network=net, mask_dataset_cls=MaskDataset, image_size_fn=size_of)
    eng = net._get_engine()
    ref_paths = [str(tmp_path / "ref" / f"{i}.json") for i in range(len(p_images))]
    pseudo_masks.generate_pseudo_masks(eng, [torch.from_numpy(np.load(p)).to(dev) for p in p_images], [size_of(p) for p in p_images], ref_paths, n_streams=1)
    for p, r in zip(p_images, ref_paths):
        assert open(ds._convert_p_image_to_p_pseudo_mask(p)).read() == open(r).read()
