"""CPU: the C-ABI library loads and exports every symbol include/zutis_hip.h declares (no compute calls)."""
import ctypes
import os

import pytest

from zutis_amd import _lib, build


@pytest.fixture(scope="module")
def lib():
    build.build(verbose=False)        # hipcc cross-compiles gfx950 without a GPU
    return _lib.load()


def test_library_exports_every_declared_symbol(lib):
    syms = _lib.declared_symbols()
    assert len(syms) >= 18
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, missing


def test_every_declared_symbol_has_a_binding():
    assert not [s for s in _lib.declared_symbols() if s not in _lib._SIGS]


def test_version_arch_and_error_text(lib):
    assert lib.zh_version() == _lib.header_abi_version() >= 210        # a stale build is refused by _lib.load()
    assert lib.zh_arch() == b"gfx950"
    assert isinstance(lib.zh_last_error(), bytes)


def test_argument_validation_without_gpu(lib):
    # shape/alignment checks happen before any launch, so they are testable on CPU
    rc = lib.zh_gemm_f16(None, 0, 0, None, 0, 0, None, 0, 0, 0, None, None, 0, 0, 0, None, None, 0, 0, 0, 0, 0, 8, 8, 64, 1, None)
    assert rc == -1 and b"null" in lib.zh_last_error()
    rc = lib.zh_attention_f16(16, 8, 8, 16, 8, 8, 16, 8, 8, 16, 8, 8, 1, 1, 4, 4, 80, 1.0, 0, 0, 0, 0, None)
    assert rc == -1 and b"head_dim" in lib.zh_last_error()
    rc = lib.zh_attention_f16(16, 8, 8, 16, 8, 8, 16, 8, 8, 16, 8, 8, 1, 1, 4, 4, 64, 1.0, 64, 0, 64, 0, None)
    assert rc == -1 and b"all three or none" in lib.zh_last_error()
    # the reference-equivalent GEMM refuses plain fp16 operands (no silent precision downgrade)
    rc = lib.zh_gemm_f16x3(16, 64, 0, 0, 16, 64, 0, 0, 16, 8, 0, 0, 0, 1.0, None, None, 0, 0, 0, None, None, 0, 0, 0, 0, 0, 8, 8, 64, 1, 0, None)
    assert rc == -1 and b"A must be a split pair" in lib.zh_last_error()
    assert lib.zh_global_ln_l2_workspace_size(2, 1764, 512) == 2 * ((1764 * 512 + 4095) // 4096) * 16


def test_stale_library_is_refused(monkeypatch):
    """A library built for another ABI version must not be bound: ctypes would hand the new argument lists to old entry points."""
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "header_abi_version", lambda: 99999)
    with pytest.raises(_lib.ZutisHipError, match="built for ABI"):
        _lib.load()


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.ZutisHipError, match="REQUIRED"):
        _lib.load()


def test_ops_refuse_cpu_tensors(lib):
    import torch
    from zutis_amd import ops
    with pytest.raises(_lib.ZutisHipError):
        ops.gemm(torch.zeros(8, 64, dtype=torch.float16), torch.zeros(8, 64, dtype=torch.float16), torch.zeros(8, 8))


def test_launch_plan_dispatcher_matches_header_and_bindings(tmp_path):
    """The native launch-plan dispatcher is generated from include/zutis_hip.h: every plannable entry point (stream last,
    device-pointer / scalar arguments) must exist in the ctypes table with the same arity, the generated C must call it with
    one argument word per parameter, and the library must report the same op-id -> name mapping (no GPU needed)."""
    import struct
    from zutis_amd import _lib, plan
    ops_ = plan.parse_header()
    names = [n for n, _ in ops_]
    assert "zh_gemm_f16" in names and "zh_attention_f16" in names and "zh_bilateral_solve" in names
    assert not any(n.startswith("zh_plan_") for n in names) and "zh_denormalize_u8" not in names
    for name, sig in ops_:
        res, args = _lib._SIGS[name]
        assert len(args) == len(sig) <= plan.MAX_ARGS + 1, name
        assert sig[-1][0] == "zh_stream_t"
    out = tmp_path / "gen.inc"
    assert plan.generate_dispatch(str(out)) == names
    txt = out.read_text()
    for i, (name, sig) in enumerate(ops_):
        line = [ln for ln in txt.splitlines() if ln.strip().startswith(f"case {i}:")][0]
        assert f"return {name}(" in line and line.count("c.a[") == len(sig) - 1
    L = _lib.load(raw=True)
    for i, name in enumerate(names):
        assert L.zh_plan_op_name(i).decode() == name
    assert L.zh_plan_op_name(len(names)) is None
    # argument words: floats by bit pattern, pointers / ints zero-extended, None -> 0
    assert plan._word("float", 1.5) == struct.unpack("<I", struct.pack("<f", 1.5))[0]
    assert plan._word("double", -2.0) == struct.unpack("<Q", struct.pack("<d", -2.0))[0]
    assert plan._word("const float*", None) == 0 and plan._word("int", -1) == 0xFFFFFFFFFFFFFFFF


def test_precision_site_sets():
    """Host logic (no kernels): named precisions and the closure of custom site sets."""
    from zutis_amd import engine as E
    assert E.resolve_precision("f16") == frozenset()
    assert E.resolve_precision("exact") == frozenset(E.ALL_SITES)
    assert E.resolve_precision("fast") == frozenset(E.HEAD_SITES)
    assert E.resolve_precision(["attn"]) == {"attn", "qkv"}
    assert E.resolve_precision(["mask"]) == {"mask", "ffn1"}
    with pytest.raises(E.ZutisHipError):
        E.resolve_precision("bf16")
    with pytest.raises(E.ZutisHipError):
        E.resolve_precision(["nope"])


def test_split_weight_packing_host_logic():
    """ops.split_weight (pack-time, plain torch — runs on CPU): W * 2^s with max|W| in [2^13, 2^14), hi + lo reproduces the
    scaled weight to ~2^-22 relative, lo is a NORMAL fp16 number wherever it is non-zero for typical weights, out_scale = 2^-s."""
    import math
    import torch
    from zutis_amd import ops
    g = torch.Generator().manual_seed(0)
    for scale in (0.03, 1.0, 1e-4, 300.0):
        w = torch.randn((64, 128), generator=g) * scale
        a = ops.split_weight(w)
        assert a.t.shape == (2, 64, 128) and a.plane == 64 * 128
        s = -math.log2(a.out_scale)
        assert s == int(s) and 2 ** 13 <= float(w.abs().max()) * 2 ** s < 2 ** 14
        rec = (a.t[0].double() + a.t[1].double()) * a.out_scale
        assert float(((rec - w.double()).abs() / w.double().abs().clamp_min(float(w.abs().max()) * 2 ** -10)).max()) < 2 ** -21
        lo = a.t[1].float().abs()
        assert float((lo[lo > 0] < 6.1e-5).float().mean()) < 0.02          # (almost) no subnormal lo halves
    z = ops.split_weight(torch.zeros((4, 64)))
    assert z.out_scale == 1.0 and float(z.t.abs().max()) == 0.0


def test_split_weight_one_plane_for_fp16_valued_weights_host_logic():
    """The f16x2 operand form (zh_gemm_f16x3 with planeW = 0): a weight whose values are fp16 numbers — what the reference's
    convert_weights leaves in the CLIP towers, clip_arch.py:566-587 — has an all-zero lo plane at ANY power-of-two scale, so it is
    packed as its hi plane alone, exactly; a single non-representable value brings the second plane back."""
    import torch
    from zutis_amd import ops
    g = torch.Generator().manual_seed(1)
    for scale in (0.03, 1.0, 2e-4, 300.0):
        w = (torch.randn((48, 64), generator=g) * scale).to(torch.float16).to(torch.float32)      # includes fp16 subnormals at 2e-4
        a = ops.split_weight(w)
        assert a.x2 and a.plane == 0 and a.t.shape == (1, 48, 64)
        assert torch.equal(a.hi.to(torch.float64) * a.out_scale, w.to(torch.float64))
        assert a.view(a.hi[:16]).x2                                                              # row slices keep the marker
        b = ops.split_weight(w, allow_x2=False)
        assert (not b.x2) and b.t.shape == (2, 48, 64) and not bool(b.t[1].any()) and torch.equal(b.t[0], a.hi)
        w2 = w.clone()
        w2[3, 5] = w2[3, 5] * (1.0 + 2.0 ** -14) + (2.0 ** -30 if w2[3, 5] == 0 else 0.0)          # not an fp16 number any more
        c = ops.split_weight(w2)
        assert (not c.x2) and c.plane != 0 and int((c.t[1] != 0).sum()) == 1
    ops.ALLOW_X2 = False
    try:
        assert not ops.split_weight(torch.ones((4, 64))).x2
    finally:
        ops.ALLOW_X2 = True
