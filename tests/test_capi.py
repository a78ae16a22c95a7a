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
    assert lib.zh_version() >= 100
    assert lib.zh_arch() == b"gfx950"
    assert isinstance(lib.zh_last_error(), bytes)


def test_argument_validation_without_gpu(lib):
    # shape/alignment checks happen before any launch, so they are testable on CPU
    rc = lib.zh_gemm_f16(None, 0, 0, None, 0, 0, None, 0, 0, 0, None, None, 0, 0, 0, 0, 8, 8, 64, 1, None)
    assert rc == -1 and b"null" in lib.zh_last_error()
    rc = lib.zh_attention_f16(16, 8, 8, 16, 8, 8, 16, 8, 8, 16, 8, 8, 1, 1, 4, 4, 80, 1.0, None)
    assert rc == -1 and b"head_dim" in lib.zh_last_error()
    assert lib.zh_global_ln_l2_workspace_size(2, 1764, 512) == 2 * ((1764 * 512 + 4095) // 4096) * 16


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
