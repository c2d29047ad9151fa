"""The C-ABI library builds, loads on a GPU-less host and exports every symbol include/fst_hip.h declares;
ops refuse to run anywhere but on the GPU (no CPU fallback)."""
import os
import re

import pytest
import torch

from feature_level_style_transfer_for_tsc_amd import _lib, ops

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "fst_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(fst_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert sorted(_lib.EXPORTED_SYMBOLS) == _declared_symbols()


@pytest.mark.skipif(not os.path.exists(_lib.LIB_PATH), reason="libfst_hip.so not built (run __graft_entry__.build())")
def test_library_loads_and_exports_every_symbol():
    lib = _lib.load()
    assert lib.fst_version() == _lib.ABI_VERSION
    for name in _declared_symbols():
        assert hasattr(lib, name), name


def test_missing_library_fails_loudly(tmp_path):
    with pytest.raises(_lib.FstLibraryError):
        _lib.load(str(tmp_path / "libfst_hip.so"))


def test_ops_refuse_cpu_tensors():
    spec = ops.ConvSpec(4, 4)
    x, w = torch.randn(1, 4, 8), torch.randn(4, 4, 1)
    with pytest.raises((_lib.FstLibraryError, RuntimeError)):
        ops.conv1d(spec, x, w, None)
