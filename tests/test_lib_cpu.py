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


def test_product_code_never_touches_the_oracle():
    """oracle/ is test infrastructure: the package and the tools must not import it; bench.py may only inside
    cpu_baseline(), __graft_entry__.py only inside smoke()."""
    import ast
    import glob
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def oracle_imports(path):
        tree = ast.parse(open(path).read())
        hits = []
        for fn in ast.walk(tree):
            if isinstance(fn, (ast.FunctionDef, ast.Module)):
                for node in fn.body if isinstance(fn, ast.Module) else ast.walk(fn):
                    if isinstance(node, ast.ImportFrom) and (node.module or "").split(".")[0] == "oracle":
                        hits.append(getattr(fn, "name", "<module>"))
                    if isinstance(node, ast.Import) and any(a.name.split(".")[0] == "oracle" for a in node.names):
                        hits.append(getattr(fn, "name", "<module>"))
        return set(hits)

    for path in glob.glob(os.path.join(root, "feature_level_style_transfer_for_tsc_amd", "**", "*.py"), recursive=True) + \
            glob.glob(os.path.join(root, "tools", "*.py")):
        assert not oracle_imports(path), f"{path} imports the oracle"
    assert oracle_imports(os.path.join(root, "bench.py")) == {"cpu_baseline"}
    assert oracle_imports(os.path.join(root, "__graft_entry__.py")) == {"smoke"}
