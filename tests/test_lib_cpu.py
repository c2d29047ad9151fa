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


def test_row_sum_handover_is_tied_to_the_tensor_it_was_computed_from():
    """ops._RowSums (BatchNorm backward → the bias gradient of the conv in front): the sums ride on the cotangent tensor object and
    are handed out only while that tensor is what they were computed from — same object, same version counter, same shape."""
    dx = torch.randn(4, 5, 32)
    rs = dx.sum(dim=2)
    assert ops._ROW_SUMS.take(dx) is None
    ops._ROW_SUMS.attach(dx, rs)
    assert ops._ROW_SUMS.take(dx) is rs
    assert ops._ROW_SUMS.take(dx.clone()) is None                 # another tensor
    assert ops._ROW_SUMS.take(dx.transpose(1, 2)) is None         # a view with other strides
    dx.add_(1.0)                                                  # rewritten in place (e.g. autograd accumulating into it)
    assert ops._ROW_SUMS.take(dx) is None
    ops._ROW_SUMS.attach(dx, torch.zeros(4, 6))                   # sums of another shape
    assert ops._ROW_SUMS.take(dx) is None


def test_dense_products_refuse_what_the_kernel_does_not_serve():
    """ops.gemm / LinearActFn run on the GPU library only: CPU tensors, other dtypes and mismatched reduction lengths are refused
    before anything is launched; ops.linear_act falls back to the framework's Linear only outside the GPU split-bf16 mode."""
    a, b = torch.zeros(4, 8), torch.zeros(3, 8)
    assert not ops.gemm_ok(a, b)
    with pytest.raises(ValueError):
        ops.gemm(a, False, b, False)
    lin = torch.nn.Linear(8, 3)
    x = torch.randn(5, 8)
    want = torch.nn.functional.leaky_relu(lin(x), 0.2)
    assert torch.equal(ops.linear_act(x, lin, ops.ACT_LEAKY, 0.2), want)          # CPU tensors: the framework's composition
    assert torch.equal(ops.linear_act(x, lin, ops.ACT_RELU), torch.relu(lin(x)))
    assert torch.equal(ops.linear_act(x, lin), lin(x))
