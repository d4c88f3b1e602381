"""The C-ABI library loads on a CPU-only host and exports every symbol include/bgan.h declares."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "bgan.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(bg_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from building_gan_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "build with __graft_entry__.build() / make -C building-gan_amd/csrc"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), n


def test_python_binding_covers_the_header():
    from building_gan_amd import _lib
    # (functions that do not return an int status are bound by hand in _lib.load())
    declared = set(_declared()) - {"bg_version", "bg_error_string", "bg_abi_sizeof", "bg_generator_forward_workspace",
                                    "bg_d_coef_backward_workspace"}
    assert declared == set(_lib.SIGNATURES.keys())
    lib = _lib.load()
    assert lib.bg_version() >= 1
    assert lib.bg_error_string(-2).decode().startswith("shape")
    # struct layouts of the host binding and of the compiled library agree (load() refuses a mismatch)
    assert lib.bg_abi_sizeof(0) == ctypes.sizeof(_lib.GeneratorWeights)
    assert lib.bg_abi_sizeof(1) == ctypes.sizeof(_lib.GeneratorPlan) and lib.bg_abi_sizeof(7) == -1


def test_no_cpu_fallback():
    import pytest
    import torch
    from building_gan_amd import _lib, ops
    with pytest.raises(_lib.BganError):
        ops.gemm_raw([torch.zeros(8, 8)], torch.zeros(8, 8), False, True)
    with pytest.raises(_lib.BganError):
        from building_gan_amd.plan import build_plan
        from building_gan_amd.synthetic import make_batch
        build_plan(make_batch("tiny"))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "building-gan_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
