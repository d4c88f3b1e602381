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
                                    "bg_d_coef_backward_workspace", "bg_critic_head_workspace", "bg_gstack_workspace", "bg_pstack_workspace", "bg_gloss_head_workspace"}
    assert declared == set(_lib.SIGNATURES.keys())
    lib = _lib.load()
    assert lib.bg_version() >= 1
    assert lib.bg_error_string(-2).decode().startswith("shape")
    # struct layouts of the host binding and of the compiled library agree (load() refuses a mismatch)
    assert lib.bg_abi_sizeof(0) == ctypes.sizeof(_lib.GeneratorWeights)
    assert lib.bg_abi_sizeof(1) == ctypes.sizeof(_lib.GeneratorPlan) and lib.bg_abi_sizeof(7) == -1
    assert lib.bg_abi_sizeof(2) == ctypes.sizeof(_lib.PlanIn)
    assert lib.bg_abi_sizeof(3) == ctypes.sizeof(_lib.CriticHead)
    assert lib.bg_abi_sizeof(4) == ctypes.sizeof(_lib.GeneratorBackPlan)


def test_plan_piece_enums_match_the_header():
    """The binding's piece / statistic names are the header's BGP_* / BGS_* enums, in order."""
    from building_gan_amd import _lib
    text = open(os.path.join(ROOT, "include", "bgan.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    bgp = re.findall(r"\bBGP_([A-Z0-9_]+)\b", text)
    i32 = bgp[: bgp.index("N_I32")]
    f32 = bgp[bgp.index("N_I32") + 1: bgp.index("N_F32")]
    assert tuple(i32) == _lib.PLAN_I32 and tuple(f32) == _lib.PLAN_F32
    bgs = re.findall(r"\bBGS_([A-Z0-9_]+)\b", text)
    assert tuple(bgs[: bgs.index("N")]) == _lib.PLAN_STATS


def test_plan_layout_is_aligned_disjoint_and_rejects_what_it_is_not_laid_out_for():
    """bg_plan_layout is host arithmetic: pieces are 256-byte aligned, in order, and large enough."""
    from building_gan_amd import _lib
    lib = _lib.load()
    pin = _lib.PlanIn(Np=37, Nv=501, Ep=90, Ev=2400, Ec=1300, B=3, S=17, n_slot=18)
    oi, of, tot = (ctypes.c_int64 * len(_lib.PLAN_I32))(), (ctypes.c_int64 * len(_lib.PLAN_F32))(), (ctypes.c_int64 * 2)()
    assert lib.bg_plan_layout(ctypes.byref(pin), oi, of, tot) == 0
    oi, of = list(oi), list(of)
    assert all(o % 64 == 0 for o in oi + of) and len(set(oi)) == len(oi) and of == sorted(of)
    need = {"PF_COL": 90, "VFW_ROWPTR": 502, "VFW_COL": 2400, "CV_PERM": 1300, "SS_COL": 17 * 17, "PP_COL": 90 + 37 * 37,
            "ARANGE": 1301, "STORYGRAPH64": 34, "FLAGS": 44, "STATS": len(_lib.PLAN_STATS),
            "CURSOR": 3 * 38 + 19 + 3 * 4 + 3 * 502 + 18}
    order = sorted(oi) + [tot[0]]
    for name, n in need.items():
        k = _lib.PLAN_I32.index(name)
        assert order[order.index(oi[k]) + 1] - oi[k] >= n, name
    assert oi[_lib.PLAN_I32.index("STATS")] - oi[_lib.PLAN_I32.index("FLAGS")] == 64      # read back with one copy
    assert tot[1] >= of[-1] + 37
    for bad in (dict(Np=0), dict(S=2049), dict(Np=2049), dict(Ev=0)):
        kw = dict(Np=37, Nv=501, Ep=90, Ev=2400, Ec=1300, B=3, S=17, n_slot=18)
        kw.update(bad)
        assert lib.bg_plan_layout(ctypes.byref(_lib.PlanIn(**kw)), (ctypes.c_int64 * 64)(), (ctypes.c_int64 * 16)(), tot) == -2
    assert lib.bg_plan_build(None, None, None, None) == -1


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
