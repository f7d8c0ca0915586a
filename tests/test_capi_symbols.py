"""CPU suite: the C-ABI library builds, loads and exports every symbol include/*.h declares
(no compute calls: there is no GPU here), and the product path refuses to run without it."""
import ctypes
import glob
import os
import re

import pytest
import torch

from conftest import PKG_DIR, ROOT


def _declared():
    names = set()
    for h in glob.glob(os.path.join(ROOT, "include", "*.h")):
        txt = open(h).read()
        txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
        names |= set(re.findall(r"\b(hp_[a-z0-9_]+)\s*\(", txt))
    return sorted(names)


@pytest.fixture(scope="module")
def lib():
    import importlib.util
    spec = importlib.util.spec_from_file_location("hp_build", os.path.join(PKG_DIR, "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    so = mod.build(verbose=False)
    return ctypes.CDLL(so)


def test_header_declares_the_reference_launchers():
    names = _declared()
    # the five launchers of structural_loss.cpp:11-15 + the model entry points
    for n in ["hp_approxmatch", "hp_matchcost", "hp_matchcostgrad", "hp_nndistance", "hp_nndistancegrad",
              "hp_chamfer_forward", "hp_chamfer_backward", "hp_gemm_f32", "hp_encoder_forward", "hp_encoder_backward",
              "hp_hypernet_forward", "hp_hypernet_backward", "hp_target_forward", "hp_target_backward",
              "hp_sample_points", "hp_kld_forward", "hp_kld_backward", "hp_adam_step"]:
        assert n in names, n


def test_library_exports_every_declared_symbol(lib):
    for n in _declared():
        assert hasattr(lib, n), f"{n} declared in include/ but not exported"


def test_workspace_queries_run_on_host(lib):
    lib.hp_approxmatch_workspace_floats.restype = ctypes.c_long
    lib.hp_target_theta_size.restype = ctypes.c_long
    # packed candidate records of emd.hip: (N+8)*(4+16) + (M+8)*(4+1+16) floats per cloud, N and M padded to multiples of 64;
    # round 6: + per set the k-d permutation (1 per point), block boxes (6 per 8 points), tile boxes (6 per 64 points), + 16 (flag)
    extra = lambda P: P + 6 * (P // 8) + 6 * (P // 64)
    assert lib.hp_approxmatch_workspace_floats(64, 2048, 2048) == 64 * (2056 * 20 + 2056 * 21 + 2 * extra(2048) + 16)
    assert lib.hp_approxmatch_workspace_floats(1, 100, 37) == (128 + 8) * 20 + (64 + 8) * 21 + extra(128) + extra(64) + 16
    ch = (ctypes.c_int * 4)(32, 64, 128, 64)
    assert lib.hp_target_theta_size(4, ch) == 19011
    assert lib.hp_target_theta_size(0, ch) == -1


def test_invalid_arguments_are_rejected_without_a_gpu(lib):
    # argument validation happens before any HIP call
    assert lib.hp_nndistance(-1, 1, None, 1, None, None, None, None, None, None) == -1
    assert lib.hp_chamfer_forward(0, 1, None, 1, None, None, None, None, None, None, None, None) == -1
    assert lib.hp_gemm_f32(None, None) == -1


def test_product_path_has_no_cpu_fallback():
    from hyperpocket_amd import HipExtensionError
    from hyperpocket_amd.losses.champfer_loss import ChamferLoss
    from hyperpocket_amd.utils.pytorch_structural_losses.StructuralLossesBackend import NNDistance
    a = torch.rand(1, 8, 3)
    with pytest.raises(HipExtensionError):
        NNDistance(a, a)
    with pytest.raises(HipExtensionError):
        ChamferLoss()(a, a)


def test_product_package_never_imports_the_oracle():
    for path in glob.glob(os.path.join(PKG_DIR, "**", "*.py"), recursive=True):
        src = open(path).read()
        assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), path
        assert "structural_losses_ref" not in src and "hyperpocket_ref" not in src, path
    for path in glob.glob(os.path.join(PKG_DIR, "csrc", "*")):
        assert "oracle/" not in open(path).read(), path


def test_strict_fp32_flips_every_piece_arithmetic_switch_and_restores_it(lib):
    """ops.strict_fp32(): inside the block every kernel family that forms fp32 products from f16 / bf16 pieces is on its fp32
    form (the library's process-wide switches read 0); afterwards each switch holds what it held before.  Host-only calls."""
    from hyperpocket_amd import ops
    loaded = ops.load_library()
    before = []
    for name in ops._PIECE_SWITCHES:
        fn = getattr(loaded, name)
        was = fn(1)
        fn(was)
        before.append(was)
    with ops.strict_fp32():
        for name in ops._PIECE_SWITCHES:
            fn = getattr(loaded, name)
            inside = fn(0)
            assert inside == 0, name
    for name, was in zip(ops._PIECE_SWITCHES, before):
        fn = getattr(loaded, name)
        now = fn(was)
        assert now == was, (name, now, was)
    # the final-sweep derivation of the EMD is its own switch (not an fp32-vs-pieces question): default on
    assert loaded.hp_emd_set_final_derive(1) == 1


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="needs the reference tree (build container only)")
def test_no_same_named_python_file_is_the_references_text():
    """The boundary forces names, sizes, registration order — not the text (three model shells were 56-81 % the
    reference's lines).  Share of the reference file's code lines found verbatim in the same-named file here: the model
    shells under 30 %, every other same-named file under 40 % (tools/line_overlap.py is the measure)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("line_overlap", os.path.join(ROOT, "tools", "line_overlap.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    pkg = os.path.join(PKG_DIR, "hyperpocket_amd")
    seen = 0
    for root, _, files in os.walk(pkg):
        for f in files:
            mine = os.path.join(root, f)
            rel = os.path.relpath(mine, pkg)
            ref = os.path.join("/root/reference", rel)
            if not f.endswith(".py") or not os.path.exists(ref):
                continue
            r, m = mod.code_lines(ref), set(mod.code_lines(mine))
            if len(r) < 10:
                continue
            share = sum(1 for ln in r if ln in m) / len(r)
            seen += 1
            assert share < (0.30 if rel.startswith("model/") else 0.40), (rel, round(share, 2))
    assert seen >= 8
