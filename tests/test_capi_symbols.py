"""-m "not gpu": the C-ABI library loads without a GPU, exports every symbol include/vs_amd.h declares,
the ctypes table covers the same set, and compute calls fail loudly (no CPU fallback) when no device exists."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "vs_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = set(re.findall(r"\b(vs_[a-z0-9_]+)\s*\(", text))
    return names


def test_header_declares_the_hot_path():
    names = _declared()
    for n in ("vs_pyr_down", "vs_grad_xy", "vs_grad_argmax", "vs_sparse_jac", "vs_sparse_warpdiff", "vs_sparse_ica",
              "vs_image_warp", "vs_bgr_image_warp", "vs_aligner_align_next", "vs_aligner_align_batch",
              "vs_stabilizer_process"):
        assert n in names


def test_library_exports_every_declared_symbol(vs):
    L = ctypes.CDLL(vs.LIB_PATH)
    missing = [n for n in sorted(_declared()) if not hasattr(L, n)]
    assert not missing, missing


def test_ctypes_table_matches_header(vs):
    assert set(vs.SIGNATURES) == _declared()


def test_no_cpu_fallback_without_device(vs):
    if vs.device_count() > 0:
        pytest.skip("a HIP device is present")
    with pytest.raises(vs.VsError, match="no usable HIP device"):
        vs.pyr_down(np.zeros((16, 16), np.uint8))
    with pytest.raises(vs.VsError):
        vs.Aligner()
    with pytest.raises(vs.VsError):
        vs.bgr_image_warp(np.zeros((8, 8, 3), np.uint8), vs.Transform.of())


def test_product_does_not_touch_the_oracle():
    # the product path must never import, link or call anything under oracle/
    pkg = os.path.join(ROOT, "video_stabilizer_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".hpp", ".h", ".sh")):
                text = open(os.path.join(dp, f), errors="ignore").read()
                assert "vs_oracle" not in text and "libvs_oracle" not in text and "from oracle" not in text, os.path.join(dp, f)
    out = os.popen("ldd %s 2>/dev/null" % os.path.join(pkg, "libvs_amd.so")).read()
    assert "oracle" not in out
