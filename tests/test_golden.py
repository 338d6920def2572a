"""Committed golden fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py from the oracle):
the oracle must still reproduce them (CPU), and the GPU path must match them through the C ABI."""
import os

import numpy as np
import pytest

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _k():
    return np.load(os.path.join(G, "kernels_160x120.npz"))


def _kernel_outputs(m, d):
    """every kernel-level output for the fixture's inputs, computed by module m (oracle or capi)"""
    T = m.Transform.of
    out = {"pyr1": m.pyr_down(d["key"])}
    gx, gy = m.grad_xy(d["key"])
    ts, lmx, lmy = m.grad_argmax(gx, gy)
    jx, jy = m.sparse_jac(gx, gy, lmx, lmy)
    out.update(ts=np.int32(ts), lmx=lmx, lmy=lmy, jx=jx, jy=jy)
    t = T(*d["t"])
    out["wdx"] = m.sparse_warpdiff(d["tmpl"], d["key"], lmx, t)
    out["wdy"] = m.sparse_warpdiff(d["tmpl"], d["key"], lmy, t)
    out["image_warp"] = m.image_warp(d["key"], t)
    for mode in (0, 1):
        for border in (0, 1):
            out["warp_m%d_b%d" % (mode, border)] = m.bgr_image_warp(d["bgr"], T(*d["tw"]), mode, border)
    out["gray"] = m.bgr_to_gray(d["bgr"])
    return out


def test_oracle_reproduces_kernel_fixture(oracle):
    d = _k()
    out = _kernel_outputs(oracle, d)
    for k, v in out.items():
        assert np.array_equal(v, d[k]), k
    assert np.array_equal(oracle.select_smallest(d["wdx"]), d["idx_x"])
    assert np.array_equal(oracle.select_smallest(d["wdy"]), d["idx_y"])
    selx, sely = d["lmx"].reshape(2, -1)[:, d["idx_x"]], d["lmy"].reshape(2, -1)[:, d["idx_y"]]
    sjx, sjy = d["jx"].reshape(4, -1)[:, d["idx_x"]], d["jy"].reshape(4, -1)[:, d["idx_y"]]
    assert np.array_equal(oracle.sparse_ica(d["tmpl"], d["key"], selx, sely, sjx, sjy, oracle.Transform.of(*d["t"])), d["ica"])
    assert np.array_equal(oracle.hessian(sjx, sjy), d["H"])


def test_oracle_reproduces_aligner_and_stabilizer_fixtures(oracle):
    from video_stabilizer_amd import synth
    a = np.load(os.path.join(G, "aligner_320x240.npz"))
    frames, _ = synth.make_clip(320, 240, len(a["path"]), seed=int(a["seed"]), path=[tuple(p) for p in a["path"]])
    al = oracle.Aligner()
    for i, f in enumerate(frames):
        ok, t = al.align_next(f)
        assert ok == bool(a["ok"][i]) and t.tup() == tuple(a["transforms"][i])
    s = np.load(os.path.join(G, "stabilizer_160x128.npz"))
    frames, _ = synth.make_clip(160, 128, len(s["meas"]), seed=int(s["seed"]), channels=3)
    st = oracle.Stabilizer(lag=3, smoother_memory=1, crop_pixels=8, warp_mode=oracle.WARP_LANCZOS2)
    for i, f in enumerate(frames):
        o = st.process(f)
        m, acc, _ = st.state()
        assert m.tup() == tuple(s["meas"][i]) and acc.tup() == tuple(s["accum"][i])
        assert (-1 if o is None else int(o.astype(np.uint32).sum())) == int(s["out_checksum"][i])
    assert np.array_equal(o, s["last"])


@pytest.mark.gpu
def test_gpu_matches_kernel_fixture(gpu_vs):
    d = _k()
    out = _kernel_outputs(gpu_vs, d)
    for k, v in out.items():
        assert np.array_equal(v, d[k]), k            # bit-exact: integer outputs and order-preserving fp32
    got, status = gpu_vs.select_smallest(np.stack([d["wdx"], d["wdy"]]))
    assert not status.any() and np.array_equal(got[0], d["idx_x"]) and np.array_equal(got[1], d["idx_y"])
    selx, sely = d["lmx"].reshape(2, -1)[:, d["idx_x"]], d["lmy"].reshape(2, -1)[:, d["idx_y"]]
    sjx, sjy = d["jx"].reshape(4, -1)[:, d["idx_x"]], d["jy"].reshape(4, -1)[:, d["idx_y"]]
    ica = gpu_vs.sparse_ica(d["tmpl"], d["key"], selx, sely, sjx, sjy, gpu_vs.Transform.of(*d["t"]))
    assert np.abs(ica - d["ica"]).max() <= 1e-12 * np.abs(d["ica"]).max() + 1e-9


@pytest.mark.gpu
def test_gpu_matches_aligner_and_stabilizer_fixtures(gpu_vs):
    from video_stabilizer_amd import synth
    a = np.load(os.path.join(G, "aligner_320x240.npz"))
    frames, _ = synth.make_clip(320, 240, len(a["path"]), seed=int(a["seed"]), path=[tuple(p) for p in a["path"]])
    for mode in (gpu_vs.SELECT_STL_HOST, gpu_vs.SELECT_DEVICE):
        al = gpu_vs.Aligner(device=0, select_mode=mode)
        st, ts = al.align_batch(frames)
        for i in range(len(frames)):
            assert bool(st[i]) == bool(a["ok"][i])
            assert np.abs(np.array(ts[i].tup()) - a["transforms"][i]).max() < 1e-4       # north_star tolerance
            assert list(al.info(i).iterations[:a["iterations"].shape[1]]) == list(a["iterations"][i])
    s = np.load(os.path.join(G, "stabilizer_160x128.npz"))
    frames, _ = synth.make_clip(160, 128, len(s["meas"]), seed=int(s["seed"]), channels=3)
    sg = gpu_vs.Stabilizer(device=0, lag=3, smoother_memory=1, crop_pixels=8, warp_mode=gpu_vs.WARP_LANCZOS2)
    for i, f in enumerate(frames):
        o = sg.process(f)
        m, acc, _ = sg.state()
        assert np.abs(np.array(m.tup()) - s["meas"][i]).max() < 1e-4
        assert (o is None) == (int(s["out_checksum"][i]) == -1)
    d = np.abs(o.astype(np.int16) - s["last"].astype(np.int16))
    assert d.max() <= 1 and (d != 0).mean() < 1e-2


def _warp_modes_outputs(m, d):
    T = m.Transform.of
    contracted = getattr(m, "WARP_LANCZOS2_CONTRACTED", None) or m.WARP_LANCZOS2_FAST      # the oracle's / the product's name of mode 2
    out = {}
    for border in (0, 1):
        out["u8_m2_b%d" % border] = m.bgr_image_warp(d["bgr8"], T(*d["tw"]), contracted, border)
        for mode in (0, 1, 2):
            out["u10_m%d_b%d" % (mode, border)] = m.bgr_image_warp(d["bgr10"], T(*d["tw"]), mode, border, max_value=1023)
    return out


def test_oracle_reproduces_warp_modes_fixture(oracle):
    """round-4 fixture: the contracted Lanczos2 twin, 10-bit frames in all three modes, the two selection rules"""
    d = np.load(os.path.join(G, "warp_modes_96x64.npz"))
    for k, v in _warp_modes_outputs(oracle, d).items():
        assert np.array_equal(v, d[k]), k
    assert np.array_equal(oracle.select_smallest(d["wd"], 0.8), d["sel_stl"])
    assert np.array_equal(oracle.select_smallest_stable(d["wd"], 0.8), d["sel_stable"])
    assert int(d["u10_m1_b0"].max()) > 255                  # (really 10-bit content)


def _warp_modes_r05_outputs(m, d4, d5):
    T = m.Transform.of
    sep = getattr(m, "WARP_LANCZOS2_SEPARABLE", None) or m.WARP_LANCZOS2_SEP                # the oracle's / the product's name of mode 3
    out = {}
    for border in (0, 1):
        for name, mode in (("sep", sep), ("cv", m.WARP_BILINEAR_CV)):
            out["u8_%s_b%d" % (name, border)] = m.bgr_image_warp(d4["bgr8"], T(*d5["tw"]), mode, border)
            out["u10_%s_b%d" % (name, border)] = m.bgr_image_warp(d4["bgr10"], T(*d5["tw"]), mode, border, max_value=1023)
    return out


def test_oracle_reproduces_warp_modes_r05_fixture(oracle):
    """round-5 fixture: the separable Lanczos2 twin and cv::warpAffine's fixed-point bilinear, 8- and 10-bit, both borders"""
    d4, d5 = np.load(os.path.join(G, "warp_modes_96x64.npz")), np.load(os.path.join(G, "warp_modes_r05_96x64.npz"))
    for k, v in _warp_modes_r05_outputs(oracle, d4, d5).items():
        assert np.array_equal(v, d5[k]), k


@pytest.mark.gpu
def test_gpu_matches_warp_modes_r05_fixture(gpu_vs):
    d4, d5 = np.load(os.path.join(G, "warp_modes_96x64.npz")), np.load(os.path.join(G, "warp_modes_r05_96x64.npz"))
    for k, v in _warp_modes_r05_outputs(gpu_vs, d4, d5).items():
        assert np.array_equal(v, d5[k]), k


@pytest.mark.gpu
def test_gpu_matches_warp_modes_fixture(gpu_vs):
    d = np.load(os.path.join(G, "warp_modes_96x64.npz"))
    for k, v in _warp_modes_outputs(gpu_vs, d).items():
        assert np.array_equal(v, d[k]), k
    got, status = gpu_vs.select_smallest(d["wd"], 0.8)
    assert not status.any() and np.array_equal(got[0], d["sel_stl"])
    assert np.array_equal(gpu_vs.select_smallest_stable(d["wd"], 0.8)[0], d["sel_stable"])
