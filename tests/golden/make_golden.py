#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the CPU oracle (oracle/, the restatement of the reference path).

The reference ships no golden vectors and cannot be built in this image (needs Halide/OpenCV/Eigen), so the
fixtures are produced by the oracle, which is itself pinned by the hand-derived known answers in
tests/test_oracle_known_answers.py.  They freeze the oracle's outputs so that (a) an accidental change of
the oracle is caught on CPU and (b) the GPU path is compared with committed data, not only with a live run.
    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import oracle as O                      # noqa: E402
from video_stabilizer_amd import synth              # noqa: E402


def kernels_fixture():
    frames, _ = synth.make_clip(160, 120, 2, seed=101, path=[(0, 0, 0, 0), (0.004, -0.003, 1.6, -0.9)], margin=16)
    tmpl, key = frames
    d = {"tmpl": tmpl, "key": key}
    d["pyr1"] = O.pyr_down(key)
    gx, gy = O.grad_xy(key)
    ts, lmx, lmy = O.grad_argmax(gx, gy)
    jx, jy = O.sparse_jac(gx, gy, lmx, lmy)
    d.update(ts=np.int32(ts), lmx=lmx, lmy=lmy, jx=jx, jy=jy)
    t = (0.002, -0.001, 0.7, 0.4)
    d["t"] = np.array(t)
    wdx = O.sparse_warpdiff(tmpl, key, lmx, O.Transform.of(*t))
    wdy = O.sparse_warpdiff(tmpl, key, lmy, O.Transform.of(*t))
    ix, iy = O.select_smallest(wdx), O.select_smallest(wdy)
    selx, sely = lmx.reshape(2, -1)[:, ix], lmy.reshape(2, -1)[:, iy]
    sjx, sjy = jx.reshape(4, -1)[:, ix], jy.reshape(4, -1)[:, iy]
    d.update(wdx=wdx, wdy=wdy, idx_x=ix, idx_y=iy)
    d["ica"] = O.sparse_ica(tmpl, key, selx, sely, sjx, sjy, O.Transform.of(*t))
    H = O.hessian(sjx, sjy)
    cond, _, Hinv = O.condition_and_invert(H)
    d.update(H=H, Hinv=Hinv, cond=np.float64(cond))
    d["image_warp"] = O.image_warp(key, O.Transform.of(*t))
    bgr, _ = synth.make_clip(96, 64, 1, seed=102, channels=3, path=[(0, 0, 0, 0)], margin=8)
    d["bgr"] = bgr[0]
    tw = (0.01, 0.006, 2.3, -1.4)
    d["tw"] = np.array(tw)
    for mode in (0, 1):
        for border in (0, 1):
            d["warp_m%d_b%d" % (mode, border)] = O.bgr_image_warp(bgr[0], O.Transform.of(*tw), mode, border)
    d["gray"] = O.bgr_to_gray(bgr[0])
    np.savez_compressed(os.path.join(HERE, "kernels_160x120.npz"), **d)


def aligner_fixture():
    path = [(0, 0, 0, 0), (0.0, 0.0, 3.25, -2.5), (0.004, 0.002, 1.0, 2.0), (-0.003, 0.001, -2.0, 0.5), (0.001, -0.002, 0.5, -1.5)]
    frames, _ = synth.make_clip(320, 240, len(path), seed=103, path=path)
    al = O.Aligner()
    ts, oks, iters = [], [], []
    for f in frames:
        ok, t = al.align_next(f)
        d = al.debug()
        oks.append(ok)
        ts.append(t.tup())
        iters.append(list(d.iterations[:d.levels]))
    np.savez_compressed(os.path.join(HERE, "aligner_320x240.npz"), seed=np.int32(103), path=np.array(path), ok=np.array(oks),
                        transforms=np.array(ts), iterations=np.array(iters))
    # the frames themselves are regenerated from the seed by the test (synth is deterministic numpy)


def stabilizer_fixture():
    frames, _ = synth.make_clip(160, 128, 9, seed=104, channels=3)
    st = O.Stabilizer(lag=3, smoother_memory=1, crop_pixels=8, warp_mode=O.WARP_LANCZOS2)   # the fixture predates the bilinear default
    meas, acc, outs = [], [], []
    for f in frames:
        o = st.process(f)
        m, a, ok = st.state()
        meas.append(m.tup())
        acc.append(a.tup())
        outs.append(-1 if o is None else int(o.astype(np.uint32).sum()))
    np.savez_compressed(os.path.join(HERE, "stabilizer_160x128.npz"), seed=np.int32(104), meas=np.array(meas), accum=np.array(acc),
                        out_checksum=np.array(outs, np.int64), last=o)


def warp_modes_fixture():
    """round 4: the forms of bgr_image_warp that the first fixture predates -- the contracted Lanczos2 twin (VSO_WARP_LANCZOS2_CONTRACTED), 10-bit
    frames in all three modes (the word-tile bilinear path and the u16 Lanczos kernels on the GPU side), a window -- and the two selection rules"""
    d = {}
    bgr8, _ = synth.make_clip(96, 64, 1, seed=105, channels=3, path=[(0, 0, 0, 0)], margin=8)
    bgr10, _ = synth.make_clip(83, 47, 1, seed=106, channels=3, bits=10, path=[(0, 0, 0, 0)], margin=8)
    d["bgr8"], d["bgr10"] = bgr8[0], bgr10[0]
    tw = (0.012, -0.007, -1.7, 2.45)
    d["tw"] = np.array(tw)
    for border in (0, 1):
        d["u8_m2_b%d" % border] = O.bgr_image_warp(bgr8[0], O.Transform.of(*tw), O.WARP_LANCZOS2_CONTRACTED, border)
        for mode in (0, 1, 2):
            d["u10_m%d_b%d" % (mode, border)] = O.bgr_image_warp(bgr10[0], O.Transform.of(*tw), mode, border, max_value=1023)
    rng = np.random.default_rng(107)
    wd = np.minimum(rng.poisson(3.0, (27, 31)), 65535).astype(np.uint16)
    d["wd"] = wd
    d["sel_stl"] = O.select_smallest(wd, 0.8)
    d["sel_stable"] = O.select_smallest_stable(wd, 0.8)
    np.savez_compressed(os.path.join(HERE, "warp_modes_96x64.npz"), **d)


def warp_modes_r05_fixture():
    """round 5: the separable member of the Lanczos2 family (VSO_WARP_LANCZOS2_SEPARABLE) and cv::warpAffine's fixed-point bilinear
    (VSO_WARP_BILINEAR_CV: the transform is the FORWARD map there), 8- and 10-bit, both borders; the same frames and transform as the round-4 file"""
    r4 = np.load(os.path.join(HERE, "warp_modes_96x64.npz"))
    d = {"tw": r4["tw"]}
    for border in (0, 1):
        for name, mode in (("sep", O.WARP_LANCZOS2_SEPARABLE), ("cv", O.WARP_BILINEAR_CV)):
            d["u8_%s_b%d" % (name, border)] = O.bgr_image_warp(r4["bgr8"], O.Transform.of(*r4["tw"]), mode, border)
            d["u10_%s_b%d" % (name, border)] = O.bgr_image_warp(r4["bgr10"], O.Transform.of(*r4["tw"]), mode, border, max_value=1023)
    np.savez_compressed(os.path.join(HERE, "warp_modes_r05_96x64.npz"), **d)


if __name__ == "__main__":
    if "--only-new" not in sys.argv and "--only-r05" not in sys.argv:   # (the first three files are frozen: regenerate them only on purpose)
        kernels_fixture()
        aligner_fixture()
        stabilizer_fixture()
    if "--only-r05" not in sys.argv:
        warp_modes_fixture()
    warp_modes_r05_fixture()
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))
