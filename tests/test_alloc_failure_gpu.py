"""Allocation failure is survivable: vs_test_fail_alloc(k) makes the k-th device / pinned allocation of the library fail, and k is
walked over EVERY allocation of the engine-level calls (align_batch on device- and host-resident frames, with and without phase
correlation; process_batch; process_clips).  For each k:
  * the call reports the failure (VsError carrying VS_ERR_HIP = -2 and "out of memory") -- it does not crash, hang or corrupt;
  * the NEXT call on the same handle succeeds and equals a fresh handle's result bit for bit (a failed call ends the running
    sequence, the reference's protocol for a failed kernel call: alignment.cpp:357-367 -- LastWidth = -1, re-initialise);
  * destroying the handle afterwards neither double-frees nor leaks (free device memory returns to where it was).
The walk ends at the first k that no longer fires: then every allocation of the call has been failed once.
"""
import gc

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

W, H = 320, 240


def _clip(n, seed=3, bits=8, w=W, h=H):
    from video_stabilizer_amd import synth
    frames, _ = synth.make_clip(w, h, n, seed=seed, channels=3, bits=bits)
    return frames


def _tr(ts):
    return [t.tup() for t in ts]


def _walk(vs, make, call, min_fired, throwing=False, stride=1):
    """returns the number of allocations that were failed; asserts the contract at every one of them.
    throwing: the allocation THROWS std::bad_alloc instead (vs_test_fail_alloc(-k)) -- a host allocation failing at that point of the call: the
    exception must stop at the C boundary (VS_ERR_NOMEM = -5) and leave the handle exactly as an error code does; stride: every stride-th k only"""
    vs.test_fail_alloc(0)
    call(make())                                   # process-wide one-time allocations (the warp's parameter ring) happen here
    ref = call(make())
    fired, k = 0, 1
    while True:
        h = make()
        vs.test_fail_alloc(-k if throwing else k)
        try:
            got, failed = call(h), False
        except vs.VsError as e:
            failed = True
            if throwing:
                assert "error -5" in str(e) and "bad_alloc" in str(e), str(e)
            else:
                assert "error -2" in str(e) and "out of memory" in str(e).lower(), str(e)
        seen = vs.test_fail_alloc(0)
        if not failed:
            assert seen < k, "allocation %d was failed (of %d made) but the call reported success" % (k, seen)
            assert got == ref
            break
        assert seen >= k
        fired += 1
        assert call(h) == ref, "k = %d: the call after the failed one differs from a fresh handle" % k
        assert call(h) is not None                 # and the handle keeps working (state carried over from a good call)
        del h
        gc.collect()
        k += stride
        assert k < 400, "the walk does not terminate"
    assert fired >= min_fired, "only %d allocations were failed: the hook does not see the call's allocations" % fired
    return fired


def test_align_batch_device_resident_every_allocation(gpu_vs):
    import torch
    vs = gpu_vs
    frames = _clip(6)
    dev = torch.from_numpy(frames).cuda()

    def call(h):
        st, ts = h.align_batch_device(dev.data_ptr(), 6, W, H, vs.FMT_BGR8)
        return list(st), _tr(ts)
    n = _walk(vs, lambda: vs.Aligner(device=0), call, min_fired=14)      # 10 device + 4 pinned slabs of ensure_capacity
    print("align_batch (device frames): %d allocations failed one by one" % n)


def test_align_batch_host_frames_and_host_selection(gpu_vs):
    vs = gpu_vs
    frames = _clip(5, seed=4)

    def call(h):
        st, ts = h.align_batch(frames)
        return list(st), _tr(ts)
    n = _walk(vs, lambda: vs.Aligner(device=0, select_mode=vs.SELECT_STL_HOST), call, min_fired=15)   # + the upload area
    print("align_batch (host frames, host selection): %d allocations" % n)


def test_align_batch_phase_correlate_every_allocation(gpu_vs):
    vs = gpu_vs
    frames = _clip(5, seed=5)

    def call(h):
        st, ts = h.align_batch(frames)
        return list(st), _tr(ts)
    n = _walk(vs, lambda: vs.Aligner(device=0, phase_correlate=1), call, min_fired=20)    # + twiddles, spectra, surfaces, candidates
    print("align_batch (phase_correlate): %d allocations" % n)


def test_regrow_with_carry_over_leaves_the_handle_as_it_was(gpu_vs):
    """ensure_capacity while a sequence is running: the failed regrow must not free or lose the old slabs (the frame carried over
    from the previous call lives in them) -- after it, the handle still aligns, and a sequence restarted on it equals a fresh one."""
    import torch
    vs = gpu_vs
    frames = _clip(12, seed=6)
    dev = torch.from_numpy(frames).cuda()
    fresh = vs.Aligner(device=0)
    want_tail = fresh.align_batch_device(dev[4:].data_ptr(), 8, W, H, vs.FMT_BGR8)
    want_tail = (list(want_tail[0]), _tr(want_tail[1]))
    for k in range(1, 15):
        h = vs.Aligner(device=0)
        h.align_batch_device(dev.data_ptr(), 4, W, H, vs.FMT_BGR8)           # cap = 4, sequence running
        vs.test_fail_alloc(k)
        with pytest.raises(vs.VsError, match="out of memory"):
            h.align_batch_device(dev[4:].data_ptr(), 8, W, H, vs.FMT_BGR8)   # needs cap 8: regrow -> allocation k fails
        assert vs.test_fail_alloc(0) >= k
        st, ts = h.align_batch_device(dev[4:].data_ptr(), 8, W, H, vs.FMT_BGR8)
        assert (list(st), _tr(ts)) == want_tail                               # frame 4 is a first frame again: a new sequence
        st, ts = h.align_batch_device(dev[8:].data_ptr(), 4, W, H, vs.FMT_BGR8)   # smaller batches still fit the (new) slabs
        assert len(st) == 4
        del h


@pytest.mark.parametrize("bits", [8, 10])
def test_stabilizer_process_batch_every_allocation(gpu_vs, bits):
    vs = gpu_vs
    frames = _clip(16, seed=7, bits=bits)

    def call(s):
        out, has = s.process_batch(frames)
        return list(has), out.tobytes()
    n = _walk(vs, lambda: vs.Stabilizer(device=0, lag=4, smoother_memory=2, crop_pixels=8, warp_mode=vs.WARP_LANCZOS2), call, min_fired=17)
    print("process_batch (%d-bit host frames): %d allocations" % (bits, n))


def test_stabilizer_process_clips_device_every_allocation(gpu_vs):
    """device-resident clips: the overlapped path (warps on their own stream, the next group's alignment prefetched)"""
    import torch
    vs = gpu_vs
    n_clips, fpc = 4, 34                                      # 33 pairs per clip -> 4 groups of one clip: prefetch + overlap + the small solver build
    frames = np.concatenate([_clip(fpc, seed=20 + c) for c in range(n_clips)])
    dev = torch.from_numpy(frames).cuda()
    crop = 8
    out = torch.zeros((n_clips * fpc, H - 2 * crop, W - 2 * crop, 3), dtype=torch.uint8, device="cuda")

    def call(s):
        out.zero_()
        r, has = s.process_clips_device(dev.data_ptr(), n_clips, fpc, W, H, vs.FMT_BGR8, out.data_ptr())
        torch.cuda.synchronize()
        return r, list(has), out.cpu().numpy().tobytes()
    n = _walk(vs, lambda: vs.Stabilizer(device=0, lag=5, crop_pixels=crop, warp_mode=vs.WARP_LANCZOS2), call, min_fired=14)
    print("process_clips (device frames, overlapped groups): %d allocations" % n)


def test_exceptions_stop_at_the_c_boundary(gpu_vs):
    """include/vs_amd.h: "No exceptions cross".  The same walks with the allocation THROWING std::bad_alloc: the call returns VS_ERR_NOMEM, the next
    call on the handle equals a fresh handle's, nothing leaks."""
    import torch
    vs = gpu_vs
    frames = _clip(6)
    dev = torch.from_numpy(frames).cuda()

    def align(h):
        st, ts = h.align_batch_device(dev.data_ptr(), 6, W, H, vs.FMT_BGR8)
        return list(st), _tr(ts)
    n1 = _walk(vs, lambda: vs.Aligner(device=0), align, min_fired=14, throwing=True)
    host = _clip(16, seed=7)

    def batch(s):
        out, has = s.process_batch(host)
        return list(has), out.tobytes()
    n2 = _walk(vs, lambda: vs.Stabilizer(device=0, lag=4, smoother_memory=2, crop_pixels=8), batch, min_fired=17, throwing=True)   # (library defaults: the fixed-point bilinear warp)
    # device-resident clips: the overlapped groups (warps on their own stream, the next group's alignment prefetched; exclusive solver build beside this warp)
    n_clips, fpc, crop = 4, 34, 8
    cdev = torch.from_numpy(np.concatenate([_clip(fpc, seed=20 + c) for c in range(n_clips)])).cuda()
    cout = torch.zeros((n_clips * fpc, H - 2 * crop, W - 2 * crop, 3), dtype=torch.uint8, device="cuda")

    def clips(st):
        cout.zero_()
        r, has = st.process_clips_device(cdev.data_ptr(), n_clips, fpc, W, H, vs.FMT_BGR8, cout.data_ptr())
        torch.cuda.synchronize()
        return r, list(has), cout.cpu().numpy().tobytes()
    n3 = _walk(vs, lambda: vs.Stabilizer(device=0, lag=5, crop_pixels=crop), clips, min_fired=14, throwing=True)
    print("exceptions stopped at the boundary: %d (align_batch, device frames) + %d (process_batch, host frames) + %d (process_clips, device frames)" % (n1, n2, n3))
    # a kernel-level call: the parameter ring's first allocation on a fresh thread would be the natural case; any allocation will do
    vs.test_fail_alloc(-1)
    with pytest.raises(vs.VsError, match="error -5"):
        vs.phase_correlate(np.zeros((64, 64), np.uint8), np.zeros((64, 64), np.uint8))
    vs.test_fail_alloc(0)
    assert vs.phase_correlate(np.zeros((64, 64), np.uint8), np.zeros((64, 64), np.uint8)) is not None


def test_failed_regrows_do_not_leak_device_memory(gpu_vs):
    import torch
    vs = gpu_vs
    frames = _clip(8, seed=9, w=1920, h=1080)
    dev = torch.from_numpy(frames).cuda()
    h = vs.Aligner(device=0, pyramid_min_width=256)
    h.align_batch_device(dev.data_ptr(), 8, 1920, 1080, vs.FMT_BGR8)
    del h
    gc.collect()
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for rep in range(3):
        for k in range(2, 15):                                # each leaves k - 1 freshly allocated slabs behind if they leak (~25 MB the set)
            h = vs.Aligner(device=0, pyramid_min_width=256)
            vs.test_fail_alloc(k)
            with pytest.raises(vs.VsError):
                h.align_batch_device(dev.data_ptr(), 8, 1920, 1080, vs.FMT_BGR8)
            vs.test_fail_alloc(0)
            del h
    gc.collect()
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info()[0]
    assert free0 - free1 < 64 << 20, "device memory went down by %.1f MB over 39 failed regrows" % ((free0 - free1) / 1e6)


def test_environment_variable_arms_the_hook():
    """VS_TEST_FAIL_ALLOC=k for programs that cannot call the hook (the harness programs): a child process, one GPU user"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import numpy as np\n"
            "from video_stabilizer_amd import capi, synth\n"
            "f, _ = synth.make_clip(160, 120, 3, seed=1, channels=3)\n"
            "a = capi.Aligner(device=0)\n"
            "try:\n"
            "    a.align_batch(f); print('NOFAIL')\n"
            "except capi.VsError as e:\n"
            "    print('FAILED', e)\n"
            "st, ts = a.align_batch(f); print('THEN', sum(st))\n") % root
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, VS_TEST_FAIL_ALLOC="3", VS_TEST_HOOKS="1"), capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    assert "FAILED" in out.stdout and "out of memory" in out.stdout.lower() and "THEN" in out.stdout, out.stdout
