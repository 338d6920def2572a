"""N > 1 path on CPU: two gloo ranks shard the clips, time a (dummy) step loop between barriers and aggregate
exactly as bench.py does on the GPU box (nccl = RCCL there)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, time, json
sys.path.insert(0, %r)
from video_stabilizer_amd import dist as D
world, rank, local = D.env_world()
d = D.init("gloo", rank, world)
clips = D.shard_clips(7, rank, world)
d.barrier()
t0 = time.perf_counter()
frames = aligned = 0
for c in clips:
    frames += 10 + c
    aligned += 9 + c
time.sleep(0.05 * (rank + 1))
d.barrier()
dt = time.perf_counter() - t0
secs, tot_f, tot_a = D.aggregate(dt, frames, aligned)
per_rank = D.gather_seconds(dt)
roll = D.roll_call({"rank": rank, "device_key": "box/gpu%%d" %% (0 if os.environ.get("SAME_DEVICE") else rank)})
print(json.dumps({"rank": rank, "clips": clips, "secs": secs, "frames": tot_f, "aligned": tot_a, "own": dt, "per_rank": per_rank,
                  "roll": roll, "clash_8_visible": D.shared_devices(roll, 8), "clash_1_visible": D.shared_devices(roll, 1)}))
d.destroy_process_group()
'''


def test_two_rank_gloo_shard_and_aggregate(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(WORKER % ROOT)
    port = str(29600 + os.getpid() % 300)
    procs = []
    for r in range(2):
        env = dict(os.environ, WORLD_SIZE="2", RANK=str(r), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=120)
        assert p.returncode == 0, e
        import json
        outs.append(json.loads(o.strip().splitlines()[-1]))
    outs.sort(key=lambda d: d["rank"])
    assert outs[0]["clips"] == [0, 2, 4, 6] and outs[1]["clips"] == [1, 3, 5]
    total = sum(10 + c for c in range(7))
    for o in outs:
        assert o["frames"] == total and o["aligned"] == total - 7
        assert o["secs"] >= max(x["own"] for x in outs) - 1e-9        # max over ranks, identical on every rank
    assert outs[0]["secs"] == outs[1]["secs"]
    # the per-rank view (bench.py's c4_strong.per_rank_seconds): every rank sees every rank's own seconds, in rank order
    for o in outs:
        assert o["per_rank"] == [outs[0]["own"], outs[1]["own"]] and max(o["per_rank"]) == o["secs"]
        # the roll call bench.py makes before anything is timed: every rank has every rank's record, in rank order; distinct devices: no clash
        assert [e["rank"] for e in o["roll"]] == [0, 1] and o["clash_8_visible"] == [] and o["clash_1_visible"] == []


def test_roll_call_finds_two_ranks_on_one_device(tmp_path):
    """both ranks report the same physical device: a launch error when the node shows a device for each (bench.py ends the job with exit code 5),
    an allowed rehearsal when there are more ranks than visible devices"""
    import json
    script = tmp_path / "w.py"
    script.write_text(WORKER % ROOT)
    port = str(29900 + os.getpid() % 90)
    procs = []
    for r in range(2):
        env = dict(os.environ, WORLD_SIZE="2", RANK=str(r), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=port, SAME_DEVICE="1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    for p in procs:
        o, e = p.communicate(timeout=120)
        assert p.returncode == 0, e
        j = json.loads(o.strip().splitlines()[-1])
        assert j["clash_8_visible"] == [["box/gpu0", [0, 1]]] and j["clash_1_visible"] == []


def test_roll_call_without_a_process_group():
    from video_stabilizer_amd import dist as D
    assert D.roll_call({"rank": 0, "device_key": "x"}) == [{"rank": 0, "device_key": "x"}]


FALLBACK_WORKER = r'''
import os, sys, json
sys.path.insert(0, %r)
import torch
from video_stabilizer_amd import dist as D
world, rank, local = D.env_world()
# no GPU in this process: RCCL cannot come up, on either rank -- the group must come back on gloo, with the reason
d, used, why = D.init_with_fallback("nccl", rank, world, device_id=torch.device("cuda", 0), probe_seconds=30)
secs, f, a = D.aggregate(1.0 + rank, 10, 9)
print(json.dumps({"rank": rank, "used": used, "why": why, "secs": secs, "frames": f, "aligned": a, "world": d.get_world_size()}))
d.destroy_process_group()
'''


def test_rccl_init_failure_falls_back_to_gloo(tmp_path):
    """if RCCL cannot be initialised the report scalars are aggregated on the host and the line says so"""
    import json
    import pytest
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: RCCL comes up")
    script = tmp_path / "f.py"
    script.write_text(FALLBACK_WORKER % ROOT)
    port = str(29950 + os.getpid() % 40)
    procs = []
    for r in range(2):
        env = dict(os.environ, WORLD_SIZE="2", RANK=str(r), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    for p in procs:
        o, e = p.communicate(timeout=240)
        assert p.returncode == 0, e[-2000:]
        j = json.loads(o.strip().splitlines()[-1])
        assert j["used"] == "gloo" and j["why"] and j["world"] == 2
        assert j["secs"] == 2.0 and j["frames"] == 20 and j["aligned"] == 18


DISAGREE_WORKER = r'''
import os, sys, json
sys.path.insert(0, %r)
import torch
from video_stabilizer_amd import dist as D
world, rank, local = D.env_world()
fail = os.environ.get("FAIL_RANK")
# the "RCCL" probe group is a second gloo group here (no GPU in the container): it works on every rank unless a failure is injected
d, used, why = D.init_with_fallback("nccl", rank, world, probe_seconds=30, _probe_backend="gloo",
                                    _fail_probe_on_rank=int(fail) if fail else None)
D.barrier()
secs, f, a = D.aggregate(1.0 + rank, 10, 9, device=D.report_device())
per = D.gather_seconds(1.0 + rank, device=D.report_device())
print(json.dumps({"rank": rank, "used": used, "why": why, "secs": secs, "frames": f, "per": per}))
d.destroy_process_group()
'''


def _run_ranks(script, world, port, extra_env=None):
    import json
    procs = []
    for r in range(world):
        env = dict(os.environ, WORLD_SIZE=str(world), RANK=str(r), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=port, **(extra_env or {}))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=240)
        assert p.returncode == 0, e[-2000:]
        outs.append(json.loads(o.strip().splitlines()[-1]))
    return sorted(outs, key=lambda j: j["rank"])


def test_ranks_agree_on_the_backend_when_the_probe_fails_on_one_rank_only(tmp_path):
    """the advisor's case: RCCL comes up on some ranks and not on others.  No rank may go on alone: every rank ends on gloo, with
    the failing rank's reason -- on the ONE rendezvous port the launcher gave (no port + 1)."""
    script = tmp_path / "d.py"
    script.write_text(DISAGREE_WORKER % ROOT)
    outs = _run_ranks(script, 3, str(29800 + os.getpid() % 100), {"FAIL_RANK": "1"})
    for j in outs:
        assert j["used"] == "gloo" and "rank 1" in j["why"] and "injected" in j["why"]
        assert j["secs"] == 3.0 and j["frames"] == 30 and j["per"] == [1.0, 2.0, 3.0]


def test_ranks_agree_on_the_probed_backend_when_it_works_everywhere(tmp_path):
    script = tmp_path / "d.py"
    script.write_text(DISAGREE_WORKER % ROOT)
    outs = _run_ranks(script, 2, str(29900 + os.getpid() % 40))
    for j in outs:
        assert j["used"] == "nccl" and j["why"] is None          # (the name of the requested backend: the probe group stood in for it)
        assert j["secs"] == 2.0 and j["frames"] == 20 and j["per"] == [1.0, 2.0]


def test_gather_seconds_without_a_process_group():
    from video_stabilizer_amd import dist as D
    assert D.gather_seconds(1.25) == [1.25]


def test_shard_covers_every_clip_once():
    from video_stabilizer_amd import dist as D
    for world in (1, 2, 4, 8):
        got = sorted(i for r in range(world) for i in D.shard_clips(64, r, world))
        assert got == list(range(64))
        assert max(len(D.shard_clips(64, r, world)) for r in range(world)) == 64 // world


def test_bench_launcher_starts_ranks_and_fails_loudly_without_a_gpu():
    """`bench.py --gpus 2` with no launcher environment spawns its two ranks (each gets RANK / LOCAL_RANK / WORLD_SIZE and a
    shared rendezvous port).  In this container there is no GPU, so the ranks must fail loudly -- and the launcher must hand
    that failure back as its own exit code instead of printing a result line."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("a GPU is present: the launcher is exercised by tests/test_bench_gpu.py::test_bench_spawns_its_own_ranks")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--frames", "4", "--steps", "1", "--warmup", "0",
                          "--no-cpu-baseline", "--no-roofline-4k", "--dist-backend", "gloo"], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode != 0
    assert "{" not in out.stdout


def test_spawn_ranks_environment(tmp_path, monkeypatch):
    """the launcher's contract: N children of the same script, RANK = LOCAL_RANK = 0..N-1, WORLD_SIZE = N, one port for all"""
    import bench
    import sys
    probe = tmp_path / "probe.py"
    probe.write_text("import os, sys\n"
                     "open(os.path.join(sys.argv[1], 'r' + os.environ['RANK']), 'w').write(' '.join(os.environ[k] for k in "
                     "('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')))\n"
                     "print('{\"rank\": %s}' % os.environ['RANK'])\n")
    monkeypatch.setattr(bench, "__file__", str(probe))
    rc = bench.spawn_ranks(3, [str(tmp_path)])
    assert rc == 0
    seen = [open(tmp_path / ("r%d" % r)).read().split() for r in range(3)]
    assert [s[0] for s in seen] == ["0", "1", "2"] and [s[1] for s in seen] == ["0", "1", "2"]
    assert {s[2] for s in seen} == {"3"} and {s[3] for s in seen} == {"127.0.0.1"} and len({s[4] for s in seen}) == 1


EIGHT_WORKER = r'''
import os, sys, time, json
sys.path.insert(0, %r)
from video_stabilizer_amd import dist as D
world, rank, local = D.env_world()
d = D.init("gloo", rank, world)
if os.environ.get("VS_TEST_FAIL_RANK", "") == str(rank):
    os._exit(7)                                     # a rank that dies after the group is up: the launcher must end the job with ITS code
mine = D.shard_clips(64, rank, world)               # BASELINE configs[3]: 64 clips, clip i -> rank i mod N
D.barrier()
t0 = time.perf_counter()
time.sleep(0.02 * (1 + rank %% 3))
D.barrier()
dt = time.perf_counter() - t0
secs, frames, aligned = D.aggregate(dt, 120 * len(mine), 119 * len(mine))
per_rank = D.gather_seconds(dt)
if rank == 0:
    print(json.dumps({"clips_per_rank": [len(D.shard_clips(64, r, world)) for r in range(world)], "mine": mine, "secs": secs, "frames": frames,
                      "aligned": aligned, "per_rank": per_rank}))
d.destroy_process_group()
'''


def _launch_eight(tmp_path, fail_rank=None):
    """eight ranks through bench.py's own launcher loop (spawn_ranks watches every child; the first failing rank ends the job)"""
    import importlib.util
    script = tmp_path / "e.py"
    script.write_text(EIGHT_WORKER % ROOT)
    driver = tmp_path / "drive.py"
    driver.write_text(
        "import sys, os\n"
        "sys.path.insert(0, %r)\n"
        "import bench\n"
        "bench.__file__ = %r\n"                      # spawn_ranks starts `python <bench.__file__> argv`: point it at the stand-in worker
        "sys.exit(bench.spawn_ranks(8, []))\n" % (ROOT, str(script)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    if fail_rank is not None:
        env["VS_TEST_FAIL_RANK"] = str(fail_rank)
    return subprocess.run([sys.executable, str(driver)], env=env, capture_output=True, text=True, timeout=300)


def test_eight_ranks_meet_before_the_first_real_eight_gpu_run(tmp_path):
    """The driver's 8-GPU run must not be the first time eight ranks meet.  The GPU box allows six GPU processes, so the eight-rank case
    runs here on the CPU: the launcher loop of bench.py (spawn_ranks), the gloo group, the clip split of BASELINE configs[3]
    (64 clips -> 8 per rank), barrier-bracketed timing, aggregate() and gather_seconds() with eight ranks, one JSON line from rank 0."""
    import json
    out = _launch_eight(tmp_path)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1                                           # (gloo's own banner is kept off stdout by bench.py itself: tests/test_bench_gpu.py)
    j = json.loads(lines[0])
    assert j["clips_per_rank"] == [8] * 8 and j["mine"] == list(range(0, 64, 8))
    assert j["frames"] == 64 * 120 and j["aligned"] == 64 * 119
    assert len(j["per_rank"]) == 8 and all(x > 0 for x in j["per_rank"]) and abs(max(j["per_rank"]) - j["secs"]) < 1e-12


def test_a_failing_rank_ends_the_eight_rank_job_with_its_exit_code(tmp_path):
    """rank 5 dies after the process group is up; ranks 0-4, 6, 7 are waiting in a barrier that will never complete: the launcher stops
    them and the job's exit code is rank 5's -- within seconds, not after a collective's timeout"""
    import time
    t0 = time.perf_counter()
    out = _launch_eight(tmp_path, fail_rank=5)
    assert out.returncode == 7, (out.returncode, out.stderr[-1000:])
    assert "rank 5 ended with exit code 7" in out.stderr
    assert "{" not in out.stdout                                     # no half-finished report line
    assert time.perf_counter() - t0 < 120
