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
print(json.dumps({"rank": rank, "clips": clips, "secs": secs, "frames": tot_f, "aligned": tot_a, "own": dt}))
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


def test_shard_covers_every_clip_once():
    from video_stabilizer_amd import dist as D
    for world in (1, 2, 4, 8):
        got = sorted(i for r in range(world) for i in D.shard_clips(64, r, world))
        assert got == list(range(64))
        assert max(len(D.shard_clips(64, r, world)) for r in range(world)) == 64 // world
