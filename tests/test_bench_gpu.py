"""bench.py contract on the GPU box: one JSON line with the driver's keys, the roofline and cpu_baseline objects;
and the N > 1 flow rehearsed with two gloo ranks sharing the one GPU (the real run uses nccl = RCCL, one GPU per rank)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
        "dtype", "data", "config"}


def _last_json(out):
    return json.loads([l for l in out.strip().splitlines() if l.startswith("{")][-1])


def test_bench_line_contract(gpu_vs):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--frames", "8", "--steps", "2", "--warmup", "1"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    j = _last_json(out.stdout)
    assert KEYS <= set(j)
    assert j["metric"] == "aligned frames/sec" and j["unit"] == "frames/s" and j["n_gpus"] == 1 and j["steps"] == 2
    assert j["vs_baseline"] is None and j["dtype"] == "u8" and j["data"] == "synthetic" and j["scaling"] == "weak"
    assert "workload" in j["config"] and "model" not in j["config"]
    assert j["value"] > 0 and j["aligned_per_step"] == 7
    r = j["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    c = j["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
    assert j["align_only"]["value"] > j["value"]


def test_bench_two_ranks_gloo_rehearsal(gpu_vs):
    port = str(29700 + os.getpid() % 200)
    procs = []
    for r in range(2):
        env = dict(os.environ, WORLD_SIZE="2", RANK=str(r), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--frames", "6", "--steps", "2",
                                       "--warmup", "1", "--no-cpu-baseline", "--dist-backend", "gloo", "--device", "0"],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e[-2000:]
    j = _last_json(outs[0][0])
    assert j["n_gpus"] == 2 and j["aligned_per_step"] == 2 * 5      # both ranks' clips are counted
    assert "{" not in outs[1][0]                                    # only rank 0 prints
