"""Multi-GPU plumbing: clips are independent units, so ranks share NOTHING on the data path.
Clip i goes to rank i mod world (static round robin); the only collectives are the barrier that brackets the
timed region and one tiny all-reduce (max of seconds, sums of frame counters) for the whole-job report --
RCCL over xGMI on the GPU box (backend "nccl"), gloo in the CPU tests."""
import os


def env_world():
    return int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))


def shard_clips(n_clips, rank, world):
    """clip indices owned by `rank`: i mod world == rank"""
    return list(range(rank, n_clips, world))


def init(backend, rank, world, device_id=None):
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    kw = {}
    if device_id is not None:
        kw["device_id"] = device_id
    dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return dist


def init_with_fallback(backend, rank, world, device_id=None, probe_seconds=300):
    """"if RCCL init fails on the box, fall back to host-side aggregation and say so".  Initialises `backend`
    and proves it with one tiny all-reduce (communicator creation and the first collective are where a broken RCCL setup shows,
    on every rank alike); if either raises, the group is torn down and re-created on gloo (CPU tensors, next port), and the
    reason is returned for the report.  -> (dist, backend in use, None or the reason)"""
    import datetime
    import torch
    import torch.distributed as dist
    if backend != "nccl":
        return init(backend, rank, world, device_id), backend, None
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    why = None
    try:
        kw = {"device_id": device_id} if device_id is not None else {}
        dist.init_process_group("nccl", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=probe_seconds), **kw)
        t = torch.ones(1, dtype=torch.int64, device=device_id)
        dist.all_reduce(t)
        torch.cuda.synchronize()
        if int(t.item()) != world:
            raise RuntimeError("probe all-reduce returned %d for %d ranks" % (int(t.item()), world))
        return dist, "nccl", None
    except Exception as e:          # noqa: BLE001 -- whatever RCCL raises
        why = "%s: %s" % (type(e).__name__, str(e).splitlines()[0][:200] if str(e) else "")
        try:
            if dist.is_initialized():
                dist.destroy_process_group()
        except Exception:           # noqa: BLE001
            pass
    os.environ["MASTER_PORT"] = str(int(os.environ["MASTER_PORT"]) + 1)      # the first store may still hold its port
    dist.init_process_group("gloo", rank=rank, world_size=world)
    return dist, "gloo", why


def aggregate(seconds, frames, aligned, device=None):
    """whole-job numbers: (max seconds over ranks, total frames, total aligned).  Works on any backend."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return float(seconds), int(frames), int(aligned)
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=device)
    c = torch.tensor([int(frames), int(aligned)], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(c, op=dist.ReduceOp.SUM)
    return float(t.item()), int(c[0].item()), int(c[1].item())


def gather_seconds(seconds, device=None):
    """every rank's seconds, in rank order (the per-rank view beside aggregate()'s maximum: a rank that lags shows up here)"""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return [float(seconds)]
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=device)
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [float(x.item()) for x in out]
