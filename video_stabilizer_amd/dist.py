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


# the process group (and device) the report collectives run on: None = the default group.  Set by init_with_fallback when RCCL
# came up on EVERY rank: the default group stays gloo (the side channel the ranks agree over), RCCL is a sub-group beside it.
_REPORT_GROUP = None
_REPORT_DEVICE = None


def report_device():
    """where aggregate() / gather_seconds() / barrier() place their tensors (the rank's GPU when they run over RCCL, else None)"""
    return _REPORT_DEVICE


def init_with_fallback(backend, rank, world, device_id=None, probe_seconds=120, _probe_backend=None, _fail_probe_on_rank=None):
    """"if RCCL init fails on the box, fall back to host-side aggregation and say so" -- decided by ALL ranks
    together, never by each rank for itself.

    1. Every rank joins a gloo group on the launcher's MASTER_PORT: the host-side channel, which needs no GPU and is what the
       report falls back to.  It is the default group for the rest of the run (no second rendezvous, no second port).
    2. Every rank tries RCCL as a sub-group (dist.new_group("nccl")) and proves it with one tiny all-reduce -- communicator creation
       and the first collective are where a broken RCCL setup shows.  Whatever happens is caught, per rank.
    3. The ranks all-reduce (MIN) their "RCCL ok" flags over gloo.  All ones: the report collectives (aggregate, gather_seconds,
       barrier) run over RCCL on every rank.  Anything else: over gloo on every rank, and every rank gets every failing rank's
       reason for the report line.
    A probe that HANGS (a peer never joins the communicator) must end in a Python exception on the waiting ranks, not in the NCCL
    watchdog's default action, which is to abort the process: the sub-group is therefore created with TORCH_NCCL_BLOCKING_WAIT=1 and
    TORCH_NCCL_ASYNC_ERROR_HANDLING=0 in the environment (set here, before the first NCCL group exists; a launcher's own setting
    wins), and the probe is an async all-reduce waited for with an explicit timeout (work.wait(timeout)), which raises in that mode.
    The waiting ranks then reach the flag all-reduce with "not ok" and the outcome is common to all.  What has been EXERCISED: RCCL up
    on one rank; both ranks of a two-rank job on one GPU refused by RCCL (symmetric failure -> gloo on both); a failure injected on one
    of three gloo ranks standing in for the probe (tests/test_dist_cpu.py).  An asymmetric failure of real RCCL across GPUs has not
    been available to test: if the flag all-reduce itself cannot complete (a rank died), the job ends with a non-zero exit code and
    the message below rather than a fallback.
    -> (dist, backend in use: "nccl" | "gloo", None or the reason).  (_probe_backend / _fail_probe_on_rank: test hooks.)"""
    global _REPORT_GROUP, _REPORT_DEVICE
    import datetime
    import torch
    import torch.distributed as dist
    _REPORT_GROUP, _REPORT_DEVICE = None, None
    if backend != "nccl":
        return init(backend, rank, world, device_id), backend, None
    # a timed-out wait raises (blocking-wait mode) instead of the watchdog aborting the process; read when the first NCCL group is made
    os.environ.setdefault("TORCH_NCCL_BLOCKING_WAIT", "1")
    os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "0")
    init("gloo", rank, world)
    why, group = None, None
    try:
        group = dist.new_group(backend=_probe_backend or "nccl", timeout=datetime.timedelta(seconds=probe_seconds))
        if _fail_probe_on_rank is not None and rank == _fail_probe_on_rank:
            raise RuntimeError("probe failure injected on rank %d" % rank)
        t = torch.ones(1, dtype=torch.int64, device=device_id if _probe_backend is None else None)
        work = dist.all_reduce(t, group=group, async_op=True)
        if not work.wait(datetime.timedelta(seconds=probe_seconds)):
            raise RuntimeError("probe all-reduce did not complete within %d s" % probe_seconds)
        if t.is_cuda:
            torch.cuda.synchronize()
        if int(t.item()) != world:
            raise RuntimeError("probe all-reduce returned %d for %d ranks" % (int(t.item()), world))
    except Exception as e:          # noqa: BLE001 -- whatever RCCL raises
        why = "rank %d: %s: %s" % (rank, type(e).__name__, str(e).splitlines()[0][:200] if str(e) else "")
    # ---- agreement over the host-side channel ----
    flag = torch.tensor([0 if why else 1], dtype=torch.int64)
    try:
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    except Exception as e:          # noqa: BLE001 -- a rank is gone: no common decision can be reached
        raise SystemExit("dist.init_with_fallback: rank %d could not agree with the other ranks over gloo (%s: %s); this rank's own RCCL probe: %s"
                         % (rank, type(e).__name__, str(e).splitlines()[0][:200] if str(e) else "", why or "ok"))
    if int(flag.item()) == 1:
        _REPORT_GROUP, _REPORT_DEVICE = group, (device_id if _probe_backend is None else None)
        return dist, "nccl", None
    reasons = [None] * world
    dist.all_gather_object(reasons, why)
    # every failing rank's reason (a rank that merely timed out waiting for a failed peer says so too)
    return dist, "gloo", "; ".join(r for r in reasons if r) or "unknown"


def barrier():
    """all ranks, over the report group (RCCL when it is up on every rank: a one-element all-reduce, then a device synchronise)"""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return
    if _REPORT_GROUP is None:
        dist.barrier()
        return
    t = torch.zeros(1, dtype=torch.int32, device=_REPORT_DEVICE)
    dist.all_reduce(t, group=_REPORT_GROUP)
    if t.is_cuda:
        torch.cuda.synchronize()


def aggregate(seconds, frames, aligned, device=None):
    """whole-job numbers: (max seconds over ranks, total frames, total aligned).  Works on any backend."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return float(seconds), int(frames), int(aligned)
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=device)
    c = torch.tensor([int(frames), int(aligned)], dtype=torch.int64, device=device)
    grp = _REPORT_GROUP if device is not None else None          # CPU tensors always travel over the default (gloo) group
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=grp)
    dist.all_reduce(c, op=dist.ReduceOp.SUM, group=grp)
    return float(t.item()), int(c[0].item()), int(c[1].item())


def gather_seconds(seconds, device=None):
    """every rank's seconds, in rank order (the per-rank view beside aggregate()'s maximum: a rank that lags shows up here)"""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return [float(seconds)]
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=device)
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t, group=_REPORT_GROUP if device is not None else None)
    return [float(x.item()) for x in out]


def roll_call(info):
    """every rank's `info` (a small dict), in rank order -- over the host-side channel (gloo), before anything is timed.  One rank: [info]."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return [info]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, info)
    return out


def shared_devices(infos, visible_devices):
    """ranks that sit on the SAME physical device although the node shows enough devices for one each: [(device key, [ranks])].
    A job with more ranks than visible devices (a rehearsal on a one-GPU box) is allowed to share."""
    if len(infos) > visible_devices:
        return []
    seen = {}
    for i in infos:
        seen.setdefault(i["device_key"], []).append(i["rank"])
    return [(k, r) for k, r in sorted(seen.items()) if len(r) > 1]
