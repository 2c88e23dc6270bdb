"""Multi-GPU sampling: one process per GPU, the batch dimension sharded over the ranks.

The reference's only multi-GPU mechanism is torch.nn.DataParallel around each score network
(ccsd/src/utils/loader.py:134-135, 649-650): every forward scatters the batch, replicates the module and
gathers the outputs on device 0, six times per PC step.  Here every rank owns a contiguous shard of the
batch and runs the whole reverse diffusion locally on its GPU; the only per-step traffic is (in exact mode)
one 6-float all-reduce of the Langevin norm sums per corrector half-step, and the samples are all-gathered
over RCCL / xGMI once at the end.

    # inside each rank (started by torch.distributed.run, or by `launch_workers` below)
    rank, world, device = distributed.init()
    sampling_fn = distributed.load_sampling_fn_sharded(configt, config.sampler, config.sample, device,
                                                       is_cc=True, d_min=3, d_max=9)
    x, adj, rank2, nfe, traj = sampling_fn(model_x, model_adj, model_rank2, init_flags)   # FULL batch in, FULL batch out

`init_flags` is the full-batch tensor on every rank (the harness draws it from the seeded numpy stream, so all
ranks hold the same one); the closure slices its shard, and returns the gathered full-batch tensors on every
rank -- the same contract as the reference's DataParallel call, so Sampler_*.sample() needs no change.

Modes:
  exact=True   the six norm sums are all-reduced, Philox is keyed by the global sample index: the sharded run
               reproduces the single-process run of the whole batch (tests/test_multiprocess_gloo.py);
  exact=False  per-shard Langevin norms, no per-step communication (what the reference's own `divide_batch`
               does to the statistics); bench.py measures this mode.
"""
from __future__ import annotations

import os
import socket
import sys
from typing import Callable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

from . import loader


def init(backend: Optional[str] = None, force_group: bool = False, timeout_s: Optional[float] = None) -> Tuple[int, int, str]:
    """Join the process group described by RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT (set by
    torch.distributed.run or by `launch_workers`).  Returns (rank, world, device).  backend: "nccl" (= RCCL on ROCm) when a GPU
    is visible, else "gloo".  A single process normally needs no group; `force_group` creates the 1-rank group anyway (the
    RCCL code path -- init with device_id, all-gather / all-reduce on device tensors -- is then exercised on one GPU)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    gpu = torch.cuda.is_available()
    device = f"cuda:{local}" if gpu else "cpu"
    if gpu:
        torch.cuda.set_device(local)
    if (world > 1 or force_group) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            os.environ.setdefault("MASTER_PORT", str(free_port()))
        be = backend or ("nccl" if gpu else "gloo")
        kw = {}
        if timeout_s is not None:
            import datetime

            kw["timeout"] = datetime.timedelta(seconds=timeout_s)
        if be == "nccl":
            dist.init_process_group(be, rank=rank, world_size=world, device_id=torch.device(device), **kw)
        else:
            dist.init_process_group(be, rank=rank, world_size=world, **kw)
    return rank, world, device


def all_gather_samples(tensors: Sequence[Optional[torch.Tensor]], group=None, force: bool = False) -> List[Optional[torch.Tensor]]:
    """Final sample collection: every rank contributes its shard (dim 0) and receives the whole batch.  A 1-rank group is a
    no-op unless `force` (which runs the collective anyway: the RCCL path on a single GPU)."""
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not force):
        return list(tensors)
    world = dist.get_world_size(group)
    out: List[Optional[torch.Tensor]] = []
    for t in tensors:
        if t is None:
            out.append(None)
            continue
        t = t.contiguous()
        full = torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        if t.is_cuda:
            dist.all_gather_into_tensor(full, t, group=group)
        else:                                                    # gloo: list form
            parts = list(full.chunk(world, dim=0))
            dist.all_gather(parts, t, group=group)
        out.append(full)
    return out


def load_sampling_fn_sharded(config_train, config_module, config_sample, device, is_cc: bool = False,
                             d_min: Optional[int] = None, d_max: Optional[int] = None, divide_batch: Optional[int] = None,
                             *, group=None, exact: bool = True, **extra) -> Callable:
    """loader.load_sampling_fn (ccsd/src/utils/loader.py:337-458) for a batch sharded over the ranks of `group`.
    The returned closure has the reference's signature; it takes the full-batch flags and returns full-batch tensors."""
    if not (dist.is_available() and dist.is_initialized()):
        return loader.load_sampling_fn(config_train, config_module, config_sample, device, is_cc, d_min, d_max, divide_batch, **extra)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    data = loader._get(config_train, "data")
    total = (loader._get(config_sample, "n_samples") if loader._get(data, "data") in ("QM9", "ZINC250k")
             else loader._get(data, "batch_size")) // (divide_batch or 1)
    if total % world:
        raise ValueError(f"batch of {total} does not split over {world} ranks")
    local = total // world
    base = int(extra.pop("sample_offset", 0))
    inner = loader.load_sampling_fn(config_train, config_module, config_sample, device, is_cc, d_min, d_max,
                                    (divide_batch or 1) * world, sample_offset=base + rank * local, call_stride=total,
                                    group=(group if group is not None else dist.group.WORLD) if exact else None, **extra)

    def sampling_fn(*args):
        models, flags = args[:-1], args[-1]
        if flags.shape[0] != total:
            raise ValueError(f"init_flags must hold the whole batch of {total}, got {flags.shape[0]}")
        res = inner(*models, flags[rank * local:(rank + 1) * local])
        nt = len(res) - 2
        if not sampling_fn.timing:
            return (*all_gather_samples(res[:nt], group), res[nt], res[nt + 1])
        # diagnostic split of the call: [local loop | final all-gather], device-synchronised on both sides
        import time

        cuda = res[0].is_cuda
        if cuda:
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        full = all_gather_samples(res[:nt], group)
        if cuda:
            torch.cuda.synchronize()
        sampling_fn.gather_seconds += time.perf_counter() - t0
        return (*full, res[nt], res[nt + 1])

    sampling_fn.local_batch, sampling_fn.inner = local, inner
    sampling_fn.timing, sampling_fn.gather_seconds = False, 0.0      # bench.py --gpus N: time the final all-gather separately
    return sampling_fn


# ---- launching: fresh worker processes, started before the parent touches a GPU ------------------------------------------
def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_workers(argv: Sequence[str], nproc: int, env: Optional[dict] = None, relay=sys.stdout) -> int:
    """Run `python <argv>` as `nproc` ranks of one node through torch.distributed.run (127.0.0.1 rendezvous) in a child
    process; returns the exit code.  Result lines of the workers (JSON objects, one per line) are relayed to `relay`,
    anything else they write to stdout (library banners such as gloo's connection messages) goes to stderr.  The caller
    must not have initialised a GPU: the workers are fresh processes, nothing is re-exec'ed."""
    import subprocess

    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), *argv]
    e = dict(os.environ if env is None else env)
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=e, text=True)
    for line in p.stdout:
        out = relay if line.lstrip().startswith("{") else sys.stderr
        out.write(line)
        out.flush()
    return p.wait()
