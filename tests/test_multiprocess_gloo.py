"""N > 1 path on CPU: two gloo ranks each drive a batch shard (through the host emulation of the kernels)
and all-reduce the six Langevin norm sums every corrector half-step; the union of the shards must equal the
single-process run of the whole batch (exact mode)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests import parity_cases as pc
from tests.emu_util import emu_library
from tests.helpers import load_golden

STEPS = 3


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(flags, lib, seed, off, group, Bshard):
    g = load_golden("g5_ccsd_qm9_CC.npz")
    fn, models, _, _ = pc.sampler_from_golden(g, "ccsd_qm9_CC", f"n1000_first{STEPS}", lib, "cpu", rng="philox", seed=seed,
                                              sample_offset=off, group=group, shape_override=Bshard)
    return fn(*models, flags)


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    lib = emu_library()
    g = load_golden("g5_ccsd_qm9_CC.npz")
    flags = torch.from_numpy(g["flags"])
    B = flags.shape[0] // world
    res = _run(flags[rank * B:(rank + 1) * B], lib, 5, rank * B, dist.group.WORLD, B)
    # final sample collection: all_gather of the shards
    outs = []
    for t in res[:3]:
        buf = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(buf, t)
        outs.append(torch.cat(buf, 0))
    if rank == 0:
        q.put([o.numpy() for o in outs])
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_exact_mode_matches_single_process():
    lib = emu_library()   # build once in the parent
    g = load_golden("g5_ccsd_qm9_CC.npz")
    flags = torch.from_numpy(g["flags"])
    single = _run(flags, lib, 5, 0, None, flags.shape[0])
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    outs = q.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for name, a, b in zip(["x", "adj", "rank2"], single[:3], outs):
        pc.assert_close(torch.from_numpy(b), a, f"2-rank vs single {name}", rtol=2e-6)
