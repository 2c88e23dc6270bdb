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


def _seam_configs(n_samples):
    """(config_train, config_module, config_sample) of the shipped qm9_CC sampling set-up, as load_sampling_fn takes them."""
    from tests.helpers import load_ckpt_np

    meta, parts = load_ckpt_np("ccsd_qm9_CC")
    sample = dict(n_samples=n_samples, probability_flow=False, noise_removal=True, eps=1e-4)
    module = dict(predictor="Reverse", corrector="Langevin", snr=0.2, scale_eps=0.7, n_steps=1)
    return meta, parts, meta["config"], module, sample


def _seam_run(flags, lib, sharded):
    """The same run through the drop-in seam: loader.load_sampling_fn in one process, distributed.load_sampling_fn_sharded
    (full flags in, gathered full batch out) under torch.distributed."""
    from ccsd_amd import distributed, loader

    meta, parts, cfgt, module, sample = _seam_configs(flags.shape[0])
    models = [loader.load_model_from_ckpt(meta[f"params_{p}"], parts[p], "cpu") for p in ("x", "adj", "rank2")]
    make = distributed.load_sampling_fn_sharded if sharded else loader.load_sampling_fn
    fn = make(cfgt, module, sample, "cpu", is_cc=True, d_min=3, d_max=9, rng="philox", seed=5, max_steps=STEPS, lib=lib)
    return fn(*models, flags)


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from ccsd_amd import distributed

    r, w, dev = distributed.init()            # gloo here (no GPU); "nccl" = RCCL on the GPU box
    assert (r, w, dev) == (rank, world, "cpu")
    torch.set_num_threads(2)
    lib = emu_library()
    g = load_golden("g5_ccsd_qm9_CC.npz")
    flags = torch.from_numpy(g["flags"])
    B = flags.shape[0] // world
    res = _run(flags[rank * B:(rank + 1) * B], lib, 5, rank * B, dist.group.WORLD, B)
    outs = distributed.all_gather_samples(res[:3])        # final sample collection
    seam = _seam_run(flags, lib, sharded=True)            # the same thing through load_sampling_fn_sharded
    if rank == 0:
        q.put([o.numpy() for o in outs] + [o.numpy() for o in seam[:3]])
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
    for name, a, b in zip(["x", "adj", "rank2"], single[:3], outs[:3]):
        pc.assert_close(torch.from_numpy(b), a, f"2-rank vs single {name}", rtol=2e-6)
    # ccsd_amd.distributed.load_sampling_fn_sharded: full flags in, full batch out on every rank, == one process
    seam_single = _seam_run(flags, lib, sharded=False)
    for name, a, b in zip(["x", "adj", "rank2"], seam_single[:3], outs[3:]):
        assert b.shape == tuple(a.shape)
        pc.assert_close(torch.from_numpy(b), a, f"sharded seam vs single-process seam {name}", rtol=2e-6)
