"""Row "drop-in harness on N GPUs" (reference: DataParallel from CCSD.run(), loader.py:58-68, 134-135, 649-650; sampler.py:1185-1211).
CPU twins over the host emulation of the kernels + gloo: (1) two ranks running Sampler_mol_CC.sample() with divide_batch 2 give the
single-process harness output (exact mode: all-reduced Langevin norms, Philox keyed by the global sample index) and bit-equal
flags; (2) CCSD.run(gpus=2) starts its own workers and returns rank 0's result."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from ccsd_amd import sampler as S
from ccsd_amd.diffusion import CCSD
from tests.test_harness import QM9_CC_YAML, write_cfg

STEPS = 2
KEYS = ("x", "adj", "rank2", "flags", "adj_int", "rank2_int", "rank2_cell_count")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _harness(folder, save=False):
    from tests.emu_util import emu_library

    c = CCSD("sample", "sample_qm9_CC", folder=folder, seed=42)
    c.sampler = S.get_sampler_from_config(c.cfg)
    c.sampler.extra = dict(lib=emu_library(), max_steps=STEPS)
    return c.sampler.sample(save=save), c.sampler


def _worker(rank, world, port, folder, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      CUDA_VISIBLE_DEVICES="")
    torch.set_num_threads(2)
    out, smp = _harness(folder, save=True)
    assert (smp.rank, smp.world) == (rank, world) and hasattr(smp.sampling_fn, "local_batch")      # the sharded seam
    assert smp.sampling_fn.local_batch == 2           # n_samples 8 / divide_batch 2 / 2 ranks
    if rank == 0:
        q.put({k: out[k].numpy() for k in KEYS})
    import torch.distributed as dist

    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_harness_matches_single_process(tmp_path):
    from tests.emu_util import emu_library

    emu_library()                                     # build once in the parent
    write_cfg(tmp_path, "sample_qm9_CC", QM9_CC_YAML)
    single, smp = _harness(str(tmp_path))
    assert smp.world == 1 and not hasattr(smp.sampling_fn, "local_batch")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, str(tmp_path), q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert np.array_equal(got["flags"], single["flags"].numpy()), "every rank must draw the single-process init_flags"
    for k in ("x", "adj", "rank2"):
        ref = single[k].numpy()
        err = np.abs(got[k] - ref).max()
        assert err <= 2e-6 * max(1.0, np.abs(ref).max()), f"{k}: sharded harness differs from the single-process one by {err}"
    for k in ("adj_int", "rank2_int", "rank2_cell_count"):
        assert np.array_equal(got[k], single[k].numpy()), k
    assert len(os.listdir(tmp_path / "samples")) == 1, "rank 0 alone writes the samples"


def test_ccsd_run_starts_its_own_workers(tmp_path):
    """CCSD.run(gpus=2): two fresh rank processes (torch.distributed.run on 127.0.0.1), rank 0's tensors come back."""
    write_cfg(tmp_path, "sample_qm9_CC", QM9_CC_YAML)
    c = CCSD("sample", "sample_qm9_CC", folder=str(tmp_path), seed=42)
    assert c.chunk_batch() == 4 and c.worker_count(gpus=2) == 2 and c.worker_count(gpus=3) == 2 and c.worker_count(gpus=1) == 1
    out = c._run_workers(2, {"save": False}, worker_args=["--emulate-steps", str(STEPS)])
    single, _ = _harness(str(tmp_path))
    assert torch.equal(out["flags"], single["flags"])
    for k in ("x", "adj", "rank2"):
        assert (out[k] - single[k]).abs().max() <= 2e-6 * max(1.0, single[k].abs().max().item()), k
    assert torch.equal(out["adj_int"], single["adj_int"])
    with pytest.raises(ValueError):
        c._run_workers(2, {"node_counts": [9, 9]})
