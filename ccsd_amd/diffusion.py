"""CCSD: the reference's public entry class (ccsd/diffusion.py:27-200) for the sampling path.

    from ccsd_amd.diffusion import CCSD
    CCSD(type="sample", config="sample_qm9_CC", folder="./").run()

Same constructor arguments and the same YAML surface (<folder>/config/<config>.yaml with the keys `is_cc`, `data.*`,
`ckpt`, `sampler.{predictor,corrector,snr,scale_eps,n_steps}`, `sample.{divide_batch,n_samples,use_ema,noise_removal,
probability_flow,eps,seed}`; <folder>/config/general_config.yaml is read when present).  `type="train"` is outside this
build's scope (training is not on the sampling path) and raises NotImplementedError.

Several GPUs.  The reference's run() uses every visible GPU through DataParallel (loader.py:58-68, 134-135, 649-650).  Here
`run()` does the same with one process per GPU: when more than one GPU is visible (or `gpus=N` / CCSD_GPUS=N asks for N) and
the process is not already a rank of a torch.distributed job, it starts N worker processes (ccsd_amd.distributed.launch_workers:
torch.distributed.run on 127.0.0.1, started BEFORE this process makes any GPU call), every worker runs the same sampler over
its shard of each chunk (all-reduced Langevin norms = the statistics of the DataParallel run, samples all-gathered over RCCL
at the end of each chunk), rank 0 writes the result and `run()` returns it.  Started under torch.distributed.run
(`python -m torch.distributed.run --nproc-per-node N -m ccsd_amd.diffusion --config sample_qm9_CC`), every rank is such a
worker.  `gpus=1` keeps everything in this process.
"""
from __future__ import annotations

import json
import os
import sys
import tempfile
import time
from typing import Optional

import yaml

from .loader import AttrDict
from .sampler import Sampler, get_sampler_from_config


def get_config(config: str, seed: int, folder: str = "./") -> AttrDict:
    """ccsd/src/parsers/config.py:15-30."""
    with open(os.path.join(folder, "config", f"{config}.yaml")) as f:
        cfg = AttrDict(yaml.load(f, Loader=yaml.FullLoader))
    cfg.seed = seed
    return cfg


def get_general_config(folder: str = "./") -> AttrDict:
    """ccsd/src/parsers/config.py:33-45 (an absent file gives an empty config)."""
    path = os.path.join(folder, "config", "general_config.yaml")
    if not os.path.exists(path):
        return AttrDict({})
    with open(path) as f:
        return AttrDict(yaml.load(f, Loader=yaml.FullLoader) or {})


class CCSD:
    """CCSD class for sampling (training is out of scope here)."""

    def __init__(self, type: str, config: str, folder: str = "./", comment: str = "", seed: int = 42) -> None:
        assert type in ("train", "sample"), f"Unknown type: {type}. Please select from [train, sample]."
        if config[-5:] == ".yaml":
            config = config[:-5]
        assert os.path.exists(os.path.join(folder, "config", f"{config}.yaml")), f"Config {config} not found."
        self.type, self.config, self.folder, self.comment, self.seed = type, config, folder, comment, seed
        self.cfg = get_config(config, seed, folder)
        self.cfg.current_time = time.strftime("%b%d-%H-%M-%S", time.gmtime())     # get_time(), diffusion.py:71
        self.cfg.experiment_type = type
        self.cfg.config_name = config
        self.cfg.general_config = get_general_config(folder)
        self.cfg.folder = folder
        self.cfg.comment = comment
        self.sampler: Optional[Sampler] = None
        self.result = None

    def __repr__(self) -> str:
        return (f"{self.__class__.__name__}(type={self.type}, config={self.config}, comment={self.comment}, "
                f"seed={self.seed}, folder={self.folder})")

    def get_config(self) -> AttrDict:
        return self.cfg

    def get_sampler(self) -> Optional[Sampler]:
        return self.sampler

    def is_trained(self) -> bool:
        return False

    # -- several GPUs ------------------------------------------------------------------------------------------------
    def chunk_batch(self) -> Optional[int]:
        """Complexes per sampling_fn call (loader.py:387-416 + the divide_batch loop): what has to split over the ranks."""
        sample = self.cfg.get("sample", {})
        div = sample.get("divide_batch", 1) or 1
        if self.cfg["data"]["data"] in ("QM9", "ZINC250k"):
            n = sample.get("n_samples")
        else:                                    # generic datasets: the TRAINING config's batch size, stored in the checkpoint
            from . import loader

            n = loader.load_ckpt(self.cfg, "cpu", is_cc=bool(self.cfg.get("is_cc", False)))["config"]["data"]["batch_size"]
        return None if n is None else int(n) // int(div)

    def worker_count(self, gpus: Optional[int] = None) -> int:
        """How many rank processes run() starts: `gpus`, else CCSD_GPUS, else every visible GPU -- reduced to the largest count
        that divides the chunk batch; 1 inside a torch.distributed job (the process is a rank already) or without GPUs."""
        if int(os.environ.get("WORLD_SIZE", "1")) > 1:
            return 1
        if gpus is None and os.environ.get("CCSD_GPUS"):
            gpus = int(os.environ["CCSD_GPUS"])
        if gpus is None:
            import torch

            gpus = torch.cuda.device_count()     # (counting devices does not initialise the GPU: the workers must be started first)
        gpus = max(1, int(gpus))
        if gpus > 1:
            chunk = self.chunk_batch()
            while gpus > 1 and (chunk is None or chunk % gpus):
                gpus -= 1
        return gpus

    def run(self, gpus: Optional[int] = None, **sample_kw):
        """diffusion.py:100-168 for type == "sample": build the sampler from the config and sample (on `gpus` GPUs: see the
        module docstring; default = every visible GPU, like the reference)."""
        if self.type == "train":
            raise NotImplementedError("training is not part of this build (the reverse-SDE sampling path only)")
        n = self.worker_count(gpus)
        if n > 1:
            return self._run_workers(n, sample_kw)
        self.sampler = get_sampler_from_config(self.cfg)
        self.result = self.sampler.sample(**sample_kw)
        return self.result

    def _run_workers(self, n: int, sample_kw: dict, worker_args=()):
        """Start n rank processes (this process has made no GPU call), wait, and return rank 0's result as CPU tensors."""
        import numpy as np
        import torch

        from .distributed import launch_workers

        for k, v in sample_kw.items():
            if k not in ("save", "rounds") or not isinstance(v, (bool, int, type(None))):
                raise ValueError(f"run(gpus > 1) forwards save= and rounds= to the workers, not {k}=")
        with tempfile.TemporaryDirectory(prefix="ccsd_run_") as tmp:
            out = os.path.join(tmp, "result.npz")
            argv = ["-m", "ccsd_amd.diffusion", "--config", self.config, "--folder", os.path.abspath(self.folder), "--seed", str(self.seed),
                    "--comment", self.comment, "--out", out, "--sample-kw", json.dumps(sample_kw), *worker_args]
            env = dict(os.environ)
            root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
            env["PYTHONPATH"] = root + (os.pathsep + env["PYTHONPATH"] if env.get("PYTHONPATH") else "")
            rc = launch_workers(argv, n, env=env, relay=sys.stderr)
            if rc != 0 or not os.path.exists(out):
                raise RuntimeError(f"CCSD.run: the {n} sampling workers failed (exit code {rc})")
            with np.load(out) as z:
                self.result = {k: torch.from_numpy(z[k]) for k in z.files}
        return self.result


def _worker_main(argv=None) -> int:
    """One rank of CCSD.run(gpus=N) / of `torch.distributed.run -m ccsd_amd.diffusion`: the whole sampler over this rank's shard."""
    import argparse

    import numpy as np

    ap = argparse.ArgumentParser(prog="python -m ccsd_amd.diffusion")
    ap.add_argument("--config", required=True)
    ap.add_argument("--folder", default="./")
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--comment", default="")
    ap.add_argument("--out", default=None, help="rank 0 writes the result tensors here (.npz)")
    ap.add_argument("--sample-kw", default="{}")
    ap.add_argument("--emulate-steps", type=int, default=0,
                    help="TEST ONLY: run over the host emulation of the kernels on the CPU with gloo, this many PC steps (exercises the "
                         "launcher / sharding path on a GPU-less box)")
    a = ap.parse_args(argv)
    if a.emulate_steps:
        os.environ.setdefault("CUDA_VISIBLE_DEVICES", "")
    c = CCSD("sample", a.config, folder=a.folder, comment=a.comment, seed=a.seed)
    c.sampler = get_sampler_from_config(c.cfg)          # joins the process group (Sampler.__init__ -> distributed.init)
    if a.emulate_steps:
        from tests.emu_util import emu_library

        c.sampler.extra = dict(lib=emu_library(), max_steps=a.emulate_steps)
    res = c.sampler.sample(**json.loads(a.sample_kw))
    if c.sampler.rank == 0 and a.out:
        np.savez(a.out, **{k: v.detach().cpu().numpy() for k, v in res.items()})
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(_worker_main())
