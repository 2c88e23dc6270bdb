"""CCSD: the reference's public entry class (ccsd/diffusion.py:27-200) for the sampling path.

    from ccsd_amd.diffusion import CCSD
    CCSD(type="sample", config="sample_qm9_CC", folder="./").run()

Same constructor arguments and the same YAML surface (<folder>/config/<config>.yaml with the keys `is_cc`, `data.*`,
`ckpt`, `sampler.{predictor,corrector,snr,scale_eps,n_steps}`, `sample.{divide_batch,n_samples,use_ema,noise_removal,
probability_flow,eps,seed}`; <folder>/config/general_config.yaml is read when present).  `type="train"` is outside this
build's scope (training is not on the sampling path) and raises NotImplementedError.
"""
from __future__ import annotations

import os
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

    def run(self, **sample_kw):
        """diffusion.py:100-168 for type == "sample": build the sampler from the config and sample."""
        if self.type == "train":
            raise NotImplementedError("training is not part of this build (the reverse-SDE sampling path only)")
        self.sampler = get_sampler_from_config(self.cfg)
        self.result = self.sampler.sample(**sample_kw)
        return self.result
