"""Shared helpers for the test-suite (fixtures are data only: tests/golden/*.npz)."""
import json
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
CKPT = os.path.join(ROOT, "ccsd_amd", "checkpoints")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def load_ckpt_np(name):
    """Neutral-format checkpoint -> (meta dict, {part: {key: torch tensor}})."""
    with open(os.path.join(CKPT, name + ".json")) as f:
        meta = json.load(f)
    z = np.load(os.path.join(CKPT, name + ".npz"))
    parts = {}
    for k in z.files:
        part, key = k.split("/", 1)
        # requires_grad mirrors nn.Parameter: ATen's linear() picks the same (non-XNNPACK) CPU kernel as
        # the reference's modules do, which makes the oracle bit-identical to it under torch.no_grad()
        parts.setdefault(part, {})[key] = torch.from_numpy(z[k]).requires_grad_(True)
    return meta, parts


def rng_matches(g):
    """The goldens regenerate inputs/noise from torch's CPU generator; check that this torch
    build reproduces the stream the fixtures were made with."""
    torch.manual_seed(int(g["seed"]))
    return np.array_equal(torch.randn(8).numpy(), g["rng_probe"])


def make_flags(B, N, counts):
    f = torch.zeros(B, N)
    for b in range(B):
        f[b, : counts[b % len(counts)]] = 1.0
    return f


def parse_case(case):
    """Name of a g5 sampler case -> (num_scales override or None, number of steps run or None = all): "k10" = a 10-scale SDE run to
    the end, "n1000_first3" = the checkpoint's 1000 scales, first 3 steps, "n1000" = the checkpoint's 1000 scales from the prior to
    the last step."""
    if case.startswith("k"):
        return int(case[1:]), None
    if "first" in case:
        return None, int(case.split("first")[1])
    return None, None
