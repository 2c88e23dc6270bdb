"""Regenerate ccsd_amd/csrc/ccsd_baked_qm9.h: the plan (PlanD) of the qm9_CC configuration at batch 1024 as a compile-time constant.

    python tools/bake_plan.py            (CPU only: uses the host emulation of the kernel source, tests/emu)

k_xa<false, XA_BAKED9> reads every plan field from this constant (index arithmetic, loop bounds and LDS offsets fold into
immediates); the host selects that instance only for a plan whose architecture bytes (ccsd_plan_arch_bytes: the plan with the
weight-derived affine fold zeroed) equal the baked ones, so any other configuration, batch or planner version simply runs the
run-time-plan instances.  Re-run after changing PlanD or the planner (ccsd_plan.h); tests/test_gpu_parity.py checks that the
headline plan really selects the baked instance and that it agrees bit for bit with the run-time-plan instance.
The configuration is read from the shipped checkpoint's own config (tests/golden) with the sampler settings of
config/sample_qm9_CC.yaml (bench.py WORKLOADS["qm9_CC"])."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# (macro name, bench workload, header): the configurations whose k_xa / k_r2 instances have the plan compiled in
TARGETS = [("QM9", "qm9_CC", "ccsd_baked_qm9.h"), ("CS", "community_small_CC", "ccsd_baked_cs.h"), ("Z", "zinc250k", "ccsd_baked_z.h"),
           ("ENZ", "enzymes_small_CC", "ccsd_baked_enz.h")]
# (not qm9_Base_CC: with the plan baked hipcc unrolls the HodgeBaseline layers into 300 spilled VGPRs)

import bench  # noqa: E402
from ccsd_amd import loader  # noqa: E402
from ccsd_amd.engine import PCEngine  # noqa: E402
from tests.emu_util import emu_library  # noqa: E402
from tests.helpers import load_ckpt_np  # noqa: E402


def make_engine(workload, lib):
    """The engine (hence the plan) of a bench workload at its bench batch, on the host emulation."""
    wl = bench.WORKLOADS[workload]
    meta, parts = load_ckpt_np(wl["ckpt"])
    cfg, is_cc = meta["config"], meta["is_cc"]
    names = ["x", "adj"] + (["rank2"] if is_cc else [])
    sdes = [loader.load_sde(cfg["sde"][p]) for p in names]
    kw = dict(d_min=cfg["data"]["d_min"], d_max=cfg["data"]["d_max"]) if is_cc else {}
    return PCEngine(meta["params_x"], parts["x"], meta["params_adj"], parts["adj"], meta.get("params_rank2") if is_cc else None,
                    parts.get("rank2") if is_cc else None, N=cfg["data"]["max_node_num"], F=cfg["data"]["max_feat_num"], is_cc=is_cc, sdes=sdes,
                    predictor=wl["predictor"], corrector=wl["corrector"], snr=wl["snr"], scale_eps=wl["scale_eps"], n_steps=1, denoise=True,
                    eps=1e-4, device="cpu", batch_hint=wl["batch"], lib=lib, **kw)


def bake(name, workload, out_path, lib):
    os.environ["CCSD_DUMP_PLAN"] = out_path
    os.environ["CCSD_DUMP_PLAN_NAME"] = name
    try:
        eng = make_engine(workload, lib)
        del eng
    finally:
        os.environ.pop("CCSD_DUMP_PLAN", None)
        os.environ.pop("CCSD_DUMP_PLAN_NAME", None)


def main():
    lib = emu_library()
    for name, workload, header in TARGETS:
        out = os.path.join(ROOT, "ccsd_amd", "csrc", header)
        bake(name, workload, out + ".tmp", lib)
        os.replace(out + ".tmp", out)
        print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
