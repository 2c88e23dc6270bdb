"""The drop-in seam accepts the reference's own model objects ("Models are nn.Module (possibly
DataParallel)"): get_pc_sampler reads hyper-parameters and weights through models.params_from_module + state_dict(), the
pair the reference's load_model_from_ckpt builds a model from (ccsd/src/utils/loader.py:619-657).

* duck-typed stand-ins (always run): plain objects that carry exactly the attributes the reference constructors store
  (ScoreNetwork_X.py:47-51, ScoreNetwork_A_CC.py:83-106, ScoreNetwork_F.py:64-76) and a state_dict(), inside a
  DataParallel-like wrapper;
* the real thing (build container only, skipped where /root/reference is absent): modules built by the reference's own
  load_model_from_ckpt, handed unchanged to this package's get_pc_sampler.
Both run the product code over the host emulation of the kernels and must reproduce the reference goldens."""
import json
import os
import sys

import numpy as np
import pytest
import torch

from ccsd_amd import loader, models, solver
from ccsd_amd.plan import rank2_dim
from tests import parity_cases as pc
from tests.emu_util import emu_library
from tests.helpers import ROOT, load_ckpt_np, load_golden, rng_matches

torch.set_num_threads(8)


class _Wrapped:
    """What torch.nn.DataParallel looks like to the seam: the model sits in `.module`, state_dict keys get a prefix."""

    def __init__(self, module):
        self.module = module

    def eval(self):
        self.module.eval()
        return self

    def state_dict(self):
        return {"module." + k: v for k, v in self.module.state_dict().items()}


def _stand_in(params, weights):
    """An object named like the reference class, with the attributes its constructor stores and nothing else."""
    # keyword defaults of the reference constructors (checkpoints only record what the training config spelled out)
    defaults = dict(use_bn=False, is_cc=False, num_heads=4, num_heads_h=4, conv="GCN", conv_hodge="HCN", use_hodge_mask=True)
    attrs = {a: params.get(k, defaults.get(k)) for k, a in models._ATTRS[params["model_type"]].items()}
    assert None not in attrs.values(), attrs

    def state_dict(self):
        return dict(weights)

    cls = type(params["model_type"], (), {"state_dict": state_dict, "eval": lambda self: self})
    obj = cls()
    for a, v in attrs.items():
        setattr(obj, a, v)
    return obj


def _run_against_golden(gname, ckpt, case, make_models):
    g = load_golden(f"g5_{gname}.npz")
    assert rng_matches(g)
    meta, parts = load_ckpt_np(ckpt)
    fn, _, flags, names = pc.sampler_from_golden(g, ckpt, case, emu_library(), "cpu")
    ms = make_models(meta, parts, names)
    torch.manual_seed(int(g["seed"]))
    res = fn(*ms, flags)
    for p, v in zip(names, res):
        pc.assert_close(v, g[f"{case}/{p}"], f"{gname} {case} {p} (foreign model objects)")


def test_params_from_module_round_trip():
    for name in ("ccsd_qm9_CC", "ccsd_qm9_Base_CC", "gdss_community_small"):
        meta, parts = load_ckpt_np(name)
        for p in ("x", "adj") + (("rank2",) if meta["is_cc"] else ()):
            want = dict(meta[f"params_{p}"])
            got = models.params_from_module(_Wrapped(_stand_in(want, parts[p])))
            assert {k: got[k] for k in want} == want, (name, p)
            # this package's own containers pass through
            assert models.params_from_module(loader.load_model(want))["model_type"] == want["model_type"]
    with pytest.raises(ValueError):
        models.params_from_module(object())


def test_gmh_hyper_parameters_from_state_dict_shapes():
    """ScoreNetworkX_GMH keeps no hyper-parameter attributes but depth / c_init (ScoreNetwork_X.py:198-201): the rest is
    recovered from the weight shapes."""
    g = load_golden("kat_gmh_models.npz")
    meta = json.loads(str(g["meta"]))
    for tag in ("small", "wide", "mlpconv_x"):
        want = meta[tag]
        sd = {k[len(tag) + 3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith(f"{tag}/w/")}
        obj = type("ScoreNetworkX_GMH", (), {"state_dict": lambda self, sd=sd: sd, "eval": lambda self: self})()
        obj.depth, obj.c_init, obj.use_bn, obj.is_cc = want["depth"], want["c_init"], want["use_bn"], want["is_cc"]
        got = models.params_from_module(obj)
        got["num_heads"] = want["num_heads"]          # kept by the Attention sub-modules only; 4 (the default) otherwise
        assert {k: got[k] for k in want} == want, tag


def test_sampler_accepts_duck_typed_reference_modules():
    _run_against_golden("ccsd_qm9_CC", "ccsd_qm9_CC", "k10",
                        lambda meta, parts, names: [_Wrapped(_stand_in(meta[f"params_{p}"], parts[p])) for p in names])
    _run_against_golden("gdss_community_small", "gdss_community_small", "n1000_first3",
                        lambda meta, parts, names: [_stand_in(meta[f"params_{p}"], parts[p]) for p in names])


@pytest.mark.skipif(not os.path.isdir("/root/reference/ccsd"), reason="the upstream reference is only present in the build container")
def test_sampler_accepts_the_reference_own_modules():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import refshim

    refshim.install()
    from ccsd.src.utils import cc_utils as ref_cc
    from ccsd.src.utils import loader as ref_loader

    def make(meta, parts, names):
        ref_cc.default_mask.cache_clear()      # see tools/make_golden.py::build_models
        ms = []
        for p in names:
            sd = {k: v.detach() for k, v in parts[p].items()}
            m = ref_loader.load_model_from_ckpt(refshim.EasyDict(meta[f"params_{p}"]), sd, "cpu")
            assert isinstance(m, torch.nn.Module) and not hasattr(m, "params")
            ms.append(m)
        ms[1] = torch.nn.DataParallel(ms[1])   # loader.py:649-650 wraps on multi-GPU hosts
        return ms

    _run_against_golden("ccsd_qm9_CC", "ccsd_qm9_CC", "k10", make)
    _run_against_golden("ccsd_qm9_Base_CC", "ccsd_qm9_Base_CC", "n1000_first3", make)
