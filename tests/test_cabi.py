"""The C-ABI library: loads without a GPU, exports every symbol include/ccsd_hip.h declares, and its
host-only entry points (config validation, sizes) behave.  No compute calls here."""
import ctypes as C
import os
import re

import pytest

from ccsd_amd import _lib, plan
from tests.helpers import ROOT, load_ckpt_np


def header_functions():
    src = open(os.path.join(ROOT, "include", "ccsd_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ccsd_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as ge

    ge.build()
    return _lib.get_library()


def test_exports_match_header(lib):
    names = header_functions()
    assert names, "no functions parsed from the header"
    for n in names:
        assert hasattr(lib.c, n), f"libccsd_hip.so does not export {n}"
    assert set(names) == set(_lib.EXPORTS), (set(names) ^ set(_lib.EXPORTS))


def test_weight_count_and_dims_host_only(lib):
    meta, parts = load_ckpt_np("ccsd_qm9_CC")
    cfg = plan.make_config(meta["params_x"], meta["params_adj"], meta["params_rank2"], predictor="Reverse",
                           corrector="Langevin", snr=0.2, scale_eps=0.7, diff_steps=1000)
    blob = plan.pack_weights(meta["params_x"], parts["x"], meta["params_adj"], parts["adj"], meta["params_rank2"], parts["rank2"])
    assert lib.ccsd_weight_count(C.byref(cfg)) == blob.size == 42764      # 42,764 parameters
    E, K = C.c_int32(), C.c_int64()
    lib.ccsd_rank2_dims(C.byref(cfg), C.byref(E), C.byref(K))
    assert (E.value, K.value) == (36, 466)
    meta, parts = load_ckpt_np("ccsd_community_small_CC")
    cfg = plan.make_config(meta["params_x"], meta["params_adj"], meta["params_rank2"])
    assert lib.ccsd_weight_count(C.byref(cfg)) == 228764
    lib.ccsd_rank2_dims(C.byref(cfg), C.byref(E), C.byref(K))
    assert (E.value, K.value) == (190, 1140)


def test_invalid_configs_are_rejected(lib):
    meta, _ = load_ckpt_np("ccsd_qm9_CC")
    cfg = plan.make_config(meta["params_x"], meta["params_adj"], meta["params_rank2"])
    cfg.abi_version = 99
    assert lib.ccsd_weight_count(C.byref(cfg)) == 0
    assert b"abi_version" in lib.ccsd_last_error()
    cfg = plan.make_config(meta["params_x"], dict(meta["params_adj"], num_layers_h=9), meta["params_rank2"])
    assert lib.ccsd_weight_count(C.byref(cfg)) == 0           # 9 hodge layers: outside the HIP envelope (1..8 are built)
    assert b"HodgeAdjAttentionLayers" in lib.ccsd_last_error()
    cfg = plan.make_config(meta["params_x"], dict(meta["params_adj"], num_layers_h=3), meta["params_rank2"])
    assert lib.ccsd_weight_count(C.byref(cfg)) > 0
    cfg = plan.make_config(meta["params_x"], dict(meta["params_adj"], num_layers_h=3, num_linears_h=2), meta["params_rank2"])
    assert lib.ccsd_weight_count(C.byref(cfg)) > 0            # (true-MLP mlp_value behind dense hodge layers: the general hodge stack)
    # use_bn / conv_hodge="MLP": the reference's own forward fails on these shapes, with these exception types
    with pytest.raises(RuntimeError, match="running_mean should contain 9 elements not 48"):
        plan.make_config(dict(meta["params_x"], use_bn=True), meta["params_adj"], meta["params_rank2"])
    with pytest.raises(RuntimeError, match="mat1 and mat2 shapes cannot be multiplied"):
        plan.make_config(meta["params_x"], dict(meta["params_adj"], conv_hodge="MLP"), meta["params_rank2"])
    with pytest.raises(NotImplementedError):
        plan.make_config(meta["params_x"], dict(meta["params_adj"], conv_hodge="GAT"), meta["params_rank2"])
    # ... while use_bn on a network whose MLPs are single Linears creates no BatchNorm at all (layers.py:205-224): it runs
    assert plan.make_config(meta["params_x"], meta["params_adj"], dict(meta["params_rank2"], use_bn=True)).f_num_linears == 1
    with pytest.raises(NotImplementedError):
        plan.make_config(meta["params_x"], dict(meta["params_adj"], conv="GAT"), meta["params_rank2"])
    assert plan.make_config(meta["params_x"], dict(meta["params_adj"], conv="MLP"), meta["params_rank2"]).a_conv_mlp == 1
    with pytest.raises(NotImplementedError):
        plan.make_config(meta["params_x"], meta["params_adj"], meta["params_rank2"], predictor="Heun")


def test_product_path_has_no_cpu_fallback():
    """Without a GPU the engine must refuse to run rather than fall back to anything."""
    import torch

    from ccsd_amd.engine import PCEngine

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    meta, parts = load_ckpt_np("gdss_community_small")
    with pytest.raises(_lib.CcsdError):
        PCEngine(meta["params_x"], parts["x"], meta["params_adj"], parts["adj"], None, None, N=20, F=10, is_cc=False, device="cpu")
    with pytest.raises(_lib.CcsdError):
        PCEngine(meta["params_x"], parts["x"], meta["params_adj"], parts["adj"], None, None, N=20, F=10, is_cc=False, device="cuda")


def test_pack_weights_errors():
    meta, parts = load_ckpt_np("ccsd_qm9_CC")
    bad = dict(parts["x"])
    bad.pop("final.linears.0.bias")
    with pytest.raises(ValueError):
        plan.pack_weights(meta["params_x"], bad, meta["params_adj"], parts["adj"], meta["params_rank2"], parts["rank2"])


def test_unshipped_switches_fail_like_the_reference():
    """use_bn=True and conv_hodge="MLP": tests/golden/reference_variant_status.json holds what the REFERENCE's forward does for a
    set of configurations (captured by tools/make_golden.py): the product raises the same exception type with the same
    message where the reference raises, and says NotImplementedError for the two degenerate shapes on which the reference's
    code happens to type-check (N == hidden width; E == K)."""
    import json

    st = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_variant_status.json")))
    nodes = {"bn_x_N5_hidden22": 5, "bn_x_N8_hidden8": 8, "bn_x_gmh": 8}
    assert len(st) >= 8
    for tag, v in st.items():
        p = v["params"]
        err = plan.reference_forward_error(p, p.get("max_node_num") or nodes[tag], 2)
        if v["result"] == "error":
            assert type(err).__name__ == v["type"] and str(err) == v["message"], (tag, err, v)
        else:
            assert isinstance(err, NotImplementedError), (tag, err)
    # the model containers construct (as the reference's modules do) and fail at the first forward / sampler build
    from ccsd_amd import loader

    m = loader.load_model(dict(st["bn_f"]["params"]))
    assert m.params["use_bn"] is True
