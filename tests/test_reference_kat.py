"""The known-answer vectors the REFERENCE's own tests hold for the path (tests/models/test_ScoreNetwork_A_CC.py:119-162,
test_ScoreNetwork_A_Base_CC.py:115-158, test_ScoreNetwork_F.py:69-106, test_hodge_attention.py:96-209, test_hodge_layers.py:143-375)
as a committed fixture (tests/golden/kat_reference_held.npz, tools/make_golden.py::kat_reference_held: each reference test was run
unmodified with a recorder around the class it builds): constructor arguments, the weights the reference constructor drew at
torch.manual_seed(42), the call's tensors, the full output, and the EXPECTED literals as the reference test writes them with the
output slice and atol (1e-4) it applies.  The oracle must meet every one; the three whole-network vectors are also checked on the
HIP path through the C ABI (the layer-level classes have no entry point of their own in the ABI: the oracle covers them)."""
import json

import numpy as np
import pytest
import torch

from oracle import ccsd_oracle as O
from tests.helpers import load_golden

DEV = "cuda:0"


def fixture():
    g = load_golden("kat_reference_held.npz")
    return g, json.loads(str(g["index"]))


def weights(g, key):
    pre = f"{key}/w/"
    return {k[len(pre):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(pre)}


def inputs(g, key, n):
    return [torch.from_numpy(g[f"{key}/in/{i}"]) if f"{key}/in/{i}" in g.files else None for i in range(n)]


def check_expected(g, key, entry, outs, names):
    """The reference test's own assertions: torch.allclose(<slice of an output>, expected literal, atol)."""
    env = dict(zip(names, outs))
    for expr, lit, atol in entry["checks"]:
        got = eval(expr, {}, env)                      # e.g. "out_rank2[0, 0]" -- an index expression on a named output
        want = torch.from_numpy(g[f"{key}/{lit}"])
        assert torch.allclose(got.cpu(), want, atol=atol), f"{entry['source']}: {expr} differs from the reference's {lit} by {(got.cpu() - want).abs().max():.2e}"


def check_full(g, key, outs, what, rtol=2e-5):
    for i, o in enumerate(outs):
        ref = torch.from_numpy(g[f"{key}/out/{i}"])
        err = (o.cpu() - ref).abs().max().item()
        assert err <= rtol * max(1.0, ref.abs().max().item()), f"{what} {key} output {i}: {err:.2e} from the reference's output"


NETWORKS = {"ScoreNetworkA_CC": "adj", "ScoreNetworkA_Base_CC": "adj", "ScoreNetworkF": "rank2"}


def net_params(entry):
    p = dict(entry["args"])
    p["model_type"] = entry["cls"]
    return p


@pytest.mark.parametrize("key", sorted(NETWORKS))
def test_oracle_meets_the_reference_held_network_vectors(key):
    g, idx = fixture()
    e = idx[key]
    x, adj, rank2 = inputs(g, key, 3)
    with torch.no_grad():
        out = O.run_network(net_params(e), weights(g, key), x, adj, rank2, None)
    check_expected(g, key, e, [out], ["out"])
    check_full(g, key, [out], "oracle")


def test_oracle_meets_the_reference_held_layer_vectors():
    g, idx = fixture()
    # DenseHCNConv (hodge_layers.py:163-199)
    e = idx["DenseHCNConv"]
    h, r = inputs(g, "DenseHCNConv", 2)
    with torch.no_grad():
        out = O.dense_hcn(weights(g, "DenseHCNConv"), h, r)
    check_expected(g, "DenseHCNConv", e, [out], ["out"])
    check_full(g, "DenseHCNConv", [out], "oracle")
    # HodgeAttention (hodge_attention.py:80-129): value = H . rank2, attention
    e = idx["HodgeAttention"]
    h, r = inputs(g, "HodgeAttention", 2)
    with torch.no_grad():
        val, att = O.hodge_attention(weights(g, "HodgeAttention"), h, r, e["args"]["num_heads"], e["args"].get("conv", "HCN"))
    check_expected(g, "HodgeAttention", e, [val, att], ["out_value", "out_attention"])
    check_full(g, "HodgeAttention", [val, att], "oracle")
    # HodgeAdjAttentionLayer (hodge_attention.py:290-325)
    e = idx["HodgeAdjAttentionLayer"]
    h, r = inputs(g, "HodgeAdjAttentionLayer", 2)
    a = e["args"]
    with torch.no_grad():
        oh, orr = O.hodge_adj_attention_layer(weights(g, "HodgeAdjAttentionLayer"), h, r, None, a["N"], a["d_min"], a["d_max"], a["num_heads"], a.get("conv", "HCN"))
    check_expected(g, "HodgeAdjAttentionLayer", e, [oh, orr], ["out_hodge_adj", "out_rank2"])
    check_full(g, "HodgeAdjAttentionLayer", [oh, orr], "oracle")
    # HodgeBaselineLayer (hodge_layers.py:385-416)
    e = idx["HodgeBaselineLayer"]
    h, r = inputs(g, "HodgeBaselineLayer", 2)
    a = e["args"]
    with torch.no_grad():
        oh, orr = O.hodge_baseline_layer(weights(g, "HodgeBaselineLayer"), h, r, None, a["N"], a["d_min"], a["d_max"])
    check_expected(g, "HodgeBaselineLayer", e, [oh, orr], ["out_hodge_adj", "out_rank2"])
    check_full(g, "HodgeBaselineLayer", [oh, orr], "oracle")


def test_fixture_holds_every_reference_known_answer_test_of_the_path():
    g, idx = fixture()
    assert sorted(idx) == ["BaselineBlock", "DenseHCNConv", "HodgeAdjAttentionLayer", "HodgeAttention", "HodgeBaselineLayer",
                           "HodgeNetworkLayer", "ScoreNetworkA_Base_CC", "ScoreNetworkA_CC", "ScoreNetworkF"]
    for key, e in idx.items():
        assert e["checks"] and all(atol == 1e-4 for _, _, atol in e["checks"])
        # the recorded reference outputs satisfy the literals themselves (the capture is the reference test's own run)
        outs = [torch.from_numpy(g[f"{key}/out/{i}"]) for i in range(e["n_out"])]
        names = sorted({c[0].split("[")[0] for c in e["checks"]}, key=[c[0].split("[")[0] for c in e["checks"]].index)
        if e["n_out"] == 2 and len(names) == 1:        # (one of two outputs is asserted on)
            continue
        check_expected(g, key, e, outs, names if len(names) == e["n_out"] else ["out"])


@pytest.mark.gpu
@pytest.mark.parametrize("key", sorted(NETWORKS))
def test_hip_path_meets_the_reference_held_network_vectors(key):
    from ccsd_amd import _lib
    from ccsd_amd.engine import PCEngine
    from tests.parity_cases import assert_close

    lib = _lib.get_library()
    g, idx = fixture()
    e = idx[key]
    x, adj, rank2 = (t.to(DEV) for t in inputs(g, key, 3))
    p = net_params(e)
    N, Fd = adj.shape[-1], x.shape[-1]
    flags = torch.ones(x.shape[0], N, device=DEV)           # flags=None in the reference = nothing masked (graph_utils.py:25-59)
    args = [None, None, None, None, None, None]
    slot = 1 if NETWORKS[key] == "adj" else 2
    args[2 * slot], args[2 * slot + 1] = p, weights(g, key)
    eng = PCEngine(*args, N=N, F=Fd, is_cc=True, d_min=p["d_min"], d_max=p["d_max"], device=DEV, lib=lib)
    out = eng.score(slot, x, adj, rank2, flags)
    check_expected(g, key, e, [out], ["out"])
    assert_close(out, g[f"{key}/out/0"], f"HIP {key} vs the reference's full output")
