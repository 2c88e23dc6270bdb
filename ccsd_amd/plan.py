"""Model hyper-parameters -> ccsd_config_t, and state_dicts -> the canonical flat weight blob.

The blob order is the contract with ccsd_amd/csrc/ccsd_plan.h::ccsd_build_plan (the C side walks the
same sequence and checks the total length):

  ScoreNetworkX     layers.{l}.weight [in][nhid], layers.{l}.bias ; final.linears.{0,1,2}.{weight [out][in], bias}
  ScoreNetworkA(_CC) per AttentionLayer l: per input channel c: gnn_q.{weight,bias}, gnn_k.{weight,bias},
                    gnn_v.{weight,bias}; then mlp.*, multi_channel.*
                    per HodgeAdjAttentionLayer l: Wcat [K][cin*2*adim] (column c*2*adim + d: d < adim -> ccnn_q of
                    channel c, else ccnn_k), bcat; mlp_value.*; mlp_attention.*
                    final.linears.{0,1,2}
  ScoreNetworkF     layers.{l}.layer.* ; final.*
An MLP with one layer is stored by the reference as `linear`, otherwise as `linears.{i}` (layers.py:205-218).
State-dict key names are the reference's (checkpoint compatibility).
"""
from __future__ import annotations

from math import comb
from typing import Dict, List, Optional, Tuple

import numpy as np

from . import _lib

Shapes = List[Tuple[str, Tuple[int, ...]]]


def rank2_dim(N: int, d_min: int, d_max: int) -> Tuple[int, int]:
    """(rows, cols) of the rank-2 incidence matrix (reference cc_utils.py:281-283)."""
    return (N * (N - 1)) // 2, sum(comb(N, i) for i in range(d_min, d_max + 1))


def _mlp_shapes(prefix: str, n: int, din: int, hid: int, dout: int) -> Shapes:
    if n < 1:
        raise ValueError("Number of layers should be greater of equal to 1.")
    if n == 1:
        return [(f"{prefix}linear.weight", (dout, din)), (f"{prefix}linear.bias", (dout,))]
    out = []
    for i in range(n):
        a = din if i == 0 else hid
        b = dout if i == n - 1 else hid
        out += [(f"{prefix}linears.{i}.weight", (b, a)), (f"{prefix}linears.{i}.bias", (b,))]
    return out


def attn_layer_dims(p: dict):
    """(cin, cout, fin, adim, fout) per AttentionLayer (ScoreNetwork_A.py:398-443)."""
    L = p["num_layers"]
    dims = []
    for l in range(L):
        first, last = l == 0, (l == L - 1 and l != 0)
        dims.append((p["c_init"] if first else p["c_hid"], p["c_final"] if last else p["c_hid"],
                     p["max_feat_num"] if first else p["nhid"], p["nhid"] if first else p["adim"], p["nhid"]))
    return dims


def gmh_layer_dims(p: dict):
    """(cin, cout, fin, adim, fout) per AttentionLayer of ScoreNetworkX_GMH (ScoreNetwork_X.py:208-252)."""
    L = p["depth"]
    dims = []
    for l in range(L):
        first, last = l == 0, (l == L - 1 and l != 0)
        dims.append((p["c_init"] if first else p["c_hid"], p["c_final"] if last else p["c_hid"],
                     p["max_feat_num"] if first else p["nhid"], p["nhid"] if first else p["adim"], p["nhid"]))
    return dims


def _attn_layer_shapes(l: int, cin: int, cout: int, fin: int, ad: int, fo: int, num_linears: int, conv: str = "GCN") -> Shapes:
    s: Shapes = []
    for c in range(cin):
        for g, o in (("q", ad), ("k", ad), ("v", fo)):
            if conv == "MLP" and g != "v":      # attention.py:168-178: MLP(2, in_dim, 2 * attn_dim, attn_dim, tanh)
                s += _mlp_shapes(f"layers.{l}.attn.{c}.gnn_{g}.", 2, fin, 2 * ad, ad)
                continue
            s += [(f"layers.{l}.attn.{c}.gnn_{g}.weight", (fin, o)), (f"layers.{l}.attn.{c}.gnn_{g}.bias", (o,))]
    hid = 2 * max(cin, cout)
    s += _mlp_shapes(f"layers.{l}.mlp.", num_linears, 2 * cin, hid, cout)
    s += _mlp_shapes(f"layers.{l}.multi_channel.", 2, cin * fo, hid, fo)
    return s


def hodge_layer_dims(p: dict):
    """(cin, cout, adim) per HodgeAdjAttentionLayer (ScoreNetwork_A_CC.py:155-205)."""
    L = p["num_layers_h"]
    dims = []
    for l in range(L):
        first, last = l == 0, (l == L - 1 and l != 0)
        dims.append((p["c_init"] if first else p["c_hid_h"], p["c_final_h"] if last else p["c_hid_h"],
                     p["nhid_h"] if first else p["adim_h"]))
    return dims


def hodge_base_dims(p: dict):
    """(cin, cout, hidden) per HodgeBaselineLayer (ScoreNetwork_A_Base_CC.py:153-195)."""
    L = p["num_layers_h"]
    dims = []
    for l in range(L):
        first, last = l == 0, (l == L - 1 and l != 0)
        dims.append((p["c_init"] if first else p["c_hid_h"], p["c_final_h"] if last else p["c_hid_h"],
                     p["nhid_h"] if first else p["hidden_h"]))
    return dims


def fnet_layer_dims(p: dict):
    L = p["num_layers"]
    dims = []
    for l in range(L):
        first, last = l == 0, (l == L - 1 and l != 0)
        dims.append((p["cnum"] if first else p["c_hid"], p["c_final"] if last else p["c_hid"]))
    return dims


def state_dict_shapes(params: dict) -> Shapes:
    """Key names and shapes of a network's state_dict, in the reference's registration order."""
    t = params["model_type"]
    s: Shapes = []
    if t == "ScoreNetworkX":
        F, H = params["max_feat_num"], params["nhid"]
        for l in range(params["depth"]):
            s += [(f"layers.{l}.weight", (F if l == 0 else H, H)), (f"layers.{l}.bias", (H,))]
        fdim = F + params["depth"] * H
        s += _mlp_shapes("final.", 3, fdim, 2 * fdim, F)
    elif t == "ScoreNetworkX_GMH":
        F, H = params["max_feat_num"], params["nhid"]
        for l, dims in enumerate(gmh_layer_dims(params)):
            s += _attn_layer_shapes(l, *dims, params["num_linears"], params.get("conv", "GCN"))
        fdim = F + params["depth"] * H
        s += _mlp_shapes("final.", 3, fdim, 2 * fdim, F)
    elif t in ("ScoreNetworkA", "ScoreNetworkA_CC", "ScoreNetworkA_Base_CC"):
        for l, dims in enumerate(attn_layer_dims(params)):
            s += _attn_layer_shapes(l, *dims, params["num_linears"], params.get("conv", "GCN"))
        fdim = params["c_hid"] * (params["num_layers"] - 1) + params["c_final"] + params["c_init"]
        if t == "ScoreNetworkA_CC":
            _, K = rank2_dim(params["max_node_num"], params["d_min"], params["d_max"])
            for l, (cin, cout, ad) in enumerate(hodge_layer_dims(params)):
                for c in range(cin):
                    for g in ("q", "k"):
                        s += [(f"layers_hodge.{l}.attn.{c}.ccnn_{g}.weight", (K, ad)),
                              (f"layers_hodge.{l}.attn.{c}.ccnn_{g}.bias", (ad,))]
                hid = 2 * max(cin, cout)
                s += _mlp_shapes(f"layers_hodge.{l}.mlp_value.", params["num_linears_h"], cin, hid, 1)
                s += _mlp_shapes(f"layers_hodge.{l}.mlp_attention.", params["num_linears_h"], cin, hid, cout)
            fdim += params["c_hid_h"] * (params["num_layers_h"] - 1) + params["c_final_h"] + params["c_init"]
        if t == "ScoreNetworkA_Base_CC":
            E, _ = rank2_dim(params["max_node_num"], params["d_min"], params["d_max"])
            for l, (cin, cout, hd) in enumerate(hodge_base_dims(params)):
                for c in range(cin):
                    s += _mlp_shapes(f"layers_hodge.{l}.layers.{c}.mlp_layer.", 2, E, hd, E)
                hid = 2 * max(cin, cout)
                s += _mlp_shapes(f"layers_hodge.{l}.mlp_rank2.", params["num_linears_h"], cin, hid, 1)
                s += _mlp_shapes(f"layers_hodge.{l}.mlp_hodge.", params["num_linears_h"], cin, hid, cout)
            fdim += params["c_hid_h"] * (params["num_layers_h"] - 1) + params["c_final_h"] + params["c_init"]
        s += _mlp_shapes("final.", 3, fdim, 2 * fdim, 1)
    elif t == "ScoreNetworkF":
        for l, (cin, cout) in enumerate(fnet_layer_dims(params)):
            s += _mlp_shapes(f"layers.{l}.layer.", params["num_linears"], cin, params["nhid"], cout)
        fdim = params["c_hid"] * (params["num_layers"] - 1) + params["c_final"] + params["cnum"]
        s += _mlp_shapes("final.", params["num_layers_mlp"], fdim, 2 * fdim, 1)
    else:
        raise ValueError(
            f"Model Name <{t}> is unknown. Please select from [ScoreNetworkX, ScoreNetworkX_GMH, ScoreNetworkA, "
            "ScoreNetworkA_CC, ScoreNetworkA_Base_CC, ScoreNetworkF]")
    return s


_DUMMY_X = dict(model_type="ScoreNetworkX", depth=1, nhid=1, use_bn=False)
_DUMMY_A = dict(model_type="ScoreNetworkA", nhid=1, num_layers=1, num_linears=1, c_init=1, c_hid=1, c_final=1, adim=1,
                num_heads=1, conv="GCN", use_bn=False)
_DUMMY_F = dict(model_type="ScoreNetworkF", num_layers_mlp=1, num_layers=1, num_linears=1, nhid=1, c_hid=1, c_final=1,
                cnum=1, use_hodge_mask=True, use_bn=False)


def complete_params(px: Optional[dict], pa: Optional[dict], pf: Optional[dict], N: int, F: int, is_cc: bool,
                    d_min: int, d_max: int):
    """Fill in inert stand-ins for networks that are not part of a call (single-network forwards)."""
    if px is None:
        px = dict(_DUMMY_X, max_feat_num=F, is_cc=is_cc)
    if pa is None:
        pa = dict(_DUMMY_A, max_feat_num=F, max_node_num=N, is_cc=is_cc)
    if pf is None and is_cc:
        pf = dict(_DUMMY_F, max_node_num=N, d_min=d_min, d_max=d_max, is_cc=True)
    return px, pa, pf


def _check_supported(p: dict):
    """Constructor-time checks (the reference's constructors raise the same, attention.py:180, hodge_attention.py:181)."""
    if p.get("conv", "GCN") not in ("GCN", "MLP"):
        raise NotImplementedError(f"Convolution layer {p.get('conv')} not implemented.")
    if p.get("conv_hodge", "HCN") not in ("HCN", "MLP"):
        raise NotImplementedError(f"Convolution layer {p.get('conv_hodge')} not implemented.")


def reference_forward_error(p: Optional[dict], N: int, B: int = 1) -> Optional[Exception]:
    """The exception the reference's forward raises for this network (None: it runs) -- for the two switches no shipped config
    sets and whose reference implementation only type-checks on degenerate shapes (captured from the reference itself in
    tests/golden/reference_variant_status.json, tools/make_golden.py::reference_variant_status):

    use_bn=True   MLP puts BatchNorm1d(hidden) behind every hidden Linear (layers.py:219-224, 262-275; single-Linear MLPs have
                  none).  On the (B, N, hidden) activations of ScoreNetworkX's head, ScoreNetworkX_GMH's head and every
                  multi_channel MLP torch normalises over dim 1 = N and raises unless N == hidden; on the 4-D activations of
                  ScoreNetworkA*'s final MLP and of ScoreNetworkF's MLPs it raises "expected 2D or 3D input".
    conv_hodge="MLP"  HodgeAttention applies an MLP with input width K to the E x E hodge adjacency (hodge_attention.py:100-102,
                  235-241): a matmul shape error unless E == K.
    The shapes that do type-check (ScoreNetworkX with N == 2 (F + depth nhid); E == K) are not built: NotImplementedError."""
    if p is None:
        return None
    t = p.get("model_type")
    bn_rt = lambda hid: RuntimeError(f"running_mean should contain {N} elements not {hid}")
    bn_4d = ValueError("expected 2D or 3D input (got 4D input)")
    if p.get("use_bn", False):
        if t == "ScoreNetworkX":
            hid = 2 * (p["max_feat_num"] + p["depth"] * p["nhid"])
            return bn_rt(hid) if N != hid else NotImplementedError(
                "use_bn=True on ScoreNetworkX with N == 2 (F + depth nhid): BatchNorm1d over the node index is not built")
        if t in ("ScoreNetworkX_GMH", "ScoreNetworkA", "ScoreNetworkA_CC", "ScoreNetworkA_Base_CC"):
            dims = gmh_layer_dims(p) if t == "ScoreNetworkX_GMH" else attn_layer_dims(p)
            for cin, cout, *_ in dims:                       # multi_channel: MLP(2, ...) on (B, N, cin * fout)
                if N != 2 * max(cin, cout):
                    return bn_rt(2 * max(cin, cout))
            if t == "ScoreNetworkX_GMH":
                hid = 2 * (p["max_feat_num"] + p["depth"] * p["nhid"])
                return bn_rt(hid) if N != hid else NotImplementedError("use_bn=True on ScoreNetworkX_GMH is not built")
            return bn_4d                                     # final MLP (3 Linears) on (B, N, N, fdim)
        if t == "ScoreNetworkF" and (p["num_linears"] > 1 or p["num_layers_mlp"] > 1):
            return bn_4d                                     # HodgeNetworkLayer / final MLP on (B, E, K, C)
    if t == "ScoreNetworkA_CC" and p.get("conv_hodge", "HCN") == "MLP":
        E, K = rank2_dim(p["max_node_num"], p["d_min"], p["d_max"])
        if E != K:
            return RuntimeError(f"mat1 and mat2 shapes cannot be multiplied ({B * E}x{E} and {K}x{2 * p['nhid_h']})")
        return NotImplementedError('conv_hodge="MLP" with E == K (the only shape the reference\'s MLP over the hodge adjacency '
                                   "type-checks on) is not built")
    return None


def make_config(px: dict, pa: dict, pf: Optional[dict], *, predictor="Euler", corrector="None", snr=0.1, scale_eps=1.0,
                n_steps=1, probability_flow=False, denoise=True, diff_steps=1, batch_hint=0) -> _lib.Config:
    for p in (px, pa, pf):
        if p is not None:
            _check_supported(p)
            err = reference_forward_error(p, pa["max_node_num"], max(1, int(batch_hint)))
            if err is not None:
                raise err
    if px["model_type"] not in ("ScoreNetworkX", "ScoreNetworkX_GMH"):
        raise NotImplementedError(f"{px['model_type']} is not supported by the HIP path yet")
    if pa["model_type"] not in ("ScoreNetworkA", "ScoreNetworkA_CC", "ScoreNetworkA_Base_CC"):
        raise NotImplementedError(f"{pa['model_type']} is not supported by the HIP path yet")
    is_cc = pf is not None
    c = _lib.Config()
    c.abi_version = _lib.ABI_VERSION
    c.N, c.F, c.is_cc = pa["max_node_num"], px["max_feat_num"], int(is_cc)
    c.d_min, c.d_max = (pf["d_min"], pf["d_max"]) if is_cc else (0, 0)
    c.x_depth, c.x_nhid = px["depth"], px["nhid"]
    if px["model_type"] == "ScoreNetworkX_GMH":
        c.x_gmh = 1
        c.x_num_linears, c.x_c_init, c.x_c_hid, c.x_c_final = px["num_linears"], px["c_init"], px["c_hid"], px["c_final"]
        c.x_adim, c.x_num_heads = px["adim"], px.get("num_heads", 4)
        c.x_conv_mlp = int(px.get("conv", "GCN") == "MLP")
    c.a_conv_mlp = int(pa.get("conv", "GCN") == "MLP")
    c.a_num_layers, c.a_num_linears = pa["num_layers"], pa["num_linears"]
    c.a_c_init, c.a_c_hid, c.a_c_final = pa["c_init"], pa["c_hid"], pa["c_final"]
    c.a_nhid, c.a_adim, c.a_num_heads = pa["nhid"], pa["adim"], pa.get("num_heads", 4)
    c.a_is_cc_net = int(pa["model_type"] == "ScoreNetworkA_CC")
    if c.a_is_cc_net:
        if not pa.get("is_cc", True):
            raise ValueError("ScoreNetworkA_CC is only for combinatorial complexes")
        c.h_num_layers, c.h_num_linears = pa["num_layers_h"], pa["num_linears_h"]
        c.h_nhid, c.h_adim, c.h_c_hid, c.h_c_final = pa["nhid_h"], pa["adim_h"], pa["c_hid_h"], pa["c_final_h"]
        c.h_num_heads = pa.get("num_heads_h", 4)
    elif pa["model_type"] == "ScoreNetworkA_Base_CC":
        # a_is_cc_net = 2: HodgeBaselineLayer stack (h_adim carries hidden_h; heads unused)
        if not pa.get("is_cc", True):
            raise ValueError("ScoreNetworkA_Base_CC is only for combinatorial complexes")
        c.a_is_cc_net = 2
        c.h_num_layers, c.h_num_linears = pa["num_layers_h"], pa["num_linears_h"]
        c.h_nhid, c.h_adim, c.h_c_hid, c.h_c_final = pa["nhid_h"], pa["hidden_h"], pa["c_hid_h"], pa["c_final_h"]
        c.h_num_heads = 1
    if is_cc:
        if pf["model_type"] != "ScoreNetworkF":
            raise NotImplementedError(f"{pf['model_type']} is not supported by the HIP path yet")
        c.f_num_layers, c.f_num_linears, c.f_nhid = pf["num_layers"], pf["num_linears"], pf["nhid"]
        c.f_c_hid, c.f_c_final, c.f_cnum = pf["c_hid"], pf["c_final"], pf["cnum"]
        c.f_num_layers_mlp, c.f_use_hodge_mask = pf["num_layers_mlp"], int(pf.get("use_hodge_mask", True))
    preds = {"Euler": _lib.PRED_EULER, "Reverse": _lib.PRED_REVERSE, "S4": _lib.PRED_S4}
    corrs = {"None": _lib.CORR_NONE, "Langevin": _lib.CORR_LANGEVIN}
    if predictor not in preds:
        raise NotImplementedError(f"Predictor {predictor} not yet supported. Select from [Reverse, Euler].")
    if corrector not in corrs and predictor != "S4":
        raise NotImplementedError(f"Corrector {corrector} not yet supported. Select from [Langevin, None].")
    if predictor == "S4":
        corrector = "None"      # S4_solver ignores the corrector / n_steps / probability_flow knobs (solver.py:1221-1226)
        n_steps, probability_flow = 1, False
    c.predictor, c.corrector = preds[predictor], corrs[corrector]
    c.n_corr_steps, c.probability_flow, c.denoise = int(n_steps), int(probability_flow), int(denoise)
    c.snr, c.scale_eps, c.diff_steps = float(snr), float(scale_eps), int(diff_steps)
    c.batch_hint = int(batch_hint)
    return c


def _np(v) -> np.ndarray:
    if hasattr(v, "detach"):
        v = v.detach().cpu().numpy()
    return np.ascontiguousarray(v, dtype=np.float32)


def pack_weights(px: dict, sdx: Optional[Dict], pa: dict, sda: Optional[Dict], pf: Optional[dict],
                 sdf: Optional[Dict]) -> np.ndarray:
    """Flatten the three state_dicts in canonical order.  A missing state_dict packs zeros."""
    chunks: List[np.ndarray] = []

    def get(sd, key, shape):
        if sd is None:
            return np.zeros(shape, np.float32)
        if key not in sd:
            raise ValueError(f"missing key {key} in state_dict")
        a = _np(sd[key])
        if tuple(a.shape) != tuple(shape):
            raise ValueError(f"size mismatch for {key}: checkpoint {tuple(a.shape)} vs model {tuple(shape)}")
        return a

    def strip(sd):
        if sd is None:
            return None
        return {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}

    sdx, sda, sdf = strip(sdx), strip(sda), strip(sdf)
    if px["model_type"] == "ScoreNetworkX_GMH":
        # AttentionLayers in the A-network's block order (per channel q, k, v; then mlp, multi_channel), then final
        shapes_x = dict(state_dict_shapes(px))
        for l, (cin, cout, fin, ad, fo) in enumerate(gmh_layer_dims(px)):
            for c in range(cin):
                for g in ("q", "k", "v"):
                    for k, shp in shapes_x.items():
                        if k.startswith(f"layers.{l}.attn.{c}.gnn_{g}."):
                            chunks.append(get(sdx, k, shp).ravel())
            for pre in (f"layers.{l}.mlp.", f"layers.{l}.multi_channel."):
                for k, shp in shapes_x.items():
                    if k.startswith(pre):
                        chunks.append(get(sdx, k, shp).ravel())
        for k, shp in shapes_x.items():
            if k.startswith("final."):
                chunks.append(get(sdx, k, shp).ravel())
    else:
        for key, shape in state_dict_shapes(px):
            chunks.append(get(sdx, key, shape).ravel())
    # A-network: everything in registration order except that the hodge q/k weights are concatenated
    shapes_a = dict(state_dict_shapes(pa))
    for l, (cin, cout, fin, ad, fo) in enumerate(attn_layer_dims(pa)):
        for c in range(cin):
            for g in ("q", "k", "v"):
                for k, shp in shapes_a.items():
                    if k.startswith(f"layers.{l}.attn.{c}.gnn_{g}."):
                        chunks.append(get(sda, k, shp).ravel())
        for pre in (f"layers.{l}.mlp.", f"layers.{l}.multi_channel."):
            for k, shp in shapes_a.items():
                if k.startswith(pre):
                    chunks.append(get(sda, k, shp).ravel())
    if pa["model_type"] == "ScoreNetworkA_CC":
        for l, (cin, cout, ad) in enumerate(hodge_layer_dims(pa)):
            ws, bs = [], []
            for c in range(cin):
                for g in ("q", "k"):
                    kw, kb = f"layers_hodge.{l}.attn.{c}.ccnn_{g}.weight", f"layers_hodge.{l}.attn.{c}.ccnn_{g}.bias"
                    ws.append(get(sda, kw, shapes_a[kw]))
                    bs.append(get(sda, kb, shapes_a[kb]))
            chunks.append(np.concatenate(ws, axis=1).ravel())   # [K][cin*2*ad]
            chunks.append(np.concatenate(bs).ravel())
            for pre in (f"layers_hodge.{l}.mlp_value.", f"layers_hodge.{l}.mlp_attention."):
                for k, shp in shapes_a.items():
                    if k.startswith(pre):
                        chunks.append(get(sda, k, shp).ravel())
    if pa["model_type"] == "ScoreNetworkA_Base_CC":
        # per layer: the BaselineBlocks' row MLPs channel by channel, then mlp_hodge.  mlp_rank2 only feeds the rank-2
        # output of the layer, which never reaches the score (ScoreNetwork_A_Base_CC.py:303-321): checked, not packed.
        for l, (cin, cout, hd) in enumerate(hodge_base_dims(pa)):
            for pre in [f"layers_hodge.{l}.layers.{c}.mlp_layer." for c in range(cin)] + [f"layers_hodge.{l}.mlp_hodge."]:
                for k, shp in shapes_a.items():
                    if k.startswith(pre):
                        chunks.append(get(sda, k, shp).ravel())
            for k, shp in shapes_a.items():
                if k.startswith(f"layers_hodge.{l}.mlp_rank2."):
                    get(sda, k, shp)
    for k, shp in shapes_a.items():
        if k.startswith("final."):
            chunks.append(get(sda, k, shp).ravel())
    if pf is not None:
        for key, shape in state_dict_shapes(pf):
            chunks.append(get(sdf, key, shape).ravel())
    return np.ascontiguousarray(np.concatenate(chunks), dtype=np.float32)
