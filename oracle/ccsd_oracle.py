"""CPU ORACLE for the CCSD reverse-SDE predictor-corrector sampling path.

THIS FILE IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  It restates, on the CPU with
plain torch fp32 tensor ops, the algorithm of the upstream reference
(AdrienC21/CCSD v0.3.3) for the one hot path this repository accelerates.  Only
`tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import
it, and only as the checker / the timed CPU baseline.  The product path
(`ccsd_amd`) never imports it and fails loudly without its HIP library.

Parity status: PINNED.  tools/make_golden.py imports the real reference in the
build container and stores its outputs under tests/golden/*.npz; tests/test_oracle_*.py
require this restatement to reproduce them (bit-for-bit where the same ATen op
sequence is used, 1e-6 otherwise), together with the reference's own known-answer
vectors (reference tests/models/*.py, tests/utils/*.py).

Every function cites the reference file:line it follows (paths relative to the
reference root).  The code is a functional re-statement over a flat weight dict
(keys = the reference checkpoint's state_dict keys); it is not a copy of the
reference's nn.Module classes.
"""
from __future__ import annotations

import math
from functools import lru_cache
from itertools import combinations
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor
Weights = Dict[str, Tensor]


# --------------------------------------------------------------------------------------
# K1: cell / edge enumeration tables            (ccsd/src/utils/cc_utils.py:44-96, 268-283)
# --------------------------------------------------------------------------------------
@lru_cache(maxsize=None)
def cell_tables(N: int, d_min: int, d_max: int):
    """Enumeration contract of get_cells (cc_utils.py:72-94).

    cells: itertools.combinations(range(N), k) for k = d_min..d_max (lexicographic within k);
    edges: combinations(range(N), 2) == row-major triu_indices(N, N, 1).
    Returns (edge_index (E,2) int64, cell_incidence (K,N) bool).
    """
    edges = np.array(list(combinations(range(N), 2)), dtype=np.int64).reshape(-1, 2)
    cells = []
    for k in range(d_min, d_max + 1):
        cells.extend(combinations(range(N), k))
    inc = np.zeros((len(cells), N), dtype=bool)
    for c, nodes in enumerate(cells):
        inc[c, list(nodes)] = True
    return torch.from_numpy(edges), torch.from_numpy(inc)


def get_rank2_dim(N: int, d_min: int, d_max: int) -> Tuple[int, int]:
    """(rows, cols) of the rank-2 incidence matrix (cc_utils.py:281-283)."""
    return (N * (N - 1)) // 2, sum(math.comb(N, i) for i in range(d_min, d_max + 1))


# --------------------------------------------------------------------------------------
# K2-K4: masks                 (graph_utils.py:25-59; cc_utils.py:527-591, 1591-1641)
# --------------------------------------------------------------------------------------
def mask_x(x: Tensor, flags: Optional[Tensor]) -> Tensor:
    """graph_utils.py:35-37."""
    if flags is None:
        return x * torch.ones(x.shape[0], x.shape[1])[:, :, None]
    return x * flags[:, :, None]


def mask_adjs(adjs: Tensor, flags: Optional[Tensor]) -> Tensor:
    """graph_utils.py:52-59 (row mask then column mask; 3-D or 4-D)."""
    if flags is None:
        flags = torch.ones(adjs.shape[0], adjs.shape[-1])
    if adjs.dim() == 4:
        flags = flags.unsqueeze(1)
    adjs = adjs * flags.unsqueeze(-1)
    adjs = adjs * flags.unsqueeze(-2)
    return adjs


def rank2_flags(flags: Tensor, N: int, d_min: int, d_max: int) -> Tuple[Tensor, Tensor]:
    """get_rank2_flags (cc_utils.py:545-557) without the Python loop.

    flags_left[b,e] = 0 iff an endpoint of edge e has flag == 0; flags_right[b,c] = 0 iff a
    node of cell c has flag == 0 (the reference tests `flags == 0`, nothing else).
    """
    edges, inc = cell_tables(N, d_min, d_max)
    off = (flags == 0)
    fl = 1.0 - (off[:, edges[:, 0]] | off[:, edges[:, 1]]).to(torch.float32)
    fr = 1.0 - ((off.to(torch.float32) @ inc.to(torch.float32).t()) > 0).to(torch.float32)
    return fl, fr


def mask_rank2(rank2: Tensor, N: int, d_min: int, d_max: int, flags: Optional[Tensor]) -> Tensor:
    """cc_utils.py:581-591: (flags_left * rank2) * flags_right, 3-D or 4-D."""
    if flags is None:
        flags = torch.ones(rank2.shape[0], N)
    fl, fr = rank2_flags(flags, N, d_min, d_max)
    if rank2.dim() == 4:
        fl, fr = fl.unsqueeze(1), fr.unsqueeze(1)
    return fl.unsqueeze(-1) * rank2 * fr.unsqueeze(-2)


def hodge_adj_flags(flags: Tensor) -> Tensor:
    """get_hodge_adj_flags (cc_utils.py:1605-1612)."""
    edges, _ = cell_tables(flags.shape[1], 1, 1)
    off = (flags == 0)
    return 1.0 - (off[:, edges[:, 0]] | off[:, edges[:, 1]]).to(torch.float32)


def mask_hodge_adjs(h: Tensor, flags: Optional[Tensor]) -> Tensor:
    """cc_utils.py:1630-1641."""
    if flags is None:
        return h * 1.0
    fh = hodge_adj_flags(flags)
    if h.dim() == 4:
        fh = fh.unsqueeze(1)
    h = h * fh.unsqueeze(-1)
    h = h * fh.unsqueeze(-2)
    return h


def node_flags(adj: Tensor, eps: float = 1e-5) -> Tensor:
    """graph_utils.py:73-77."""
    flags = torch.abs(adj).sum(-1).gt(eps).to(dtype=torch.float32)
    if flags.dim() == 3:
        flags = flags[:, 0, :]
    return flags


# --------------------------------------------------------------------------------------
# K5: noise                                   (graph_utils.py:158-178; cc_utils.py:594-615)
# --------------------------------------------------------------------------------------
def sym_noise_from_raw(z: Tensor, flags: Optional[Tensor]) -> Tensor:
    """graph_utils.py:173-175 applied to an already drawn z = randn_like(adj)."""
    z = z.triu(1)
    z = z + z.transpose(-1, -2)
    return mask_adjs(z, flags)


class NoiseSource:
    """Draw order contract of the sampler: each call to `.draw(shape)` returns the next raw
    standard-normal tensor.  Default draws with torch.randn (== randn_like on CPU for the
    global generator); `Recorded` replays a list (used for injected-noise parity)."""

    def draw(self, like: Tensor) -> Tensor:
        return torch.randn_like(like)


class RecordedNoise(NoiseSource):
    def __init__(self, tensors: Sequence[Tensor]):
        self.tensors = list(tensors)
        self.i = 0

    def draw(self, like: Tensor) -> Tensor:
        z = self.tensors[self.i]
        self.i += 1
        assert z.shape == like.shape, (z.shape, like.shape)
        return z


class RecordingNoise(NoiseSource):
    """Draws from torch's global generator and keeps every draw."""

    def __init__(self):
        self.tensors: List[Tensor] = []

    def draw(self, like: Tensor) -> Tensor:
        z = torch.randn_like(like)
        self.tensors.append(z)
        return z


def gen_noise(x: Tensor, flags: Optional[Tensor], sym: bool, src: NoiseSource) -> Tensor:
    """graph_utils.py:171-178."""
    z = src.draw(x)
    return sym_noise_from_raw(z, flags) if sym else mask_x(z, flags)


def gen_noise_rank2(x: Tensor, N: int, d_min: int, d_max: int, flags: Optional[Tensor], src: NoiseSource) -> Tensor:
    """cc_utils.py:613-615."""
    return mask_rank2(src.draw(x), N, d_min, d_max, flags)


# --------------------------------------------------------------------------------------
# Q1/Q2: quantisation                                         (graph_utils.py:181-213)
# --------------------------------------------------------------------------------------
def quantize(t: Tensor, thr: float = 0.5) -> Tensor:
    """graph_utils.py:191."""
    return torch.where(t < thr, torch.zeros_like(t), torch.ones_like(t))


def quantize_mol(adjs) -> np.ndarray:
    """graph_utils.py:205-213 (thresholds 0.5 / 1.5 / 2.5, int64 output)."""
    a = adjs.detach().cpu().clone() if isinstance(adjs, torch.Tensor) else torch.tensor(adjs)
    out = torch.zeros_like(a)
    out[a >= 0.5] = 1
    out[a >= 1.5] = 2
    out[a >= 2.5] = 3
    return out.to(torch.int64).numpy()


# --------------------------------------------------------------------------------------
# A1/A2/A8/F1: small tensor utilities
# --------------------------------------------------------------------------------------
def pow_tensor(x: Tensor, cnum: int) -> Tensor:
    """graph_utils.py:285-292: [A, A^2, ...] by repeated bmm(x_, x)."""
    xs = [x.unsqueeze(1)]
    cur = x
    for _ in range(cnum - 1):
        cur = torch.bmm(cur, x)
        xs.append(cur.unsqueeze(1))
    return torch.cat(xs, dim=1)


def adj_to_hodgedual(adj: Tensor) -> Tensor:
    """cc_utils.py:1514-1538: upper-triangular entries -> diagonal (E,E) matrix."""
    N = adj.shape[-1]
    r, c = torch.triu_indices(N, N, offset=1)
    return torch.diag_embed(adj[..., r, c])


def hodgedual_to_adj(h: Tensor) -> Tensor:
    """cc_utils.py:1552-1588: only the diagonal is scattered back, symmetrically."""
    E = h.shape[-1]
    N = int((1 + np.sqrt(1 + 8 * E)) / 2)
    diag = h.diagonal(dim1=-2, dim2=-1)
    adj = torch.zeros(*h.shape[:-2], N, N)
    r, c = torch.triu_indices(N, N, offset=1)
    adj[..., r, c] = diag
    adj[..., c, r] = diag
    return adj


def pow_tensor_cc(x: Tensor, cnum: int, hodge_mask: Optional[Tensor]) -> Tensor:
    """cc_utils.py:961-979: H = (F F^T) * hodge_mask; channels [F, HF, H(HF), ...]."""
    H = x @ x.transpose(-1, -2)
    if hodge_mask is not None:
        H = H * hodge_mask
    xs = [x.unsqueeze(1)]
    cur = x
    for _ in range(cnum - 1):
        cur = torch.bmm(H, cur)
        xs.append(cur.unsqueeze(1))
    return torch.cat(xs, dim=1)


# --------------------------------------------------------------------------------------
# M1/G1/A5: layers                       (models/layers.py:115-158, 246-275; hodge_layers.py:163-199)
# --------------------------------------------------------------------------------------
def _sub(w: Weights, prefix: str) -> Weights:
    n = len(prefix)
    return {k[n:]: v for k, v in w.items() if k.startswith(prefix)}


def mlp(w: Weights, x: Tensor, act: Callable[[Tensor], Tensor]) -> Tensor:
    """layers.py:260-275 (use_bn=False path).  `w` holds either linear.* or linears.{i}.*"""
    if "linear.weight" in w:
        return F.linear(x, w["linear.weight"], w["linear.bias"])
    n = 0
    while f"linears.{n}.weight" in w:
        n += 1
    h = x
    for i in range(n - 1):
        h = act(F.linear(h, w[f"linears.{i}.weight"], w[f"linears.{i}.bias"]))
    return F.linear(h, w[f"linears.{n-1}.weight"], w[f"linears.{n-1}.bias"])


def dense_gcn(w: Weights, x: Tensor, adj: Tensor) -> Tensor:
    """layers.py:134-158 with add_loop=True, improved=False, mask=None."""
    adj = adj.clone()
    idx = torch.arange(adj.shape[-1])
    adj[:, idx, idx] = 1
    out = torch.matmul(x, w["weight"])
    dis = adj.sum(dim=-1).clamp(min=1).pow(-0.5)
    adj = dis.unsqueeze(-1) * adj * dis.unsqueeze(-2)
    out = torch.matmul(adj, out)
    return out + w["bias"]


def dense_hcn(w: Weights, hodge_adj: Tensor, rank2: Tensor) -> Tensor:
    """hodge_layers.py:185-199 (no self loops)."""
    out = torch.matmul(rank2, w["weight"])
    dis = hodge_adj.sum(dim=-1).clamp(min=1).pow(-0.5)
    hodge_adj = dis.unsqueeze(-1) * hodge_adj * dis.unsqueeze(-2)
    out = torch.matmul(hodge_adj, out)
    return out + w["bias"]


def _head_attention(Q: Tensor, K: Tensor, dim_split: int, scale_dim: int, like: Tensor) -> Tensor:
    """attention.py:111-130 / hodge_attention.py:108-127: chunk, tanh(QK^T/sqrt(d)), mean, symmetrise."""
    Q_ = torch.cat(Q.split(dim_split, 2), 0)
    K_ = torch.cat(K.split(dim_split, 2), 0)
    A = torch.tanh(Q_.bmm(K_.transpose(1, 2)) / math.sqrt(scale_dim))
    A = A.view(-1, *like.shape).mean(dim=0)
    return (A + A.transpose(-1, -2)) / 2


# --------------------------------------------------------------------------------------
# A3/A4: Attention + AttentionLayer                       (models/attention.py:84-132, 270-304)
# --------------------------------------------------------------------------------------
def attention(w: Weights, x: Tensor, adj: Tensor, num_heads: int, conv: str = "GCN") -> Tuple[Tensor, Tensor]:
    if conv == "GCN":
        Q = dense_gcn(_sub(w, "gnn_q."), x, adj)
        K = dense_gcn(_sub(w, "gnn_k."), x, adj)
    elif conv == "MLP":
        Q = mlp(_sub(w, "gnn_q."), x, torch.tanh)
        K = mlp(_sub(w, "gnn_k."), x, torch.tanh)
    else:
        raise NotImplementedError(f"Convolution layer {conv} not implemented.")
    wv = _sub(w, "gnn_v.")
    V = dense_gcn(wv, x, adj)
    attn_dim = Q.shape[-1]
    out_dim = wv["weight"].shape[1]
    A = _head_attention(Q, K, attn_dim // num_heads, out_dim, adj)
    return V, A


def attention_layer(w: Weights, x: Tensor, adj: Tensor, flags: Optional[Tensor], num_heads: int, conv: str = "GCN") -> Tuple[Tensor, Tensor]:
    cin = adj.shape[1]
    masks, xs = [], []
    for k in range(cin):
        v, a = attention(_sub(w, f"attn.{k}."), x, adj[:, k, :, :], num_heads, conv)
        masks.append(a.unsqueeze(-1))
        xs.append(v)
    x_out = torch.tanh(mask_x(mlp(_sub(w, "multi_channel."), torch.cat(xs, dim=-1), F.elu), flags))
    mlp_in = torch.cat([torch.cat(masks, dim=-1), adj.permute(0, 2, 3, 1)], dim=-1)
    shape = mlp_in.shape
    out = mlp(_sub(w, "mlp."), mlp_in.view(-1, shape[-1]), F.elu)
    _adj = out.view(shape[0], shape[1], shape[2], -1).permute(0, 3, 1, 2)
    _adj = _adj + _adj.transpose(-1, -2)
    return x_out, mask_adjs(_adj, flags)


# --------------------------------------------------------------------------------------
# A6/A7: HodgeAttention + HodgeAdjAttentionLayer          (models/hodge_attention.py:80-129, 290-325)
# --------------------------------------------------------------------------------------
def hodge_attention(w: Weights, hodge_adj: Tensor, rank2: Tensor, num_heads: int, conv: str = "HCN") -> Tuple[Tensor, Tensor]:
    if conv == "HCN":
        Q = dense_hcn(_sub(w, "ccnn_q."), hodge_adj, rank2)
        K = dense_hcn(_sub(w, "ccnn_k."), hodge_adj, rank2)
    elif conv == "MLP":
        Q = mlp(_sub(w, "ccnn_q."), hodge_adj, torch.tanh)
        K = mlp(_sub(w, "ccnn_k."), hodge_adj, torch.tanh)
    else:
        raise NotImplementedError(f"Convolution layer {conv} not implemented.")
    V = torch.bmm(hodge_adj, rank2)  # ccnn_v is Identity (hodge_attention.py:107,164)
    attn_dim = Q.shape[-1]
    Kdim = rank2.shape[-1]  # out_dim == K (hodge_attention.py:235-241)
    A = _head_attention(Q, K, attn_dim // num_heads, Kdim, hodge_adj)
    return V, A


def hodge_adj_attention_layer(w: Weights, hodge_adj: Tensor, rank2: Tensor, flags: Optional[Tensor],
                              N: int, d_min: int, d_max: int, num_heads: int, conv: str = "HCN") -> Tuple[Tensor, Tensor]:
    cin = hodge_adj.shape[1]
    vals, atts = [], []
    for k in range(cin):
        v, a = hodge_attention(_sub(w, f"attn.{k}."), hodge_adj[:, k, :, :], rank2, num_heads, conv)
        vals.append(v.unsqueeze(-1))
        atts.append(a.unsqueeze(-1))
    h = mask_hodge_adjs(mlp(_sub(w, "mlp_attention."), torch.cat(atts, dim=-1), F.elu).permute(0, 3, 1, 2), flags)
    h = torch.tanh(h)
    h = h + h.transpose(-1, -2)
    r = mlp(_sub(w, "mlp_value."), torch.cat(vals, dim=-1), F.elu).squeeze(-1)
    return h, mask_rank2(r, N, d_min, d_max, flags)


# --------------------------------------------------------------------------------------
# X1 / A9 / A9' / F2 / F3: the three score networks
# --------------------------------------------------------------------------------------
def _count(w: Weights, fmt: str) -> int:
    n = 0
    while any(k.startswith(fmt.format(n)) for k in w):
        n += 1
    return n


def score_network_x(w: Weights, x: Tensor, adj: Tensor, flags: Optional[Tensor]) -> Tensor:
    """ScoreNetwork_X.py:102-153 (forward_graph == forward_cc; rank2 ignored)."""
    depth = _count(w, "layers.{}.")
    xs = [x]
    for k in range(depth):
        x = torch.tanh(dense_gcn(_sub(w, f"layers.{k}."), x, adj))
        xs.append(x)
    cat = torch.cat(xs, dim=-1)
    out = mlp(_sub(w, "final."), cat, F.elu).view(adj.shape[0], adj.shape[1], -1)
    return mask_x(out, flags)


def score_network_x_gmh(w: Weights, p: dict, x: Tensor, adj: Tensor, flags: Optional[Tensor]) -> Tensor:
    """ScoreNetworkX_GMH.forward_graph (ScoreNetwork_X.py:290-318): AttentionLayers instead of GCN layers."""
    adjc = pow_tensor(adj, p["c_init"])
    x_list = [x]
    for k in range(p["depth"]):
        x, adjc = attention_layer(_sub(w, f"layers.{k}."), x, adjc, flags, p.get("num_heads", 4), p.get("conv", "GCN"))
        x = torch.tanh(x)
        x_list.append(x)
    xs = torch.cat(x_list, dim=-1)
    out = mlp(_sub(w, "final."), xs, F.elu).view(adj.shape[0], adj.shape[1], -1)
    return mask_x(out, flags)


def _nodiag_mask(n: int) -> Tensor:
    """default_mask (cc_utils.py:942)."""
    return torch.ones(n, n) - torch.eye(n)


def score_network_a(w: Weights, p: dict, x: Tensor, adj: Tensor, flags: Optional[Tensor]) -> Tensor:
    """ScoreNetwork_A.py:505-541 (graph-only A-network)."""
    adjc = pow_tensor(adj, p["c_init"])
    adj_list = [adjc]
    for k in range(p["num_layers"]):
        x, adjc = attention_layer(_sub(w, f"layers.{k}."), x, adjc, flags, p.get("num_heads", 4), p.get("conv", "GCN"))
        adj_list.append(adjc)
    adjs = torch.cat(adj_list, dim=1).permute(0, 2, 3, 1)
    score = mlp(_sub(w, "final."), adjs, F.elu).view(*adjs.shape[:-1])
    score = score * _nodiag_mask(adj.shape[-1]).unsqueeze(0)
    return mask_adjs(score, flags)


def score_network_a_cc(w: Weights, p: dict, x: Tensor, adj: Tensor, rank2: Tensor, flags: Optional[Tensor]) -> Tensor:
    """ScoreNetwork_A_CC.py:275-332."""
    N, d_min, d_max = p["max_node_num"], p["d_min"], p["d_max"]
    adjc = pow_tensor(adj, p["c_init"])
    hodge_adjc = adj_to_hodgedual(adjc)
    adj_list = [adjc]
    _x = x.clone()
    for k in range(p["num_layers"]):
        _x, adjc = attention_layer(_sub(w, f"layers.{k}."), _x, adjc, flags, p.get("num_heads", 4), p.get("conv", "GCN"))
        adj_list.append(adjc)
    hodge_list = [hodge_adjc]
    _r = rank2.clone()
    for k in range(p["num_layers_h"]):
        hodge_adjc, _r = hodge_adj_attention_layer(_sub(w, f"layers_hodge.{k}."), hodge_adjc, _r, flags,
                                                   N, d_min, d_max, p.get("num_heads_h", 4), p.get("conv_hodge", "HCN"))
        hodge_list.append(hodge_adjc)
    adjs = torch.cat(adj_list, dim=1).permute(0, 2, 3, 1)
    hodge = hodgedual_to_adj(torch.cat(hodge_list, dim=1)).permute(0, 2, 3, 1)
    out = torch.cat([adjs, hodge], dim=-1)
    score = mlp(_sub(w, "final."), out, F.elu).view(*adjs.shape[:-1])
    score = score * _nodiag_mask(N).unsqueeze(0)
    return mask_adjs(score, flags)


def hodge_baseline_layer(w: Weights, hodge_adj: Tensor, rank2: Tensor, flags: Optional[Tensor],
                         N: int, d_min: int, d_max: int) -> Tuple[Tensor, Tensor]:
    """HodgeBaselineLayer.forward (hodge_layers.py:385-416) over BaselineBlock.forward (hodge_layers.py:247-270)."""
    cin = hodge_adj.shape[1]
    r_list, h_list = [], []
    for c in range(cin):
        h = torch.tanh(mlp(_sub(w, f"layers.{c}.mlp_layer."), hodge_adj[:, c], F.elu))   # row-wise E -> hidden -> E
        r_list.append(torch.bmm(h, rank2).unsqueeze(-1))
        h_list.append(((h + h.transpose(-1, -2)) / 2).unsqueeze(-1))
    h_out = mask_hodge_adjs(mlp(_sub(w, "mlp_hodge."), torch.cat(h_list, dim=-1), F.elu).permute(0, 3, 1, 2), flags)
    h_out = torch.tanh(h_out)
    h_out = h_out + h_out.transpose(-1, -2)
    r_out = mlp(_sub(w, "mlp_rank2."), torch.cat(r_list, dim=-1), F.elu).squeeze(-1)
    return h_out, mask_rank2(r_out, N, d_min, d_max, flags)


def score_network_a_base_cc(w: Weights, p: dict, x: Tensor, adj: Tensor, rank2: Tensor, flags: Optional[Tensor]) -> Tensor:
    """ScoreNetwork_A_Base_CC.py:266-323."""
    N, d_min, d_max = p["max_node_num"], p["d_min"], p["d_max"]
    adjc = pow_tensor(adj, p["c_init"])
    hodge_adjc = adj_to_hodgedual(adjc)
    adj_list = [adjc]
    _x = x.clone()
    for k in range(p["num_layers"]):
        _x, adjc = attention_layer(_sub(w, f"layers.{k}."), _x, adjc, flags, p.get("num_heads", 4), p.get("conv", "GCN"))
        adj_list.append(adjc)
    hodge_list = [hodge_adjc]
    _r = rank2.clone()
    for k in range(p["num_layers_h"]):
        hodge_adjc, _r = hodge_baseline_layer(_sub(w, f"layers_hodge.{k}."), hodge_adjc, _r, flags, N, d_min, d_max)
        hodge_list.append(hodge_adjc)
    adjs = torch.cat(adj_list, dim=1).permute(0, 2, 3, 1)
    hodge = hodgedual_to_adj(torch.cat(hodge_list, dim=1)).permute(0, 2, 3, 1)
    out = torch.cat([adjs, hodge], dim=-1)
    score = mlp(_sub(w, "final."), out, F.elu).view(*adjs.shape[:-1])
    score = score * _nodiag_mask(N).unsqueeze(0)
    return mask_adjs(score, flags)


def score_network_f(w: Weights, p: dict, x: Tensor, adj: Tensor, rank2: Tensor, flags: Optional[Tensor]) -> Tensor:
    """ScoreNetwork_F.py:175-217 (x, adj ignored); HodgeNetworkLayer hodge_layers.py:86-92."""
    N, d_min, d_max = p["max_node_num"], p["d_min"], p["d_max"]
    E = rank2.shape[-2]
    hmask = _nodiag_mask(E).unsqueeze(0) if p.get("use_hodge_mask", True) else torch.ones(1, E, E)
    rc = pow_tensor_cc(rank2, p["cnum"], hmask)
    lst = [rc]
    cur = rc.clone()
    for k in range(p["num_layers"]):
        cur = mlp(_sub(w, f"layers.{k}.layer."), cur.permute(0, 2, 3, 1), F.elu).permute(0, 3, 1, 2)
        cur = mask_rank2(cur, N, d_min, d_max, flags)
        lst.append(cur)
    cat = torch.cat(lst, dim=1).permute(0, 2, 3, 1)
    score = mlp(_sub(w, "final."), cat, F.elu).view(*cat.shape[:-1])
    score = score * torch.ones(1, *rank2.shape[-2:])
    return mask_rank2(score, N, d_min, d_max, flags)


def run_network(params: dict, w: Weights, x: Tensor, adj: Tensor, rank2: Optional[Tensor], flags: Optional[Tensor]) -> Tensor:
    """Dispatch on params['model_type'] like loader.load_model (loader.py:83-101)."""
    t = params["model_type"]
    if t == "ScoreNetworkX":
        return score_network_x(w, x, adj, flags)
    if t == "ScoreNetworkX_GMH":
        return score_network_x_gmh(w, params, x, adj, flags)
    if t == "ScoreNetworkA":
        return score_network_a(w, params, x, adj, flags)
    if t == "ScoreNetworkA_CC":
        return score_network_a_cc(w, params, x, adj, rank2, flags)
    if t == "ScoreNetworkA_Base_CC":
        return score_network_a_base_cc(w, params, x, adj, rank2, flags)
    if t == "ScoreNetworkF":
        return score_network_f(w, params, x, adj, rank2, flags)
    raise ValueError(f"Model Name <{t}> is unknown.")


# --------------------------------------------------------------------------------------
# S1-S3: SDEs                                                        (ccsd/src/sde.py)
# --------------------------------------------------------------------------------------
class SDE:
    T = 1

    def __init__(self, kind: str, bmin: float, bmax: float, N: int):
        self.kind, self.bmin, self.bmax, self.N = kind, bmin, bmax, N
        if kind in ("VP", "subVP"):
            self.discrete_betas = torch.linspace(bmin / N, bmax / N, N)  # sde.py:364, 689
            self.alphas = 1.0 - self.discrete_betas
        elif kind == "VE":
            self.discrete_sigmas = torch.exp(torch.linspace(np.log(bmin), np.log(bmax), N))  # sde.py:523-525
        else:
            raise NotImplementedError(f"SDE class {kind} not (yet) supported.")

    # forward sde(): drift, diffusion
    def sde(self, v: Tensor, t: Tensor) -> Tuple[Tensor, Tensor]:
        if self.kind == "VP":  # sde.py:401-404
            beta_t = self.bmin + t * (self.bmax - self.bmin)
            return -0.5 * beta_t[:, None, None] * v, torch.sqrt(beta_t)
        if self.kind == "subVP":  # sde.py:722-728
            beta_t = self.bmin + t * (self.bmax - self.bmin)
            disc = 1.0 - torch.exp(-2 * self.bmin * t - (self.bmax - self.bmin) * t**2)
            return -0.5 * beta_t[:, None, None] * v, torch.sqrt(beta_t * disc)
        sigma = self.bmin * (self.bmax / self.bmin) ** t  # sde.py:558-565
        return torch.zeros_like(v), sigma * torch.sqrt(torch.tensor(2 * (np.log(self.bmax) - np.log(self.bmin))))

    def marginal_std(self, t: Tensor) -> Tensor:
        if self.kind == "VE":  # sde.py:579
            return self.bmin * (self.bmax / self.bmin) ** t
        lmc = -0.25 * t**2 * (self.bmax - self.bmin) - 0.5 * t * self.bmin
        if self.kind == "VP":  # sde.py:419-424
            return torch.sqrt(1.0 - torch.exp(2.0 * lmc))
        return 1 - torch.exp(2.0 * lmc)  # subVP sde.py:742-747

    def discretize(self, v: Tensor, t: Tensor) -> Tuple[Tensor, Tensor]:
        ts = (t * (self.N - 1) / self.T).long()
        if self.kind == "VP":  # sde.py:477-483
            beta, alpha = self.discrete_betas[ts], self.alphas[ts]
            return torch.sqrt(alpha)[:, None, None] * v - v, torch.sqrt(beta)
        if self.kind == "VE":  # sde.py:639-648
            sigma = self.discrete_sigmas[ts]
            adjacent = torch.where(ts == 0, torch.zeros_like(t), self.discrete_sigmas[ts - 1])
            return torch.zeros_like(v), torch.sqrt(sigma**2 - adjacent**2)
        dt = 1 / self.N  # base class Euler discretisation, sde.py:107-111 (subVP)
        drift, diffusion = self.sde(v, t)
        return drift * dt, diffusion * torch.sqrt(torch.tensor(dt))

    def transition(self, v: Tensor, t: Tensor, dt: float) -> Tuple[Tensor, Tensor]:
        if self.kind == "VP":  # sde.py:498-503
            lmc = 0.25 * dt * (2 * self.bmin + (2 * t + dt) * (self.bmax - self.bmin))
            return torch.exp(-lmc[:, None, None]) * v, torch.sqrt(1.0 - torch.exp(2.0 * lmc))
        if self.kind == "VE":  # sde.py:664-669
            std = torch.square(self.bmin * (self.bmax / self.bmin) ** t) - torch.square(
                self.bmin * (self.bmax / self.bmin) ** (t + dt))
            return v, torch.sqrt(std)
        raise AttributeError("subVPSDE has no transition")

    def prior(self, shape) -> Tensor:
        return torch.randn(*shape)  # sde.py:436/593/759

    def prior_sym(self, shape) -> Tensor:
        z = torch.randn(*shape).triu(1)  # sde.py:448-449/606-608/772-773
        return z + z.transpose(-1, -2)


def load_sde(cfg: dict) -> SDE:
    """loader.py:254-267."""
    return SDE(cfg["type"], cfg["beta_min"], cfg["beta_max"], cfg["num_scales"])


# --------------------------------------------------------------------------------------
# W1: score functions                                              (losses.py:18-198)
# --------------------------------------------------------------------------------------
def make_score_fn(sde: SDE, net: Callable[..., Tensor]) -> Callable[..., Tensor]:
    """VE: raw network output; VP/subVP: -out / std(t) (losses.py:157-163, 189-193)."""
    if sde.kind == "VE":
        return lambda *a: net(*a[:-1])

    def fn(*a):
        t = a[-1]
        return -net(*a[:-1]) / sde.marginal_std(t)[:, None, None]

    return fn


# --------------------------------------------------------------------------------------
# L1-L4: predictors / correctors                                (solver.py:157-853)
# --------------------------------------------------------------------------------------
class _Target:
    """One of x / adj / rank2: how to pick it out of the state and draw its noise."""

    def __init__(self, obj: str, N: int = 0, d_min: int = 0, d_max: int = 0):
        if obj not in ("x", "adj", "rank2"):
            raise NotImplementedError(f"Object {obj} not yet supported. Select from [x, adj, rank2].")
        self.obj, self.N, self.d_min, self.d_max = obj, N, d_min, d_max
        self.idx = {"x": 0, "adj": 1, "rank2": 2}[obj]

    def noise(self, v: Tensor, flags: Tensor, src: NoiseSource) -> Tensor:
        if self.obj == "x":
            return gen_noise(v, flags, False, src)
        if self.obj == "adj":
            return gen_noise(v, flags, True, src)
        return gen_noise_rank2(v, self.N, self.d_min, self.d_max, flags, src)


def langevin_update(tg: _Target, sde: SDE, score_fn, snr: float, seps: float, n_steps: int,
                    state: Sequence[Tensor], flags: Tensor, t: Tensor, src: NoiseSource, trace: Optional[dict] = None):
    """LangevinCorrector.update_fn_{graph,cc} (solver.py:678-716, 746-802)."""
    if sde.kind in ("VP", "subVP"):
        alpha = sde.alphas[(t * (sde.N - 1) / sde.T).long()]
    else:
        alpha = torch.ones_like(t)
    state = list(state)
    v = state[tg.idx]
    v_mean = v
    for _ in range(n_steps):
        grad = score_fn(*state, flags, t)
        noise = tg.noise(v, flags, src)
        grad_norm = torch.norm(grad.reshape(grad.shape[0], -1), dim=-1).mean()
        noise_norm = torch.norm(noise.reshape(noise.shape[0], -1), dim=-1).mean()
        step = (snr * noise_norm / grad_norm) ** 2 * 2 * alpha
        v_mean = v + step[:, None, None] * grad
        v = v_mean + torch.sqrt(step * 2)[:, None, None] * noise * seps
        state[tg.idx] = v
        if trace is not None:
            trace[tg.obj] = dict(grad_norm=grad_norm, noise_norm=noise_norm, step_size=step[0])
    return v, v_mean


def reverse_update(tg: _Target, sde: SDE, score_fn, pflow: bool, state, flags, t, src):
    """ReverseDiffusionPredictor.update_fn_* (solver.py:385-398, 430-457) + RSDE.discretize (sde.py:229-235, 329-340)."""
    v = state[tg.idx]
    f, G = sde.discretize(v, t)
    score = score_fn(*state, flags, t)
    rev_f = f - G[:, None, None] ** 2 * score * (0.5 if pflow else 1.0)
    rev_G = torch.zeros_like(G) if pflow else G
    z = tg.noise(v, flags, src)
    v_mean = v - rev_f
    return v_mean + rev_G[:, None, None] * z, v_mean


def euler_update(tg: _Target, sde: SDE, score_fn, pflow: bool, state, flags, t, src):
    """EulerMaruyamaPredictor.update_fn_* (solver.py:227-244, 275-307) + RSDE.sde (sde.py:200-207, 290-302)."""
    dt = -1.0 / sde.N
    v = state[tg.idx]
    z = tg.noise(v, flags, src)  # Euler draws BEFORE the score evaluation
    drift, diffusion = sde.sde(v, t)
    score = score_fn(*state, flags, t)
    drift = drift - diffusion[:, None, None] ** 2 * score * (0.5 if pflow else 1.0)
    if pflow:
        # sde.py:206/301 returns the python float 0.0, which solver.py:234/284 then indexes
        raise TypeError("'float' object is not subscriptable")
    v_mean = v + drift * dt
    return v_mean + diffusion[:, None, None] * np.sqrt(-dt) * z, v_mean


def none_corrector(tg, state):
    """NoneCorrector (solver.py:545-556, 582-597)."""
    return state[tg.idx], state[tg.idx]


# --------------------------------------------------------------------------------------
# L0: the PC sampler                                         (solver.py:856-1176)
# --------------------------------------------------------------------------------------
def get_pc_sampler(sde_x: SDE, sde_adj: SDE, shape_x, shape_adj, predictor="Euler", corrector="None",
                   snr=0.1, scale_eps=1.0, n_steps=1, probability_flow=False, continuous=False, denoise=True,
                   eps=1e-3, is_cc=False, sde_rank2: Optional[SDE] = None, shape_rank2=None,
                   d_min=None, d_max=None, noise: Optional[NoiseSource] = None, keep_traj=True,
                   n_diff_steps: Optional[int] = None, prior=None, trace: Optional[list] = None):
    """Functional restatement of get_pc_sampler.  `nets` passed to the closure are callables
    net(x, adj[, rank2], flags) -> Tensor.  Extra test hooks (not in the reference): `noise`
    (draw source), `n_diff_steps` (stop after that many of the sde_adj.N steps), `prior`
    (explicit initial state), `trace` (per-step Langevin scalars)."""
    if predictor not in ("Reverse", "Euler"):
        raise NotImplementedError(f"Predictor {predictor} not yet supported. Select from [Reverse, Euler].")
    if corrector not in ("Langevin", "None"):
        raise NotImplementedError(f"Corrector {corrector} not yet supported. Select from [Langevin, None].")
    if not continuous:
        raise NotImplementedError("Discrete not supported")
    src = noise or NoiseSource()
    N = shape_adj[1]
    sdes = [sde_x, sde_adj] + ([sde_rank2] if is_cc else [])
    targets = [_Target("x"), _Target("adj")] + ([_Target("rank2", N, d_min, d_max)] if is_cc else [])
    pred = reverse_update if predictor == "Reverse" else euler_update

    def pc_sampler(*args):
        nets, init_flags = args[:-1], args[-1]
        assert len(nets) == len(sdes)
        fns = [make_score_fn(s, n) for s, n in zip(sdes, nets)]
        with torch.no_grad():
            if prior is not None:
                state = [p.clone() for p in prior]
            else:
                state = [sde_x.prior(shape_x), sde_adj.prior_sym(shape_adj)]
                if is_cc:
                    state.append(sde_rank2.prior(shape_rank2))
            flags = init_flags
            state[0] = mask_x(state[0], flags)
            state[1] = mask_adjs(state[1], flags)
            if is_cc:
                state[2] = mask_rank2(state[2], N, d_min, d_max, flags)
            diff_steps = sde_adj.N
            timesteps = torch.linspace(sde_adj.T, eps, diff_steps)
            traj = []
            means = list(state)
            for i in range(diff_steps if n_diff_steps is None else n_diff_steps):
                vec_t = torch.ones(shape_adj[0]) * timesteps[i]
                tr = {} if trace is not None else None
                # all correctors see the pre-corrector state (solver.py:1129-1137)
                s0 = list(state)
                for k, (tg, sd, fn) in enumerate(zip(targets, sdes, fns)):
                    if corrector == "Langevin":
                        state[k], means[k] = langevin_update(tg, sd, fn, snr, scale_eps, n_steps, s0, flags, vec_t, src, tr)
                    else:
                        state[k], means[k] = none_corrector(tg, s0)
                s1 = list(state)
                for k, (tg, sd, fn) in enumerate(zip(targets, sdes, fns)):
                    state[k], means[k] = pred(tg, sd, fn, probability_flow, s1, flags, vec_t, src)
                if trace is not None:
                    trace.append(tr)
                if keep_traj:
                    traj.append([(m if denoise else s)[0].detach().clone() for m, s in zip(means, state)])
            out = means if denoise else state
            return (*out, diff_steps * (n_steps + 1), traj)

    return pc_sampler


# --------------------------------------------------------------------------------------
# L5: the S4 sampler                                         (solver.py:1179-1563)
# --------------------------------------------------------------------------------------
def S4_solver(sde_x: SDE, sde_adj: SDE, shape_x, shape_adj, predictor="None", corrector="None", snr=0.1, scale_eps=1.0,
              n_steps=1, probability_flow=False, continuous=False, denoise=True, eps=1e-3, is_cc=False,
              sde_rank2: Optional[SDE] = None, shape_rank2=None, d_min=None, d_max=None,
              noise: Optional[NoiseSource] = None, keep_traj=True, n_diff_steps: Optional[int] = None):
    """Functional restatement of S4_solver: per step one joint score evaluation, Sdrift = -g(t)^2 score
    (solver.py:1290-1294, 1436-1444), a Langevin-style correction per target with the alpha index taken from sde_x
    (:1296-1334, :1446-1510), transition(v, t, dt/2) + noise, v += Sdrift*dt, transition(v, t + dt/2, dt/2) + noise
    (:1337-1352, :1512-1529).  predictor / corrector / n_steps / probability_flow are unused, as in the reference."""
    if not continuous:
        raise NotImplementedError("Discrete not supported")
    src = noise or NoiseSource()
    N = shape_adj[1]
    sdes = [sde_x, sde_adj] + ([sde_rank2] if is_cc else [])
    targets = [_Target("x"), _Target("adj")] + ([_Target("rank2", N, d_min, d_max)] if is_cc else [])

    def s4_solver(*args):
        nets, init_flags = args[:-1], args[-1]
        assert len(nets) == len(sdes)
        fns = [make_score_fn(s, n) for s, n in zip(sdes, nets)]
        with torch.no_grad():
            state = [sde_x.prior(shape_x), sde_adj.prior_sym(shape_adj)]
            if is_cc:
                state.append(sde_rank2.prior(shape_rank2))
            flags = init_flags
            state[0] = mask_x(state[0], flags)
            state[1] = mask_adjs(state[1], flags)
            if is_cc:
                state[2] = mask_rank2(state[2], N, d_min, d_max, flags)
            diff_steps = sde_adj.N
            timesteps = torch.linspace(sde_adj.T, eps, diff_steps)
            dt = -1.0 / diff_steps
            traj = []
            means = list(state)
            for i in range(diff_steps if n_diff_steps is None else n_diff_steps):
                t = timesteps[i]
                vec_t = torch.ones(shape_adj[0]) * t
                vec_dt = torch.ones(shape_adj[0]) * (dt / 2)
                scores = [fn(*state, flags, vec_t) for fn in fns]
                sdrift = [-sd.sde(v, vec_t)[1][:, None, None] ** 2 * sc for sd, v, sc in zip(sdes, state, scores)]
                timestep = (vec_t * (sde_x.N - 1) / sde_x.T).long()
                for k, (tg, sd) in enumerate(zip(targets, sdes)):           # correction step
                    z = tg.noise(state[k], flags, src)
                    grad_norm = torch.norm(scores[k].reshape(scores[k].shape[0], -1), dim=-1).mean()
                    noise_norm = torch.norm(z.reshape(z.shape[0], -1), dim=-1).mean()
                    alpha = sd.alphas[timestep] if sd.kind == "VP" else torch.ones_like(vec_t)
                    step = (snr * noise_norm / grad_norm) ** 2 * 2 * alpha
                    v_mean = state[k] + step[:, None, None] * scores[k]
                    state[k] = v_mean + torch.sqrt(step * 2)[:, None, None] * z * scale_eps
                trans = [sd.transition(v, vec_t, vec_dt) for sd, v in zip(sdes, state)]      # prediction step
                for k, tg in enumerate(targets):
                    state[k] = trans[k][0] + trans[k][1][:, None, None] * tg.noise(state[k], flags, src)
                for k in range(len(state)):
                    state[k] = state[k] + sdrift[k] * dt
                trans = [sd.transition(v, vec_t + vec_dt, vec_dt) for sd, v in zip(sdes, state)]
                for k, tg in enumerate(targets):
                    state[k] = trans[k][0] + trans[k][1][:, None, None] * tg.noise(state[k], flags, src)
                    means[k] = trans[k][0]
                if keep_traj:
                    traj.append([(m if denoise else v)[0].detach().clone() for m, v in zip(means, state)])
            out = means if denoise else state
            return (*out, 0, traj)

    return s4_solver

