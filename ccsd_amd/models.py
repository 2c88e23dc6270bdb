"""Score-network objects with the reference's constructor signatures and state_dict keys.

They are weight containers: `forward` evaluates the network with the HIP kernels (through a
single-network plan), and the PC sampler reads `params` + `state_dict()` to build its joint plan.
Reference: ccsd/src/models/ScoreNetwork_X.py:22-153, ScoreNetwork_A.py:348-561,
ScoreNetwork_A_CC.py:21-332, ScoreNetwork_F.py:22-217; factory loader.py:71-101.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Any, Dict, Optional

import torch

from . import _lib, plan as _plan


class _ScoreNetwork:
    model_type = "?"
    target = -1

    def __init__(self, **params):
        self.params = dict(params, model_type=self.model_type)
        self.params.setdefault("use_bn", False)
        _plan._check_supported(self.params)
        self._device = torch.device("cpu")
        self._sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
        self._engine = None
        self.training = True
        self.reset_parameters()

    # ---- nn.Module-like surface used by the reference's loader / sampler
    def reset_parameters(self) -> None:
        """glorot-uniform weights, zero biases (reference layers.py:20-39).  Does not reproduce the
        reference's RNG consumption order; checkpoints are the supported source of weights."""
        for k, shape in _plan.state_dict_shapes(self.params):
            if k.endswith("bias"):
                t = torch.zeros(shape)
            else:
                stdv = math.sqrt(6.0 / (shape[-2] + shape[-1]))
                t = torch.empty(shape).uniform_(-stdv, stdv)
            self._sd[k] = t.to(self._device)
        self._engine = None

    def state_dict(self) -> "OrderedDict[str, torch.Tensor]":
        return OrderedDict(self._sd)

    def load_state_dict(self, state_dict: Dict[str, Any], strict: bool = True):
        sd = {(k[7:] if k.startswith("module.") else k): v for k, v in state_dict.items()}
        expected = dict(_plan.state_dict_shapes(self.params))
        missing = [k for k in expected if k not in sd]
        # (BatchNorm1d entries of a use_bn=True checkpoint are tolerated: no such network reaches a forward, plan.reference_forward_error)
        unexpected = [k for k in sd if k not in expected and ".batch_norms." not in k]
        if strict and (missing or unexpected):
            raise RuntimeError(f"Error(s) in loading state_dict for {self.model_type}: missing {missing}, unexpected {unexpected}")
        for k, shape in expected.items():
            if k in sd:
                v = torch.as_tensor(sd[k], dtype=torch.float32)
                if tuple(v.shape) != tuple(shape):
                    raise RuntimeError(f"size mismatch for {k}: copying a param with shape {tuple(v.shape)}, the shape in current model is {tuple(shape)}")
                self._sd[k] = v.detach().clone().to(self._device)
        self._engine = None
        return self

    def parameters(self):
        return iter(self._sd.values())

    def to(self, device):
        self._device = torch.device(device)
        self._sd = OrderedDict((k, v.to(self._device)) for k, v in self._sd.items())
        self._engine = None
        return self

    def eval(self):
        self.training = False
        return self

    def train(self, mode: bool = True):
        if mode:
            raise NotImplementedError("ccsd_amd implements the sampling path only")
        return self.eval()

    def __repr__(self) -> str:
        return f"{self.model_type}({', '.join(f'{k}={v}' for k, v in self.params.items() if k != 'model_type')})"

    # ---- forward through the HIP kernels
    def _dims(self, x, adj):
        return adj.shape[-1], x.shape[-1]

    def _get_engine(self, x, adj, lib=None):
        from .engine import PCEngine

        N, F = self._dims(x, adj)
        key = (N, F, str(x.device))
        if self._engine is None or self._engine[0] != key:
            is_cc = bool(self.params.get("is_cc", False)) and "d_min" in self.params
            kw = dict(N=N, F=F, is_cc=is_cc, d_min=self.params.get("d_min", 0), d_max=self.params.get("d_max", 0),
                      device=x.device, lib=lib)
            slots = [None, None, None, None, None, None]
            slots[2 * self.target], slots[2 * self.target + 1] = self.params, self._sd
            self._engine = (key, PCEngine(*slots, **kw))
        return self._engine[1]

    def forward(self, x, adj, *rest, lib=None):
        rank2, flags = (rest + (None, None))[:2] if len(rest) != 1 else (None, rest[0])
        if flags is None:
            flags = torch.ones(x.shape[0], adj.shape[-1], device=x.device)
        eng = self._get_engine(x, adj, lib)
        if eng.is_cc and rank2 is None:
            raise ValueError("rank2 is required")
        return eng.score(self.target, x.contiguous(), adj.contiguous(), rank2.contiguous() if rank2 is not None else None,
                         flags.contiguous())

    __call__ = forward


class ScoreNetworkX(_ScoreNetwork):
    model_type, target = "ScoreNetworkX", _lib.TARGET_X

    def __init__(self, max_feat_num: int, depth: int, nhid: int, use_bn: bool = False, is_cc: bool = False):
        super().__init__(max_feat_num=max_feat_num, depth=depth, nhid=nhid, use_bn=use_bn, is_cc=is_cc)

    def _get_engine(self, x, adj, lib=None):
        # ScoreNetworkX ignores rank2 (ScoreNetwork_X.py:153): a graph-only plan is enough
        from .engine import PCEngine

        N, F = self._dims(x, adj)
        key = (N, F, str(x.device))
        if self._engine is None or self._engine[0] != key:
            self._engine = (key, PCEngine(self.params, self._sd, None, None, None, None, N=N, F=F, is_cc=False,
                                          device=x.device, lib=lib))
        return self._engine[1]

    def forward(self, x, adj, *rest, lib=None):
        flags = rest[-1] if rest else None
        if flags is None:
            flags = torch.ones(x.shape[0], adj.shape[-1], device=x.device)
        return self._get_engine(x, adj, lib).score(self.target, x.contiguous(), adj.contiguous(), None, flags.contiguous())

    __call__ = forward


class ScoreNetworkX_GMH(ScoreNetworkX):
    """ScoreNetwork_X.py:156-341: AttentionLayers in place of the GCN layers."""
    model_type = "ScoreNetworkX_GMH"

    def __init__(self, max_feat_num: int, depth: int, nhid: int, num_linears: int, c_init: int, c_hid: int, c_final: int,
                 adim: int, num_heads: int = 4, conv: str = "GCN", use_bn: bool = False, is_cc: bool = False):
        _ScoreNetwork.__init__(self, max_feat_num=max_feat_num, depth=depth, nhid=nhid, num_linears=num_linears, c_init=c_init,
                               c_hid=c_hid, c_final=c_final, adim=adim, num_heads=num_heads, conv=conv, use_bn=use_bn,
                               is_cc=is_cc)


class ScoreNetworkA(_ScoreNetwork):
    model_type, target = "ScoreNetworkA", _lib.TARGET_ADJ

    def __init__(self, max_feat_num: int, max_node_num: int, nhid: int, num_layers: int, num_linears: int, c_init: int,
                 c_hid: int, c_final: int, adim: int, num_heads: int = 4, conv: str = "GCN", use_bn: bool = False,
                 is_cc: bool = False):
        super().__init__(max_feat_num=max_feat_num, max_node_num=max_node_num, nhid=nhid, num_layers=num_layers,
                         num_linears=num_linears, c_init=c_init, c_hid=c_hid, c_final=c_final, adim=adim,
                         num_heads=num_heads, conv=conv, use_bn=use_bn, is_cc=is_cc)

    def _get_engine(self, x, adj, lib=None):
        from .engine import PCEngine

        N, F = self._dims(x, adj)
        key = (N, F, str(x.device))
        if self._engine is None or self._engine[0] != key:
            self._engine = (key, PCEngine(None, None, self.params, self._sd, None, None, N=N, F=F, is_cc=False,
                                          device=x.device, lib=lib))
        return self._engine[1]

    def forward(self, x, adj, *rest, lib=None):
        flags = rest[-1] if rest else None
        if flags is None:
            flags = torch.ones(x.shape[0], adj.shape[-1], device=x.device)
        return self._get_engine(x, adj, lib).score(self.target, x.contiguous(), adj.contiguous(), None, flags.contiguous())

    __call__ = forward


class ScoreNetworkA_CC(_ScoreNetwork):
    model_type, target = "ScoreNetworkA_CC", _lib.TARGET_ADJ

    def __init__(self, max_feat_num: int, max_node_num: int, d_min: int, d_max: int, nhid: int, nhid_h: int,
                 num_layers: int, num_layers_h: int, num_linears: int, num_linears_h: int, c_init: int, c_hid: int,
                 c_hid_h: int, c_final: int, c_final_h: int, adim: int, adim_h: int, num_heads: int = 4,
                 num_heads_h: int = 4, conv: str = "GCN", conv_hodge: str = "HCN", use_bn: bool = False, is_cc: bool = True):
        if not is_cc:
            raise ValueError("ScoreNetworkA_CC is only for combinatorial complexes")
        super().__init__(max_feat_num=max_feat_num, max_node_num=max_node_num, d_min=d_min, d_max=d_max, nhid=nhid,
                         nhid_h=nhid_h, num_layers=num_layers, num_layers_h=num_layers_h, num_linears=num_linears,
                         num_linears_h=num_linears_h, c_init=c_init, c_hid=c_hid, c_hid_h=c_hid_h, c_final=c_final,
                         c_final_h=c_final_h, adim=adim, adim_h=adim_h, num_heads=num_heads, num_heads_h=num_heads_h,
                         conv=conv, conv_hodge=conv_hodge, use_bn=use_bn, is_cc=is_cc)


class ScoreNetworkA_Base_CC(_ScoreNetwork):
    """ScoreNetwork_A_Base_CC.py:28-323: the AttentionLayer stack of ScoreNetworkA plus HodgeBaselineLayers."""
    model_type, target = "ScoreNetworkA_Base_CC", _lib.TARGET_ADJ

    def __init__(self, max_feat_num: int, max_node_num: int, d_min: int, d_max: int, nhid: int, nhid_h: int,
                 num_layers: int, num_layers_h: int, num_linears: int, num_linears_h: int, c_init: int, c_hid: int,
                 c_hid_h: int, c_final: int, c_final_h: int, adim: int, hidden_h: int, num_heads: int = 4,
                 conv: str = "GCN", use_bn: bool = False, is_cc: bool = True):
        if not is_cc:
            raise ValueError("ScoreNetworkA_Base_CC is only for combinatorial complexes")
        super().__init__(max_feat_num=max_feat_num, max_node_num=max_node_num, d_min=d_min, d_max=d_max, nhid=nhid,
                         nhid_h=nhid_h, num_layers=num_layers, num_layers_h=num_layers_h, num_linears=num_linears,
                         num_linears_h=num_linears_h, c_init=c_init, c_hid=c_hid, c_hid_h=c_hid_h, c_final=c_final,
                         c_final_h=c_final_h, adim=adim, hidden_h=hidden_h, num_heads=num_heads, conv=conv, use_bn=use_bn,
                         is_cc=is_cc)


class ScoreNetworkF(_ScoreNetwork):
    model_type, target = "ScoreNetworkF", _lib.TARGET_RANK2

    def __init__(self, num_layers_mlp: int, num_layers: int, num_linears: int, nhid: int, c_hid: int, c_final: int,
                 cnum: int, max_node_num: int, d_min: int, d_max: int, use_hodge_mask: bool = True, use_bn: bool = False,
                 is_cc: bool = True):
        super().__init__(num_layers_mlp=num_layers_mlp, num_layers=num_layers, num_linears=num_linears, nhid=nhid,
                         c_hid=c_hid, c_final=c_final, cnum=cnum, max_node_num=max_node_num, d_min=d_min, d_max=d_max,
                         use_hodge_mask=use_hodge_mask, use_bn=use_bn, is_cc=is_cc)


def _shape(sd, key):
    v = sd.get(key)
    if v is None:
        v = sd.get("module." + key)
    return None if v is None else tuple(v.shape)


def _mlp_dims(sd, prefix):
    """(number of linears, output width) of a reference MLP from its state_dict keys (layers.py:205-218)."""
    s1 = _shape(sd, prefix + "linear.weight")
    if s1 is not None:
        return 1, s1[0]
    n = 0
    while _shape(sd, f"{prefix}linears.{n}.weight") is not None:
        n += 1
    if n == 0:
        raise ValueError(f"no MLP under {prefix}")
    return n, _shape(sd, f"{prefix}linears.{n - 1}.weight")[0]


_ATTRS = {
    # model_type -> constructor keyword -> attribute the reference module stores it under
    "ScoreNetworkX": dict(max_feat_num="nfeat", depth="depth", nhid="nhid", use_bn="use_bn", is_cc="is_cc"),
    "ScoreNetworkA": {k: k for k in ("max_feat_num", "max_node_num", "nhid", "num_layers", "num_linears", "c_init", "c_hid",
                                     "c_final", "adim", "num_heads", "conv", "use_bn", "is_cc")},
    "ScoreNetworkA_CC": {k: k for k in ("max_feat_num", "max_node_num", "d_min", "d_max", "nhid", "nhid_h", "num_layers",
                                        "num_layers_h", "num_linears", "num_linears_h", "c_init", "c_hid", "c_hid_h", "c_final",
                                        "c_final_h", "adim", "adim_h", "num_heads", "num_heads_h", "conv", "conv_hodge", "use_bn",
                                        "is_cc")},
    "ScoreNetworkA_Base_CC": {k: k for k in ("max_feat_num", "max_node_num", "d_min", "d_max", "nhid", "nhid_h", "num_layers",
                                             "num_layers_h", "num_linears", "num_linears_h", "c_init", "c_hid", "c_hid_h",
                                             "c_final", "c_final_h", "adim", "hidden_h", "num_heads", "conv", "use_bn", "is_cc")},
    "ScoreNetworkF": {k: k for k in ("num_layers_mlp", "num_layers", "num_linears", "nhid", "c_hid", "c_final", "cnum",
                                     "max_node_num", "d_min", "d_max", "use_hodge_mask", "use_bn", "is_cc")},
}


def params_from_module(model) -> Dict[str, Any]:
    """Constructor keywords (+ model_type) of a score network: this package's containers carry them as `.params`; for the
    reference's own nn.Modules (ccsd/src/models/*.py, possibly inside DataParallel) they are read back from the attributes
    the constructors store (ScoreNetwork_X.py:47-51, ScoreNetwork_A.py:390-402, ScoreNetwork_A_CC.py:83-106,
    ScoreNetwork_A_Base_CC.py:82-103, ScoreNetwork_F.py:64-76) and, where a constructor keeps none (ScoreNetworkX_GMH,
    ScoreNetwork_X.py:198-201), from the shapes in state_dict().  This is the pair loader.load_model_from_ckpt
    (loader.py:619-657) builds a model from, so the seam accepts whatever that function returned."""
    m = getattr(model, "module", model)
    if hasattr(m, "params") and isinstance(getattr(m, "params"), dict):
        return dict(m.params)
    t = type(m).__name__
    if t in _ATTRS:
        missing = [a for a in _ATTRS[t].values() if not hasattr(m, a)]
        if missing:
            raise TypeError(f"{t} object lacks the attributes {missing}: cannot recover its hyper-parameters")
        p = {k: getattr(m, a) for k, a in _ATTRS[t].items()}
        for k, v in p.items():
            if isinstance(v, bool) or isinstance(v, str):
                continue
            p[k] = int(v)
        p["model_type"] = t
        return p
    if t == "ScoreNetworkX_GMH":
        sd = m.state_dict()
        depth, c_init = int(m.depth), int(m.c_init)
        conv = "MLP" if _shape(sd, "layers.0.attn.0.gnn_q.linears.0.weight") is not None else "GCN"
        fin0, nhid = _shape(sd, "layers.0.attn.0.gnn_v.weight")
        num_linears, c_hid0 = _mlp_dims(sd, "layers.0.mlp.")
        _, c_last = _mlp_dims(sd, f"layers.{depth - 1}.mlp.")
        if depth > 1:
            adim = (_shape(sd, "layers.1.attn.0.gnn_q.linears.1.weight")[0] if conv == "MLP"
                    else _shape(sd, "layers.1.attn.0.gnn_q.weight")[1])
        else:
            adim = nhid
        heads = 4
        try:
            heads = int(m.layers[0].attn[0].num_heads)
        except Exception:
            pass
        return dict(model_type=t, max_feat_num=int(fin0), depth=depth, nhid=int(nhid), num_linears=int(num_linears), c_init=c_init,
                    c_hid=int(c_hid0), c_final=int(c_last if depth > 1 else c_hid0), adim=int(adim), num_heads=heads, conv=conv,
                    use_bn=bool(getattr(m, "use_bn", False)), is_cc=bool(getattr(m, "is_cc", False)))
    raise ValueError(
        f"Model Name <{t}> is unknown. Please select from [ScoreNetworkX, ScoreNetworkX_GMH, ScoreNetworkA, ScoreNetworkA_CC, ScoreNetworkA_Base_CC, ScoreNetworkF]")


MODEL_TYPES = {c.model_type: c for c in (ScoreNetworkX, ScoreNetworkX_GMH, ScoreNetworkA, ScoreNetworkA_CC, ScoreNetworkA_Base_CC, ScoreNetworkF)}


def load_model(params: Dict[str, Any]) -> _ScoreNetwork:
    """loader.load_model (loader.py:83-101)."""
    p = dict(params)
    t = p.pop("model_type", None)
    if t not in MODEL_TYPES:
        raise ValueError(
            f"Model Name <{t}> is unknown. Please select from [ScoreNetworkX, ScoreNetworkX_GMH, ScoreNetworkA, ScoreNetworkA_CC, ScoreNetworkA_Base_CC, ScoreNetworkF]")
    return MODEL_TYPES[t](**p)
