"""Sampler harness: the reference's Sampler_{Graph,CC,mol_Graph,mol_CC}.sample() up to quantisation.

Mirrors ccsd/src/sampler.py:92-1235 for the part that is on, or directly around, the reverse-SDE path:

  load_ckpt -> load_seed(config.seed) -> load_model_from_ckpt x3 (EMA weights if sample.use_ema)
  -> load_sampling_fn(configt, config.sampler, config.sample, device, is_cc, d_min, d_max, divide_batch)
  -> load_seed(config.sample.seed) -> [init_flags -> sampling_fn] x divide_batch (x sampling rounds for the generic
  datasets) -> torch.cat -> quantize / quantize_mol (+ the molecule relabelling and one-hot of sampler.py:1216-1225)

What follows in the reference -- rdkit molecule construction, toponetx complexes, MMD / NSPDK / FCD metrics, pickles,
plots, wandb -- is out of scope; `sample()` returns the tensors those steps consume and, when
`save=True`, writes them to <folder>/samples/<log name>.npz.

init_flags (cc_utils.py:883-914): the reference draws `np.random.randint(0, len(train_list), batch)` and takes the node
flags of those training objects.  Only the node COUNT of each training object matters, so this build ships the counts
(ccsd_amd/data/node_counts.json: per-graph node counts of the pickled datasets in file order -> the same indices pick
the same flags for the same numpy seed).  QM9 / ZINC250k: the reference draws from its training molecules (sampler.py:1162-1194),
whose files the reference repository does not ship (.MISSING_LARGE_BLOBS).  When the user's dataset copy is present under
<folder>/data (<dataset>_kekulized.npz or <dataset>.csv, with valid_idx_<dataset>.json) the node counts of the training molecules
are read from it in the reference's order (mol_train_node_counts: same indices, same flags for the same numpy seed); otherwise
the flags come from a shipped histogram -- QM9: the node counts of the shipped test graphs (data/qm9_test_nx.pkl), ZINC250k: a
discretised normal fit of the dataset's published heavy-atom statistics (node_counts.json notes) -- and the run says so.
"""
from __future__ import annotations

import json
import math
import os
import time
from typing import Dict, List, Optional

import numpy as np
import torch

from . import loader
from .engine import PCEngine
from .loader import AttrDict, _get

_COUNTS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "node_counts.json")


def _dataset_key(name: str) -> str:
    return name[:-3] if name.endswith("_CC") else name


def train_node_counts(configt, with_test_size: bool = False):
    """Node counts of the training split, in the order the reference's load_data(get_list=True) returns it
    (and, on request, the size of the test split, which sets the number of sampling rounds)."""
    with open(_COUNTS) as f:
        table = json.load(f)
    data = _get(configt, "data")
    entry = table.get(_dataset_key(_get(data, "data")))
    if entry is None or "node_counts" not in entry:
        return (None, 0) if with_test_size else None
    counts = np.asarray(entry["node_counts"], dtype=np.int64)
    test_size = int(_get(data, "test_split", 0.2) * len(counts))          # data_loader.py:80-81, 109-110
    return (counts[test_size:], test_size) if with_test_size else counts[test_size:]


def init_flags(obj_counts, config, batch_size: Optional[int] = None, is_cc: bool = False) -> torch.Tensor:
    """cc_utils.py:883-914 on node counts: flags[b, :n_b] = 1 with n_b the node count of a uniformly drawn training
    object (np.random.randint on the global numpy stream, like the reference).  `obj_counts` may also be a
    {node count: frequency} histogram (QM9 / ZINC250k, see the module docstring)."""
    data = _get(config, "data")
    if batch_size is None:
        batch_size = _get(data, "batch_size")
    N = _get(data, "max_node_num")
    if isinstance(obj_counts, dict):
        ks = np.array([int(k) for k in obj_counts], dtype=np.int64)
        p = np.array([float(v) for v in obj_counts.values()], dtype=np.float64)
        counts = np.random.choice(ks, size=batch_size, p=p / p.sum())
    else:
        counts = np.asarray(obj_counts, dtype=np.int64)
        counts = counts[np.random.randint(0, len(counts), batch_size)]
    flags = torch.zeros(batch_size, N)
    for b, c in enumerate(counts):
        flags[b, : int(min(c, N))] = 1.0
    return flags


_TWO_LETTER = ("Cl", "Br", "Si", "Se", "Na", "Li", "Mg", "Ca", "Al", "Sn", "Zn", "Cu", "Fe", "As", "Te")


def smiles_heavy_atoms(smiles: str) -> int:
    """Number of non-hydrogen atoms of a SMILES string, without rdkit: one per bracket atom (other than [H] / isotopes of H), per
    two-letter element, per organic-subset letter (B C N O P S F I) and per aromatic letter (b c n o p s).  That is the node count
    of the molecule's graph in the reference (mol.GetNumAtoms() of the kekulised molecule, data_loader_mol.py / mol_utils.py)."""
    n, i, L = 0, 0, len(smiles)
    while i < L:
        ch = smiles[i]
        if ch == "[":
            j = smiles.index("]", i)
            body = smiles[i + 1:j].lstrip("0123456789")
            sym = body[:2] if body[:2] in _TWO_LETTER else body[:1]
            if sym not in ("H",):
                n += 1
            i = j + 1
        elif smiles[i:i + 2] in _TWO_LETTER:
            n += 1
            i += 2
        elif ch in "BCNOPSFI" or ch in "bcnops":
            n += 1
            i += 1
        else:
            i += 1
    return n


def mol_train_node_counts(config, configt):
    """Node counts of the TRAINING molecules of QM9 / ZINC250k in the reference's order (data_loader_mol.py:352-379: the molecules of
    <dataset>_kekulized.npz in file order minus the indices of valid_idx_<dataset>.json), from the user's own copy of the dataset --
    the reference repository does not ship these blobs (.MISSING_LARGE_BLOBS).  Looked up under <folder>/<data.dir>/:
      <dataset>_kekulized.npz   arr_0 = atomic numbers per molecule, zero padded -> count of non-zeros (what the reference itself loads), else
      <dataset>.csv             column SMILES1 (QM9) / smiles (ZINC250k) -> heavy atoms per SMILES (smiles_heavy_atoms)
    with valid_idx_<dataset>.json beside it.  Returns None when neither file (or the index file) is there."""
    data = _get(configt, "data")
    name = str(_get(data, "data")).lower()
    folder = _get(config, "folder", "./")
    ddir = _get(_get(config, "data"), "dir", None) or _get(data, "dir", "./data")
    for base in dict.fromkeys([os.path.join(folder, ddir), os.path.join(folder, "data")]):
        idx_path = os.path.join(base, f"valid_idx_{name}.json")
        npz, csv_path = os.path.join(base, f"{name}_kekulized.npz"), os.path.join(base, f"{name}.csv")
        if not os.path.exists(idx_path) or not (os.path.exists(npz) or os.path.exists(csv_path)):
            continue
        with open(idx_path) as f:
            test_idx = json.load(f)
        if isinstance(test_idx, dict):                       # QM9: {"valid_idxs": ["123", ...]}
            test_idx = test_idx["valid_idxs"]
        test_idx = {int(i) for i in test_idx}
        if os.path.exists(npz):
            with np.load(npz, allow_pickle=True) as z:
                counts = np.array([int(np.count_nonzero(np.asarray(x))) for x in z["arr_0"]], dtype=np.int64)
        else:
            import csv

            col = "SMILES1" if name == "qm9" else "smiles"
            with open(csv_path, newline="") as f:
                counts = np.array([smiles_heavy_atoms(row[col]) for row in csv.DictReader(f)], dtype=np.int64)
        keep = np.ones(len(counts), dtype=bool)
        keep[[i for i in test_idx if i < len(counts)]] = False
        return counts[keep]
    return None


class Sampler:
    """Common body of the four reference samplers.  The subclasses below carry the reference's per-class differences:
    `IS_MOL` (molecule datasets: one sampling round of sample.n_samples, quantize_mol + relabelling) and `APPLIES_EMA`
    (Sampler_Graph / Sampler_CC copy the EMA weights into the models when sample.use_ema is set, sampler.py:177-186, 458-471;
    Sampler_mol_Graph / Sampler_mol_CC never look at the switch, sampler.py:684-1240).  Instantiated directly, the class
    picks both from the dataset name."""

    IS_MOL: Optional[bool] = None
    APPLIES_EMA: Optional[bool] = None

    def __init__(self, config) -> None:
        self.config = config if isinstance(config, AttrDict) else AttrDict(config)
        self.is_cc = bool(_get(self.config, "is_cc", False))
        self.is_mol = self.IS_MOL if self.IS_MOL is not None else _get(_get(self.config, "data"), "data") in ("QM9", "ZINC250k")
        self.applies_ema = self.APPLIES_EMA if self.APPLIES_EMA is not None else not self.is_mol
        # Several GPUs (reference: load_device() returns every GPU and each network is wrapped in DataParallel, loader.py:58-68,
        # 134-135, 649-650).  Here: one process per GPU, each sampling a shard of every chunk (ccsd_amd/distributed.py).  A process
        # that belongs to a torch.distributed group -- started by torch.distributed.run, or by CCSD.run(gpus=N) -- takes the
        # sharded seam in load(); a lone process is the single-GPU harness.
        self.rank, self.world = 0, 1
        if (torch.distributed.is_available() and torch.distributed.is_initialized()) or int(os.environ.get("WORLD_SIZE", "1")) > 1:
            from . import distributed

            self.rank, self.world, dev = distributed.init()
            self.device = dev if dev == "cpu" else [dev]
            self.device0 = dev
        else:
            self.device = loader.load_device()
            self.device0 = loader._device_id(self.device)
        sample = _get(self.config, "sample")
        # sharded runs: True (default) = the Langevin norm sums are all-reduced and Philox is keyed by the global sample index, i.e. the
        # statistics of the reference's DataParallel run (norms over the whole chunk); False = per-shard norms, no per-step traffic
        self.shard_exact = bool(_get(sample, "shard_exact", True))
        # every class: sample.n_samples, else the SAMPLING config's data.batch_size, else None (sampler.py:116-118, 393-395, 705-707)
        self.n_samples = _get(sample, "n_samples", _get(_get(self.config, "data"), "batch_size", None))
        self.divide_batch = _get(sample, "divide_batch", 1) or 1
        # diff_traj has one consumer, the plotting code behind general_config.plotly_fig (sampler.py:329, 644, 983, 1402): it is
        # recorded exactly when that switch is on; `extra["keep_traj"]` overrides
        self.keep_traj = bool(_get(_get(self.config, "general_config", {}), "plotly_fig", False))
        self.extra = {}          # forwarded to get_pc_sampler / S4_solver (rng, seed, keep_traj, group, lib, ...)

    def __repr__(self) -> str:
        return f"{self.__class__.__name__}(is_cc={self.is_cc}, data={_get(_get(self.config, 'data'), 'data')})"

    # -- pieces of sample(), reusable on their own
    def load(self):
        cfg = self.config
        self.ckpt_dict = loader.load_ckpt(cfg, self.device, is_cc=self.is_cc)
        self.configt = self.ckpt_dict["config"]
        loader.load_seed(_get(cfg, "seed", 42))
        parts = ["x", "adj"] + (["rank2"] if self.is_cc else [])
        use_ema = self.applies_ema and bool(_get(_get(cfg, "sample"), "use_ema", False))
        self.models = []
        for p in parts:
            sd = dict(self.ckpt_dict[f"{p}_state_dict"])
            if use_ema:
                ema = self.ckpt_dict.get(f"ema_{p}")
                if ema is None:
                    raise KeyError(f"sample.use_ema is set but the checkpoint holds no EMA weights for {p}")
                sd.update(ema)                                            # ema.copy_to(model.parameters()), sampler.py:469-471
            self.models.append(loader.load_model_from_ckpt(self.ckpt_dict[f"params_{p}"], sd, self.device))
        data = _get(cfg, "data")
        extra = dict({"keep_traj": self.keep_traj}, **self.extra)
        if self.world > 1:
            # full-chunk flags in, all-gathered full chunk out on every rank: the contract of the reference's DataParallel call
            from . import distributed

            self.sampling_fn = distributed.load_sampling_fn_sharded(self.configt, _get(cfg, "sampler"), _get(cfg, "sample"), self.device,
                                                                    is_cc=self.is_cc, d_min=_get(data, "d_min"), d_max=_get(data, "d_max"),
                                                                    divide_batch=self.divide_batch, exact=self.shard_exact, **extra)
        else:
            self.sampling_fn = loader.load_sampling_fn(self.configt, _get(cfg, "sampler"), _get(cfg, "sample"), self.device,
                                                       is_cc=self.is_cc, d_min=_get(data, "d_min"), d_max=_get(data, "d_max"),
                                                       divide_batch=self.divide_batch, **extra)
        counts, self.n_test = train_node_counts(self.configt, with_test_size=True)
        self.node_counts_source = "shipped per-graph node counts of the training split (ccsd_amd/data/node_counts.json)"
        if counts is None and self.is_mol:
            # molecule datasets: the training molecules of the user's own dataset copy, as the reference draws them (sampler.py:1162-1194)
            counts = mol_train_node_counts(cfg, self.configt)
            self.node_counts_source = "training molecules of the dataset under <folder>/data (reference order)"
        if counts is None:
            with open(_COUNTS) as f:
                entry = json.load(f).get(_dataset_key(_get(_get(self.configt, "data"), "data")), {})
            counts = entry.get("test_histogram") or entry.get("fallback_histogram")
            if counts is None:
                raise FileNotFoundError(f"no node counts for dataset {_get(_get(self.configt, 'data'), 'data')}: pass "
                                        "`node_counts=` to sample()")
            self.node_counts_source = entry.get("note", "shipped node-count histogram")
            if self.rank == 0:
                print(f"init_flags: dataset files not found under {_get(cfg, 'folder', './')}/data -- node counts are drawn from the shipped "
                      f"histogram ({self.node_counts_source})")
        self.node_counts = counts

    def sample(self, save: bool = False, node_counts=None, rounds: Optional[int] = None) -> Dict[str, torch.Tensor]:
        cfg = self.config
        self.load()
        if node_counts is not None:
            self.node_counts = node_counts
        loader.load_seed(_get(_get(cfg, "sample"), "seed", 42))
        datat = _get(self.configt, "data")
        # flags per chunk: n_samples // divide_batch, or the training batch size when no n_samples is configured
        # (sampler.py:210-214, 498-502, 794-796, 1185-1187; init_flags' default, cc_utils.py:901-902)
        qty = self.n_samples // self.divide_batch if self.n_samples is not None else None
        if self.is_mol:
            n_rounds = 1
        else:
            bs = _get(datat, "batch_size")
            n_rounds = rounds if rounds is not None else max(1, math.ceil(self.n_test / bs))   # sampler.py:200-202, 488-490
        t0 = time.perf_counter()
        outs: List[List[torch.Tensor]] = []
        flags_all = []
        diff_traj = []
        for _ in range(n_rounds):
            parts = None
            for _d in range(self.divide_batch):
                # (sharded runs: every rank draws the same full-chunk flags from its identically seeded numpy stream)
                fl = init_flags(self.node_counts, self.configt, qty, is_cc=self.is_cc).to(self.device0)
                res = self.sampling_fn(*self.models, fl)
                nt = 3 if self.is_cc else 2
                if parts is None:
                    diff_traj = res[-1]        # the first chunk's trajectory of the (last) round is the one kept (sampler.py:220, 511)
                parts = list(res[:nt]) if parts is None else [torch.cat((a, b), dim=0) for a, b in zip(parts, res[:nt])]
                flags_all.append(fl)
            outs.append(parts)
        self.diff_traj = diff_traj
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        sampling_time = time.perf_counter() - t0
        x, adj = (torch.cat([o[k] for o in outs], dim=0) for k in (0, 1))
        rank2 = torch.cat([o[2] for o in outs], dim=0) if self.is_cc else None
        out: Dict[str, torch.Tensor] = {"x": x, "adj": adj, "flags": torch.cat(flags_all, dim=0)}
        quant = PCEngine(None, None, None, None, None, None, N=adj.shape[-1], F=1, is_cc=False, device=self.device0,
                         lib=self.extra.get("lib"))
        if self.is_mol:
            samples_int = quant.quantize(adj, -1.0)                          # quantize_mol, graph_utils.py:195-213
            samples_int = samples_int - 1
            samples_int[samples_int == -1] = 3                               # 0,1,2,3 (no,S,D,T) -> 3,0,1,2 (sampler.py:1219-1220)
            out["adj_int"] = samples_int
            out["adj_onehot"] = torch.nn.functional.one_hot(samples_int, num_classes=4).permute(0, 3, 1, 2)
            xi = torch.where(x > 0.5, 1, 0)
            out["x_onehot"] = torch.concat([xi, 1 - xi.sum(dim=-1, keepdim=True)], dim=-1)
        else:
            out["adj_int"] = quant.quantize(adj, 0.5)                        # quantize, graph_utils.py:181-192
        if self.is_cc:
            out["rank2"] = rank2
            out["rank2_int"] = quant.quantize(rank2, 0.5).to(torch.uint8)
            # sparse form for cc_from_incidence (cc_utils.py:243-262): which of the K candidate cells exist, per complex
            out["rank2_cell_bits"], out["rank2_cell_count"] = quant.rank2_cells(rank2, 0.5)
        out["sampling_time"] = torch.tensor(sampling_time)
        self.result = out
        if save and self.rank == 0:          # every rank holds the gathered samples; rank 0 writes them
            folder = os.path.join(_get(cfg, "folder", "./"), "samples")
            os.makedirs(folder, exist_ok=True)
            name = f"{_get(cfg, 'config_name', 'sample')}_{_get(cfg, 'ckpt')}-sample_{_get(cfg, 'current_time', 'now')}"
            np.savez_compressed(os.path.join(folder, name + ".npz"), **{k: v.detach().cpu().numpy() for k, v in out.items()})
        if self.rank == 0:
            print("Sampling done.")
        return out


# the reference's four class names (sampler.py:92, 369, 684, 1061) and its factory (sampler.py:1438-1468)
class Sampler_Graph(Sampler):
    IS_MOL, APPLIES_EMA = False, True


class Sampler_CC(Sampler):
    IS_MOL, APPLIES_EMA = False, True


class Sampler_mol_Graph(Sampler):
    IS_MOL, APPLIES_EMA = True, False


class Sampler_mol_CC(Sampler):
    IS_MOL, APPLIES_EMA = True, False


def get_sampler_from_config(config) -> Sampler:
    config = config if isinstance(config, AttrDict) else AttrDict(config)
    is_cc = bool(_get(config, "is_cc", False))
    is_mol = _get(_get(config, "data"), "data") in ("QM9", "ZINC250k")
    cls = (Sampler_mol_CC if is_cc else Sampler_mol_Graph) if is_mol else (Sampler_CC if is_cc else Sampler_Graph)
    return cls(config)
