"""Sampler harness / CCSD API / YAML surface (ccsd_amd.sampler, ccsd_amd.diffusion): the reference's
Sampler_*.sample() flow up to quantisation (sampler.py:1104-1235, 415-545; diffusion.py:27-200).  CPU: the product code
runs over the host emulation of the kernel source (explicit lib=); the GPU twin lives in test_gpu_parity.py."""
import json
import os

import numpy as np
import pytest
import torch
import yaml

from ccsd_amd import sampler as S
from ccsd_amd.diffusion import CCSD, get_config
from ccsd_amd.loader import AttrDict

QM9_CC_YAML = {
    "is_cc": True,
    "data": {"data": "QM9", "dir": "./data", "max_node_num": 9, "min_node_val": 6, "max_node_val": 9, "node_label": "symbol",
             "min_edge_val": 1, "max_edge_val": 3, "edge_label": "bond_type", "d_min": 3, "d_max": 9},
    "ckpt": "ccsd_qm9_CC",
    "sampler": {"predictor": "Reverse", "corrector": "Langevin", "snr": 0.2, "scale_eps": 0.7, "n_steps": 1},
    "sample": {"divide_batch": 2, "n_samples": 8, "cc_nb_eval": 1000, "use_ema": False, "noise_removal": True,
               "probability_flow": False, "eps": 1.0e-4, "seed": 42},
}
ENZYMES_YAML = {
    "is_cc": True,
    "data": {"data": "ENZYMES_small_CC", "dir": "./data", "d_min": 3, "d_max": 4},
    "ckpt": "ccsd_enzymes_small_CC",
    "sampler": {"predictor": "S4", "corrector": "None", "snr": 0.15, "scale_eps": 0.7, "n_steps": 1},
    "sample": {"use_ema": True, "noise_removal": True, "probability_flow": False, "eps": 1.0e-4, "seed": 42},
}


def write_cfg(tmp_path, name, cfg):
    os.makedirs(tmp_path / "config", exist_ok=True)
    with open(tmp_path / "config" / f"{name}.yaml", "w") as f:
        yaml.safe_dump(cfg, f)


def test_init_flags_follows_the_numpy_stream():
    """Same indices as the reference's np.random.randint over the train split (cc_utils.py:906-907)."""
    cfgt = AttrDict({"data": {"data": "community_small_CC", "max_node_num": 20, "batch_size": 16, "test_split": 0.2}})
    counts = S.train_node_counts(cfgt)
    table = json.load(open(S._COUNTS))["community_small"]["node_counts"]
    assert len(counts) == 80 and list(counts) == table[20:]
    np.random.seed(12)
    fl = S.init_flags(counts, cfgt)
    np.random.seed(12)
    idx = np.random.randint(0, 80, 16)
    want = torch.zeros(16, 20)
    for b, i in enumerate(idx):
        want[b, : counts[i]] = 1
    assert torch.equal(fl, want)
    assert S.init_flags(counts, cfgt, 5).shape == (5, 20)


def test_init_flags_matches_the_reference_golden():
    """(f)2 pinned by the reference: tests/golden/g7_init_flags.npz holds init_flags(train_graph_list, config, batch) of the
    reference (cc_utils.py:883-914) on its own load_data(get_list=True) split of the four shipped graph datasets, for several
    numpy seeds and batch sizes (tools/make_golden.py::g7_init_flags).  The product reproduces the flags bit for bit from the
    shipped node counts and leaves the numpy stream at the same position."""
    from tests.helpers import load_golden

    g = load_golden("g7_init_flags.npz")
    meta = json.loads(str(g["meta"]))
    assert set(meta) == {"community_small", "ego_small", "ENZYMES_small", "grid_small"}
    for name, m in meta.items():
        for is_cc in (False, True):          # the *_CC datasets are the same graphs in the same order (cc_utils.py:909-913)
            cfgt = AttrDict({"data": {"data": name + ("_CC" if is_cc else ""), "max_node_num": m["max_node_num"], "batch_size": 24,
                                      "test_split": 0.2}})
            counts, n_test = S.train_node_counts(cfgt, with_test_size=True)
            assert len(counts) == m["n_train"] and n_test == m["n_test"]
            for case in m["cases"]:
                np.random.seed(case["seed"])
                fl = S.init_flags(counts, cfgt, case["batch"], is_cc=is_cc)
                assert fl.dtype == torch.float32
                assert torch.equal(fl, torch.from_numpy(g[case["key"]])), (name, case)
                assert int(np.random.randint(0, 1 << 30)) == case["next_randint"], "numpy stream position differs"


def check_flags_against_reference_golden(out, dataset, seed, batch):
    """out["flags"] of a harness run against the reference's own init_flags for the same numpy seed (g7_init_flags.npz): the
    first chunk of the first sampling round is drawn right after load_seed(sample.seed) (sampler.py:198-218)."""
    from tests.helpers import load_golden

    g = load_golden("g7_init_flags.npz")
    want = torch.from_numpy(g[f"{dataset}/s{seed}_b{batch}"])
    assert torch.equal(out["flags"][:batch].cpu(), want), f"init_flags of {dataset} differs from the reference's"


def run_harness(tmp_path, lib, device_patch, name, cfg, max_steps, **kw):
    write_cfg(tmp_path, name, cfg)
    c = CCSD("sample", name + ".yaml", folder=str(tmp_path), seed=42)
    assert get_config(name, 42, str(tmp_path)).ckpt == cfg["ckpt"]
    with pytest.raises(NotImplementedError):
        CCSD("train", name, folder=str(tmp_path)).run()
    c.sampler = S.get_sampler_from_config(c.cfg)
    c.sampler.extra = dict(lib=lib, max_steps=max_steps)
    return c.sampler.sample(save=True, **kw), c


def test_ccsd_sample_qm9_cc_yaml(tmp_path):
    from tests.emu_util import emu_library

    out, c = run_harness(tmp_path, emu_library(), None, "sample_qm9_CC", QM9_CC_YAML, max_steps=2)
    assert type(c.sampler).__name__ == "Sampler_mol_CC"
    assert out["x"].shape == (8, 9, 4) and out["adj"].shape == (8, 9, 9) and out["rank2"].shape == (8, 36, 466)
    assert out["flags"].shape == (8, 9) and set(out["flags"].unique().tolist()) <= {0.0, 1.0}
    assert out["adj_int"].dtype == torch.int64 and set(out["adj_int"].unique().tolist()) <= {0, 1, 2, 3}
    assert out["adj_onehot"].shape == (8, 4, 9, 9) and out["x_onehot"].shape == (8, 9, 5)
    assert out["rank2_int"].dtype == torch.uint8
    # relabelling of sampler.py:1219-1220: "no bond" (quantize_mol 0) -> 3
    from oracle import ccsd_oracle as O
    q = torch.as_tensor(O.quantize_mol(out["adj"].clone()))
    assert torch.equal(out["adj_int"], torch.where(q == 0, torch.full_like(q, 3), q - 1))
    assert len(os.listdir(tmp_path / "samples")) == 1
    # masked entries stay masked through the whole flow
    fl = out["flags"]
    assert torch.equal(out["adj"], out["adj"] * fl[:, :, None] * fl[:, None, :])


def test_ccsd_sample_enzymes_s4_ema_yaml(tmp_path):
    """The shipped ENZYMES_small_CC sampling config: S4 solver + EMA weights; one round of the generic-dataset loop."""
    from tests.emu_util import emu_library

    out, c = run_harness(tmp_path, emu_library(), None, "sample_enzymes_small_CC", ENZYMES_YAML, max_steps=1, rounds=1)
    assert type(c.sampler).__name__ == "Sampler_CC"
    B = c.sampler.configt.data.batch_size
    assert out["adj"].shape == (B, 12, 12) and out["rank2"].shape[0] == B
    assert set(out["adj_int"].unique().tolist()) <= {0, 1}
    check_flags_against_reference_golden(out, "ENZYMES_small", 42, 64)
    ema = c.sampler.ckpt_dict["ema_adj"]
    sd = c.sampler.models[1].state_dict()
    k = next(iter(ema))
    assert torch.equal(sd[k].cpu(), ema[k]) and not torch.equal(ema[k], c.sampler.ckpt_dict["adj_state_dict"][k])


def test_ccsd_sample_qm9_base_cc_yaml(tmp_path):
    """The shipped sample_qm9_Base_CC configuration (ScoreNetworkA_Base_CC checkpoint) through the same harness."""
    from tests.emu_util import emu_library

    cfg = dict(QM9_CC_YAML, ckpt="ccsd_qm9_Base_CC")
    out, c = run_harness(tmp_path, emu_library(), None, "sample_qm9_Base_CC", cfg, max_steps=2)
    assert type(c.sampler).__name__ == "Sampler_mol_CC"
    assert type(c.sampler.models[1]).__name__ == "ScoreNetworkA_Base_CC"
    assert out["adj"].shape == (8, 9, 9) and out["rank2"].shape == (8, 36, 466)
    assert torch.isfinite(out["adj"]).all() and torch.isfinite(out["rank2"]).all()
    assert set(out["adj_int"].unique().tolist()) <= {0, 1, 2, 3}


def test_harness_identical_seed_vs_oracle(tmp_path):
    """CPU twin of the GPU harness-parity test: Sampler_mol_CC.sample() (divide_batch = 2, every draw from torch's CPU
    generator) over the emulation library against the oracle driven through the same seeds."""
    from tests import parity_cases as pc
    from tests.emu_util import emu_library

    pc.case_harness_vs_oracle(emu_library(), tmp_path, "sample_qm9_CC_parity", QM9_CC_YAML, "ccsd_qm9_CC", max_steps=3)


def test_mol_samplers_ignore_use_ema(tmp_path):
    """Sampler_mol_Graph / Sampler_mol_CC never read sample.use_ema (the reference's classes contain no EMA code at all,
    sampler.py:684-1240), Sampler_Graph / Sampler_CC copy the EMA weights (sampler.py:177-186, 458-471): a molecule YAML with
    use_ema: True gives the samples of the plain run."""
    from tests.emu_util import emu_library

    plain, _ = run_harness(tmp_path, emu_library(), None, "sample_qm9_CC", QM9_CC_YAML, max_steps=2)
    cfg = dict(QM9_CC_YAML, sample=dict(QM9_CC_YAML["sample"], use_ema=True))
    ema, c = run_harness(tmp_path, emu_library(), None, "sample_qm9_CC_ema", cfg, max_steps=2)
    assert type(c.sampler).__name__ == "Sampler_mol_CC" and not c.sampler.applies_ema
    for k in ("x", "adj", "rank2", "adj_int"):
        assert torch.equal(plain[k], ema[k]), k
    k = next(iter(c.sampler.ckpt_dict["adj_state_dict"]))
    assert torch.equal(c.sampler.models[1].state_dict()[k].cpu(), c.sampler.ckpt_dict["adj_state_dict"][k])
    assert S.Sampler_CC.APPLIES_EMA and S.Sampler_Graph.APPLIES_EMA and not S.Sampler_mol_Graph.APPLIES_EMA


def test_plotly_fig_switches_diff_traj(tmp_path):
    """diff_traj is recorded exactly when general_config.plotly_fig is on -- its only consumer is the plotting code behind that
    switch (sampler.py:329, 644, 983, 1402); one [x[0], adj[0], rank2[0]] entry per executed PC step
    (solver.py:1150-1157), the first divide_batch chunk's trajectory (sampler.py:1195)."""
    from tests.emu_util import emu_library

    out, c = run_harness(tmp_path, emu_library(), None, "sample_qm9_CC", QM9_CC_YAML, max_steps=2)
    assert c.sampler.keep_traj is False and c.sampler.diff_traj == []
    with open(tmp_path / "config" / "general_config.yaml", "w") as f:
        yaml.safe_dump({"plotly_fig": True, "print_initial": False}, f)
    out, c = run_harness(tmp_path, emu_library(), None, "sample_qm9_CC", QM9_CC_YAML, max_steps=2)
    assert c.sampler.keep_traj is True
    traj = c.sampler.diff_traj
    assert len(traj) == 2 and [tuple(t.shape) for t in traj[0]] == [(9, 4), (9, 9), (36, 466)]
    # the last entry holds sample 0 of the first chunk's returned (denoised) tensors
    assert torch.equal(traj[-1][1].cpu(), out["adj"][0].cpu()) and torch.equal(traj[-1][2].cpu(), out["rank2"][0].cpu())


def _mol_dataset(tmp_path, name, col, smiles, test_idx, as_dict):
    os.makedirs(tmp_path / "data", exist_ok=True)
    with open(tmp_path / "data" / f"{name}.csv", "w") as f:
        f.write(f"idx,{col},extra\n" + "".join(f"{i},{s},0\n" for i, s in enumerate(smiles)))
    with open(tmp_path / "data" / f"valid_idx_{name}.json", "w") as f:
        json.dump({"valid_idxs": [str(i) for i in test_idx]} if as_dict else test_idx, f)


def test_molecule_node_counts_come_from_the_users_dataset(tmp_path):
    """QM9 / ZINC250k: the reference draws init_flags from its TRAINING molecules (sampler.py:1162-1194, cc_utils.py:883-914;
    data_loader_mol.py:352-379: file order minus valid_idx).  With the user's dataset copy under <folder>/data the product does the
    same -- heavy atoms per SMILES of <dataset>.csv, or the atomic-number rows of <dataset>_kekulized.npz -- and the harness draws
    the same indices from the numpy stream."""
    smiles = ["C", "CCO", "c1ccccc1", "CC(=O)Nc1ccc(Cl)cc1", "[nH]1cccc1", "C[N+](C)(C)C", "O=C1C=CC(=O)N1", "CC(C)CO", "N#CC1CC1", "OCC(O)CO"]
    want = [1, 3, 6, 11, 5, 5, 7, 5, 5, 6]
    assert [S.smiles_heavy_atoms(s) for s in smiles] == want
    _mol_dataset(tmp_path, "qm9", "SMILES1", smiles, [1, 4, 7], as_dict=True)
    cfg = AttrDict({"folder": str(tmp_path), "data": {"data": "QM9", "dir": "./data"}})
    cfgt = AttrDict({"data": {"data": "QM9", "max_node_num": 9}})
    counts = S.mol_train_node_counts(cfg, cfgt)
    assert counts.tolist() == [want[i] for i in (0, 2, 3, 5, 6, 8, 9)]
    # the kekulised arrays take precedence (what the reference itself loads): arr_0 = zero-padded atomic numbers
    x = np.zeros((10, 9), np.int64)
    for i, n in enumerate([2, 3, 4, 5, 6, 7, 8, 9, 1, 2]):
        x[i, :n] = 6
    np.savez(tmp_path / "data" / "qm9_kekulized.npz", x, np.zeros((10, 4, 9, 9), np.int8))
    assert S.mol_train_node_counts(cfg, cfgt).tolist() == [2, 4, 5, 7, 8, 1, 2]
    # ZINC250k: a plain list of indices, column "smiles"; nothing under <folder>/data -> None
    _mol_dataset(tmp_path, "zinc250k", "smiles", smiles, [0, 9], as_dict=False)
    cz = AttrDict({"folder": str(tmp_path), "data": {"data": "ZINC250k", "dir": "./data"}})
    assert S.mol_train_node_counts(cz, AttrDict({"data": {"data": "ZINC250k", "max_node_num": 38}})).tolist() == want[1:9]
    assert S.mol_train_node_counts(AttrDict({"folder": str(tmp_path / "nowhere"), "data": {"data": "QM9"}}), cfgt) is None


def test_harness_draws_flags_from_the_users_training_molecules(tmp_path):
    from tests.emu_util import emu_library

    smiles = ["CCO", "c1ccccc1", "CC(=O)NC", "CCCCCCCCC", "C", "CC", "OCC(O)CO", "CC(C)CO"]
    _mol_dataset(tmp_path, "qm9", "SMILES1", smiles, [4, 5], as_dict=True)
    out, c = run_harness(tmp_path, emu_library(), None, "sample_qm9_CC", QM9_CC_YAML, max_steps=1)
    train = np.array([3, 6, 5, 9, 6, 5])
    assert "training molecules" in c.sampler.node_counts_source and np.array_equal(np.asarray(c.sampler.node_counts), train)
    np.random.seed(QM9_CC_YAML["sample"]["seed"])
    want = np.concatenate([train[np.random.randint(0, len(train), 4)] for _ in range(2)])       # two chunks of 4 (divide_batch 2)
    assert np.array_equal(out["flags"].sum(1).cpu().numpy().astype(np.int64), want)


def test_zinc250k_has_a_documented_fallback_histogram(tmp_path, capsys):
    """No dataset files: QM9 falls back to the node counts of the shipped test graphs, ZINC250k to a documented approximation --
    neither raises (the shipped sample_zinc250k.yaml runs without `node_counts=`), and the run says which source it used."""
    table = json.load(open(S._COUNTS))
    h = table["ZINC250k"]["fallback_histogram"]
    assert "APPROXIMATION" in table["ZINC250k"]["note"] and min(map(int, h)) == 6 and max(map(int, h)) == 38
    cfgt = AttrDict({"data": {"data": "ZINC250k", "max_node_num": 38, "batch_size": 8}})
    np.random.seed(3)
    fl = S.init_flags(h, cfgt, 64)
    n = fl.sum(1)
    assert fl.shape == (64, 38) and 6 <= n.min() and n.max() <= 38 and 18 < n.mean() < 28
    cfg = AttrDict(dict(QM9_CC_YAML, folder=str(tmp_path)))
    smp = S.get_sampler_from_config(cfg)
    smp.load()
    assert isinstance(smp.node_counts, dict)
    assert "shipped" in capsys.readouterr().out
