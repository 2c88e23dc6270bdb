"""CPU suite: the product code (ccsd_amd.*) driven through the C ABI of the HOST EMULATION of the kernel
source (tests/emu, see ccsd_amd/csrc/ccsd_rt.h) against the reference goldens and the oracle.  This checks
indexing, weight layout, masks, SDE tables, draw order and the step orchestration on the GPU-less build
box; the MFMA / wave-level code paths are covered by tests/test_gpu_parity.py (-m gpu)."""
import pytest
import torch

from tests import parity_cases as pc
from tests.emu_util import emu_library

torch.set_num_threads(8)
DEV = "cpu"


@pytest.fixture(scope="module")
def lib():
    return emu_library()


@pytest.mark.parametrize("name", ["ccsd_qm9_CC", "gdss_community_small", "ccsd_qm9_Base_CC"])
def test_forward_vs_reference_golden(lib, name):
    pc.case_forward_vs_reference_golden(name, lib, DEV)


@pytest.mark.parametrize("name", ["ccsd_community_small_CC", "ccsd_community_small_Base_CC"])
def test_forward_community_small_cc(lib, name):
    pc.case_forward_vs_reference_golden(name, lib, DEV)


@pytest.mark.parametrize("name", ["ccsd_enzymes_small_CC", "gdss_zinc250k"])
def test_forward_large_nets(lib, name):
    """ENZYMES_small_CC (E = 66 > 64: tiled rank-2 kernels, 2-linear hodge MLPs) and zinc250k (N = 38: channel stack in
    the HBM workspace)."""
    pc.case_forward_vs_reference_golden(name, lib, DEV)


def test_model_objects(lib):
    pc.case_model_objects_forward(lib, DEV)


def test_kat_gmh(lib):
    pc.case_kat_gmh(lib, DEV)


def test_kat_small_general_paths(lib):
    pc.case_kat_small_general(lib, DEV)


@pytest.mark.parametrize("gname,ckpt,case", [
    ("ccsd_qm9_CC", "ccsd_qm9_CC", "k10"),
    ("ccsd_qm9_CC", "ccsd_qm9_CC", "n1000_first3"),
    ("ccsd_qm9_Base_CC", "ccsd_qm9_Base_CC", "n1000_first3"),
    ("gdss_community_small", "gdss_community_small", "k10"),
    ("gdss_community_small", "gdss_community_small", "n1000_first3"),
    ("ccsd_qm9_CC_nsteps2_none", "ccsd_qm9_CC", "k6"),
    ("ccsd_qm9_CC_langevin2", "ccsd_qm9_CC", "k4"),
    ("ccsd_community_small_CC", "ccsd_community_small_CC", "n1000_first2"),
    ("gdss_zinc250k", "gdss_zinc250k", "k5"),
    ("s4_ccsd_qm9_CC", "ccsd_qm9_CC", "k6"),
    ("s4_gdss_community_small", "gdss_community_small", "k5"),
    ("s4_ccsd_enzymes_small_CC", "ccsd_enzymes_small_CC", "n1000_first2"),
    # subVPSDE (Euler / Reverse), probability_flow + Reverse (CC and graph-only), subVP(x) + VE mixed with n_steps = 2
    ("ccsd_qm9_CC_subvp_euler", "ccsd_qm9_CC", "k6"),
    ("ccsd_qm9_CC_subvp_reverse", "ccsd_qm9_CC", "k6"),
    ("ccsd_qm9_CC_pflow", "ccsd_qm9_CC", "k6"),
    ("gdss_community_small_pflow", "gdss_community_small", "k5"),
    ("ccsd_qm9_CC_subvp_mixed", "ccsd_qm9_CC", "k4"),
])
def test_pc_sampler_identical_seed(lib, gname, ckpt, case):
    pc.case_pc_sampler_identical_seed(gname, ckpt, case, lib, DEV)


@pytest.mark.parametrize("case", ["k4", "k20"])
def test_fp64_arbiter_s4_enzymes(lib, case):
    """The two cases whose tolerance against the fp32 reference golden is wider than 1e-4 (parity_cases.TRAJ_RTOL), judged
    against the float64 trajectory: the product may not be further from it than the reference is."""
    r = pc.case_fp64_arbiter("s4_ccsd_enzymes_small_CC", "ccsd_enzymes_small_CC", case, lib, DEV)
    e_ref, e_mine, e_mut = r["rank2"]
    assert e_mut <= e_ref + e_mine + 1e-7
    assert e_ref > 0.9e-4, "the reference itself is no longer > 1e-4 from the exact trajectory: tighten TRAJ_RTOL"


def test_fp64_arbiter_qm9(lib):
    """Control: on a well-conditioned case both the reference and the product sit within 2e-5 of the float64 trajectory."""
    r = pc.case_fp64_arbiter("ccsd_qm9_CC", "ccsd_qm9_CC", "k10", lib, DEV)
    for p, (e_ref, e_mine, _) in r.items():
        assert e_ref < 5e-5 and e_mine < 5e-5, (p, e_ref, e_mine)


def test_kat_cnum_more_hodge_powers(lib):
    pc.case_kat_cnum(lib, DEV)


def test_zinc5b_substitute(lib):
    pc.case_zinc5b(lib, DEV)


def test_philox_properties(lib):
    pc.case_philox_properties(lib, DEV)


def test_philox_calls_are_independent(lib):
    pc.case_philox_calls_are_independent(lib, DEV)


def test_philox_langevin_nsteps2(lib):
    pc.case_philox_langevin_nsteps2(lib, DEV)


def test_philox_prior_statistics(lib):
    pc.case_philox_prior_statistics(lib, DEV)


def test_error_behaviour(lib):
    pc.case_error_behaviour(lib, DEV)


def test_rank2_cells_sparse_output(lib):
    pc.case_rank2_cells(lib, DEV)


@pytest.mark.parametrize("env", [{"CCSD_NO_FUSED_R2": "1"}, {"CCSD_XA_PASS": "1"}, {"CCSD_NO_FUSED_APPLY": "1"}, {"CCSD_XA_GCH": "1"}])
def test_alternative_kernel_paths_qm9(lib, env, monkeypatch):
    """The general tiled rank-2 kernels / the LDS-staged-weights variant / the unfused apply pass / the HBM channel
    stack on the qm9_CC cases."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    pc.case_forward_vs_reference_golden("ccsd_qm9_CC", lib, DEV)
    pc.case_pc_sampler_identical_seed("ccsd_qm9_CC", "ccsd_qm9_CC", "k10", lib, DEV)
    pc.case_philox_properties(lib, DEV)


def test_one_step_vs_oracle_edge_batches(lib):
    """Single-complex batch; a batch holding an empty graph, a 1-node and a 2-node graph (no rank-2 cell fits)."""
    pc.case_one_step_vs_oracle_large("ccsd_qm9_CC", lib, DEV, 1, [7], "Reverse", "Langevin", 0.2, 0.7)
    pc.case_one_step_vs_oracle_large("ccsd_qm9_CC", lib, DEV, 5, [9, 0, 1, 2, 3], "Euler", "Langevin", 0.2, 0.7, seed=8)


def test_kat_hodge_layers_three_and_four(lib):
    pc.case_kat_hodge_layers(lib, DEV)


def test_kat_hodge_general_stack(lib):
    pc.case_kat_hodge_general(lib, DEV)


def test_production_loop_vs_oracle(lib):
    """ccsd_sampler_run (Philox in the kernels, fused Langevin apply) against the oracle fed with the exported draws; the GPU
    twin runs the BASELINE batch."""
    pc.case_production_loop_vs_oracle("ccsd_qm9_CC", lib, DEV, 5, [9, 7, 8, 0, 4], 2, "Reverse", "Langevin", 0.2, 0.7, expect_fused=True)
    pc.case_production_loop_vs_oracle("gdss_community_small", lib, DEV, 3, [20, 12, 16], 2, "Euler", "Langevin", 0.05, 0.7,
                                      expect_fused=False)


def test_production_loop_edge_flags(lib):
    """Edge cases of the flags (init_flags can draw any node count 1..N; 0 = an EMPTY complex is what the masks make of a padded
    batch slot): complexes with no node, one node (no edge, no cell), two nodes (one edge, no cell of rank 2) beside full ones, and a
    batch of ONE -- the production loop against the oracle on the exported draws, and bit equality with the step-wise loop."""
    pc.case_production_loop_vs_oracle("ccsd_qm9_CC", lib, DEV, 7, [0, 1, 2, 9, 3, 0, 9], 3, "Reverse", "Langevin", 0.2, 0.7, seed=31)
    pc.case_production_loop_vs_oracle("ccsd_qm9_CC", lib, DEV, 1, [5], 2, "Reverse", "Langevin", 0.2, 0.7, seed=37)


def test_fused_r2_serves_nonaffine_shapes(lib):
    pc.case_fused_r2_nonaffine_shapes(lib, DEV)


def test_zinc5b_production_loop_vs_oracle(lib):
    pc.case_zinc5b_production_loop(lib, DEV)


def test_geometry_switch_is_inert_on_the_emulation(lib):
    """CPU twin of the GPU test: the CCSD_NO_GEO plan option changes nothing in the results (the emulation compiles both forms too)."""
    pc.case_geometry_instances_bitwise(lib, DEV, B=6, steps=2)


def test_baked_plan_headers_are_current(lib, tmp_path):
    """ccsd_amd/csrc/ccsd_baked_*.h (the plans of the qm9_CC and community_small_CC bench configurations as compile-time constants,
    tools/bake_plan.py) equal what the planner produces today: a change to PlanD or the planner without a re-bake would silently
    retire the baked kernel instances (the host falls back to the run-time-plan ones), so it is caught here."""
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    import bake_plan
    for name, workload, header in bake_plan.TARGETS:
        out = tmp_path / header
        bake_plan.bake(name, workload, str(out), lib)
        committed = open(os.path.join(root, "ccsd_amd", "csrc", header)).read()
        assert out.read_text() == committed, f"{header}: re-run python tools/bake_plan.py"
    assert bake_plan.make_engine("qm9_CC", lib).query("xa_variant") == 7, "the headline plan does not select the baked k_xa instance"
    assert bake_plan.make_engine("community_small_CC", lib).query("xa_variant") == 8
