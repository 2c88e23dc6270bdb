"""The PRODUCT's per-step scalar table (ccsd_amd.sde.step_coefficients -> ccsd_plan_create) against the reference's SDE tables
(tests/golden/g3_sde_tables.npz, written by tools/make_golden.py::g3_sde_tables from ccsd/src/sde.py) for ALL 1000 steps of the
shipped time grid linspace(1, 1e-4, 1000), VE / VP / subVP x Reverse / Euler / S4.  g3 holds sde() / discretize() / transition() at
v = 0.5; drift, f and the transition mean are linear in v, so their value at v = 1 (what the table stores) is exactly twice that.
Derivations follow solver.py:275-307, 430-457, 752-756, 1296-1352; losses.py:157-163; sde.py:290-340.
Tolerance: a few float32 ulps -- the tables in g3 were evaluated on all 1000 times at once (ATen's vectorised pow / exp), the product
evaluates one step at a time as the reference's loop does (vec_t = ones(B) * t: scalar or vector kernels depending on B), and the two
kernels differ in the last bit; index lookups (timestep -> alpha) must be exact."""
import numpy as np
import pytest
import torch

from ccsd_amd import sde as S
from tests.helpers import load_golden

KINDS = {"VP": ("VP", 0.1, 1.0), "VE": ("VE", 0.1, 1.0), "VE2": ("VE", 0.2, 1.0), "subVP": ("subVP", 0.1, 1.0)}
N = 1000
F32 = np.float32


def make(kind):
    t, a, b = KINDS[kind]
    return {"VP": S.VPSDE, "VE": S.VESDE, "subVP": S.subVPSDE}[t](a, b, N)


def close(got, want, what, ulps=2, amp=1.0):
    """|got - want| <= ulps * ulp(want) * amp; amp = the condition number of the closed form with respect to a last-bit change of its
    inputs (1 unless the formula cancels)."""
    got, want = np.asarray(got, np.float32), np.asarray(want, np.float32)
    tol = ulps * np.spacing(np.abs(want).astype(np.float32)) * amp
    bad = np.abs(got.astype(np.float64) - want.astype(np.float64)) > tol
    assert not bad.any(), f"{what}: {int(bad.sum())} of {bad.size} steps differ by more than {ulps} ulp (first at step {int(np.argmax(bad))}: {got[bad][0]} vs {want[bad][0]})"


@pytest.mark.parametrize("kind", sorted(KINDS))
@pytest.mark.parametrize("predictor", ["Reverse", "Euler", "S4"])
def test_step_coefficients_all_1000_steps_against_the_reference_tables(kind, predictor):
    g = load_golden("g3_sde_tables.npz")
    s = make(kind)
    if predictor == "S4" and kind == "subVP":
        # subVPSDE has no transition() in the reference (sde.py:672-786): its S4_solver fails with this AttributeError, and so does the product
        with pytest.raises(AttributeError):
            S.step_coefficients([s, s, s], predictor, False, 1e-4)
        return
    tab = S.step_coefficients([s, s, s], predictor, False, 1e-4)
    assert tab.shape == (N, 3, 10) and tab.dtype == np.float32
    assert np.array_equal(tab[:, 0], tab[:, 1]) and np.array_equal(tab[:, 0], tab[:, 2])
    c = tab[:, 0]
    ve = KINDS[kind][0] == "VE"
    std = g[f"{kind}/marginal_std"].astype(F32)
    sscale = np.ones(N, F32) if ve else (F32(-1.0) / std).astype(F32)              # losses.py:157-163
    alpha = np.ones(N, F32) if ve else g[f"{kind}/alphas"][g[f"{kind}/timestep_idx"]]   # solver.py:752-756
    close(c[:, 0], sscale, f"{kind} sscale")
    assert np.array_equal(c[:, 1], alpha), f"{kind} Langevin alpha (timestep lookup) differs"
    if predictor == "Reverse":
        f1 = (F32(2.0) * g[f"{kind}/disc_f"].reshape(N)).astype(F32)             # f(v = 1)
        G = g[f"{kind}/disc_G"].astype(F32)
        close(c[:, 2], (F32(1.0) - f1).astype(F32), f"{kind} Reverse pa")
        close(c[:, 3], ((G * G) * F32(1.0) * sscale).astype(F32), f"{kind} Reverse pb")
        assert np.array_equal(c[:, 4], G), f"{kind} Reverse pc = G"
        # probability flow: half the score term, no noise (sde.py:329-340)
        pf = S.step_coefficients([s, s, s], "Reverse", True, 1e-4)[:, 0]
        close(pf[:, 3], ((G * G) * F32(0.5) * sscale).astype(F32), f"{kind} Reverse pflow pb")
        assert not pf[:, 4].any() and np.array_equal(pf[:, 2], c[:, 2])
    elif predictor == "Euler":
        dt = F32(-1.0 / N)
        drift1 = (F32(2.0) * g[f"{kind}/sde_drift"].reshape(N)).astype(F32)
        gd = g[f"{kind}/sde_diffusion"].astype(F32)
        close(c[:, 2], (F32(1.0) + drift1 * dt).astype(F32), f"{kind} Euler pa")
        close(c[:, 3], (-(gd * gd) * F32(1.0) * dt * sscale).astype(F32), f"{kind} Euler pb", ulps=6)
        close(c[:, 4], (gd * F32(np.sqrt(1.0 / N))).astype(F32), f"{kind} Euler pc", ulps=4)
        with pytest.raises(TypeError):             # the reference's own failure for Euler + probability flow (sde.py:301, solver.py:284)
            S.step_coefficients([s, s, s], "Euler", True, 1e-4)
    else:
        dt = F32(-1.0 / N)
        gd = g[f"{kind}/sde_diffusion"].astype(F32)
        close(c[:, 5], (F32(2.0) * g[f"{kind}/trans_mean"].reshape(N)).astype(F32), f"{kind} S4 m1 (transition over dt / 2)")
        # VE: s1 = sqrt(sigma(t)^2 - sigma(t + dt/2)^2) (sde.py:664-669) cancels two nearly equal squares -- a last-bit change of
        # sigma moves it by sigma^2 / s1^2 ulps (~400 at 1000 scales); VP's closed form does not cancel
        ts1 = g[f"{kind}/trans_std"].astype(np.float64)
        amp = (std.astype(np.float64) / ts1) ** 2 if ve else 1.0
        close(c[:, 6], g[f"{kind}/trans_std"], f"{kind} S4 s1", ulps=4, amp=amp)
        close(c[:, 7], (-(gd * gd) * sscale * dt).astype(F32), f"{kind} S4 score drift", ulps=6)
        assert not c[:, 2:5].any()
        # second half-step kernel at t + dt / 2: same closed forms, checked against the product's own SDE class at that time (the
        # class itself is pinned by the rows above and by test_oracle_golden's g3 comparison of the oracle)
        ts = torch.linspace(1.0, 1e-4, N)
        m2, s2 = s.transition(torch.ones(N, 1, 1), ts + float(dt) / 2, torch.ones(N) * (float(dt) / 2))
        close(c[:, 8], m2.reshape(N).numpy(), f"{kind} S4 m2")
        close(c[:, 9], s2.numpy(), f"{kind} S4 s2", ulps=4, amp=amp)


def test_mixed_sdes_take_each_targets_own_table():
    """x: VP, adj / rank2: VE (the zinc250k / ENZYMES set-ups): every target's column comes from ITS sde (solver.py:1028-1036); the S4
    Langevin alpha is looked up with sde_x's index for every target (solver.py:1296)."""
    vp, ve = make("VP"), make("VE2")
    for pred in ("Reverse", "Euler", "S4"):
        mixed = S.step_coefficients([vp, ve, ve], pred, False, 1e-4)
        assert np.array_equal(mixed[:, 0], S.step_coefficients([vp, vp, vp], pred, False, 1e-4)[:, 0])
        assert np.array_equal(mixed[:, 1], S.step_coefficients([ve, ve, ve], pred, False, 1e-4)[:, 1])
        assert np.array_equal(mixed[:, 1], mixed[:, 2])
