"""Debug helper (GPU box): ScoreNetworkF of qm9_CC through k_r2 against the reference golden, error per row / column block."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ccsd_amd import _lib
from tests import parity_cases as pc
from tests.helpers import load_golden

lib = _lib.get_library()
g = load_golden("g1_ccsd_qm9_CC.npz")
eng, meta, _ = pc.engine_from_ckpt("ccsd_qm9_CC", lib, "cuda:0")
flags = torch.from_numpy(g["flags"])
x, adj, rank2 = pc.masked_state(int(g["seed"]), 4, 9, 4, True, 3, 9, flags)
dv = lambda t: t.to("cuda:0")
out = eng.score(2, dv(x), dv(adj), dv(rank2), dv(flags)).cpu()
want = torch.from_numpy(g["unit/net_rank2"])
err = (out - want).abs()
print("scale", want.abs().max().item(), "max err", err.max().item())
for b in range(4):
    print("sample", b, "rows 0-15:", err[b, :16].max().item(), "16-31:", err[b, 16:32].max().item(), "32-35:", err[b, 32:].max().item())
    e = err[b]
    bad = (e > 1e-3).nonzero()
    print("  bad count", len(bad), "first", bad[:8].tolist())
    print("  per row max", [round(v, 3) for v in e.max(dim=1).values.tolist()])
    cols = e.max(dim=0).values
    print("  bad col tiles", sorted(set((c // 16) for c in (cols > 1e-3).nonzero().flatten().tolist())))
