"""Diagnostic (GPU box): phase stamps of the LAST k_r2 launch of a ccsd_sampler_run step = the predictor launch with the fused
corrector apply (mode "pred"), or of a norms launch (mode "norms": step-wise corrector_norms call)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, numpy as np, ctypes as C
import bench
from ccsd_amd import loader
from ccsd_amd.engine import PCEngine
from tests.helpers import load_ckpt_np

mode = sys.argv[1] if len(sys.argv) > 1 else "pred"
meta, parts = load_ckpt_np("ccsd_qm9_CC")
cfg = meta["config"]
sdes = [loader.load_sde(cfg["sde"][p]) for p in ("x", "adj", "rank2")]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
eng = PCEngine(meta["params_x"], parts["x"], meta["params_adj"], parts["adj"], meta["params_rank2"], parts["rank2"], N=9, F=4, is_cc=True,
               d_min=3, d_max=9, sdes=sdes, predictor="Reverse", corrector="Langevin", snr=0.2, scale_eps=0.7, n_steps=1, denoise=True,
               eps=1e-4, device="cuda:0", batch_hint=B)
flags = bench.hist_flags(B, 9, bench.QM9_HIST).cuda()
st, sc, rs = eng.alloc_state(B), eng.alloc_state(B), eng.alloc_state(B)
eng.init_state(flags, st, None, 1, 0)
eng.run(flags, st, sc, rs, 1, 0, 0, 200)
dbg = torch.zeros(B + 256, 64, dtype=torch.int64, device="cuda")
eng.lib.check(eng.lib.ccsd_debug_stamps(eng.handle, C.c_void_p(dbg.data_ptr())))
if mode == "pred":
    eng.run(flags, st, sc, rs, 1, 0, 200, 203)
else:
    sums = torch.zeros(8, device="cuda")
    eng.run(flags, st, sc, rs, 1, 0, 200, 202)
    eng.corrector_norms(202, 0, st, st, flags, None, 1, 0, sums)
torch.cuda.synchronize()
d = dbg.cpu().numpy()[:B, :32].astype(np.float64)
t0 = d[:, 0].min()
names = ["phase 0: mask tables + barrier", "phase 0: block load loop (thread 0)", "phase 0: barrier after the load", "-> phase 1 start", "phase 1 (own tasks)", "wait for H", "phase 2 (column tiles, leftovers)", "norm sums"]
slots = [(0, 8), (8, 9), (9, 1), (1, 2), (2, 7), (7, 3), (3, 4), (4, 5)]
start = d[:, 0] - t0
order = np.argsort(start)
first = order[:min(512, B)]; second = order[512:] if B > 512 else order[:1]
rt = (dbg.cpu().numpy()[:B, 31].max() - dbg.cpu().numpy()[:B, 30].min()) / 100.0
span = d[:, 5].max() - t0
print(f"mode {mode}: span {span:.0f} cycles = {rt:.1f} us -> {span / rt / 1000:.2f} GHz")
for nm, grp in (("first 512 workgroups", first), ("last 512 workgroups", second)):
    print(nm, f": start {np.median(start[grp]):.0f} (min {start[grp].min():.0f}, max {start[grp].max():.0f}), life {np.median(d[grp, 5] - d[grp, 0]):.0f}")
    for (a, b), n in zip(slots, names):
        print(f"    {n:42s} {np.median(d[grp, b] - d[grp, a]):9.0f}")
