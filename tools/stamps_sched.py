"""Diagnostic: workgroup schedule of k_r2's predictor launch on qm9_CC (start / end clocks per workgroup)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np, ctypes as C
import bench
from ccsd_amd import loader
from ccsd_amd.engine import PCEngine
from tests.helpers import load_ckpt_np
meta, parts = load_ckpt_np("ccsd_qm9_CC")
cfg = meta["config"]
sdes = [loader.load_sde(cfg["sde"][p]) for p in ("x", "adj", "rank2")]
B = 1024
eng = PCEngine(meta["params_x"], parts["x"], meta["params_adj"], parts["adj"], meta["params_rank2"], parts["rank2"],
               N=9, F=4, is_cc=True, d_min=3, d_max=9, sdes=sdes, predictor="Reverse", corrector="Langevin", snr=0.2, scale_eps=0.7,
               n_steps=1, denoise=True, eps=1e-4, device="cuda:0", batch_hint=B)
flags = bench.hist_flags(B, 9, bench.QM9_HIST).cuda()
st, sc, rs = eng.alloc_state(B), eng.alloc_state(B), eng.alloc_state(B)
eng.init_state(flags, st, None, 1, 0)
eng.run(flags, st, sc, rs, 1, 0, 0, 3)
dbg = torch.zeros(B, 64, dtype=torch.int64, device="cuda")
eng.lib.check(eng.lib.ccsd_debug_stamps(eng.handle, C.c_void_p(dbg.data_ptr())))
eng.predictor(5, st, flags, None, 1, 0, sc, None)
torch.cuda.synchronize()
d = dbg.cpu().numpy()
t0 = d[:, 0].min()
start, end = d[:, 0] - t0, d[:, 5] - t0
order = np.argsort(start)
print("k_r2 launch span (cycles):", end.max(), " workgroup duration median / p10 / p90:", np.median(end - start), np.percentile(end - start, 10), np.percentile(end - start, 90))
print("start-time deciles:", [int(np.percentile(start, q)) for q in range(0, 101, 10)])
print("first-wave starts (sorted, every 32nd):", [int(start[order[i]]) for i in range(0, 1024, 32)])
for name, a, b_ in (("load+apply", 0, 1), ("prep", 1, 2), ("phase1", 2, 3), ("phase2", 3, 4)):
    v = d[:, b_] - d[:, a]
    first, later = v[start < np.percentile(start, 45)], v[start > np.percentile(start, 55)]
    print(f"{name:12s} early WGs median {np.median(first):8.0f}   late WGs median {np.median(later):8.0f}")
dur = d[:, 5] - d[:, 0]
print("duration by dispatch round (blockIdx // 256):", [int(np.median(dur[i * 256:(i + 1) * 256])) for i in range(4)])
print("duration by blockIdx % 8 (XCD under round-robin):", [int(np.median(dur[np.arange(B) % 8 == x])) for x in range(8)])
for name, a, b_ in (("load+apply", 0, 1), ("phase1", 2, 3), ("phase2", 3, 4)):
    v = d[:, b_] - d[:, a]
    print(f"{name:12s} by round:", [int(np.median(v[i * 256:(i + 1) * 256])) for i in range(4)], " p10/p90", int(np.percentile(v, 10)), int(np.percentile(v, 90)))
# per-XCD timeline: clocks of one XCD are comparable
for x in range(2):
    sel = np.arange(B) % 8 == x
    s0 = d[sel, 0].min()
    print(f"XCD {x}: span {int(d[sel, 5].max() - s0)}  starts (sorted, every 8th):", [int(v) for v in np.sort(d[sel, 0] - s0)[::8]])
