"""Diagnostic: per-phase shader-clock stamps of k_r2 / k_xa on the qm9_CC workload (run on the GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np, ctypes as C
import bench
from ccsd_amd import loader
from ccsd_amd.engine import PCEngine

from tests.helpers import load_ckpt_np
meta, parts = load_ckpt_np(os.environ.get("STAMPS_CKPT", "ccsd_qm9_CC"))   # STAMPS_CKPT=ccsd_qm9_Base_CC: the ablation checkpoint
cfg = meta["config"]
is_cc = bool(meta.get("is_cc", True))
sdes = [loader.load_sde(cfg["sde"][p]) for p in (("x", "adj", "rank2") if is_cc else ("x", "adj"))]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
Nn, Ff = cfg["data"]["max_node_num"], cfg["data"]["max_feat_num"]
eng = PCEngine(meta["params_x"], parts["x"], meta["params_adj"], parts["adj"], meta.get("params_rank2") if is_cc else None, parts.get("rank2") if is_cc else None,
               N=Nn, F=Ff, is_cc=is_cc, d_min=cfg["data"].get("d_min", 0) if is_cc else 0, d_max=cfg["data"].get("d_max", 0) if is_cc else 0, sdes=sdes, predictor="Reverse",
               corrector="Langevin", snr=0.2, scale_eps=0.7, n_steps=1, denoise=True, eps=1e-4, device="cuda:0", batch_hint=B)
flags = (bench.hist_flags(B, 9, bench.QM9_HIST) if Nn == 9 else bench.hist_flags(B, Nn, {Nn: 3, Nn - 2: 2, Nn - 5: 1})).cuda()
st, sc, rs = eng.alloc_state(B), eng.alloc_state(B), eng.alloc_state(B)
eng.init_state(flags, st, None, 1, 0)
eng.run(flags, st, sc, rs, 1, 0, 0, int(os.environ.get("STAMPS_WARM_STEPS", "3")))      # (sustained-load clocks: STAMPS_WARM_STEPS=300)
dbg = torch.zeros(B + 256, 64, dtype=torch.int64, device="cuda")   # (+256 rows: barrier arrival tables of the diagnostic build)
eng.lib.check(eng.lib.ccsd_debug_stamps(eng.handle, C.c_void_p(dbg.data_ptr())))
eng.predictor(5, st, flags, None, 1, 0, sc, None)
torch.cuda.synchronize()
dfull = dbg.cpu().numpy()
d = dfull[:B]
names_r2 = ["load", "prep(masks,acoef,u)", "->phase1", "gemm tiles (H,P)", "HF+epilogue", "store"]
print("k_r2 per-phase cycles (median over workgroups), total", np.median(d[:, 5] - d[:, 0]))
for i in range(5):
    print(f"  {names_r2[i]:24s} {np.median(d[:, i + 1] - d[:, i]):10.0f}")
print("phase 1 (wave 0): k loop + epilogue", int(np.median(d[:, 6] - d[:, 2])), " loop exit", int(np.median(d[:, 7] - d[:, 6])), " barrier wait", int(np.median(d[:, 3] - d[:, 7])))
x = d[:, 32:]
lab = {0: "start", 1: "X-net done", 2: "L0 start", 3: "L0 gcn/att done", 4: "L0 edge MLP done", 5: "L1 start", 6: "L1 gcn/att done",
       7: "L1 edge MLP done", 8: "L2 start", 9: "L2 gcn/att done", 10: "L2 edge MLP done", 12: "hodge start", 13: "final MLP start", 14: "end"}
print("k_xa total", np.median(x[:, 14] - x[:, 0]))
prev = 0
for k in sorted(lab):
    if k == 0:
        continue
    print(f"  {lab[prev]:18s} -> {lab[k]:18s} {np.median(x[:, k] - x[:, prev]):10.0f}")
    prev = k
span = (d[:, 5].max() - d[:, 0].min())
hs = [12, 16, 17, 18, 19, 20, 13]
print("hodge: fill/hq0, dense pairs, deg, MFMA proj, diag att, scatter:", [int(np.median(x[:, hs[i + 1]] - x[:, hs[i]])) for i in range(6)])
print("X-network: inputs + conv layers, head (final MLP), epilogue:", [int(np.median(x[:, b2] - x[:, a2])) for a2, b2 in ((0, 26), (26, 27), (27, 1))])
print("final MLP: chain (wave 0)", int(np.median(x[:, 11] - x[:, 13])), " wait barrier", int(np.median(x[:, 15] - x[:, 11])), " epilogue", int(np.median(x[:, 14] - x[:, 15])))
print("k_r2 first-start to last-end cycles:", span, " k_xa:", x[:, 14].max() - x[:, 0].min())
for nm, v, s0, s1 in (("k_r2", d[:, :32], 0, 5), ("k_xa", x, 0, 14)):
    ok = v[:, s0] > 0
    if not ok.any():
        continue            # (a plan without the fused rank-2 kernel has no k_r2 stamps)
    cyc = float(v[ok, s1].max() - v[ok, s0].min())
    rt = float(v[ok, 31].max() - v[ok, 30].min()) / 100.0        # us (100 MHz counter)
    print(f"{nm}: span {cyc:.0f} cycles = {rt:.1f} us of real time -> shader clock {cyc / rt / 1000:.2f} GHz; per-workgroup median {np.median(v[:, s1] - v[:, s0]):.0f} cycles")
if os.environ.get("STAMPS_CKPT", "").endswith("Base_CC"):
    hb = [12, 16, 17, 18, 19, 20, 21, 13]
    print("baseline hodge: stage+hidden0, (to chunk 0), fill S, mlp_hodge chain, hidden rows + diag of layer 1, remaining chunks, head+scatter:",
          [int(np.median(x[:, hb[i + 1]] - x[:, hb[i]])) for i in range(7)], " rows per chunk", "see plan")
print("layer 1, first channel group: bias fill + dinv, gcn tiles (+barrier), attention pairs (wave 0), multi_channel accumulate (+barrier):", [int(np.median(x[:, b] - x[:, a])) for a, b in ((5, 22), (22, 23), (23, 25), (25, 24))])
print("layer 1 rest: edge MLP chain + node linear (+barrier), symmetrise/tanh:", [int(np.median(x[:, b] - x[:, a])) for a, b in ((24, 6), (6, 7))])
if os.environ.get("STAMPS_BARRIERS"):
    life = x[:, 14] - x[:, 0]
    print("k_xa barriers passed (wave 0):", int(np.median(x[:, 21])), " share of life inside __syncthreads(), waves 0..3:",
          [round(float(np.median(x[:, 26 + k] / life)), 3) for k in range(4)])
    nb = int(np.median(x[:, 21]))
    arr = dfull[B:B + 256].reshape(64, 4, 64)[:, :, :nb].astype(np.float64)     # [workgroup][wave][barrier]
    rel = arr.max(axis=1)                                                       # release ~ last arrival
    start = x[:64, 0].astype(np.float64)
    prev = np.concatenate([start[:, None], rel[:, :-1]], axis=1)
    dur = np.median(rel - prev, axis=0)
    busy = np.median(arr - prev[:, None, :], axis=0)                            # [wave][barrier]: time from the previous release to arrival
    print("interval  duration   busy(w0 w1 w2 w3)   (cycles, median of 64 workgroups; interval k ends at the k-th __syncthreads)")
    for k in range(nb):
        print(f"  {k:3d} {dur[k]:9.0f}   " + " ".join(f"{busy[w, k]:7.0f}" for w in range(4)) + f"   at {np.median(rel[:, k] - start):8.0f}")
    ct = dfull[B + 255, :16].astype(np.int64)
    print("layer 1 edge-MLP interval, workgroup 0 wave 0: chain tile (entry, gather, linear 1, middle, last + epilogue), node MLP second Linear:", [int(ct[i + 1] - ct[i]) for i in range(0, 5)], int(ct[7] - ct[6]), int(ct[8] - ct[7]))
life = x[:, 14] - x[:, 0]
print("k_xa workgroup life percentiles (min, 10, 50, 90, 99, max):", [int(np.percentile(life, q)) for q in (0, 10, 50, 90, 99, 100)])
lr = d[:, 5] - d[:, 0]
print("k_r2 workgroup life percentiles (min, 10, 50, 90, 99, max):", [int(np.percentile(lr, q)) for q in (0, 10, 50, 90, 99, 100)])
hw = x[:, 26:30].astype(np.int64)
simd = (hw >> 4) & 3
cu = (hw[:, 0] >> 8) & 15
se = (hw[:, 0] >> 13) & 7        # (gfx9 HW_ID: [3:0] wave, [5:4] simd, [7:6] pipe, [11:8] cu, [12] sh, [15:13] se)
xcc = (hw[:, 0] >> 32) & 15
distinct = np.array([len(set(r)) for r in simd])
for k in (4, 3, 2, 1):
    m = distinct == k
    if m.any():
        print(f"workgroups with waves on {k} distinct SIMDs: {int(m.sum())}, median life {int(np.median(life[m]))}, p90 {int(np.percentile(life[m], 90))}")
slow = life > np.percentile(life, 85)
print("slow workgroups (top 15 %): distinct-SIMD histogram", np.bincount(distinct[slow], minlength=5)[1:], " all:", np.bincount(distinct, minlength=5)[1:])
key = (xcc.astype(np.int64) << 16) | (se << 8) | (((hw[:, 0] >> 12) & 1) << 4) | cu
groups = {}
for i, kk in enumerate(key):
    groups.setdefault(int(kk), []).append(i)
sizes = np.array([len(v) for v in groups.values()])
print("workgroups per (xcc, se, sh, cu):", np.bincount(sizes))
gl = np.array([[life[i] for i in v][:4] + [0] * (4 - min(4, len(v))) for v in groups.values()])
print("per-CU: median of (max life - min life) among its workgroups:", int(np.median(gl.max(1) - np.where(gl > 0, gl, 10**9).min(1))))
cs = np.array([np.mean([life[i] for i in v]) for v in groups.values()])
print("per-CU mean life percentiles (min, 50, 90, max):", [int(np.percentile(cs, q)) for q in (0, 50, 90, 100)])
labs = {0: "start", 1: "X-net done", 2: "L0 start", 5: "L1 start", 8: "L2 start", 12: "hodge start", 13: "final MLP start", 14: "end"}
ks = sorted(labs)
fast = life < np.percentile(life, 30)
print("phase durations, fast (bottom 30 %) vs slow (top 15 %) workgroups:")
for a_, b_ in zip(ks[:-1], ks[1:]):
    print(f"  {labs[a_]:16s} -> {labs[b_]:16s} fast {int(np.median(x[fast, b_] - x[fast, a_])):7d}   slow {int(np.median(x[slow, b_] - x[slow, a_])):7d}")
print("co-resident workgroups of a few CUs: (block id, life, start - min start):")
for kk in list(groups)[:6]:
    v = groups[kk]
    s0 = min(x[i, 0] for i in v)
    print("  ", sorted((int(i), int(life[i]), int(x[i, 0] - s0)) for i in v))
w0 = np.array([len(set(int(simd[i, 0]) for i in v)) for v in groups.values() if len(v) == 4])
print("co-resident workgroups: distinct SIMDs among their wave-0s (histogram over CUs, 1..4):", np.bincount(w0, minlength=5)[1:])
print("SIMD of waves 0..3 of the co-resident workgroups of a few CUs (by block id):")
for kk in list(groups)[:6]:
    print("  ", [(int(i), [int(t) for t in simd[i]]) for i in sorted(groups[kk])])
order_ok = 0
slow_is_last = 0
for v in groups.values():
    ids = sorted(v)
    if len(ids) == 4 and all((ids[j + 1] - ids[j]) % 256 == 0 for j in range(3)):
        order_ok += 1
    st = sorted(v, key=lambda i: x[i, 0])
    if life[st[-1]] == max(life[i] for i in v):
        slow_is_last += 1
print("CUs whose four block ids are congruent mod 256:", order_ok, "of", len(groups), "; CUs whose last-started workgroup is the slowest:", slow_is_last)
print("hodge MFMA projection (wave 0): operands ready after", int(np.median(x[:, 22] - x[:, 18])), ", first row tile", int(np.median(x[:, 23] - x[:, 22])), ", rest + barrier", int(np.median(x[:, 19] - x[:, 23])))
