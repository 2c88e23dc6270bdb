"""Diagnostic (GPU box, diagnostic library of tools/dev/phase_mix.sh): per-phase instruction mix of k_xa and k_r2.
launch: one predictor half-step per stop point (the grid ends at that stamp);  report: difference the per-dispatch counters."""
import sys, os, csv, glob
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

# stamps every thread of the workgroup reaches, in program order (k_xa: slots of ccsd_k_xa.h / ccsd_attn_stack.inc; k_r2: ccsd_k_r2.h)
XA_STOPS = [(1, "X-net"), (2, "-> L0 start"), (3, "L0 gcn/att"), (4, "L0 edge MLP"), (5, "-> L1 start"), (6, "L1 gcn/att"), (7, "L1 edge MLP"),
            (8, "-> L2 start"), (9, "L2 gcn/att"), (10, "L2 edge MLP"), (12, "-> hodge start"), (13, "hodge branch"), (15, "final MLP chain"), (14, "epilogue")]
R2_STOPS = [(8, "mask tables + first batch issued"), (9, "block load loop"), (1, "(adjacency powers) + barrier"), (2, "tables/prep"), (3, "phase 1 (H, P tiles)"), (4, "phase 2 (HF tiles + epilogue)"), (5, "store")]

ABLATIONS = [(0, "full predictor launch"), (1, "- projection k loops"), (2, "- H k loops"), (8, "- noise in the epilogue"), (4, "- column-tile epilogue"),
             (16, "- column tiles"), (1 | 2 | 16, "- all tiles: load + tables + store left")]
if sys.argv[1] == "launch":
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
                   N=9, F=cfg["data"]["max_feat_num"], is_cc=True, d_min=cfg["data"]["d_min"], d_max=cfg["data"]["d_max"], sdes=sdes,
                   predictor="Reverse", corrector="Langevin", snr=0.2, scale_eps=0.7, n_steps=1, denoise=True, eps=1e-4, device="cuda:0", batch_hint=B)
    flags = bench.hist_flags(B, 9, bench.QM9_HIST).cuda()
    st, sc, rs = eng.alloc_state(B), eng.alloc_state(B), eng.alloc_state(B)
    eng.init_state(flags, st, None, 1, 0)
    dbg = torch.zeros(B + 256, 64, dtype=torch.int64, device="cuda")
    eng.lib.check(eng.lib.ccsd_debug_stamps(eng.handle, C.c_void_p(dbg.data_ptr())))
    for _ in range(3):
        eng.predictor(5, st, flags, None, 1, 0, sc, None)        # warm-up, full launches
    torch.cuda.synchronize()
    for slot, _ in XA_STOPS:
        dbg[B + 255, 31] = slot + 1                               # k_xa's dbg pointer is 32 slots in: row B + 254, slot 63 + 32
        torch.cuda.synchronize()
        eng.predictor(5, st, flags, None, 1, 0, sc, None)
        torch.cuda.synchronize()
    dbg[B + 255, 31] = 0
    for slot, _ in R2_STOPS:
        dbg[B + 254, 63] = slot + 1
        torch.cuda.synchronize()
        eng.predictor(5, st, flags, None, 1, 0, sc, None)
        torch.cuda.synchronize()
    dbg[B + 254, 63] = 0
    torch.cuda.synchronize()
    # k_r2 ablations of a full predictor launch (diagnostic bits, see ccsd_k_r2.h): 0 is the reference launch
    for bits, _ in ABLATIONS:
        dbg[B + 254, 62] = bits
        torch.cuda.synchronize()
        eng.predictor(5, st, flags, None, 1, 0, sc, None)
        torch.cuda.synchronize()
    dbg[B + 254, 62] = 0
    torch.cuda.synchronize()
    # full launches of the other modes: score (no noise), norms, predictor, then two steps of the production loop (merged k_r2)
    eng.score(2, st[0], st[1], st[2], flags)
    sums = torch.zeros(6, device="cuda")
    eng.corrector_norms(5, 0, st, st, flags, None, 1, 0, sums)
    eng.predictor(5, st, flags, None, 1, 0, sc, None)
    torch.cuda.synchronize()
    eng.run(flags, st, sc, rs, 1, 0, 0, 2)
    torch.cuda.synchronize()
    print("launched", len(XA_STOPS) + len(R2_STOPS), "stop points")
else:
    import numpy as np
    tot = {}
    for d in sys.argv[2:]:
        f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
        rows = list(csv.DictReader(open(f)))
        per = {}
        for r in rows:
            kn = "k_xa" if "k_xa" in r["Kernel_Name"] else "k_r2" if "k_r2" in r["Kernel_Name"] else None
            if kn is None:
                continue
            per.setdefault((kn, int(r["Dispatch_Id"])), {})[r["Counter_Name"]] = float(r["Counter_Value"])
        for kn in ("k_xa", "k_r2"):
            ids = sorted(i for (k, i) in per if k == kn)
            tot.setdefault(kn, [dict() for _ in ids])
            for n, i in enumerate(ids):
                tot[kn][n].update(per[(kn, i)])
    nx, nr = len(XA_STOPS), len(R2_STOPS)
    for kn, stops, sel in (("k_xa", XA_STOPS, lambda L: L[3:3 + nx]), ("k_r2", R2_STOPS, lambda L: L[3 + nx:3 + nx + nr])):
        L = tot[kn]
        assert len(L) >= 3 + nx + nr, (kn, len(L))
        runs = sel(L)
        names = sorted(runs[0])
        print(f"== {kn}: counters per launch up to each stop point, differenced into phases (thousands of wave-instructions / quad-cycles); full launch in the last row")
        print(f"{'phase':32s}" + "".join(f"{n.replace('SQ_', '').replace('INSTS_', '')[:14]:>15s}" for n in names))
        prev = {n: 0.0 for n in names}
        for (slot, label), r in zip(stops, runs):
            print(f"{label:32s}" + "".join(f"{(r[n] - prev[n]) / 1e3:15.0f}" for n in names))
            prev = r
        full = L[2]
        print(f"{'(whole launch)':32s}" + "".join(f"{full[n] / 1e3:15.0f}" for n in names))
        na = len(ABLATIONS)
        if kn == "k_r2":
            print("-- ablations of the predictor launch (difference to the full launch)")
            ab = L[3 + nx + nr:3 + nx + nr + na]
            for (bits, label), r in zip(ABLATIONS, ab):
                print(f"{label:32s}" + "".join(f"{(r[n] - (ab[0][n] if bits else 0)) / 1e3:15.0f}" for n in names))
        L = L[:3 + nx + nr] + L[3 + nx + nr + na:]
        print("-- full launches after the stop points, in launch order (k_r2: score(rank2), norms, predictor, then the production loop's; k_xa: norms, predictor, loop)")
        for n_, r in enumerate(L[3 + nx + nr:]):
            print(f"{'launch ' + str(n_):32s}" + "".join(f"{r[n] / 1e3:15.0f}" for n in names))
