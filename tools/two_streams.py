"""Diagnostic (GPU box): do two half-batch sampler runs on two HIP streams overlap (k_r2 of one under k_xa of the other)?
Compares one engine at B = 1024 with two engines at B = 512 on separate streams, same total work."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from ccsd_amd import loader
from ccsd_amd.engine import PCEngine
from tests.helpers import load_ckpt_np

meta, parts = load_ckpt_np("ccsd_qm9_CC")
cfg = meta["config"]
sdes = [loader.load_sde(cfg["sde"][p]) for p in ("x", "adj", "rank2")]
STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 300


def engine(B):
    return PCEngine(meta["params_x"], parts["x"], meta["params_adj"], parts["adj"], meta["params_rank2"], parts["rank2"], N=9, F=4,
                    is_cc=True, d_min=3, d_max=9, sdes=sdes, predictor="Reverse", corrector="Langevin", snr=0.2, scale_eps=0.7,
                    n_steps=1, denoise=True, eps=1e-4, device="cuda:0", batch_hint=1024)


def setup(B, off):
    eng = engine(B)
    flags = bench.hist_flags(1024, 9, bench.QM9_HIST)[off:off + B].cuda()
    st, sc, rs = eng.alloc_state(B), eng.alloc_state(B), eng.alloc_state(B)
    eng.init_state(flags, st, None, 1, off)
    eng._workspace(B)
    return eng, flags, st, sc, rs


def timed(fn, n=3):
    best = 1e9
    for _ in range(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best


one = setup(1024, 0)
t1 = timed(lambda: one[0].run(one[1], one[2], one[3], one[4], 1, 0, 0, STEPS))
print(f"one stream, B=1024: {t1 / STEPS * 1e3:.4f} ms / step")
a, b = setup(512, 0), setup(512, 512)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def both(skew):
    with torch.cuda.stream(s1):
        a[0].run(a[1], a[2], a[3], a[4], 1, 0, 0, STEPS)
    with torch.cuda.stream(s2):
        if skew:
            torch.cuda._sleep(int(skew))
        b[0].run(b[1], b[2], b[3], b[4], 1, 512, 0, STEPS)


for skew in (0, 100_000, 200_000):
    t2 = timed(lambda: both(skew))
    print(f"two streams, 2 x B=512, skew {skew} cycles: {t2 / STEPS * 1e3:.4f} ms / step for the same 1024 complexes")
h = setup(512, 0)
th = timed(lambda: h[0].run(h[1], h[2], h[3], h[4], 1, 0, 0, STEPS))
print(f"one stream, B=512 alone: {th / STEPS * 1e3:.4f} ms / step")
