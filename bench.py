"""bench.py -- sampled complexes/sec of the CCSD reverse-SDE predictor-corrector sampler on MI355X.

Workload (BASELINE.json configs[2], the configuration the metric is quoted on): qm9_CC checkpoint,
N=9, F=4, E=36, K=466, batch 1024 complexes per GPU, VE SDEs, Reverse predictor + Langevin corrector
(snr 0.2, scale_eps 0.7, n_steps 1), eps 1e-4, QM9 node-count flag mix, in-kernel Philox noise.
A "step" is one PC step (corrector + predictor = 2 joint score evaluations + 2 state updates) over the
whole batch.  value = complexes / (time of 1000 such steps) = B_total / (ms_per_step).

  python bench.py [--gpus N] [--steps K] [--warmup W]
N > 1 is launched by torch.distributed.run, one rank per GPU; the batch dimension is sharded (weak
scaling, 1024 per rank, per-shard Langevin norms like the reference's divide_batch) with no per-step
collective and one RCCL all-gather of the samples at the end, inside the timed region.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

QM9_HIST = {9: 10949, 8: 1757, 7: 294, 6: 60, 5: 15, 4: 5, 3: 1, 2: 1}   # data/qm9_test_nx.pkl node counts
FLOP_X, FLOP_A, FLOP_F = 71_424, 10_710_522, 3_220_992   # dense-as-written GEMM FLOPs / complex / forward
FLOP_PER_UNIT = 2 * (FLOP_X + FLOP_A + FLOP_F)            # per complex per PC step
BYTES_PER_UNIT = 4 * 16_893 * 4                           # 2 x (read + write) of x, adj, rank2
PEAK_F32_MFMA_TFLOPS = 157.3                              # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 dense peak
PEAK_HBM_GBPS = 8000.0
# Kernels of the step (profiles/README.md): algorithmic work of ONE launch for a batch of B complexes (SURVEY 8d figures).
# k_xa evaluates ScoreNetworkX + ScoreNetworkA_CC once per launch (fp32 MFMA / issue bound): dominant kernel.
# k_r2 is the rank-2 side: reads rank2 (E*K fp32) once and writes it once per launch (HBM bound), ScoreNetworkF in between.
KERNELS = {
    "k_xa": {"bound": "mfma", "flops_per_complex": FLOP_X + FLOP_A},
    "k_r2": {"bound": "hbm", "bytes_per_complex": 2 * 36 * 466 * 4, "flops_per_complex": FLOP_F},
}
DOMINANT = "k_xa"
PMC_FILE = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")   # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes


def pmc_traffic(kernel: str):
    """HBM bytes per launch from the committed PMC passes (FETCH_SIZE doubled: gfx950 reports half of a wide
    coalesced read, MI355X_MICROARCH.md section HBM; WRITE_SIZE as is); None when the file is absent."""
    try:
        d = json.load(open(PMC_FILE))[kernel]
        return {"bytes": d["hbm_bytes_per_launch"], "source": "profiles/r01_pmc_traffic.json"}
    except Exception:
        return None


def qm9_flags(B: int, seed: int = 42) -> torch.Tensor:
    rs = np.random.RandomState(seed)
    ks = np.array(list(QM9_HIST.keys()))
    p = np.array(list(QM9_HIST.values()), dtype=np.float64)
    counts = rs.choice(ks, size=B, p=p / p.sum())
    f = torch.zeros(B, 9)
    for b, c in enumerate(counts):
        f[b, :c] = 1.0
    return f


def log(msg: str):
    print(f"[bench] {msg}", file=sys.stderr, flush=True)


def host_threads() -> int:
    """CPU share of this process: affinity mask, capped by the cgroup quota and by 16 (the GPU box's share)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, 16))


def load_qm9():
    from tests.helpers import load_ckpt_np

    return load_ckpt_np("ccsd_qm9_CC")


def cpu_baseline(B: int, steps: int = 2, warm: int = 1):
    """The oracle (CPU restatement certified bit-identical to the reference) timed on the host cores."""
    from oracle import ccsd_oracle as O

    threads = host_threads()
    torch.set_num_threads(threads)
    log(f"cpu_baseline: oracle on {threads} host threads, B={B}")
    meta, parts = load_qm9()
    cfg = meta["config"]
    names = ["x", "adj", "rank2"]
    so = [O.load_sde(cfg["sde"][p]) for p in names]
    nets = [(lambda x, a, r, f, p=p: O.run_network(meta[f"params_{p}"], parts[p], x, a, r, f)) for p in names]
    flags = qm9_flags(B)
    kw = dict(sde_x=so[0], sde_adj=so[1], sde_rank2=so[2], shape_x=(B, 9, 4), shape_adj=(B, 9, 9), shape_rank2=(B, 36, 466),
              predictor="Reverse", corrector="Langevin", snr=0.2, scale_eps=0.7, n_steps=1, continuous=True, denoise=True,
              eps=1e-4, is_cc=True, d_min=3, d_max=9, keep_traj=False)
    torch.manual_seed(0)
    t0 = time.perf_counter()
    O.get_pc_sampler(n_diff_steps=warm, **kw)(*nets, flags)
    log(f"cpu_baseline: warm-up step took {time.perf_counter() - t0:.1f} s")
    t0 = time.perf_counter()
    O.get_pc_sampler(n_diff_steps=steps, **kw)(*nets, flags)
    dt = (time.perf_counter() - t0) / steps
    log(f"cpu_baseline: {dt:.2f} s / PC step")
    return {"value": B / (dt * 1000.0), "unit": "complexes/s at 1000 PC steps", "cores": threads, "kind": "port",
            "sample": f"oracle (torch CPU, {threads} threads), B={B}, {steps} PC steps after {warm} warm-up, scaled to 1000 steps"}


def roofline_obj(kname, ktimes, B, dt):
    if kname not in ktimes or not ktimes[kname][0]:
        return None
    launches, kms = ktimes[kname]
    avg_s = kms / launches * 1e-3
    k = KERNELS[kname]
    tr = pmc_traffic(kname)
    o = {"kernel": kname, "bound": k["bound"], "launches": launches, "avg_launch_us": avg_s * 1e6,
         "share_of_step": kms / (dt * 1e3), "traffic": tr["bytes"] if tr else None}
    if k["bound"] == "mfma":
        a = k["flops_per_complex"] * B / avg_s / 1e12
        o.update(achieved=a, peak=PEAK_F32_MFMA_TFLOPS, unit="TFLOP/s", frac=a / PEAK_F32_MFMA_TFLOPS,
                 note="achieved = dense-as-written GEMM FLOPs per launch (SURVEY 8d: ScoreNetworkX + ScoreNetworkA_CC) / "
                      "mean launch time (HIP events on the launch stream); fp32 MFMA peak")
    else:
        a = k["bytes_per_complex"] * B / avg_s / 1e9
        o.update(achieved=a, peak=PEAK_HBM_GBPS, unit="GB/s", frac=a / PEAK_HBM_GBPS,
                 note="achieved = algorithmic bytes per launch (read + write of rank2) / mean launch time")
    return o


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=1024, help="complexes per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true", help="do not time the dominant kernel with HIP events")
    args = ap.parse_args()

    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    torch.cuda.set_device(local)
    dev = f"cuda:{local}"
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device(dev))

    import __graft_entry__ as ge

    if rank == 0:
        ge.build()
    if world > 1:
        dist.barrier()
    from ccsd_amd import loader
    from ccsd_amd.engine import PCEngine

    meta, parts = load_qm9()
    cfg = meta["config"]
    names = ["x", "adj", "rank2"]
    sdes = [loader.load_sde(cfg["sde"][p]) for p in names]
    B = args.batch
    eng = PCEngine(meta["params_x"], parts["x"], meta["params_adj"], parts["adj"], meta["params_rank2"], parts["rank2"],
                   N=9, F=4, is_cc=True, d_min=3, d_max=9, sdes=sdes, predictor="Reverse", corrector="Langevin", snr=0.2,
                   scale_eps=0.7, n_steps=1, probability_flow=False, denoise=True, eps=1e-4, device=dev)
    flags = qm9_flags(B * world)[rank * B:(rank + 1) * B].to(dev)
    state, scratch, result = eng.alloc_state(B), eng.alloc_state(B), eng.alloc_state(B)
    gathered = [torch.empty((world,) + tuple(t.shape), device=dev) for t in result] if world > 1 else None
    seed, off = 42, rank * B
    diff = eng.diff_steps

    def run_steps(k0, k1):
        s = k0
        while s < k1:       # K may exceed the 1000 scales: wrap around
            a, b = s % diff, min(diff, s % diff + (k1 - s))
            eng.run(flags, state, scratch, result, seed, off, a, b)
            s += b - a

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    log(f"rank {rank}/{world}: plan built, B={B} per GPU, warm-up {args.warmup} steps")
    eng.init_state(flags, state, None, seed, off)
    run_steps(0, args.warmup)
    eng.init_state(flags, state, None, seed, off)          # the timed region starts from a fresh prior, inputs resident
    if not args.no_kernel_events:
        for kname in KERNELS:                              # HIP events around those kernels' launches, on their stream
            eng.profile_kernel(kname)
    sync()
    t0 = time.perf_counter()
    run_steps(0, args.steps)
    if world > 1:                                          # final sample collection
        for g, t in zip(gathered, result):
            dist.all_gather_into_tensor(g, t)
    sync()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = tmax.item()
    log(f"timed region: {args.steps} steps in {dt:.3f} s")
    ktimes = {k: eng.profile_read(k) for k in KERNELS} if not args.no_kernel_events else {}
    eng.profile_kernel(None)
    ok = all(torch.isfinite(t).all().item() for t in result)

    if rank == 0:
        ms_per_step = dt * 1e3 / args.steps
        value = (B * world) / (ms_per_step)                # complexes per (1000 steps x ms_per_step / 1000 s)
        units_per_s = B * world * args.steps / dt          # complex-steps per second
        line = {
            "metric": "sampled complexes/sec at 1000 PC steps, QM9_CC batch=1024, 1/2/4/8 GPU",
            "value": value, "unit": "complexes/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic (QM9 node-count histogram flags, Philox N(0,1) prior/noise; shipped ccsd_qm9_CC weights)",
            "config": {"workload": "qm9_CC N=9 F=4 E=36 K=466, B=1024 per GPU, VE x3, Reverse+Langevin snr=0.2 scale_eps=0.7 n_steps=1, 1000 scales",
                       "global_batch": B * world, "parallelism": f"batch-sharded x{world}, per-shard Langevin norms, all-gather at end",
                       "finite": ok},
            "roofline": roofline_obj(DOMINANT, ktimes, B, dt),
            "roofline_k_r2": roofline_obj("k_r2", ktimes, B, dt),
            "achieved_model_tflops": FLOP_PER_UNIT * units_per_s / 1e12,
            "achieved_state_gbps": BYTES_PER_UNIT * units_per_s / 1e9,
            "cpu_baseline": None,
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(B)
            line["speedup_vs_cpu_baseline"] = value / line["cpu_baseline"]["value"]
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
