"""bench.py -- sampled complexes/sec of the CCSD reverse-SDE predictor-corrector sampler on MI355X.

Workload (BASELINE.json configs[2], the configuration the metric is quoted on): qm9_CC checkpoint,
N=9, F=4, E=36, K=466, batch 1024 complexes per GPU, VE SDEs, Reverse predictor + Langevin corrector
(snr 0.2, scale_eps 0.7, n_steps 1), eps 1e-4, QM9 node-count flag mix, in-kernel Philox noise.
A "step" is one PC step (corrector + predictor = 2 joint score evaluations + 2 state updates) over the
whole batch.  value = complexes / (time of 1000 such steps) = B_total / (ms_per_step).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload qm9_CC|community_small_CC|community_small|zinc250k|enzymes_small_CC]
(the default workload is the metric's; the others are BASELINE.json's remaining configs, same JSON line)
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
COMMUNITY_HIST = {12: 29, 14: 14, 16: 23, 18: 25, 20: 9}                 # data/community_small.pkl
PEAK_F32_MFMA_TFLOPS = 157.3                              # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 dense peak
PEAK_HBM_GBPS = 8000.0
# Workloads = BASELINE.json configs.  flop_* : dense-as-written GEMM FLOPs / complex / forward (FlopCounterMode on
# the reference).  The default (the configuration the metric is quoted on) is qm9_CC, B = 1024 per GPU.
WORKLOADS = {
    "qm9_CC": dict(ckpt="ccsd_qm9_CC", batch=1024, hist=QM9_HIST, predictor="Reverse", corrector="Langevin", snr=0.2, scale_eps=0.7,
                   flop_x=71_424, flop_a=10_710_522, flop_f=3_220_992,
                   desc="qm9_CC N=9 F=4 E=36 K=466, B={B} per GPU, VE x3, Reverse+Langevin snr=0.2 scale_eps=0.7 n_steps=1, 1000 scales"),
    "community_small_CC": dict(ckpt="ccsd_community_small_CC", batch=512, hist=COMMUNITY_HIST, predictor="Euler", corrector="Langevin",
                               snr=0.05, scale_eps=0.7, flop_x=3_014_720, flop_a=192_613_760, flop_f=168_081_600,
                               desc="community_small_CC N=20 F=11 E=190 K=1140, B={B} per GPU, VP x3, Euler+Langevin snr=0.05 scale_eps=0.7, 1000 scales"),
    "community_small": dict(ckpt="gdss_community_small", batch=16, hist=COMMUNITY_HIST, predictor="Euler", corrector="Langevin",
                            snr=0.05, scale_eps=0.7, flop_x=2_952_960, flop_a=16_628_480, flop_f=0,
                            desc="community_small (graph-only) N=20 F=10, B={B} per GPU, VP x2, Euler+Langevin snr=0.05 scale_eps=0.7, 1000 scales"),
    "zinc250k": dict(ckpt="gdss_zinc250k", batch=256, hist={38: 1, 30: 2, 24: 4, 23: 4, 20: 2}, predictor="Reverse", corrector="Langevin",
                     snr=0.2, scale_eps=0.9, flop_x=945_440, flop_a=58_489_296, flop_f=0,
                     desc="zinc250k (graph-only substitute for the infeasible zinc250k_CC, SURVEY 8d 5a) N=38 F=9, B={B} per GPU, "
                          "VP(x)/VE(adj), Reverse+Langevin snr=0.2 scale_eps=0.9, 1000 scales; synthetic node-count mix"),
    "qm9_Base_CC": dict(ckpt="ccsd_qm9_Base_CC", batch=1024, hist=QM9_HIST, predictor="Reverse", corrector="Langevin", snr=0.2,
                        scale_eps=0.7, flop_x=None, flop_a=None, flop_f=None,
                        desc="qm9_Base_CC (ScoreNetworkA_Base_CC ablation) N=9 F=4 E=36 K=466, B={B} per GPU, VE x3, Reverse+Langevin "
                             "snr=0.2 scale_eps=0.7 n_steps=1, 1000 scales"),
    "enzymes_small_CC": dict(ckpt="ccsd_enzymes_small_CC", batch=64, hist={12: 6, 11: 5, 10: 5, 9: 4, 8: 4, 6: 3, 4: 2}, predictor="S4",
                             corrector="None", snr=0.15, scale_eps=0.7, flop_x=None, flop_a=None, flop_f=None,
                             desc="ENZYMES_small_CC N=12 F=10 E=66 K=715, B={B} per GPU, VP(x)/VE/VE, S4 solver snr=0.15 scale_eps=0.7, 1000 scales"),
}
# Kernels of the step (profiles/README.md).  k_xa evaluates ScoreNetworkX + ScoreNetworkA(_CC) once per launch (fp32 MFMA /
# issue bound); the rank-2 side (ScoreNetworkF + hodge projections) reads rank2 (E*K fp32) once and writes it once per
# half-step (HBM bound): fused k_r2 when the block fits LDS (qm9_CC), k_gemm_h + k_gemm_p + k_hf_score otherwise.
KERNEL_BOUND = {"k_xa": "mfma", "k_r2": "hbm", "k_hf_score": "hbm", "k_gemm_h": "mfma", "k_gemm_p": "mfma", "k_langevin_apply": "hbm",
                "k_s4_apply": "hbm"}
PMC_FILE = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")   # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes


def pmc_traffic(kernel: str):
    """HBM bytes per launch from the committed PMC passes (FETCH_SIZE doubled: gfx950 reports half of a wide
    coalesced read, MI355X_MICROARCH.md section HBM; WRITE_SIZE as is); None when the file is absent."""
    try:
        d = json.load(open(PMC_FILE))[kernel]
        return {"bytes": d["hbm_bytes_per_launch"], "source": "profiles/r01_pmc_traffic.json"}
    except Exception:
        return None


def hist_flags(B: int, N: int, hist: dict, seed: int = 42) -> torch.Tensor:
    rs = np.random.RandomState(seed)
    ks = np.array(list(hist.keys()))
    p = np.array(list(hist.values()), dtype=np.float64)
    counts = rs.choice(ks, size=B, p=p / p.sum())
    f = torch.zeros(B, N)
    for b, c in enumerate(counts):
        f[b, :c] = 1.0
    return f


def qm9_flags(B: int, seed: int = 42) -> torch.Tensor:
    return hist_flags(B, 9, QM9_HIST, seed)


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


def workload_setup(name: str):
    """(meta, parts, names, N, F, d_min, d_max, is_cc) of a workload's checkpoint."""
    from tests.helpers import load_ckpt_np

    meta, parts = load_ckpt_np(WORKLOADS[name]["ckpt"])
    cfg, is_cc = meta["config"], meta["is_cc"]
    N, F = cfg["data"]["max_node_num"], cfg["data"]["max_feat_num"]
    d_min, d_max = (cfg["data"]["d_min"], cfg["data"]["d_max"]) if is_cc else (0, 0)
    names = ["x", "adj"] + (["rank2"] if is_cc else [])
    return meta, parts, names, N, F, d_min, d_max, is_cc


def cpu_baseline(wname: str, B: int, steps: int = 2, warm: int = 1):
    """The oracle (CPU restatement certified bit-identical to the reference) timed on the host cores, on a bounded
    sample of the same workload: the full batch, `steps` PC steps after `warm` warm-up, scaled to 1000 steps."""
    from oracle import ccsd_oracle as O

    wl = WORKLOADS[wname]
    threads = host_threads()
    torch.set_num_threads(threads)
    log(f"cpu_baseline: oracle on {threads} host threads, {wname} B={B}")
    meta, parts, names, N, F, d_min, d_max, is_cc = workload_setup(wname)
    cfg = meta["config"]
    so = [O.load_sde(cfg["sde"][p]) for p in names]
    if is_cc:
        nets = [(lambda x, a, r, f, p=p: O.run_network(meta[f"params_{p}"], parts[p], x, a, r, f)) for p in names]
    else:
        nets = [(lambda x, a, f, p=p: O.run_network(meta[f"params_{p}"], parts[p], x, a, None, f)) for p in names]
    flags = hist_flags(B, N, wl["hist"])
    kw = dict(sde_x=so[0], sde_adj=so[1], shape_x=(B, N, F), shape_adj=(B, N, N), snr=wl["snr"], scale_eps=wl["scale_eps"],
              continuous=True, denoise=True, eps=1e-4, keep_traj=False)
    if is_cc:
        kw.update(is_cc=True, sde_rank2=so[2], shape_rank2=(B, *O.get_rank2_dim(N, d_min, d_max)), d_min=d_min, d_max=d_max)
    if wl["predictor"] == "S4":
        make = lambda n: O.S4_solver(n_diff_steps=n, **kw)
    else:
        make = lambda n: O.get_pc_sampler(n_diff_steps=n, predictor=wl["predictor"], corrector=wl["corrector"], n_steps=1, **kw)
    torch.manual_seed(0)
    t0 = time.perf_counter()
    make(warm)(*nets, flags)
    log(f"cpu_baseline: warm-up step took {time.perf_counter() - t0:.1f} s")
    t0 = time.perf_counter()
    make(steps)(*nets, flags)
    dt = (time.perf_counter() - t0) / steps
    log(f"cpu_baseline: {dt:.2f} s / PC step")
    return {"value": B / (dt * 1000.0), "unit": "complexes/s at 1000 PC steps", "cores": threads, "kind": "port",
            "sample": f"oracle (torch CPU, {threads} threads), {wname} B={B}, {steps} PC steps after {warm} warm-up, scaled to 1000 steps"}


def kernel_work(wname: str, kname: str, E: int, K: int):
    """Algorithmic work of ONE launch per complex: (flops, bytes, what)."""
    wl = WORKLOADS[wname]
    if kname == "k_xa":
        f = None if wl["flop_x"] is None else wl["flop_x"] + wl["flop_a"]
        return f, None, "dense-as-written GEMM FLOPs of ScoreNetworkX + ScoreNetworkA(_CC) (SURVEY 8a)"
    if kname in ("k_r2", "k_hf_score"):
        return wl["flop_f"], 2 * E * K * 4, "read + write of rank2 (E*K fp32 each)"
    if kname in ("k_langevin_apply", "k_s4_apply"):
        return None, 3 * E * K * 4, "read state + raw score, write state (rank2 dominates)"
    return None, None, ""


def roofline_obj(wname, kname, ktimes, B, dt, E, K):
    if kname not in ktimes or not ktimes[kname][0]:
        return None
    launches, kms = ktimes[kname]
    avg_s = kms / launches * 1e-3
    flops, nbytes, what = kernel_work(wname, kname, E, K)
    tr = pmc_traffic(kname) if wname == "qm9_CC" else None
    bound = KERNEL_BOUND[kname]
    o = {"kernel": kname, "bound": bound, "launches": launches, "avg_launch_us": avg_s * 1e6,
         "share_of_step": kms / (dt * 1e3), "traffic": tr["bytes"] if tr else None}
    if bound == "mfma" and flops:
        a = flops * B / avg_s / 1e12
        o.update(achieved=a, peak=PEAK_F32_MFMA_TFLOPS, unit="TFLOP/s", frac=a / PEAK_F32_MFMA_TFLOPS,
                 note=f"achieved = {what} per launch / mean launch time (HIP events on the launch stream); fp32 MFMA peak")
    elif bound == "hbm" and nbytes:
        a = nbytes * B / avg_s / 1e9
        o.update(achieved=a, peak=PEAK_HBM_GBPS, unit="GB/s", frac=a / PEAK_HBM_GBPS,
                 note=f"achieved = algorithmic bytes per launch ({what}) / mean launch time (HIP events on the launch stream)")
    else:
        o.update(achieved=None, peak=None, unit=None, frac=None, note="no algorithmic-work figure for this kernel / workload")
    return o


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="qm9_CC", choices=sorted(WORKLOADS), help="BASELINE.json config (default: the metric's)")
    ap.add_argument("--batch", type=int, default=0, help="complexes per GPU (default: the workload's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true", help="do not time the kernels with HIP events")
    ap.add_argument("--event-stride", type=int, default=17, help="bracket every n-th launch of a kernel with HIP events (odd: a kernel's norms / predictor launches alternate)")
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

    wname = args.workload
    wl = WORKLOADS[wname]
    meta, parts, names, N, F, d_min, d_max, is_cc = workload_setup(wname)
    cfg = meta["config"]
    sdes = [loader.load_sde(cfg["sde"][p]) for p in names]
    B = args.batch or wl["batch"]
    eng = PCEngine(meta["params_x"], parts["x"], meta["params_adj"], parts["adj"], meta.get("params_rank2"), parts.get("rank2"),
                   N=N, F=F, is_cc=is_cc, d_min=d_min, d_max=d_max, sdes=sdes, predictor=wl["predictor"], corrector=wl["corrector"],
                   snr=wl["snr"], scale_eps=wl["scale_eps"], n_steps=1, probability_flow=False, denoise=True, eps=1e-4, device=dev,
                   batch_hint=B)
    E, K = eng.E, eng.K
    flags = hist_flags(B * world, N, wl["hist"])[rank * B:(rank + 1) * B].to(dev)
    state, scratch, result = eng.alloc_state(B), eng.alloc_state(B), eng.alloc_state(B)
    outs = [t for t in result if t is not None]
    gathered = [torch.empty((world,) + tuple(t.shape), device=dev) for t in outs] if world > 1 else None
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

    log(f"rank {rank}/{world}: {wname} plan built, B={B} per GPU, warm-up {args.warmup} steps")
    eng.init_state(flags, state, None, seed, off)
    run_steps(0, args.warmup)
    eng.init_state(flags, state, None, seed, off)          # the timed region starts from a fresh prior, inputs resident
    knames = ["k_xa", "k_r2", "k_hf_score", "k_gemm_h", "k_gemm_p", "k_langevin_apply", "k_s4_apply"]
    if not args.no_kernel_events:
        for kname in knames:                               # HIP events around those kernels' launches, on their stream
            eng.profile_kernel(kname)
        # every n-th launch: dense bracketing costs ~6 % of the step.  Short runs bracket more densely so that the roofline
        # object always has samples.
        eng.profile_stride(args.event_stride if args.steps >= 200 else 3 if args.steps >= 20 else 1)
    sync()
    t0 = time.perf_counter()
    run_steps(0, args.steps)
    if world > 1:                                          # final sample collection
        for g, t in zip(gathered, outs):
            dist.all_gather_into_tensor(g, t)
    sync()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = tmax.item()
    log(f"timed region: {args.steps} steps in {dt:.3f} s")
    ktimes = {k: eng.profile_read(k) for k in knames} if not args.no_kernel_events else {}
    eng.profile_kernel(None)
    ok = all(torch.isfinite(t).all().item() for t in outs)

    if rank == 0:
        ms_per_step = dt * 1e3 / args.steps
        value = (B * world) / (ms_per_step)                # complexes per (1000 steps x ms_per_step / 1000 s)
        units_per_s = B * world * args.steps / dt          # complex-steps per second
        evals = 1 if wl["predictor"] == "S4" else 2        # joint score evaluations per step
        live = sorted((k for k in ktimes if ktimes[k][0]), key=lambda k: -ktimes[k][1])
        dominant = live[0] if live else None
        state_floats = N * F + N * N + E * K
        line = {
            "metric": "sampled complexes/sec at 1000 PC steps, QM9_CC batch=1024, 1/2/4/8 GPU",
            "value": value, "unit": "complexes/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": f"synthetic (node-count histogram flags, Philox N(0,1) prior/noise; shipped {wl['ckpt']} weights)",
            "config": {"workload": wl["desc"].format(B=B), "global_batch": B * world,
                       "parallelism": f"batch-sharded x{world}, per-shard Langevin norms, all-gather at end", "finite": ok},
            "roofline": roofline_obj(wname, dominant, ktimes, B, dt, E, K) if dominant else None,
            "cpu_baseline": None,
        }
        for k in live[1:]:
            line[f"roofline_{k}"] = roofline_obj(wname, k, ktimes, B, dt, E, K)
        if wl["flop_x"] is not None:
            line["achieved_model_tflops"] = evals * (wl["flop_x"] + wl["flop_a"] + wl["flop_f"]) * units_per_s / 1e12
        line["achieved_state_gbps"] = 2 * evals * state_floats * 4 * units_per_s / 1e9
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(wname, B)
            line["speedup_vs_cpu_baseline"] = value / line["cpu_baseline"]["value"]
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
