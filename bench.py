"""bench.py -- sampled complexes/sec of the CCSD reverse-SDE predictor-corrector sampler on MI355X.

Workload (BASELINE.json configs[2], the configuration the metric is quoted on): qm9_CC checkpoint,
N=9, F=4, E=36, K=466, batch 1024 complexes per GPU, VE SDEs, Reverse predictor + Langevin corrector
(snr 0.2, scale_eps 0.7, n_steps 1), eps 1e-4, QM9 node-count flag mix, in-kernel Philox noise.
A "step" is one PC step (corrector + predictor = 2 joint score evaluations + 2 state updates) over the
whole batch.  value = complexes / (time of 1000 such steps) = B_total / ms_per_step.

The timed region is the reference's own span (sampler.py:1185-1211: init_flags on the host, then the call of
the closure load_sampling_fn returned): `sampling_fn(model_x, model_adj, model_rank2, init_flags)` through
ccsd_amd.loader.load_sampling_fn, i.e. prior draw (ccsd_init_state), K PC steps, and -- for N > 1 -- the final
all-gather of the samples, device-synchronised.  Checkpoint load, plan creation and quantisation are outside,
as they are in the reference.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload qm9_CC|community_small_CC|...]
`--gpus N` (N > 1) without a launcher starts N fresh worker processes itself (torch.distributed.run on
127.0.0.1, before this process touches a GPU) and relays rank 0's JSON line; under torch.distributed.run it
is a rank.  The batch dimension is sharded (weak scaling, 1024 complexes per rank, per-shard Langevin norms
like the reference's divide_batch, Philox keyed by the global sample index); no per-step collective, one RCCL
all-gather of the samples at the end, inside the timed region.
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

QM9_HIST = {9: 10949, 8: 1757, 7: 294, 6: 60, 5: 15, 4: 5, 3: 1, 2: 1}   # data/qm9_test_nx.pkl node counts
COMMUNITY_HIST = {12: 29, 14: 14, 16: 23, 18: 25, 20: 9}                 # data/community_small.pkl
PEAK_F32_MFMA_TFLOPS = 157.3                              # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 dense peak
PEAK_HBM_GBPS = 8000.0
# Workloads = BASELINE.json configs.  flop_* : dense-as-written GEMM FLOPs / complex / forward (FlopCounterMode on
# the reference).  The default (the configuration the metric is quoted on) is qm9_CC, B = 1024 per GPU.
WORKLOADS = {
    "qm9_CC": dict(ckpt="ccsd_qm9_CC", data="QM9", batch=1024, hist=QM9_HIST, predictor="Reverse", corrector="Langevin", snr=0.2, scale_eps=0.7,
                   flop_x=71_424, flop_a=10_710_522, flop_f=3_220_992,
                   desc="qm9_CC N=9 F=4 E=36 K=466, B={B} per GPU, VE x3, Reverse+Langevin snr=0.2 scale_eps=0.7 n_steps=1, 1000 scales"),
    "community_small_CC": dict(ckpt="ccsd_community_small_CC", data="community_small_CC", batch=512, cpu_batch=32, wc=16, hist=COMMUNITY_HIST, predictor="Euler", corrector="Langevin",
                               snr=0.05, scale_eps=0.7, flop_x=3_014_720, flop_a=192_613_760, flop_f=168_081_600,
                               desc="community_small_CC N=20 F=11 E=190 K=1140, B={B} per GPU, VP x3, Euler+Langevin snr=0.05 scale_eps=0.7, 1000 scales"),
    "community_small": dict(ckpt="gdss_community_small", data="community_small", batch=16, hist=COMMUNITY_HIST, predictor="Euler", corrector="Langevin",
                            snr=0.05, scale_eps=0.7, flop_x=2_952_960, flop_a=16_628_480, flop_f=0,
                            desc="community_small (graph-only) N=20 F=10, B={B} per GPU, VP x2, Euler+Langevin snr=0.05 scale_eps=0.7, 1000 scales"),
    "zinc250k": dict(ckpt="gdss_zinc250k", data="ZINC250k", batch=256, hist={38: 1, 30: 2, 24: 4, 23: 4, 20: 2}, predictor="Reverse", corrector="Langevin",
                     snr=0.2, scale_eps=0.9, flop_x=945_440, flop_a=58_489_296, flop_f=0,
                     desc="zinc250k (graph-only substitute for the infeasible zinc250k_CC, SURVEY 8d 5a) N=38 F=9, B={B} per GPU, "
                          "VP(x)/VE(adj), Reverse+Langevin snr=0.2 scale_eps=0.9, 1000 scales; synthetic node-count mix"),
    "zinc250k_CC_5b": dict(ckpt="zinc250k_CC_5b", data="ZINC250k", batch=256, cpu_batch=2, hist={38: 1, 30: 2, 24: 4, 23: 4, 20: 2}, predictor="Reverse",
                           corrector="Langevin", snr=0.2, scale_eps=0.9, flop_x=None, flop_a=None, flop_f=None, wc=8,
                           desc="zinc250k_CC substitute 5b (SURVEY 8d: the config's N=38 and network hyper-parameters, d_min=d_max=3 instead of the "
                                "infeasible d_max=24) N=38 F=9 E=703 K=8436, B={B} per GPU, VP(x)/VE/VE, Reverse+Langevin snr=0.2 scale_eps=0.9, "
                                "1000 scales; reference-initialised random weights, synthetic node-count mix"),
    "qm9_Base_CC": dict(ckpt="ccsd_qm9_Base_CC", data="QM9", batch=1024, hist=QM9_HIST, predictor="Reverse", corrector="Langevin", snr=0.2,
                        scale_eps=0.7, flop_x=None, flop_a=None, flop_f=None,
                        desc="qm9_Base_CC (ScoreNetworkA_Base_CC ablation) N=9 F=4 E=36 K=466, B={B} per GPU, VE x3, Reverse+Langevin "
                             "snr=0.2 scale_eps=0.7 n_steps=1, 1000 scales"),
    "enzymes_small_CC": dict(ckpt="ccsd_enzymes_small_CC", data="ENZYMES_small_CC", batch=64, hist={12: 6, 11: 5, 10: 5, 9: 4, 8: 4, 6: 3, 4: 2}, predictor="S4",
                             corrector="None", snr=0.15, scale_eps=0.7, flop_x=None, flop_a=None, flop_f=None,
                             desc="ENZYMES_small_CC N=12 F=10 E=66 K=715, B={B} per GPU, VP(x)/VE/VE, S4 solver snr=0.15 scale_eps=0.7, 1000 scales"),
}
# Kernels of the step (profiles/README.md).  k_xa evaluates ScoreNetworkX + ScoreNetworkA(_CC) once per launch (fp32 MFMA /
# issue bound); the rank-2 side (ScoreNetworkF + hodge projections) reads rank2 (E*K fp32) once and writes it once per
# half-step (HBM bound): fused k_r2 when the block fits LDS (qm9_CC), k_gemm_h + k_gemm_p + k_hf_score otherwise.
# `bound` of a roofline object: kernels that have both an algorithmic byte count and an as-written FLOP count (k_r2, k_hf_score)
# report BOTH fractions and take the larger one as the binding roofline; the entry here is the fallback when only
# one of the two figures exists.  Neither roofline is what actually limits k_r2 / k_xa: they are bound by vector-instruction
# issue (on gfx950 an fp32 MFMA and VALU work share the SIMD's vector pipe), reported as `issue_frac`.
KERNEL_BOUND = {"k_xa": "mfma", "k_r2": "mfma", "k_hf_score": "hbm", "k_gemm_h": "mfma", "k_gemm_p": "mfma", "k_langevin_apply": "hbm",
                "k_s4_apply": "hbm", "k_ew1": "hbm"}
KERNEL_LIMITER = {"k_r2": "vector-instruction issue: fp32 MFMA (32 cycles each) and VALU (>= 2.65 cycles each at 4 waves / SIMD) share one pipe per SIMD",
                  "k_xa": "latency of one graph's critical path (barrier intervals), then vector-instruction issue"}
N_SIMDS, MAX_CLOCK_HZ = 1024, 2.4e9                       # 256 CUs x 4 SIMDs; MI355X_MICROARCH.md chip table
# cycles one wave64 VALU instruction occupies a SIMD that holds 4 waves (both kernels run at 4 waves / SIMD), measured:
# tools/ubench/valu_rate.hip, profiles/r03_c_valu_rate.txt -- v_fma_f32 2.65, v_xor_b32 2.47, 32-bit integer multiplies 4.4,
# transcendentals 8.3; the plain-FMA figure is used for every instruction, so issue_frac is a lower bound
VALU_ISSUE_CYCLES = 2.65
KERNEL_NAMES = ["k_xa", "k_r2", "k_hf_score", "k_gemm_h", "k_gemm_p", "k_langevin_apply", "k_s4_apply", "k_ew1"]
MFMA_F32_FLOP_PER_BUSY_CYCLE = 64                         # 2048 FLOP / 32 cycles (16x16x4) = 512 FLOP / 8 cycles (4x4x1 16B): both fp32 shapes
WARMUP_MIN_SECONDS = 0.25                                 # the warm-up call is repeated until the device has been busy this long


def pmc_files():
    """Committed per-launch PMC summaries (tools/pmc_traffic.py), newest round first; each names its workload in `_meta`."""
    import glob

    return sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc*.json")), reverse=True)
KERNEL_SOURCES = [os.path.join(ROOT, "ccsd_amd", "csrc", f) for f in ("ccsd_dev.h", "ccsd_rank2_common.h", "ccsd_k_rank2.h", "ccsd_k_r2.h", "ccsd_k_xa.h", "ccsd_k_update.h", "ccsd_attn_stack.inc", "ccsd_plan.h", "ccsd_api.h", "ccsd_baked_qm9.h", "ccsd_baked_cs.h", "ccsd_baked_z.h", "ccsd_baked_enz.h")]


def kernel_source_hash() -> str:
    """Identifies the kernel source a PMC pass was collected on (tools/pmc_traffic.py records it next to the counters)."""
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def pmc_counters(workload: str, kernel: str, B: int):
    """Per-launch PMC means of `kernel` from the newest committed rocprofv3 --pmc passes (separate runs: FETCH_SIZE doubled
    -- gfx950 reports half of a wide coalesced read, MI355X_MICROARCH.md HBM section --, WRITE_SIZE as is).  They are NOT
    measured in this run: the object says which file they come from and whether the kernel source has changed since."""
    for path in pmc_files():
        try:
            d = json.load(open(path))
        except Exception:
            continue
        meta = d.get("_meta", {})
        if meta.get("workload", "qm9_CC") != workload or kernel not in d:
            continue
        if meta.get("batch", WORKLOADS[workload]["batch"]) != B:       # per-launch counters belong to the batch they were collected at
            continue
        src = meta.get("kernel_src_sha16")
        return {"counters": d[kernel], "source": os.path.relpath(path, ROOT), "collected_at_commit": meta.get("commit", "round 1 (5cbf17b)"),
                "stale": (src != kernel_source_hash()) if src else True}
    return None


def hist_flags(B: int, N: int, hist: dict, seed: int = 42):
    import numpy as np
    import torch

    rs = np.random.RandomState(seed)
    ks = np.array(list(hist.keys()))
    p = np.array(list(hist.values()), dtype=np.float64)
    counts = rs.choice(ks, size=B, p=p / p.sum())
    f = torch.zeros(B, N)
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


def load_workload(wname: str, device):
    """Checkpoint -> (ckpt dict, models, names) through the product's own loader (ccsd_amd.loader.load_ckpt reads the
    packaged neutral-format copy of the shipped checkpoint)."""
    from ccsd_amd import loader

    wl = WORKLOADS[wname]
    is_cc = wname not in ("community_small", "zinc250k")
    if wl["ckpt"] == "zinc250k_CC_5b":
        pass   # packaged neutral-format file written by tools/make_golden.py::kat_zinc5b (no shipped checkpoint exists)
    ck = loader.load_ckpt({"ckpt": wl["ckpt"], "data": {"data": wl["data"]}, "folder": ROOT}, device, is_cc=is_cc)
    names = ["x", "adj"] + (["rank2"] if is_cc else [])
    return ck, names, is_cc


def cpu_baseline(wname: str, B: int, budget_s: float = 25.0):
    """The oracle (CPU restatement certified bit-identical to the reference) timed on the host cores, on a bounded sample of
    the same workload: the full batch (the workload's `cpu_batch` complexes where a full-batch step would take minutes), up
    to 10 PC steps after 1 warm-up step (fewer when a step is so slow that 10 would exceed ~25 s), scaled to 1000 steps."""
    B = min(B, WORKLOADS[wname].get("cpu_batch", B))
    import torch

    from ccsd_amd import loader
    from oracle import ccsd_oracle as O

    wl = WORKLOADS[wname]
    threads = host_threads()
    torch.set_num_threads(threads)
    log(f"cpu_baseline: oracle on {threads} host threads, {wname} B={B}")
    ck, names, is_cc = load_workload(wname, "cpu")
    cfg = ck["config"]
    data = cfg["data"]
    N, F = data["max_node_num"], data["max_feat_num"]
    d_min, d_max = (data["d_min"], data["d_max"]) if is_cc else (0, 0)
    so = [O.load_sde(cfg["sde"][p]) for p in names]
    # requires_grad mirrors nn.Parameter: ATen's linear() then takes the same CPU kernel as the reference's modules
    w = {p: {k: v.clone().requires_grad_(True) for k, v in ck[f"{p}_state_dict"].items()} for p in names}
    if is_cc:
        nets = [(lambda x, a, r, f, p=p: O.run_network(ck[f"params_{p}"], w[p], x, a, r, f)) for p in names]
    else:
        nets = [(lambda x, a, f, p=p: O.run_network(ck[f"params_{p}"], w[p], x, a, None, f)) for p in names]
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
    make(1)(*nets, flags)
    tw = time.perf_counter() - t0
    steps = max(2, min(10, int(budget_s / max(tw, 1e-3))))
    log(f"cpu_baseline: warm-up step took {tw:.1f} s -> timing {steps} steps")
    t0 = time.perf_counter()
    make(steps)(*nets, flags)
    dt = (time.perf_counter() - t0) / steps
    log(f"cpu_baseline: {dt:.2f} s / PC step")
    out = {"value": B / (dt * 1000.0), "unit": "complexes/s at 1000 PC steps", "cores": threads, "kind": "port",
           "sample": f"oracle (torch CPU, {threads} threads), {wname} B={B}, {steps} PC steps after 1 warm-up step, scaled to 1000 steps"}
    cert = cpu_baseline_cert()
    if cert:          # how the oracle's wall time relates to the real reference's (measured in the build container, tools/certify_cpu_baseline.py)
        r = cert["oracle_over_reference_time"]
        out["reference_time_ratio"] = r
        out["reference_equivalent_value"] = out["value"] * r
        out["certificate"] = cert["_file"]
        out["sample"] += (f"; the oracle is bit-identical to the reference and takes {r:.2f}x the reference's wall time "
                          f"({cert['workload']}; {cert['_file']}): the reference itself would run at ~{out['value'] * r:.3f} complexes/s here")
    return out


def cpu_baseline_cert():
    """Newest profiles/rNN_cpu_baseline_cert.json (written by tools/certify_cpu_baseline.py in the build container)."""
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_cpu_baseline_cert.json")))
    if not files:
        return None
    try:
        d = json.load(open(files[-1]))
    except Exception:
        return None
    d["_file"] = os.path.relpath(files[-1], ROOT)
    return d


MERGED_R2 = False     # set from the plan (engine.query("merged_r2")): a k_r2 launch = predictor half-step + the next norms pass


def kernel_work(wname: str, kname: str, E: int, K: int):
    """Algorithmic work of ONE launch per complex: (flops, bytes, what)."""
    wl = WORKLOADS[wname]
    if kname == "k_r2" and MERGED_R2:
        return 2 * wl["flop_f"], 4 * E * K * 4, ("merged launch = two ScoreNetworkF evaluations; FLOPs: 2 x the dense-as-written GEMM FLOPs of "
                                                "ScoreNetworkF (SURVEY 8a); bytes: read rank2 + the raw score of the norms pass, write the new rank2 + "
                                                "the next raw score (4 x E*K fp32)")
    if kname == "k_xa":
        f = None if wl["flop_x"] is None else wl["flop_x"] + wl["flop_a"]
        return f, None, "dense-as-written GEMM FLOPs of ScoreNetworkX + ScoreNetworkA(_CC) (SURVEY 8a)"
    if kname == "k_r2":
        return wl["flop_f"], 2 * E * K * 4, ("FLOPs: dense-as-written GEMM FLOPs of ScoreNetworkF (SURVEY 8a); bytes: read + write of rank2 "
                                             "(E*K fp32 each)")
    if kname == "k_hf_score":
        # tiled path: ScoreNetworkF's two contractions are two kernels -- H = F F^T in k_gemm_h, (H F) + the per-element network here
        return 2 * E * E * K, 2 * E * K * 4, ("FLOPs: the (H F) contraction as written, 2 E^2 K (H = F F^T is k_gemm_h's; together = ScoreNetworkF's "
                                              "dense-as-written GEMM FLOPs, SURVEY 8a); bytes: read + write of rank2 (E*K fp32 each)")
    if kname in ("k_langevin_apply", "k_s4_apply"):
        return None, 3 * E * K * 4, "read state + raw score, write state (rank2 dominates)"
    if kname == "k_ew1":
        return None, 2 * E * K * 4, ("element-wise ScoreNetworkF (cnum = 1): mean over the step's two launches -- the norms launch reads rank2, the "
                                     "predictor launch reads it and writes the corrected and the new rank2")
    if kname == "k_gemm_p" and wl.get("wc"):
        return 2 * E * K * wl["wc"], None, "hodge projection GEMM P_0 = rank2 . Wcat (2 E K wc FLOPs)"
    if kname == "k_gemm_h":
        return 2 * E * E * K, None, "H = F F^T as written (2 E^2 K FLOPs; the kernel computes the upper-triangle tiles only)"
    return None, None, ""


def roofline_obj(wname, kname, kt, B, ms_per_step, E, K):
    """kt = (bracketed launches, their summed ms, all launches of the kernel-event pass, steps of that pass); ms_per_step = the timed
    region's (un-instrumented) step time."""
    sampled, kms, total, ksteps = kt
    if not sampled:
        return None
    avg_s = kms / sampled * 1e-3
    flops, nbytes, what = kernel_work(wname, kname, E, K)
    pmc = pmc_counters(wname, kname, B)
    bound = KERNEL_BOUND[kname]
    hbm_frac = nbytes * B / avg_s / 1e9 / PEAK_HBM_GBPS if nbytes else None
    mfma_frac = flops * B / avg_s / 1e12 / PEAK_F32_MFMA_TFLOPS if flops else None
    if hbm_frac is not None and mfma_frac is not None:     # both figures exist: the binding roofline is the larger fraction
        bound = "mfma" if mfma_frac >= hbm_frac else "hbm"
    o = {"kernel": kname, "bound": bound, "launches_per_step": total / ksteps, "launches_timed": sampled, "event_pass_steps": ksteps,
         "avg_launch_us": avg_s * 1e6, "share_of_step": (total / ksteps) * avg_s / (ms_per_step * 1e-3), "traffic": None, "hbm_frac": hbm_frac, "mfma_frac_as_written": mfma_frac}
    if kname in KERNEL_LIMITER:
        o["limiter"] = KERNEL_LIMITER[kname]
    if pmc:
        c = pmc["counters"]
        o["traffic"] = c.get("hbm_bytes_per_launch")
        o["traffic_source"] = f"{pmc['source']} (rocprofv3 --pmc passes collected at {pmc['collected_at_commit']}; replayed, not measured in this run)"
        o["traffic_stale"] = pmc["stale"]
        mf = c.get("SQ_INSTS_VALU_MFMA_F32")
        busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES")
        if mf:
            # executed matrix FLOPs from the pipe's BUSY CYCLES (PMC), not from the instruction count: the kernels mix
            # v_mfma_f32_16x16x4_f32 (2048 FLOP, 32 cycles) with v_mfma_f32_4x4x1_16B_f32 (512 FLOP, 8 cycles) -- both 64 FLOP per busy
            # cycle --, so pricing every MFMA at 2048 FLOP overstates k_r2.  Without the busy counter: 32 cycles each.
            busy_cyc = busy if busy else mf * 32.0
            ex = busy_cyc * MFMA_F32_FLOP_PER_BUSY_CYCLE / avg_s / 1e12
            o["executed_mfma_tflops"] = ex
            o["executed_frac"] = ex / PEAK_F32_MFMA_TFLOPS       # what the matrix pipe actually did (dead GEMMs skipped, padding included)
            o["mfma_busy_frac"] = busy_cyc / (N_SIMDS * MAX_CLOCK_HZ * avg_s)   # == executed_frac: 157.3 TF = 1024 SIMDs x 64 FLOP x 2.4 GHz
            o["mfma_cycles_per_instruction"] = busy_cyc / mf
            va = c.get("SQ_INSTS_VALU")
            if va:
                # vector-issue cycles per launch: the matrix instructions' busy cycles (PMC; 32 per fp32 16x16x4 MFMA, 8.8 per 4x4x1)
                # + VALU_ISSUE_CYCLES per other VALU wave-instruction (SQ_INSTS_VALU counts the MFMAs too), over what 1024 SIMDs offer
                # at the 2.4 GHz maximum clock in the launch's time
                cyc = busy_cyc + max(va - mf, 0.0) * VALU_ISSUE_CYCLES
                o["valu_wave_instructions"] = va
                o["issue_cycles_per_launch"] = cyc
                o["issue_frac"] = cyc / (N_SIMDS * MAX_CLOCK_HZ * avg_s)
    if bound == "mfma" and flops and flops * B / avg_s / 1e12 > PEAK_F32_MFMA_TFLOPS and o.get("executed_mfma_tflops"):
        # the as-written count exceeds what the matrix pipe could do in the launch time: the kernel skips most of the dense-as-written
        # work (k_xa: unordered pairs instead of N x N, masked rows, MLP tiles of real width), so the executed MFMA rate is the honest figure
        o.update(achieved=o["executed_mfma_tflops"], peak=PEAK_F32_MFMA_TFLOPS, unit="TFLOP/s", frac=o["executed_frac"],
                 as_written_tflops=flops * B / avg_s / 1e12,
                 note=f"achieved = MFMA busy cycles per launch (PMC) x 64 FLOP / mean launch time; the as-written model-FLOPs convention ({what}) "
                      "gives as_written_tflops, above the fp32 MFMA peak because the kernel legally skips most of that work")
    elif bound == "mfma" and flops and flops * B / avg_s / 1e12 > PEAK_F32_MFMA_TFLOPS:
        # as above, but no PMC pass exists for this workload / batch: the as-written rate alone is not a utilisation figure
        o.update(achieved=None, peak=PEAK_F32_MFMA_TFLOPS, unit="TFLOP/s", frac=None, as_written_tflops=flops * B / avg_s / 1e12,
                 note=f"the as-written model-FLOPs convention ({what}) exceeds the fp32 MFMA peak -- the kernel legally skips most of that work --, "
                      "and no PMC pass was collected at this workload / batch to give the executed rate")
    elif bound == "mfma" and flops:
        a = flops * B / avg_s / 1e12
        o.update(achieved=a, peak=PEAK_F32_MFMA_TFLOPS, unit="TFLOP/s", frac=a / PEAK_F32_MFMA_TFLOPS,
                 note=f"achieved = {what} per launch / mean launch time (HIP events on the launch stream): the as-written model-FLOPs "
                      "convention, which counts GEMMs the kernel legally skips; executed_frac = MFMA busy cycles (PMC) x 64 FLOP / time "
                      "is the pipe utilisation (= mfma_busy_frac); fp32 MFMA peak")
    elif bound == "hbm" and nbytes:
        a = nbytes * B / avg_s / 1e9
        o.update(achieved=a, peak=PEAK_HBM_GBPS, unit="GB/s", frac=a / PEAK_HBM_GBPS,
                 note=f"achieved = algorithmic bytes per launch ({what}) / mean launch time (HIP events on the launch stream)")
    else:
        o.update(achieved=None, peak=None, unit=None, frac=None, note="no algorithmic-work figure for this kernel / workload")
    return o


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--warmup-seconds", type=float, default=WARMUP_MIN_SECONDS,
                    help="repeat the W-step warm-up call until the device has been busy this long (0: exactly W steps -- profiler passes, "
                         "whose per-launch medians should see the timed region's launch mix)")
    ap.add_argument("--workload", default="qm9_CC", choices=sorted(WORKLOADS), help="BASELINE.json config (default: the metric's)")
    ap.add_argument("--batch", type=int, default=0, help="complexes per GPU (default: the workload's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true", help="do not time the kernels with HIP events")
    ap.add_argument("--event-stride", type=int, default=17, help="bracket every n-th launch of a kernel with HIP events (odd: a kernel's norms / predictor launches alternate)")
    ap.add_argument("--split-bf16", type=int, default=0, choices=[0, 3],
                    help="EXPERIMENT, never the default: 3 = split-precision (bf16 x 3, fp32 accumulate) MFMA contraction in k_gemm_h_full "
                         "(community_small geometry; CCSD_SPLIT_BF16).  The line then says dtype \"f32 + bf16x3 (experiment: ...)\"")
    ap.add_argument("--emulate", action="store_true",
                    help="TEST ONLY: run the product code over the host emulation of the kernels on the CPU with gloo (exercises the launcher / "
                         "sharding / reporting path on a GPU-less box; the numbers mean nothing)")
    return ap.parse_args(argv)


def main():
    # stdout carries exactly one line (rank 0's JSON); everything else the product prints (the reference's own
    # `print(" ")` after the loop, "... loaded" of load_ckpt) goes to stderr
    import contextlib

    real_stdout = sys.stdout
    with contextlib.redirect_stdout(sys.stderr):
        _main(real_stdout)


def _main(real_stdout):
    args = parse_args()
    if args.split_bf16:
        os.environ["CCSD_SPLIT_BF16"] = str(args.split_bf16)      # (a plan switch: read when the engine creates its plan)
    else:
        os.environ.pop("CCSD_SPLIT_BF16", None)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # self-launch: N fresh rank processes, started before this process has made any GPU call (it never makes one)
        from ccsd_amd.distributed import launch_workers

        rc = launch_workers([os.path.abspath(__file__)] + sys.argv[1:], args.gpus, relay=real_stdout)
        if rc != 0:
            log(f"a worker failed (exit code {rc})")
        sys.exit(rc)

    import torch
    import torch.distributed as dist

    from ccsd_amd import distributed

    if args.emulate:
        os.environ.setdefault("CUDA_VISIBLE_DEVICES", "")
    else:
        # the library is in place BEFORE the process group exists (no collective -- and no RCCL call -- ever waits on a compiler):
        # rank 0 (re)builds when a source is newer, the other ranks wait for the finished file
        import __graft_entry__ as ge

        if int(os.environ.get("RANK", "0")) == 0:
            ge.build()
        else:
            t_wait = time.time()
            while not os.path.exists(ge.LIB) or any(os.path.getmtime(f) > os.path.getmtime(ge.LIB) for f in ge.SOURCES):
                if time.time() - t_wait > 1200:
                    raise SystemExit("bench.py: timed out waiting for rank 0 to build libccsd_hip.so")
                time.sleep(1.0)
    rank, world, dev = distributed.init("gloo" if args.emulate else None)
    if args.emulate and os.environ.get("CCSD_BENCH_FAIL_RANK") == str(rank):     # tests/test_bench_launcher.py: a dying worker
        os._exit(7)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    lib = None
    if args.emulate:
        from tests.emu_util import emu_library

        torch.set_num_threads(max(1, host_threads() // world))
        lib = emu_library() if rank == 0 else None
        if world > 1:
            dist.barrier()
        lib = lib or emu_library()
    from ccsd_amd import loader

    wname = args.workload
    wl = WORKLOADS[wname]
    ck, names, is_cc = load_workload(wname, dev)
    cfgt = ck["config"]
    data = cfgt["data"]
    N, F = data["max_node_num"], data["max_feat_num"]
    d_min, d_max = (data["d_min"], data["d_max"]) if is_cc else (None, None)
    B = args.batch or wl["batch"]
    total = B * world
    data["batch_size"] = total                      # generic datasets take the batch from the training config (loader.py:387-416)
    models = [loader.load_model_from_ckpt(ck[f"params_{p}"], ck[f"{p}_state_dict"], dev) for p in names]
    module = dict(predictor=wl["predictor"], corrector=wl["corrector"], snr=wl["snr"], scale_eps=wl["scale_eps"], n_steps=1)
    sample = dict(n_samples=total, probability_flow=False, noise_removal=True, eps=1e-4)
    # the drop-in seam: load_sampling_fn -> sampling_fn(models..., init_flags); sharded over the ranks when N > 1
    sampling_fn = distributed.load_sampling_fn_sharded(cfgt, module, sample, dev, is_cc=is_cc, d_min=d_min, d_max=d_max, exact=False,
                                                       rng="philox", seed=42, lib=lib)
    inner = getattr(sampling_fn, "inner", sampling_fn)
    flags = hist_flags(total, N, wl["hist"]).to(dev)
    diff = loader.load_sde(cfgt["sde"]["adj"]).N

    def run(k):
        out = None
        while k > 0:                                 # K may exceed the 1000 scales: further calls of the closure
            n = min(k, diff)
            inner.max_steps = n
            out = sampling_fn(*models, flags)
            k -= n
        return out

    def sync():
        if world > 1:
            dist.barrier()
        if dev != "cpu":
            torch.cuda.synchronize()

    log(f"rank {rank}/{world}: {wname}, B={B} per GPU, warm-up {args.warmup} steps (builds the plan)")
    # Warm-up: W steps (the first call also builds the plan).  A W-step call of a few milliseconds leaves the device at idle
    # clocks, so the same call is repeated until the device has been busy for WARMUP_MIN_SECONDS (untimed; reported as
    # `warmup_steps_run`): the timed region then starts at the clocks a production run -- 1000 steps per call -- sees.
    wsteps = max(1, args.warmup)
    run(wsteps)
    warm_run = wsteps
    if dev != "cpu" and args.warmup_seconds > 0:
        torch.cuda.synchronize()
        tw0 = time.perf_counter()
        run(wsteps)
        torch.cuda.synchronize()
        t1 = max(time.perf_counter() - tw0, 1e-4)
        reps = torch.tensor([min(int(args.warmup_seconds / t1), 5000)], device=dev, dtype=torch.int64)
        if world > 1:                               # every rank must make the same number of (collective) closure calls
            dist.all_reduce(reps, op=dist.ReduceOp.MAX)
        for _ in range(int(reps.item())):
            run(wsteps)
        torch.cuda.synchronize()
        warm_run += wsteps * (1 + int(reps.item()))
    eng = inner.engine()
    E, K = eng.E, eng.K
    global MERGED_R2
    MERGED_R2 = bool(eng.query("merged_r2"))
    if world > 1:
        sampling_fn.timing, sampling_fn.gather_seconds = True, 0.0   # split [local loop | final all-gather] per rank (diagnostic)
    # ---- the timed region: exactly K steps, un-instrumented (no HIP events between the dispatches)
    sync()
    t0 = time.perf_counter()
    outs = run(args.steps)
    if dev != "cpu":
        torch.cuda.synchronize()
    dt_local = time.perf_counter() - t0                              # this rank's own time, before it waits for the others
    sync()
    dt = time.perf_counter() - t0
    # ---- kernel-event pass, AFTER the timed region (an event record between two dispatches breaks back-to-back issue: dense
    # bracketing costs ~6 % of the step): the same steps again (at most 200) with every n-th launch of a kernel bracketed
    ksteps = 0
    if not args.no_kernel_events:
        for kname in KERNEL_NAMES:                   # HIP events around those kernels' launches, on their stream
            eng.profile_kernel(kname)
        ksteps = min(args.steps, 200)
        eng.profile_stride(args.event_stride if ksteps >= 200 else 3 if ksteps >= 20 else 1)
        gs = getattr(sampling_fn, "gather_seconds", 0.0)
        run(ksteps)
        if dev != "cpu":
            torch.cuda.synchronize()
        if world > 1:
            sampling_fn.gather_seconds = gs          # (the per-rank split below belongs to the timed region)
    per_rank = None
    if world > 1:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = tmax.item()
        mine = torch.tensor([dt_local, sampling_fn.gather_seconds], device=dev, dtype=torch.float64)
        allr = torch.empty(world, 2, device=dev, dtype=torch.float64)
        if mine.is_cuda:
            dist.all_gather_into_tensor(allr, mine)
        else:
            dist.all_gather(list(allr.unbind(0)), mine)
        per_rank = [{"rank": r, "ms_per_step": (allr[r, 0].item() - allr[r, 1].item()) * 1e3 / args.steps,
                     "all_gather_ms": allr[r, 1].item() * 1e3, "region_ms": allr[r, 0].item() * 1e3} for r in range(world)]
        log(f"rank {rank}: local loop {(dt_local - sampling_fn.gather_seconds) * 1e3 / args.steps:.4f} ms/step, final all-gather "
            f"{sampling_fn.gather_seconds * 1e3:.2f} ms")
    log(f"timed region: {args.steps} steps in {dt:.3f} s")
    ktimes = {}
    if not args.no_kernel_events:
        for k in KERNEL_NAMES:
            tot = eng.profile_launches(k)
            n, ms = eng.profile_read(k)
            ktimes[k] = (n, ms, tot, ksteps)
        eng.profile_kernel(None)
    nt = 3 if is_cc else 2
    ok = all(torch.isfinite(t).all().item() for t in outs[:nt]) and all(t.shape[0] == total for t in outs[:nt])

    if rank == 0:
        ms_per_step = dt * 1e3 / args.steps
        value = total / ms_per_step                  # complexes per (1000 steps x ms_per_step / 1000 s)
        units_per_s = total * args.steps / dt        # complex-steps per second
        evals = 1 if wl["predictor"] == "S4" else 2  # joint score evaluations per step
        live = sorted((k for k in ktimes if ktimes[k][0]), key=lambda k: -(ktimes[k][2] * ktimes[k][1] / ktimes[k][0]))
        dominant = live[0] if live else None
        state_floats = N * F + N * N + E * K
        line = {
            "metric": "sampled complexes/sec at 1000 PC steps, QM9_CC batch=1024, 1/2/4/8 GPU",
            "value": value, "unit": "complexes/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "warmup_steps_run": warm_run,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if not args.split_bf16 else "f32 + bf16x3 (experiment: H = F F^T of the norms pass as three bf16 MFMA terms, fp32 accumulate)",
            "data": f"synthetic (node-count histogram flags, Philox N(0,1) prior/noise; shipped {wl['ckpt']} weights)",
            "config": {"workload": wl["desc"].format(B=B), "global_batch": total,
                       "parallelism": f"batch-sharded x{world}, per-shard Langevin norms, all-gather at end",
                       "timed_region": "load_sampling_fn -> sampling_fn(models, init_flags): prior draw + PC steps (+ final all-gather)",
                       "finite": ok},
            "roofline": roofline_obj(wname, dominant, ktimes[dominant], B, ms_per_step, E, K) if dominant else None,
            "cpu_baseline": None,
        }
        if per_rank is not None:     # per-rank split of the timed region: PC-step time without the final all-gather, and the all-gather
            line["per_rank"] = per_rank
        if args.emulate:
            line["data"] = "EMULATION (host CPU, test of the launcher path only): " + line["data"]
        for k in live[1:]:
            line[f"roofline_{k}"] = roofline_obj(wname, k, ktimes[k], B, ms_per_step, E, K)
        if wl["flop_x"] is not None:
            line["achieved_model_tflops"] = evals * (wl["flop_x"] + wl["flop_a"] + wl["flop_f"]) * units_per_s / 1e12
        line["achieved_state_gbps"] = 2 * evals * state_floats * 4 * units_per_s / 1e9
        if world == 1 and not args.no_cpu_baseline and not args.emulate:
            line["cpu_baseline"] = cpu_baseline(wname, B)
            line["speedup_vs_cpu_baseline"] = value / line["cpu_baseline"]["value"]
        print(json.dumps(line), file=real_stdout, flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if not ok:
        sys.exit(3)


if __name__ == "__main__":
    main()
