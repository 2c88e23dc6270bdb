"""get_pc_sampler: the reference's predictor-corrector sampler factory, backed by the HIP kernels.

Same signature, closure signature and return tuple as ccsd/src/solver.py:856-1176:
  graph: pc_sampler(model_x, model_adj, init_flags)              -> (x, adj, nfe, diff_traj)
  CC:    pc_sampler(model_x, model_adj, model_rank2, init_flags) -> (x, adj, rank2, nfe, diff_traj)

Extra keyword-only knobs (not in the reference):
  rng        "philox" (default): counter-based noise generated inside the kernels, one C call for the loop;
             "torch": priors from the CPU generator and in-loop noise from randn_like on the state's
                      device, in the reference's draw order (what the reference itself does, solver.py:1111-1113,
                      graph_utils.py:171);
             "torch_cpu": every draw from torch's CPU generator -> identical-seed parity with the
                      reference's CPU run.
  keep_traj  record diff_traj (the reference always does; default False here because only the plotting
             code consumes it -- SURVEY.md section 7).  With keep_traj=False an empty list is returned.
  group      torch.distributed process group: all-reduce the six Langevin norm sums across ranks so that
             a sharded batch reproduces the single-process step size exactly.
  seed, sample_offset, call_stride
             Philox stream: the draws of sample b of a call are keyed by (seed, global sample index), with
             global index = sample_offset + calls_so_far * call_stride + b.  The closure counts its calls, so the
             chunks of the harness's divide_batch loop and its sampling rounds (sampler.py:1195-1211, 511-527: one
             closure, called repeatedly) get disjoint sample indices -- every call draws fresh priors and noise.
             call_stride defaults to B (B * world size when `group` is given: rank r passes sample_offset = r * B).
             `pc_sampler.calls` can be read or reset by the caller.

The models may be this package's weight containers (ccsd_amd.models) or the reference's own nn.Modules
(possibly wrapped in DataParallel): hyper-parameters and weights are read through models.params_from_module
and state_dict() (loader.py:619-657 builds them from the same pair).
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence

import torch

from . import _lib
from .engine import PCEngine
from .sde import SDE


def get_predictor(predictor: str) -> str:
    if predictor not in ("Reverse", "Euler"):
        raise NotImplementedError(f"Predictor {predictor} not yet supported. Select from [Reverse, Euler].")
    return predictor


def get_corrector(corrector: str) -> str:
    if corrector not in ("Langevin", "None"):
        raise NotImplementedError(f"Corrector {corrector} not yet supported. Select from [Langevin, None].")
    return corrector


def _unwrap(model):
    return getattr(model, "module", model)


def S4_solver(sde_x: SDE, sde_adj: SDE, shape_x: Sequence[int], shape_adj: Sequence[int], predictor: str = "None",
              corrector: str = "None", snr: float = 0.1, scale_eps: float = 1.0, n_steps: int = 1,
              probability_flow: bool = False, continuous: bool = False, denoise: bool = True, eps: float = 1e-3,
              device: str = "cuda", is_cc: bool = False, sde_rank2: Optional[SDE] = None,
              shape_rank2: Optional[Sequence[int]] = None, d_min: Optional[int] = None, d_max: Optional[int] = None,
              **extra) -> Callable:
    """The reference's S4 sampler factory (ccsd/src/solver.py:1179-1563): same signature, closure signature and return
    tuple (nfe is 0 there); predictor / corrector / n_steps / probability_flow are accepted and unused, as in the
    reference.  One step = one joint score evaluation, a Langevin-style correction with it, and two half-step transition
    kernels around the score drift.  `extra`: the keyword-only knobs of get_pc_sampler (rng, seed, keep_traj, ...)."""
    return get_pc_sampler(sde_x, sde_adj, shape_x, shape_adj, "S4", "None", snr, scale_eps, 1, False, continuous, denoise, eps,
                          device, is_cc, sde_rank2, shape_rank2, d_min, d_max, **extra)


def get_pc_sampler(sde_x: SDE, sde_adj: SDE, shape_x: Sequence[int], shape_adj: Sequence[int], predictor: str = "Euler",
                   corrector: str = "None", snr: float = 0.1, scale_eps: float = 1.0, n_steps: int = 1,
                   probability_flow: bool = False, continuous: bool = False, denoise: bool = True, eps: float = 1e-3,
                   device: str = "cuda", is_cc: bool = False, sde_rank2: Optional[SDE] = None,
                   shape_rank2: Optional[Sequence[int]] = None, d_min: Optional[int] = None, d_max: Optional[int] = None,
                   *, rng: str = "philox", keep_traj: bool = False, seed: Optional[int] = None, group=None,
                   sample_offset: int = 0, call_stride: Optional[int] = None, max_steps: Optional[int] = None,
                   lib: Optional[_lib.Library] = None) -> Callable:
    s4 = predictor == "S4"          # reached through S4_solver only (get_predictor rejects the name, as the reference does)
    if not s4:
        get_predictor(predictor)
        get_corrector(corrector)
    if rng not in ("philox", "torch", "torch_cpu"):
        raise ValueError(f"rng {rng} unknown. Select from [philox, torch, torch_cpu].")
    B, N, F = shape_x
    sdes = [sde_x, sde_adj] + ([sde_rank2] if is_cc else [])
    cache = {}

    def build_engine(models) -> PCEngine:
        if not continuous:
            raise NotImplementedError("Discrete not supported")   # losses.py:69,161
        key = tuple(id(m) for m in models)
        if key not in cache:
            from .models import params_from_module

            ms = [_unwrap(m) for m in models]
            for m in ms:
                m.eval()
            args = []
            for m in ms:
                args += [params_from_module(m), m.state_dict()]
            if not is_cc:
                args += [None, None]
            cache.clear()
            cache[key] = PCEngine(*args, N=N, F=F, is_cc=is_cc, d_min=d_min or 0, d_max=d_max or 0, sdes=sdes,
                                  predictor=predictor, corrector=corrector, snr=snr, scale_eps=scale_eps, n_steps=n_steps,
                                  probability_flow=probability_flow, denoise=denoise, eps=eps, device=device, lib=lib,
                                  batch_hint=B)
        return cache[key]

    def draw(shape, dev):
        if rng == "torch_cpu":
            return torch.randn(*shape).to(dev)
        return torch.randn(*shape, device=dev)

    def pc_sampler(*args):
        models, init_flags = args[:-1], args[-1]
        if len(models) != len(sdes):
            raise TypeError(f"pc_sampler expects {len(sdes)} models and init_flags")
        eng = build_engine(models)
        dev = eng.device
        flags = init_flags.to(dev, torch.float32).contiguous()
        if flags.shape != (B, N):
            raise ValueError(f"init_flags must have shape {(B, N)}, got {tuple(flags.shape)}")
        the_seed = int(seed if seed is not None else torch.initial_seed()) & 0xFFFFFFFFFFFFFFFF
        # global index of this call's first sample in the Philox stream: calls never share draws
        stride = call_stride
        if stride is None:
            stride = B
            if group is not None:
                import torch.distributed as dist

                if dist.is_available() and dist.is_initialized():
                    stride = B * dist.get_world_size(group)
        offset = int(sample_offset) + pc_sampler.calls * int(stride)
        pc_sampler.calls += 1
        shapes = eng.shapes(B)
        nt = 3 if is_cc else 2
        state, scratch, result = eng.alloc_state(B), eng.alloc_state(B), eng.alloc_state(B)
        diff_steps = sde_adj.N
        ms = pc_sampler.max_steps                   # (attribute: a caller may change the step budget between calls)
        last = diff_steps if ms is None else min(ms, diff_steps)
        diff_traj: List[List[torch.Tensor]] = []
        with torch.no_grad():
            if rng == "philox":
                traj = None
                if keep_traj:
                    per = sum(s[1] * s[2] for s in shapes[:nt])
                    traj = torch.empty(diff_steps, per, device=dev)
                # the single C call covers n_steps == 1; more inner Langevin steps (solver.py:1131-1137) and the exact
                # multi-GPU mode are driven step by step (same kernels, Philox noise generated in them)
                stepwise = group is not None or (corrector == "Langevin" and n_steps != 1 and not s4)
                if last == 0 or stepwise:
                    eng.init_state(flags, state, None, the_seed, offset)
                if last == 0:
                    for dst, src in zip(result, state):
                        if dst is not None:
                            dst.copy_(src)
                elif not stepwise:
                    eng.init_and_run(flags, state, scratch, result, the_seed, offset, 0, last, traj)   # (prior draw + loop, back to back)
                else:
                    traj = None
                    _stepwise(eng, flags, state, scratch, result, None, the_seed, offset, last, diff_traj, keep_traj, group)
                if traj is not None:
                    o = 0
                    for i in range(last):
                        row, o2, item = traj[i], 0, []
                        for s in shapes[:nt]:
                            n = s[1] * s[2]
                            item.append(row[o2:o2 + n].view(s[1], s[2]).clone())
                            o2 += n
                        diff_traj.append(item)
            else:
                # priors on the CPU generator, then moved (solver.py:1111-1113)
                px = sde_x.prior_sampling(shape_x)
                pa = torch.randn(*shape_adj)          # raw draw; the kernel applies triu(1)+transpose
                prior = [px.to(dev), pa.to(dev)]
                if is_cc:
                    prior.append(sde_rank2.prior_sampling(shape_rank2).to(dev))
                eng.init_state(flags, state, prior)
                noise_fn = lambda k: draw(shapes[k], dev)
                _stepwise(eng, flags, state, scratch, result, noise_fn, the_seed, offset, last, diff_traj, keep_traj, group)
        if last == 0:
            out = state
        elif rng != "philox" or stepwise:
            out = result if denoise else state
        else:
            out = result
        print(" ")
        return (*out[:nt], 0 if s4 else diff_steps * (n_steps + 1), diff_traj)

    def _stepwise(eng, flags, state, scratch, result, noise_fn, the_seed, off, last, diff_traj, keep, grp):
        """Python-driven loop: used for host-supplied noise and for the exact multi-GPU mode."""
        import torch.distributed as dist

        dev = eng.device
        sums = torch.zeros(8, device=dev)
        nt = 3 if is_cc else 2
        third = eng.alloc_state(flags.shape[0]) if (corrector == "Langevin" and n_steps > 1) else None
        a, b = state, scratch   # a = live
        for step in range(last):
            lastone = step == last - 1
            if s4:
                # draw order of one S4 step (solver.py:1299-1350): correction x, adj(, rank2); first transition; second transition
                z1, z2, z3 = ([noise_fn(k) for k in range(nt)] for _ in range(3)) if noise_fn else (None, None, None)
                eng.corrector_norms(step, 0, a, a, flags, z1, the_seed, off, sums)
                if grp is not None and dist.is_available() and dist.is_initialized():
                    dist.all_reduce(sums, group=grp)
                want_mean = denoise and (lastone or keep)
                eng.s4_apply(step, a, flags, z1, z2, z3, the_seed, off, sums, b, result if want_mean else None)
                a, b = b, a
            elif corrector == "Langevin":
                base, cur = a, a
                bufs = [b, third]
                # the reference finishes all inner steps of one target before the next target draws
                # (solver.py:1131-1137): target-major draw order
                zs = [[None] * nt for _ in range(n_steps)]
                if noise_fn:
                    for k in range(nt):
                        for it in range(n_steps):
                            zs[it][k] = noise_fn(k)
                for it in range(n_steps):
                    z = zs[it] if noise_fn else None
                    eng.corrector_norms(step, it, base, cur, flags, z, the_seed, off, sums)
                    if grp is not None and dist.is_available() and dist.is_initialized():
                        dist.all_reduce(sums, group=grp)   # six floats over RCCL / gloo: exact batch-global step size
                    out = bufs[it % 2]
                    eng.corrector_apply(step, it, cur, flags, z, the_seed, off, sums, out)
                    cur = out
                z = [noise_fn(k) for k in range(nt)] if noise_fn else None
                want_mean = denoise and (lastone or keep)
                eng.predictor(step, cur, flags, z, the_seed, off, a, result if want_mean else None)   # base is dead: reuse it
            else:
                z = [noise_fn(k) for k in range(nt)] if noise_fn else None
                want_mean = denoise and (lastone or keep)
                eng.predictor(step, a, flags, z, the_seed, off, b, result if want_mean else None)
                a, b = b, a
            if keep:
                src = result if denoise else a
                diff_traj.append([t[0].detach().clone() for t in src[:nt]])
        if a is not state:
            for dst, src in zip(state, a):
                if dst is not None:
                    dst.copy_(src)

    pc_sampler.calls = 0
    pc_sampler.max_steps = max_steps
    pc_sampler.engine = lambda: next(iter(cache.values()), None)      # the PCEngine of the last model triple (measurement hooks)
    return pc_sampler
