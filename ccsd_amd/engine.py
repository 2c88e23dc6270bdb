"""PCEngine: owns a ccsd_plan_t and drives the C ABI with torch tensors as device memory.

PyTorch is plumbing here (allocations, streams, RCCL): every score evaluation, mask, noise draw
and state update is done by the HIP kernels behind include/ccsd_hip.h.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib, plan as _plan
from .sde import SDE, step_coefficients


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


class PCEngine:
    def __init__(self, params_x: Optional[dict], sd_x, params_adj: Optional[dict], sd_adj, params_rank2: Optional[dict],
                 sd_rank2, *, N: int, F: int, is_cc: bool, d_min: int = 0, d_max: int = 0,
                 sdes: Optional[Sequence[SDE]] = None, predictor: str = "Euler", corrector: str = "None",
                 snr: float = 0.1, scale_eps: float = 1.0, n_steps: int = 1, probability_flow: bool = False,
                 denoise: bool = True, eps: float = 1e-3, device="cuda", lib: Optional[_lib.Library] = None,
                 batch_hint: int = 0):
        self.lib = lib if lib is not None else _lib.get_library()
        self.device = torch.device(device)
        if self.lib.is_hip:
            if self.device.type != "cuda":
                raise _lib.CcsdError("the HIP library needs a cuda (ROCm) device; ccsd_amd has no CPU path")
            if not torch.cuda.is_available():
                raise _lib.CcsdError("no MI355X visible: ccsd_amd has no CPU fallback")
        px, pa, pf = _plan.complete_params(params_x, params_adj, params_rank2 if is_cc else None, N, F, is_cc, d_min, d_max)
        self.is_cc, self.N, self.F = is_cc, N, F
        self.E, self.K = _plan.rank2_dim(N, d_min, d_max) if is_cc else (N * (N - 1) // 2, 0)
        if sdes is not None:
            coef = step_coefficients(list(sdes), predictor, probability_flow, eps)
            if not is_cc:
                coef[:, 2] = coef[:, 1]
            self.diff_steps = sdes[1].N
        else:
            coef = np.zeros((1, 3, 10), np.float32)
            coef[:, :, 0] = 1.0
            self.diff_steps = 1
        self.coef = np.ascontiguousarray(coef, np.float32)
        self.n_steps, self.corrector, self.denoise = n_steps, corrector, denoise
        cfg = _plan.make_config(px, pa, pf, predictor=predictor, corrector=corrector, snr=snr, scale_eps=scale_eps,
                                n_steps=n_steps, probability_flow=probability_flow, denoise=denoise,
                                diff_steps=self.diff_steps, batch_hint=batch_hint)
        blob = _plan.pack_weights(px, sd_x if params_x is not None else None, pa, sd_adj if params_adj is not None else None,
                                  pf, sd_rank2 if (is_cc and params_rank2 is not None) else None)
        self.cfg = cfg
        handle = C.c_void_p()
        with torch.cuda.device(self.device) if self.lib.is_hip else _Null():
            st = self.lib.ccsd_plan_create(C.byref(cfg), blob.ctypes.data_as(C.POINTER(C.c_float)), blob.size,
                                           self.coef.ctypes.data_as(C.POINTER(_lib.StepCoef)), C.byref(handle))
        self.lib.check(st)
        self.handle = handle
        self._ws: Optional[torch.Tensor] = None
        self._ws_B = 0

    def __del__(self):
        h = getattr(self, "handle", None)
        if h:
            try:
                self.lib.ccsd_plan_destroy(h)
            except Exception:
                pass
            self.handle = None

    # -- helpers
    def _stream(self):
        if self.lib.is_hip:
            return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        return None

    def _workspace(self, B: int) -> Tuple[C.c_void_p, int]:
        n = self.lib.ccsd_workspace_bytes(self.handle, B)
        if self._ws is None or self._ws.numel() < n:
            self._ws = torch.empty(n, dtype=torch.uint8, device=self.device)
        return C.c_void_p(self._ws.data_ptr()), self._ws.numel()

    def _check(self, t: Optional[torch.Tensor], shape, name):
        if t is None:
            return
        if tuple(t.shape) != tuple(shape) or t.dtype != torch.float32 or not t.is_contiguous() or t.device.type != self.device.type:
            raise ValueError(f"{name}: expected contiguous float32 {tuple(shape)} on {self.device}, got {tuple(t.shape)} {t.dtype} {t.device}")

    def shapes(self, B: int):
        return (B, self.N, self.F), (B, self.N, self.N), (B, self.E, self.K)

    def _state(self, x, adj, rank2, B, name="state") -> _lib.State:
        sx, sa, sr = self.shapes(B)
        self._check(x, sx, name + ".x")
        self._check(adj, sa, name + ".adj")
        if self.is_cc:
            if rank2 is None:
                raise ValueError(f"{name}.rank2 is required for combinatorial complexes")
            self._check(rank2, sr, name + ".rank2")
        return _lib.State(_ptr(x), _ptr(adj), _ptr(rank2) if self.is_cc else None)

    def _noise(self, z: Optional[Sequence[Optional[torch.Tensor]]], B: int):
        if z is None:
            return None
        sx, sa, sr = self.shapes(B)
        self._check(z[0], sx, "noise.x")
        self._check(z[1], sa, "noise.adj")
        if self.is_cc:
            self._check(z[2], sr, "noise.rank2")
        return C.byref(_lib.Noise(_ptr(z[0]), _ptr(z[1]), _ptr(z[2]) if self.is_cc and len(z) > 2 else None))

    def alloc_state(self, B: int) -> List[Optional[torch.Tensor]]:
        sx, sa, sr = self.shapes(B)
        out = [torch.empty(sx, device=self.device), torch.empty(sa, device=self.device)]
        out.append(torch.empty(sr, device=self.device) if self.is_cc else None)
        return out

    # -- API
    def score(self, target: int, x, adj, rank2, flags, sscale: float = 1.0) -> torch.Tensor:
        B = x.shape[0]
        st = self._state(x, adj, rank2, B)
        self._check(flags, (B, self.N), "flags")
        out = torch.empty(self.shapes(B)[target], device=self.device)
        ws, n = self._workspace(B)
        self.lib.check(self.lib.ccsd_score(self.handle, target, B, C.byref(st), _ptr(flags), float(sscale), _ptr(out), ws, n,
                                           self._stream()))
        return out

    def init_state(self, flags, state, prior=None, seed: int = 0, sample_offset: int = 0):
        B = flags.shape[0]
        st = self._state(*state, B)
        self.lib.check(self.lib.ccsd_init_state(self.handle, B, _ptr(flags), self._noise(prior, B), seed, sample_offset,
                                                C.byref(st), self._stream()))

    def noise_draws(self, flags, step: int, phase: int, out, seed: int = 0, sample_offset: int = 0):
        """The masked Philox noise of half-step (step, phase) as the kernels consume it (ccsd_noise_draws): phase 0..n_steps-1
        = corrector inner iterations, n_steps = predictor; S4: 0, 1, 2.  `out` = [x, adj, rank2] tensors of the state's shapes."""
        B = flags.shape[0]
        so = self._state(*out, B, "out")
        self.lib.check(self.lib.ccsd_noise_draws(self.handle, B, _ptr(flags), seed, sample_offset, int(step), int(phase),
                                                 C.byref(so), self._stream()))

    def query(self, what: str) -> int:
        """Which kernels the plan selected (ccsd_plan_query): "fused_r2", "xa_variant", "r2_lds_bytes", "xa_lds_bytes", "fused_loop", "merged_r2", "ew1"."""
        v = C.c_int64(0)
        self.lib.check(self.lib.ccsd_plan_query(self.handle, _lib.QUERIES[what], C.byref(v)))
        return v.value

    def corrector_norms(self, step, it, base, cur, flags, noise, seed, sample_offset, sums):
        B = flags.shape[0]
        sb, sc = self._state(*base, B, "base"), self._state(*cur, B, "cur")
        ws, n = self._workspace(B)
        self.lib.check(self.lib.ccsd_corrector_norms(self.handle, B, step, it, C.byref(sb), C.byref(sc), _ptr(flags),
                                                     self._noise(noise, B), seed, sample_offset, _ptr(sums), ws, n,
                                                     self._stream()))

    def corrector_apply(self, step, it, cur, flags, noise, seed, sample_offset, sums, out):
        B = flags.shape[0]
        sc, so = self._state(*cur, B, "cur"), self._state(*out, B, "out")
        ws, n = self._workspace(B)
        self.lib.check(self.lib.ccsd_corrector_apply(self.handle, B, step, it, C.byref(sc), _ptr(flags),
                                                     self._noise(noise, B), seed, sample_offset, _ptr(sums), C.byref(so),
                                                     ws, n, self._stream()))

    def predictor(self, step, inp, flags, noise, seed, sample_offset, out, mean=None):
        B = flags.shape[0]
        si, so = self._state(*inp, B, "in"), self._state(*out, B, "out")
        sm = C.byref(self._state(*mean, B, "mean")) if mean is not None else None
        ws, n = self._workspace(B)
        self.lib.check(self.lib.ccsd_predictor(self.handle, B, step, C.byref(si), _ptr(flags), self._noise(noise, B), seed,
                                               sample_offset, C.byref(so), sm, ws, n, self._stream()))

    def s4_apply(self, step, cur, flags, noise1, noise2, noise3, seed, sample_offset, sums, out, mean=None):
        """Update half of one S4_solver step; corrector_norms(step, 0, cur, cur, ...) must have filled `sums`."""
        B = flags.shape[0]
        sc, so = self._state(*cur, B, "cur"), self._state(*out, B, "out")
        sm = C.byref(self._state(*mean, B, "mean")) if mean is not None else None
        ws, n = self._workspace(B)
        self.lib.check(self.lib.ccsd_s4_apply(self.handle, B, step, C.byref(sc), _ptr(flags), self._noise(noise1, B),
                                              self._noise(noise2, B), self._noise(noise3, B), seed, sample_offset,
                                              _ptr(sums), C.byref(so), sm, ws, n, self._stream()))

    def run(self, flags, state, scratch, result, seed: int = 0, sample_offset: int = 0, first_step: int = 0,
            last_step: Optional[int] = None, traj: Optional[torch.Tensor] = None):
        B = flags.shape[0]
        s, sc, r = self._state(*state, B), self._state(*scratch, B, "scratch"), self._state(*result, B, "result")
        ws, n = self._workspace(B)
        last = self.diff_steps if last_step is None else last_step
        self.lib.check(self.lib.ccsd_sampler_run(self.handle, B, _ptr(flags), seed, sample_offset, first_step, last,
                                                 C.byref(s), C.byref(sc), C.byref(r), _ptr(traj), ws, n, self._stream()))

    def init_and_run(self, flags, state, scratch, result, seed: int = 0, sample_offset: int = 0, first_step: int = 0,
                     last_step: Optional[int] = None, traj: Optional[torch.Tensor] = None):
        """init_state (in-kernel Philox prior) followed by run, with every argument of both calls prepared BEFORE the first one is
        issued: the two C calls go out back to back, so the GPU is not left idle between the prior draw and the loop's first
        launches while Python checks shapes and builds structs (a 20-step call is ~5 ms: ~25 us of that gap is 0.5 %)."""
        B = flags.shape[0]
        s, sc, r = self._state(*state, B), self._state(*scratch, B, "scratch"), self._state(*result, B, "result")
        ws, n = self._workspace(B)
        last = self.diff_steps if last_step is None else last_step
        fp, tp, stream, lib, h = _ptr(flags), _ptr(traj), self._stream(), self.lib, self.handle
        rc0 = lib.ccsd_init_state(h, B, fp, None, seed, sample_offset, C.byref(s), stream)
        rc1 = lib.ccsd_sampler_run(h, B, fp, seed, sample_offset, first_step, last, C.byref(s), C.byref(sc), C.byref(r), tp, ws, n, stream) if rc0 == 0 else 0
        lib.check(rc0)
        lib.check(rc1)

    def profile_kernel(self, name: Optional[str]):
        """Add a kernel to the set timed with HIP events on the launch stream (None clears the set)."""
        self.lib.check(self.lib.ccsd_profile_kernel(self.handle, -1 if name is None else _lib.KERNEL_IDS[name]))

    def profile_stride(self, stride: int):
        """Bracket only every `stride`-th launch of the selected kernels with events."""
        self.lib.check(self.lib.ccsd_profile_stride(self.handle, int(stride)))

    def profile_read(self, name: str) -> Tuple[int, float]:
        n, ms = C.c_int64(0), C.c_double(0.0)
        self.lib.check(self.lib.ccsd_profile_read(self.handle, _lib.KERNEL_IDS[name], C.byref(n), C.byref(ms)))
        return n.value, ms.value

    def profile_launches(self, name: str) -> int:
        """All launches of a selected kernel since the selection / stride was set (bracketed by events or not)."""
        n = C.c_int64(0)
        self.lib.check(self.lib.ccsd_profile_launches(self.handle, _lib.KERNEL_IDS[name], C.byref(n)))
        return n.value

    def quantize(self, t: torch.Tensor, thr: float = 0.5) -> torch.Tensor:
        """thr < 0 selects quantize_mol's 0/1/2/3 bins."""
        t = t.contiguous()
        out = torch.empty(t.shape, dtype=torch.int64, device=t.device)
        self.lib.check(self.lib.ccsd_quantize(_ptr(t), t.numel(), float(thr), _ptr(out), self._stream()))
        return out


    def rank2_cells(self, rank2: torch.Tensor, thr: float = 0.5) -> Tuple[torch.Tensor, torch.Tensor]:
        """Sparse form of quantize(rank2): (bits (B, ceil(K/64)) int64 -- bit k%64 of word k//64 set iff column k holds a
        rank-2 cell --, counts (B,) int32).  cells_from_bits turns a row into the cell tuples cc_from_incidence adds."""
        rank2 = rank2.contiguous()
        B, E, K = rank2.shape
        bits = torch.zeros(B, (K + 63) // 64, dtype=torch.int64, device=rank2.device)
        counts = torch.zeros(B, dtype=torch.int32, device=rank2.device)
        self.lib.check(self.lib.ccsd_rank2_cells(_ptr(rank2), B, E, K, float(thr), _ptr(bits), _ptr(counts), self._stream()))
        return bits, counts


def cells_from_bits(bits_row, N: int, d_min: int, d_max: int):
    """Cell tuples of one complex from its bitmask row, in the reference's enumeration order (get_cells,
    cc_utils.py:72-94: itertools.combinations(range(N), d) for d = d_min..d_max)."""
    from itertools import combinations

    words = [int(w) & 0xFFFFFFFFFFFFFFFF for w in bits_row.tolist()]
    out, k = [], 0
    for d in range(d_min, d_max + 1):
        for combi in combinations(range(N), d):
            if (words[k >> 6] >> (k & 63)) & 1:
                out.append(combi)
            k += 1
    return out


class _Null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False
