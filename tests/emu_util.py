"""Build + load the CPU emulation of the kernel source (tests only; see ccsd_amd/csrc/ccsd_rt.h)."""
import os
import subprocess

from ccsd_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "emu", "ccsd_emu.cpp")
OUT = os.path.join(ROOT, "tests", "emu", "_build", "libccsd_emu.so")
DEPS = [SRC] + [os.path.join(ROOT, "ccsd_amd", "csrc", f) for f in ("ccsd_rt.h", "ccsd_plan.h", "ccsd_kernels.h", "ccsd_dev.h", "ccsd_rank2_common.h", "ccsd_k_rank2.h", "ccsd_k_r2.h", "ccsd_k_xa.h", "ccsd_k_update.h", "ccsd_attn_stack.inc", "ccsd_api.h", "ccsd_baked_qm9.h", "ccsd_baked_cs.h", "ccsd_baked_z.h", "ccsd_baked_enz.h", "ccsd_instances.h")] + \
       [os.path.join(ROOT, "include", "ccsd_hip.h")]

_emu = None


def emu_library() -> _lib.Library:
    global _emu
    if _emu is None:
        os.makedirs(os.path.dirname(OUT), exist_ok=True)
        stale = not os.path.exists(OUT) or any(os.path.getmtime(d) > os.path.getmtime(OUT) for d in DEPS)
        if stale:
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-DCCSD_EMU", "-fPIC", "-shared", "-Wno-unknown-pragmas",
                                   SRC, "-o", OUT])
        _emu = _lib.Library(OUT, is_hip=False)
    return _emu
