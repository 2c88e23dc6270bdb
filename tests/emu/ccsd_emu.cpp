// CPU emulation build of the kernel source (TEST INFRASTRUCTURE ONLY -- see ccsd_amd/csrc/ccsd_rt.h).
// Build: g++ -O2 -DCCSD_EMU -fPIC -shared tests/emu/ccsd_emu.cpp -o tests/emu/_build/libccsd_emu.so
#ifndef CCSD_EMU
#error "compile with -DCCSD_EMU"
#endif
#include "../../ccsd_amd/csrc/ccsd_api.h"
