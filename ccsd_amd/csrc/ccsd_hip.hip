// ccsd_hip.hip -- product translation unit: gfx950 kernels + C ABI (libccsd_hip.so).
// Build: hipcc --offload-arch=gfx950 -O3 -fPIC -shared ccsd_hip.hip -o libccsd_hip.so
#include "ccsd_api.h"
