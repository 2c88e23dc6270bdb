// ccsd_hip.hip -- product translation unit 1 of 6: the C ABI (libccsd_hip.so) and the small kernels.  k_r2 / k_xa are
// instantiated in ccsd_r2*.hip / ccsd_xa.hip and only declared here.
// Build (see __graft_entry__.build): hipcc --offload-arch=gfx950 -O3 -fPIC -c <unit>.hip for the units in parallel,
// then hipcc --offload-arch=gfx950 -shared *.o -o libccsd_hip.so
#include "ccsd_kernels.h"
#define CCSD_INST extern template
#define CCSD_INST_R2_A
#define CCSD_INST_R2_B
#define CCSD_INST_R2_C
#define CCSD_INST_R2_D
#define CCSD_INST_XA
#include "ccsd_instances.h"
#include "ccsd_api.h"
