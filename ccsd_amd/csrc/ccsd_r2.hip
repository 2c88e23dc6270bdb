// ccsd_r2.hip -- product translation unit 2 of 3: the instantiations of the fused rank-2 kernel k_r2.
#include "ccsd_dev.h"
#include "ccsd_k_r2.h"
#define CCSD_INST template
#define CCSD_INST_R2
#include "ccsd_instances.h"
