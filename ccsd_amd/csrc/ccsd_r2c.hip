// ccsd_r2c.hip -- product translation unit: instantiations of the fused rank-2 kernel k_r2 -- the affine shapes with one row tile (E <= 15).
#include "ccsd_dev.h"
#include "ccsd_k_r2.h"
#define CCSD_INST template
#define CCSD_INST_R2_C
#include "ccsd_instances.h"
