// ccsd_r2b.hip -- product translation unit: instantiations of the fused rank-2 kernel k_r2 -- the other affine shapes with two to four row tiles.
#include "ccsd_dev.h"
#include "ccsd_k_r2.h"
#define CCSD_INST template
#define CCSD_INST_R2_B
#include "ccsd_instances.h"
