// ccsd_r2.hip -- product translation unit: instantiations of the fused rank-2 kernel k_r2 -- the qm9_CC geometry (E = 36: MT = 3, RS = 1), affine ScoreNetworkF.
#include "ccsd_dev.h"
#include "ccsd_k_r2.h"
#define CCSD_INST template
#define CCSD_INST_R2_A
#include "ccsd_instances.h"
