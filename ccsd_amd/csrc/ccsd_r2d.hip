// ccsd_r2d.hip -- product translation unit: instantiations of the fused rank-2 kernel k_r2 -- the non-affine ScoreNetworkF path (per-element MLPs).
#include "ccsd_dev.h"
#include "ccsd_k_r2.h"
#define CCSD_INST template
#define CCSD_INST_R2_D
#include "ccsd_instances.h"
