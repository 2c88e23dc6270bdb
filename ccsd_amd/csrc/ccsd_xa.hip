// ccsd_xa.hip -- product translation unit 3 of 3: the instantiations of the graph-network kernel k_xa.
#include "ccsd_dev.h"
#include "ccsd_k_xa.h"
#define CCSD_INST template
#define CCSD_INST_XA
#include "ccsd_instances.h"
