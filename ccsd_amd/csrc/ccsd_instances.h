// ccsd_instances.h -- the instantiations of the two big kernel templates.  CCSD_INST expands to `template` in the
// translation unit that owns an instance (ccsd_r2*.hip, ccsd_xa.hip) and to `extern template` in ccsd_hip.hip (the C ABI, which
// launches them): the units compile in parallel, each kernel is compiled once.  Keep in step with R2_DISPATCH / launch_xa.
#define CCSD_R2_SIG (const PlanD* __restrict__, const float* __restrict__, const unsigned char* __restrict__, \
                     const unsigned long long* __restrict__, R2Args, RankEpi, NoiseArgs)
#define CCSD_R2_ONE(MT_, RS_) \
    CCSD_INST __global__ void k_r2<MT_, RS_, true, false> CCSD_R2_SIG; \
    CCSD_INST __global__ void k_r2<MT_, RS_, true, true> CCSD_R2_SIG;
#define CCSD_R2_GEN(MT_) \
    CCSD_INST __global__ void k_r2<MT_, 0, false, false> CCSD_R2_SIG; \
    CCSD_INST __global__ void k_r2<MT_, 0, false, true> CCSD_R2_SIG;
// k_r2 is instantiated in four units (ccsd_r2.hip: the qm9 geometry E = 36; ccsd_r2b.hip / ccsd_r2c.hip: the other affine
// shapes; ccsd_r2d.hip: the non-affine ScoreNetworkF path) so that the units compile in parallel
#ifdef CCSD_INST_R2_A
CCSD_R2_ONE(3, 1)
CCSD_INST __global__ void k_r2<3, 1, true, false, 1> CCSD_R2_SIG;     // the qm9 geometry compiled in (QM9 = 1)
CCSD_INST __global__ void k_r2<3, 1, true, false, 2> CCSD_R2_SIG;     // ... and the whole baked plan (QM9 = 2)
#endif
#ifdef CCSD_INST_R2_B
CCSD_R2_ONE(3, 0) CCSD_R2_ONE(4, 2) CCSD_R2_ONE(2, 2) CCSD_R2_ONE(2, 3)
#endif
#ifdef CCSD_INST_R2_C
CCSD_R2_ONE(1, 0) CCSD_R2_ONE(1, 1) CCSD_R2_ONE(1, 2) CCSD_R2_ONE(1, 3)
#endif
#ifdef CCSD_INST_R2_D
CCSD_R2_GEN(1) CCSD_R2_GEN(2) CCSD_R2_GEN(3) CCSD_R2_GEN(4)
CCSD_INST __global__ void k_r2<3, 0, false, false, 1> CCSD_R2_SIG;     // non-affine ScoreNetworkF, the qm9 geometry compiled in (qm9_Base_CC)
#endif
#define CCSD_XA_SIG (const PlanD* __restrict__, const float* __restrict__, const unsigned char* __restrict__, XaArgs, NoiseArgs)
#ifdef CCSD_INST_XA
CCSD_INST __global__ void k_xa<false, XA_PLAIN> CCSD_XA_SIG;
CCSD_INST __global__ void k_xa<false, XA_HB> CCSD_XA_SIG;
CCSD_INST __global__ void k_xa<false, XA_GMH> CCSD_XA_SIG;
CCSD_INST __global__ void k_xa<false, XA_GEN> CCSD_XA_SIG;
CCSD_INST __global__ void k_xa<false, XA_PLAIN9> CCSD_XA_SIG;
CCSD_INST __global__ void k_xa<false, XA_BAKED9> CCSD_XA_SIG;
CCSD_INST __global__ void k_xa<false, XA_BAKEDENZ> CCSD_XA_SIG;
CCSD_INST __global__ void k_xa<true, XA_PLAIN> CCSD_XA_SIG;
CCSD_INST __global__ void k_xa<true, XA_HB> CCSD_XA_SIG;
CCSD_INST __global__ void k_xa<true, XA_GMH> CCSD_XA_SIG;
CCSD_INST __global__ void k_xa<true, XA_GEN> CCSD_XA_SIG;
CCSD_INST __global__ void k_xa<true, XA_PLAIN20> CCSD_XA_SIG;
CCSD_INST __global__ void k_xa<true, XA_BAKED20> CCSD_XA_SIG;
CCSD_INST __global__ void k_xa<true, XA_BAKED38> CCSD_XA_SIG;
CCSD_INST __global__ void k_xa<true, XA_PLAIN38> CCSD_XA_SIG;
#endif
