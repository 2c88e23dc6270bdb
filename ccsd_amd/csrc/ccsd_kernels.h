// ccsd_kernels.h -- hand-written gfx950 kernels for the CCSD predictor-corrector sampling path (umbrella header).
//
// Data layout in HBM (all fp32, contiguous, batch-major; same as the reference tensors):
//   x (B,N,F)   adj (B,N,N)   rank2 (B,E,K)   flags (B,N) of exact 0/1
//   H (B,E,E) = F F^T (diag zeroed)            P_l (B*E, wc_l) = hodge Q|K projections of layer l
//
// Source map
//   ccsd_dev.h       device helpers: Philox4x32-10, fast math, MFMA building blocks (block_linear, mlp_chain_tile, gcn_tile)
//   ccsd_rank2_common.h  masks, epilogue modes, ScoreNetworkF per element, wave-level tile loops (shared helpers)
//   ccsd_k_rank2.h   k_flagbits, k_gemm_h (H = F F^T), k_gemm_p / k_gemm_p0 (hodge projections), k_edgecoef,
//                    k_hf_score ((H F) + ScoreNetworkF epilogue + fused predictor update / Langevin norms): the tiled rank-2 path
//   ccsd_k_r2.h      k_r2: the fused rank-2 kernel, one complex per workgroup with its rank2 block LDS-resident (E <= 64)
//   ccsd_k_xa.h      k_xa: ScoreNetworkX + ScoreNetworkA(_CC): one workgroup per graph, everything LDS-resident
//                    (+ ccsd_attn_stack.inc: the AttentionLayer stack)
//   ccsd_k_update.h  k_normsum, k_langevin_apply, k_s4_apply, k_init_state, k_quantize, k_rank2_cells
// The product library is built from several translation units compiled in parallel (ccsd_hip.hip: C ABI + the small kernels;
// ccsd_r2*.hip / ccsd_xa.hip: the explicit instantiations of the two big kernel templates); the host emulation used by the
// CPU tests includes everything in one unit.  Reference file:line citations sit next to each restated formula.
#pragma once
#include "ccsd_dev.h"
#include "ccsd_rank2_common.h"
#include "ccsd_k_rank2.h"
#include "ccsd_k_r2.h"
#include "ccsd_k_xa.h"
#include "ccsd_k_update.h"
