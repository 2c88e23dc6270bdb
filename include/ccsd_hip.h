/*
 * ccsd_hip.h -- C ABI of the MI355X-native CCSD reverse-SDE sampling path (libccsd_hip.so).
 *
 * Drop-in boundary.  The reference (AdrienC21/CCSD v0.3.3) is pure Python; its seam for this
 * path is  load_sampling_fn(...) -> sampling_fn(model_x, model_adj[, model_rank2], init_flags)
 * (ccsd/src/utils/loader.py:337-458, ccsd/src/solver.py:856-1176).  A maintainer binds this
 * library with ctypes; every pointer marked `dev` is a device pointer into a
 * caller-owned allocation (PyTorch-ROCm tensors in practice), fp32, contiguous, batch-major.
 * No torch types appear in any signature.  All calls are asynchronous on `stream`
 * (a hipStream_t passed as void*; NULL = the default stream) and return an int status.
 *
 * Entry point                  replaces (reference file:line)
 * ---------------------------  -----------------------------------------------------------------
 * ccsd_plan_create/destroy     load_model_from_ckpt + load_sde + get_pc_sampler closure setup
 *                              (loader.py:619-657, 242-267; solver.py:856-1104)
 * ccsd_score                   get_score_fn / get_score_fn_cc applied to ScoreNetworkX /
 *                              ScoreNetworkA(_CC) / ScoreNetworkF.forward
 *                              (losses.py:18-198; models/ScoreNetwork_{X,A,A_CC,F}.py)
 * ccsd_init_state              sde.prior_sampling(_sym) + mask_x/mask_adjs/mask_rank2
 *                              (solver.py:1111-1118; sde.py:426-449, 583-608)
 * ccsd_corrector_norms         first half of LangevinCorrector.update_fn_* : score, noise, the two
 *                              batch norms (solver.py:759-767, 773-780, 787-797)
 * ccsd_corrector_apply         second half: step_size, x_mean, x (solver.py:767-769, 781-783, 797-801)
 * ccsd_predictor               ReverseDiffusionPredictor / EulerMaruyamaPredictor.update_fn_*
 *                              (solver.py:210-313, 367-463) + RSDE.sde/discretize (sde.py:180-340)
 * ccsd_s4_apply                the update half of one S4_solver step (solver.py:1296-1352, 1446-1529)
 * ccsd_sampler_run             the whole pc_sampler / s4_solver loop (solver.py:1109-1174, 1266-1352)
 * ccsd_quantize                quantize_mol / quantize (graph_utils.py:181-213)
 * ccsd_rank2_cells             the rank-2 part of cc_from_incidence's input, as a cell bitmask (cc_utils.py:243-262)
 */
#ifndef CCSD_HIP_H
#define CCSD_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CCSD_ABI_VERSION 5

/* status codes; the Python shim re-raises the reference's exception types */
enum {
    CCSD_OK = 0,
    CCSD_ERR_INVALID = 1,       /* ValueError: bad argument / shape / NULL */
    CCSD_ERR_UNSUPPORTED = 2,   /* NotImplementedError: config outside the HIP path's envelope */
    CCSD_ERR_WEIGHTS = 3,       /* ValueError: weight blob size does not match the config */
    CCSD_ERR_RUNTIME = 4,       /* RuntimeError: HIP runtime failure (hipGetLastError text via ccsd_last_error) */
    CCSD_ERR_WORKSPACE = 5      /* ValueError: workspace too small */
};

enum { CCSD_SDE_VP = 0, CCSD_SDE_VE = 1, CCSD_SDE_SUBVP = 2 };
enum { CCSD_PRED_EULER = 0, CCSD_PRED_REVERSE = 1, CCSD_PRED_S4 = 2 };
enum { CCSD_CORR_NONE = 0, CCSD_CORR_LANGEVIN = 1 };
enum { CCSD_TARGET_X = 0, CCSD_TARGET_ADJ = 1, CCSD_TARGET_RANK2 = 2 };

/* Per diffusion step, per target (x, adj, rank2) scalars.  The host computes them with the same
 * fp32 arithmetic as the reference's SDE classes (sde.py) so that the device never re-derives a
 * table index or a sigma:
 *   sscale : score = sscale * net(...)      (1 for VE, -1/std(t) for VP/subVP; losses.py:157-163)
 *   alpha  : Langevin alpha                 (alphas[timestep] for VP/subVP, 1 for VE; solver.py:752-756)
 *   pa,pb,pc : predictor  v_mean = pa*v + pb*net ; v = v_mean + pc*z   (pb already includes sscale)
 *   m1,s1,d,m2,s2 : S4_solver (solver.py:1179-1563), after its Langevin-style correction v1:
 *              v2 = m1*v1 + s1*z2   (sde.transition(v, t, dt/2));   v3 = v2 + d*net   (Sdrift*dt, d = -g(t)^2*sscale*dt);
 *              v_mean = m2*v3 ; v = v_mean + s2*z3   (sde.transition(v, t + dt/2, dt/2)).  Zero for the PC predictors.
 */
typedef struct {
    float sscale, alpha, pa, pb, pc;
    float m1, s1, d, m2, s2;
} ccsd_step_coef_t;

typedef struct {
    int32_t abi_version;        /* = CCSD_ABI_VERSION */
    /* shapes */
    int32_t N, F, is_cc, d_min, d_max;   /* E and K are derived: E=N(N-1)/2, K=sum C(N,k) */
    /* ScoreNetworkX (ScoreNetwork_X.py:26-75) */
    int32_t x_depth, x_nhid;
    /* ScoreNetworkA / ScoreNetworkA_CC graph branch (ScoreNetwork_A.py:351-460) */
    int32_t a_num_layers, a_num_linears, a_c_init, a_c_hid, a_c_final, a_nhid, a_adim, a_num_heads;
    /* ScoreNetworkA_CC hodge branch (ScoreNetwork_A_CC.py:155-205); a_is_cc_net=0 -> ScoreNetworkA;
     * a_is_cc_net=2 -> ScoreNetworkA_Base_CC (ScoreNetwork_A_Base_CC.py:105-195): HodgeBaselineLayers, h_nhid = nhid_h,
     * h_adim = hidden_h, h_num_heads unused */
    int32_t a_is_cc_net, h_num_layers, h_num_linears, h_nhid, h_adim, h_c_hid, h_c_final, h_num_heads;
    /* ScoreNetworkF (ScoreNetwork_F.py:24-145) */
    int32_t f_num_layers, f_num_linears, f_nhid, f_c_hid, f_c_final, f_cnum, f_num_layers_mlp, f_use_hodge_mask;
    /* sampler (solver.py:856-875) */
    int32_t predictor, corrector, n_corr_steps, probability_flow, denoise;
    float snr, scale_eps;
    int32_t diff_steps;         /* sde_adj.N == number of rows of step_coef */
    int32_t batch_hint;         /* expected batch per launch (0 = unknown): picks the LDS layout of the graph-network kernel so
                                 * that ceil(batch / #CUs) workgroups are co-resident per CU; any batch stays correct */
    /* ScoreNetworkX_GMH (ScoreNetwork_X.py:156-341) when x_gmh=1: x_depth AttentionLayers (x_nhid wide) instead of GCN layers */
    int32_t x_gmh, x_num_linears, x_c_init, x_c_hid, x_c_final, x_adim, x_num_heads;
    /* conv="MLP" (attention.py:168-178): Q and K of every Attention are 2-layer tanh MLPs of x (no adjacency) instead of
     * DenseGCNConv; V stays a DenseGCNConv.  a_: the A-network's AttentionLayers, x_: ScoreNetworkX_GMH's */
    int32_t a_conv_mlp, x_conv_mlp;
} ccsd_config_t;

typedef struct ccsd_plan ccsd_plan_t;

/* Build a plan: validates the config, uploads weights and tables to the current HIP device.
 * `weights` is a HOST pointer to the canonical weight blob (order documented in DESIGN.md,
 * produced by ccsd_amd.plan.pack_weights); `step_coef` is a HOST array [diff_steps][3]. */
int ccsd_plan_create(const ccsd_config_t* cfg, const float* weights, size_t n_weights,
                     const ccsd_step_coef_t* step_coef, ccsd_plan_t** out);
void ccsd_plan_destroy(ccsd_plan_t* plan);

/* number of floats the canonical weight blob must hold for this config (0 on invalid config) */
size_t ccsd_weight_count(const ccsd_config_t* cfg);
/* E and K for the config */
void ccsd_rank2_dims(const ccsd_config_t* cfg, int32_t* E, int64_t* K);
/* bytes of device scratch the step/run calls need for batch B */
size_t ccsd_workspace_bytes(const ccsd_plan_t* plan, int32_t B);

const char* ccsd_last_error(void);

/* Device state of one batch shard: caller-owned, fp32, contiguous.  rank2 may be NULL when !is_cc. */
typedef struct {
    float* x;      /* dev (B,N,F) */
    float* adj;    /* dev (B,N,N) */
    float* rank2;  /* dev (B,E,K) */
} ccsd_state_t;

/* Noise of one half-step.  NULL pointers => counter-based Philox4x32-10 in-kernel, keyed by
 * (seed, draw index, global sample index, element).  Non-NULL => the RAW standard-normal draw the
 * reference would have obtained from randn_like (full (B,N,N) for adj; the kernel applies
 * triu(1)+transpose and the flag masks exactly as gen_noise does, graph_utils.py:171-178). */
typedef struct {
    const float* zx;     /* dev (B,N,F) or NULL */
    const float* zadj;   /* dev (B,N,N) or NULL */
    const float* zrank2; /* dev (B,E,K) or NULL */
} ccsd_noise_t;

/* score = sscale(t) * net(x, adj, rank2, flags) for one target; `sscale` is passed by the caller
 * (1 for VE, -1/std(t) for VP).  out has the target's shape. */
int ccsd_score(ccsd_plan_t* plan, int32_t target, int32_t B, const ccsd_state_t* in, const float* flags_dev,
               float sscale, float* out_dev, void* workspace, size_t workspace_bytes, void* stream);

/* state <- masked prior.  prior==NULL: Philox draws (draw index 0..2); else mask the given raw draws.
 * Takes no workspace: its off-bit table lives in a plan-owned buffer.  Like every call on a plan it is reentrant per handle
 * only, and all calls on one plan belong on ONE stream (the buffer is shared between calls; when it has to grow the call
 * synchronises the device). */
int ccsd_init_state(ccsd_plan_t* plan, int32_t B, const float* flags_dev, const ccsd_noise_t* prior,
                    uint64_t seed, int64_t sample_offset, ccsd_state_t* state, void* stream);

/* The masked noise of one half-step exactly as the kernels of ccsd_sampler_run / the step calls consume it with NULL noise
 * pointers: out->{x, adj, rank2} <- gen_noise / gen_noise_rank2 (graph_utils.py:158-178, cc_utils.py:594-615) of the Philox
 * draws keyed by (seed, sample_offset + b, draw index of (step, phase)).  phase: 0 .. n_steps-1 = the Langevin corrector's
 * inner iterations, n_steps = the predictor (S4 plans: 0, 1, 2 = the three draws of a step).  Test / audit hook: feeding
 * these tensors to a CPU run of the reference algorithm as its noise stream makes the production loop (in-kernel noise,
 * fused corrector apply) comparable with it value for value.  Same generator code as the kernels (philox_normal4). */
int ccsd_noise_draws(ccsd_plan_t* plan, int32_t B, const float* flags_dev, uint64_t seed, int64_t sample_offset,
                     int32_t step, int32_t phase, ccsd_state_t* out, void* stream);

/* Which kernels a plan selected (host-side facts, no device work). */
enum {
    CCSD_QUERY_FUSED_R2 = 0,      /* 1: the LDS-resident fused rank-2 kernel k_r2 serves the rank-2 side; 0: the tiled kernels */
    CCSD_QUERY_XA_VARIANT = 1,    /* instantiation of the graph-network kernel k_xa: 0 plain, 1 HodgeBaseline, 2 X_GMH, 3 general; 4 / 5 / 6 plain with
                                     the qm9 / community_small / zinc250k geometry compiled in; 7 .. 10 the whole plan of a shipped configuration
                                     compiled in (qm9_CC, community_small_CC, zinc250k, ENZYMES_small_CC at their bench batch) */
    CCSD_QUERY_R2_LDS_BYTES = 2,
    CCSD_QUERY_XA_LDS_BYTES = 3,
    CCSD_QUERY_FUSED_LOOP = 4,    /* 1: ccsd_sampler_run fuses the Langevin apply into the predictor launches */
    CCSD_QUERY_MERGED_R2 = 5,     /* 1: ... and k_r2 runs the predictor half-step of step i and the rank-2 side of the norms pass of step i + 1
                                     in one launch (one block load per PC step) */
    CCSD_QUERY_EW1 = 6            /* 1: element-wise rank-2 kernel k_ew1 (affine ScoreNetworkF without a Hodge Laplacian term, cnum = 1) */
};
int ccsd_plan_query(const ccsd_plan_t* plan, int32_t what, int64_t* value);

/* Langevin corrector, phase 1: evaluate the three scores (all correctors see the same pre-corrector
 * state `base`, solver.py:1129-1137; with n_steps > 1 each target's own tensor is taken from `cur`, its
 * current inner iterate, solver.py:760-769; cur == base for the first inner step), keep them in the
 * workspace, and write
 * norm_sums_dev[6] = { sum_b ||net_x[b]||, sum_b ||net_adj[b]||, sum_b ||net_rank2[b]||,
 *                      sum_b ||z_x[b]||,  sum_b ||z_adj[b]||,  sum_b ||z_rank2[b]|| }.
 * In multi-GPU exact mode the caller all-reduces these six floats (RCCL) between the two phases. */
int ccsd_corrector_norms(ccsd_plan_t* plan, int32_t B, int32_t step, int32_t corr_iter,
                         const ccsd_state_t* base, const ccsd_state_t* cur, const float* flags_dev, const ccsd_noise_t* noise,
                         uint64_t seed, int64_t sample_offset, float* norm_sums_dev,
                         void* workspace, size_t workspace_bytes, void* stream);
/* phase 2: step_size = (snr*zn/gn)^2*2*alpha from norm_sums_dev; out = cur + step*score + sqrt(2 step)*z*scale_eps */
int ccsd_corrector_apply(ccsd_plan_t* plan, int32_t B, int32_t step, int32_t corr_iter,
                         const ccsd_state_t* cur, const float* flags_dev, const ccsd_noise_t* noise,
                         uint64_t seed, int64_t sample_offset, const float* norm_sums_dev,
                         ccsd_state_t* out, void* workspace, size_t workspace_bytes, void* stream);

/* predictor half-step: out = pa*in + pb*net(in) + pc*z; mean (nullable) = pa*in + pb*net(in). */
int ccsd_predictor(ccsd_plan_t* plan, int32_t B, int32_t step, const ccsd_state_t* in, const float* flags_dev,
                   const ccsd_noise_t* noise, uint64_t seed, int64_t sample_offset,
                   ccsd_state_t* out, ccsd_state_t* mean, void* workspace, size_t workspace_bytes, void* stream);

/* S4_solver (plans created with predictor = CCSD_PRED_S4): one step = ccsd_corrector_norms(step, corr_iter 0, base = cur =
 * state) -- the three scores at the current state, the first noise draw and the six norm sums -- followed by
 * ccsd_s4_apply: Langevin-style correction with that score and noise (solver.py:1296-1334), transition kernel over dt/2
 * with a second draw, the score drift over dt, transition kernel over dt/2 with a third draw (solver.py:1337-1352).
 * noise1 must be the draws given to ccsd_corrector_norms; NULL noise pointers select in-kernel Philox.  `mean`
 * (nullable) receives the last transition's mean. */
int ccsd_s4_apply(ccsd_plan_t* plan, int32_t B, int32_t step, const ccsd_state_t* cur, const float* flags_dev,
                  const ccsd_noise_t* noise1, const ccsd_noise_t* noise2, const ccsd_noise_t* noise3, uint64_t seed,
                  int64_t sample_offset, const float* norm_sums_dev, ccsd_state_t* out, ccsd_state_t* mean,
                  void* workspace, size_t workspace_bytes, void* stream);

/* The whole loop with in-kernel Philox noise and per-shard Langevin norms (the reference's own
 * divide_batch semantics, sampler.py:1199-1211).  `state` holds the prior on entry (see
 * ccsd_init_state) and the last state on exit; `result` receives the means (denoise) or the state;
 * `scratch` is a second state used for ping-pong.  PRECONDITION: `state` is MASKED by `flags_dev` -- x rows, adj rows / columns
 * and rank2 rows / columns of switched-off nodes hold zeros, as in every state the reference's loop ever sees (its prior is masked,
 * solver.py:1111-1118, and every update preserves the masks) and in everything ccsd_init_state or an earlier ccsd_sampler_run
 * wrote.  The loop's rank-2 kernels rely on it (they skip re-masking rank2 in the hodge-projection loader); the step calls above
 * (ccsd_score, ccsd_corrector_norms, ccsd_predictor, ...) accept arbitrary states.  traj_dev (nullable): [diff_steps][N*F+N*N+E*K]
 * receives sample 0 of every step (diff_traj, solver.py:1150-1165).  first_step/last_step allow
 * running a sub-range [first_step, last_step) of the diff_steps steps. */
int ccsd_sampler_run(ccsd_plan_t* plan, int32_t B, const float* flags_dev, uint64_t seed, int64_t sample_offset,
                     int32_t first_step, int32_t last_step, ccsd_state_t* state, ccsd_state_t* scratch,
                     ccsd_state_t* result, float* traj_dev, void* workspace, size_t workspace_bytes, void* stream);

/* quantize_mol: >=2.5->3, [1.5,2.5)->2, [0.5,1.5)->1, <0.5->0 (int64 out); thr<0 selects it, otherwise
 * quantize(t, thr): t<thr ? 0 : 1. */
int ccsd_quantize(const float* in_dev, int64_t n, float thr, int64_t* out_dev, void* stream);

/* Sparse form of the quantised rank-2 incidence matrix, the input of cc_from_incidence (cc_utils.py:243-262): column k
 * of rank2 (B,E,K) holds a rank-2 cell iff any entry of the column is >= thr (quantize(), graph_utils.py:181-192).
 * bits_dev: (B, ceil(K/64)) uint64, bit (k % 64) of word k / 64; counts_dev: (B,) int32 number of cells.  The column index
 * k enumerates itertools.combinations(range(N), d) for d = d_min..d_max (get_cells, cc_utils.py:72-94).  Replaces the
 * (B,E,K) fp32 device-to-host copy after sampling by ~K/8 bytes per complex. */
int ccsd_rank2_cells(const float* rank2_dev, int32_t B, int32_t E, int64_t K, float thr, uint64_t* bits_dev,
                     int32_t* counts_dev, void* stream);

/* Measurement hooks (bench.py): time every launch of selected kernels with HIP events on the launch stream.
 * kernel_id: 0 k_xa, 1 k_gemm_p, 2 k_hf_score, 3 k_gemm_h, 4 k_langevin_apply, 5 k_r2, 6 k_s4_apply, 7 k_ew1; each call adds one kernel to the
 * selection, -1 clears it.  ccsd_profile_read synchronises on that kernel's events and returns launches + summed ms. */
int ccsd_profile_kernel(ccsd_plan_t* plan, int32_t kernel_id);
/* bracket only every stride-th launch of the selected kernels (default 1 = every launch): event records break
 * back-to-back dispatch, so dense bracketing perturbs the timed region (~6 % of a qm9_CC step). */
int ccsd_profile_stride(ccsd_plan_t* plan, int32_t stride);
int ccsd_profile_read(ccsd_plan_t* plan, int32_t kernel_id, int64_t* launches, double* total_ms);
/* ALL launches of a selected kernel since the selection / stride was last set (bracketed or not): with the mean of the
 * bracketed ones this gives the kernel's share of a timed region */
int ccsd_profile_launches(ccsd_plan_t* plan, int32_t kernel_id, int64_t* launches);
/* Diagnostic: when dev_buffer (B x 64 int64, device) is non-NULL, thread 0 of every workgroup of k_r2 (slots 0-31)
 * and k_xa (slots 32-63) stores the shader clock at its phase boundaries (tools/stamps.py).  NULL disables. */
int ccsd_debug_stamps(ccsd_plan_t* plan, void* dev_buffer);

#ifdef __cplusplus
}
#endif
#endif /* CCSD_HIP_H */
