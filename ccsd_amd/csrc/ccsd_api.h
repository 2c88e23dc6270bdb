// ccsd_api.h -- host side of the C ABI declared in include/ccsd_hip.h: plan construction,
// workspace carve-up and the launch sequences of one corrector / predictor half-step.
// Included once by ccsd_hip.hip (product) and by tests/emu/ccsd_emu.cpp (CPU emulation of the same
// kernels, test infrastructure only).
#pragma once
#include "ccsd_kernels.h"
#include <stdio.h>
#include <stdlib.h>
#include <string>
#include <vector>

static thread_local std::string g_last_error;
static int set_err(int st, const std::string& m) { g_last_error = m; return st; }

// the k_xa variant a plan needs, and its instantiation (for hipFuncSetAttribute)
static inline bool plan_is_baked(const PlanD& p, const unsigned char* baked, size_t baked_size) {
    if (baked_size != sizeof(PlanD) || p.geo_off != 0) return false;
    unsigned char bytes[sizeof(PlanD)];
    ccsd_plan_arch_bytes(p, bytes);
    return memcmp(bytes, baked, sizeof(PlanD)) == 0;
}
static inline int xa_variant_sem(const PlanD& p) {          // what the network needs: XA_PLAIN / XA_HB / XA_GMH / XA_GEN
    bool conv_mlp = false;
    for (int l = 0; l < p.a_L; ++l) conv_mlp = conv_mlp || p.al[l].conv_mlp;
    if (p.x_gmh) for (int l = 0; l < p.x_depth; ++l) conv_mlp = conv_mlp || p.gl[l].conv_mlp;
    if (conv_mlp || (p.hb_L && p.x_gmh) || p.h_L > 2) return XA_GEN;
    // the small-graph XA_PLAIN / XA_GMH variants are compiled without the two widest final-MLP chain shapes (ccsd_k_xa.h: they cost
    // them their registers -- 30 VGPRs spilled around the final MLP of every launch); the HodgeBaseline networks need them
    if (!p.chan_global && !p.hb_L && p.a_fin.chain >= 5) return XA_GEN;
    if (p.hb_L) return XA_HB;
    if (p.x_gmh) return XA_GMH;
    return XA_PLAIN;
}
static inline int xa_variant(const PlanD& p) {
    const int sem = xa_variant_sem(p);
    // instances with the WHOLE plan as a compile-time constant: only for a plan whose architecture bytes equal a baked one's
    // (tools/bake_plan.py; ccsd_baked_*.h), and only in the instantiation the bake was made for
    if (sem == XA_PLAIN && !p.chan_global && plan_is_baked(p, CCSD_BAKED_QM9_PLAN, CCSD_BAKED_QM9_SIZE)) return XA_BAKED9;
    if (sem == XA_PLAIN && p.chan_global && plan_is_baked(p, CCSD_BAKED_CS_PLAN, CCSD_BAKED_CS_SIZE)) return XA_BAKED20;
    if (sem == XA_PLAIN && p.chan_global && plan_is_baked(p, CCSD_BAKED_Z_PLAN, CCSD_BAKED_Z_SIZE)) return XA_BAKED38;
    if (sem == XA_GEN && !p.chan_global && plan_is_baked(p, CCSD_BAKED_ENZ_PLAN, CCSD_BAKED_ENZ_SIZE)) return XA_BAKEDENZ;
    if (sem != XA_PLAIN || p.geo_off == 1) return sem;
    // instances with the dataset geometry compiled in
    if (!p.chan_global && p.N == 9 && p.F == 4 && p.E == 36 && p.ldn == 16) return XA_PLAIN9;
    if (p.chan_global && p.N == 20 && p.E == 190 && p.ldn == 24) return XA_PLAIN20;
    if (p.chan_global && p.N == 38 && p.E == 703 && p.ldn == 40) return XA_PLAIN38;
    return XA_PLAIN;
}
static inline const void* xa_kernel(const PlanD& p) {
    const int v = xa_variant(p);
#define XA_FN(G_) (v == XA_HB ? (const void*)k_xa<G_, XA_HB> : v == XA_GMH ? (const void*)k_xa<G_, XA_GMH> : \
                   v == XA_GEN ? (const void*)k_xa<G_, XA_GEN> : (const void*)k_xa<G_, XA_PLAIN>)
    if (v == XA_PLAIN9) return (const void*)k_xa<false, XA_PLAIN9>;
    if (v == XA_BAKED9) return (const void*)k_xa<false, XA_BAKED9>;
    if (v == XA_BAKEDENZ) return (const void*)k_xa<false, XA_BAKEDENZ>;
    if (v == XA_PLAIN20) return (const void*)k_xa<true, XA_PLAIN20>;
    if (v == XA_BAKED20) return (const void*)k_xa<true, XA_BAKED20>;
    if (v == XA_PLAIN38) return (const void*)k_xa<true, XA_PLAIN38>;
    if (v == XA_BAKED38) return (const void*)k_xa<true, XA_BAKED38>;
    return p.chan_global ? XA_FN(true) : XA_FN(false);
#undef XA_FN
}

// widest layer of ScoreNetworkF's per-element MLPs
static inline int fnet_width(const PlanD& p) {
    int fw = p.f_fdim > p.f_cnum ? p.f_fdim : p.f_cnum;
    for (int l = 0; l < p.f_L; ++l) {
        const MlpD& m = p.fl[l];
        const int wd = m.n > 1 && m.hid > m.in ? (m.hid > m.out ? m.hid : m.out) : (m.in > m.out ? m.in : m.out);
        if (wd > fw) fw = wd;
    }
    if (p.f_fin.n > 1 && p.f_fin.hid > fw) fw = p.f_fin.hid;
    return fw;
}

struct ccsd_plan {
    ccsd_config_t cfg;
    PlanD h;                    // host copy
    PlanD* d = nullptr;         // device copy
    float* w = nullptr;         // device weights
    float* wp = nullptr;        // device: zero-padded copies of the chain MLPs' linears (mlp_chain_tile)
    unsigned char* hpairs = nullptr;   // device: (e, e2), e <= e2, row-major: unordered pairs of the dense hodge layer
    size_t npacked = 0;
    unsigned char* edges = nullptr;
    unsigned long long* cells = nullptr;
    std::vector<ccsd_step_coef_t> coef;  // [diff_steps][3]
    size_t nweights = 0;
    long long* dbg = nullptr;   // diagnostic cycle stamps (ccsd_debug_stamps)
    unsigned long long* init_off = nullptr;   // device: off-bit table of ccsd_init_state (which takes no workspace), grown on demand
    size_t init_off_cap = 0;
    // fused rank-2 kernel (k_r2): eligibility and LDS geometry
    int fused_r2 = 0, r2_ldk = 0, r2_ldh = 0;
    size_t r2_lds = 0;
    // element-wise rank-2 side (k_ew1): affine ScoreNetworkF without a Hodge Laplacian term (cnum = 1), tiled path, PC samplers
    int ew1 = 0;
    // general hodge stack: more than two HodgeAdjAttentionLayers whose later projections cannot be folded into rank2's (a non-affine
    // mlp_value, or no fused rank-2 kernel for the geometry): R_l is materialised layer by layer (k_hodge_value) from the dense hodge
    // adjacencies k_xa<., XA_GEN> dumps (launch_xa); CCSD_HODGE_GENERAL forces it for any plan with more than two layers (diagnostic)
    int h_general = 0;
    // diagnostic knobs, read from the environment ONCE at plan creation (never on the launch path):
    // CCSD_OLD_GEMM_P, CCSD_XA_THREADS, CCSD_NO_FUSED_APPLY (CCSD_NO_FUSED_R2 / CCSD_XA_PASS / CCSD_XA_GCH / CCSD_NO_CHAIN shape the plan itself)
    int opt_old_gemm_p = 0, opt_xa_threads = 0, opt_no_fused_apply = 0;   // opt_xa_threads: 0 = by batch (launch_xa)
    int opt_r2_stagger_mask = 0, opt_r2_stagger_sleep = 0;     // CCSD_R2_STAGGER="mask,sleep" (diagnostic)
    int opt_xa_prio = 0;                                       // CCSD_XA_PRIO (diagnostic: k_xa issue-priority scheme)
    int opt_xa_stagger_mask = 0, opt_xa_stagger_sleep = 0;     // CCSD_XA_STAGGER="mask,sleep" (diagnostic)
    int opt_no_merge = 0;                                      // CCSD_NO_MERGE (diagnostic: separate norms / predictor k_r2 launches)
    int opt_hp_full_norms = 0;                                 // CCSD_HP_FULL_NORMS (diagnostic: k_hp_full in the norms pass too)
    int opt_no_hp_full = 0;                                    // CCSD_NO_HP_FULL (diagnostic: k_gemm_p0<., ., 1 / 2> + k_gemm_h_full instead of the one fused pass)
    // EXPERIMENT, never the default (CCSD_SPLIT_BF16=3): split-precision ("bf16 x 3") MFMA contraction in k_gemm_h_full, the norms pass's
    // H = F F^T of the community_small geometry (split_frag / split_mma, ccsd_k_rank2.h); results are NOT bit-identical to fp32
    int opt_split_bf16 = 0;
    int opt_no_h_full = 0;                                     // CCSD_NO_H_FULL (diagnostic: k_gemm_h's 64 x 64 tiles for the community_small geometry too)
    int opt_no_tiled_fuse = 0;                                 // CCSD_NO_TILED_FUSE (diagnostic: k_noise_norm / k_langevin_apply as launches of their own on the tiled path)
    int opt_r2_masked = 1;                                     // CCSD_NO_R2_MASKED clears it (diagnostic: the loop's k_r2 launches re-mask rank2 in the Q_1 loader)
    // optional per-kernel timing with HIP events on the launch stream (bench.py roofline leg)
    unsigned prof_mask = 0;
    size_t prof_used[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    // every prof_stride-th launch of a selected kernel is bracketed by events (event records break back-to-back dispatch:
    // bracketing every launch costs ~6 % of the step)
    int prof_stride = 1;
    size_t prof_calls[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#ifndef CCSD_EMU
    std::vector<hipEvent_t> prof_ev[8];
#endif
};

enum { KID_XA = 0, KID_GEMM_P = 1, KID_HF = 2, KID_GEMM_H = 3, KID_LANGEVIN = 4, KID_R2 = 5, KID_S4 = 6, KID_EW1 = 7 };
static void prof_mark(ccsd_plan* pl, int kid, void* stream) {
#ifndef CCSD_EMU
    if (!(pl->prof_mask & (1u << kid))) return;
    const size_t call = pl->prof_calls[kid]++;                 // two calls per launch: before and after
    if (pl->prof_stride > 1 && (call >> 1) % (size_t)pl->prof_stride != 0) return;
    std::vector<hipEvent_t>& ev = pl->prof_ev[kid];
    if (pl->prof_used[kid] == ev.size()) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return;
        ev.push_back(e);
    }
    (void)hipEventRecord(ev[pl->prof_used[kid]++], (hipStream_t)stream);
#else
    (void)pl; (void)kid; (void)stream;
#endif
}
#define RT_CHECK(expr)                                                                    \
    do {                                                                                  \
        rtError_t _e = (expr);                                                            \
        if (_e != RT_OK) return set_err(CCSD_ERR_RUNTIME, std::string(#expr) + ": " + rt_error_string(_e)); \
    } while (0)
#define LAUNCH_CHECK()                                                                    \
    do {                                                                                  \
        rtError_t _e = rt_last_error();                                                   \
        if (_e != RT_OK) return set_err(CCSD_ERR_RUNTIME, std::string("kernel launch: ") + rt_error_string(_e)); \
    } while (0)

// Instantiation of the fused rank-2 kernel for a plan: (MT, RS) from E -- MT = ceil(E / 16) row tiles, RS = plain MFMA steps
// covering E mod 16 behind the full 16-wide blocks of phase 2's contraction index (0: last block taken whole) --, AFFINE
// ScoreNetworkF, general hodge mlp_value.  RS only shapes the affine phase 2; the non-affine kernels are instantiated with RS = 0.
static inline void r2_shape(const ccsd_plan* pl, int* MT, int* RS, bool* aff, bool* gen1) {
    const int E = pl->h.E, rem = E & 15;
    *MT = (E + 15) / 16;
    *aff = pl->h.f_affine != 0;
    *gen1 = pl->h.h_L > 1 && pl->h.hl[0].mval.n > 1;
    *RS = (!*aff || rem == 0 || rem > 12) ? 0 : (rem + 3) / 4;
}
// X(MT, RS, AFFINE, GEN1) is expanded for the plan's combination
#define R2_CASE(MT_, RS_, X) \
    if (mt_ == MT_ && rs_ == RS_) { \
        if (aff_ && !gen1_) { X(MT_, RS_, true, false); } else if (aff_) { X(MT_, RS_, true, true); } \
        else if (!gen1_) { X(MT_, 0, false, false); } else { X(MT_, 0, false, true); } \
    }
// non-affine ScoreNetworkF: RS is always 0 (r2_shape), every MT has its instance (CCSD_R2_GEN)
#define R2_CASE_GEN(MT_, X) \
    if (!aff_ && mt_ == MT_) { if (!gen1_) { X(MT_, 0, false, false); } else { X(MT_, 0, false, true); } }
#define R2_DISPATCH(pl_, X) \
    do { \
        int mt_, rs_; bool aff_, gen1_; \
        r2_shape(pl_, &mt_, &rs_, &aff_, &gen1_); \
        /* E = N (N - 1) / 2 <= 64, i.e. E in {1, 3, 6, 10, 15, 21, 28, 36, 45, 55}: the (MT, RS) pairs that occur */ \
        R2_CASE_GEN(1, X) else R2_CASE_GEN(2, X) else R2_CASE_GEN(3, X) else R2_CASE_GEN(4, X) \
        else R2_CASE(1, 0, X) else R2_CASE(1, 1, X) else R2_CASE(1, 2, X) else R2_CASE(1, 3, X) \
        else R2_CASE(2, 2, X) else R2_CASE(2, 3, X) else R2_CASE(3, 0, X) else R2_CASE(3, 1, X) else R2_CASE(4, 2, X) \
    } while (0)
// the instances with the qm9 geometry compiled in: 1 = k_r2<3, 1, true, false, 1> (geometry only), 2 = <..., 2> (the whole baked plan:
// only when the plan's architecture bytes equal ccsd_baked_qm9.h), 0 = the run-time-geometry instances
static inline int r2_qm9(const ccsd_plan* pl) {
    const PlanD& p = pl->h;
    const bool gen1 = p.h_L > 1 && p.hl[0].mval.n > 1;
    if (p.geo_off == 1 || gen1 || p.E != 36 || p.K != 466 || p.N != 9 || pl->r2_ldk != 488 || pl->r2_ldh != 36) return 0;
    if (!p.f_affine) return 3;                           // k_r2<3, 0, false, false, 1>: the non-affine network on the qm9 geometry
    return plan_is_baked(p, CCSD_BAKED_QM9_PLAN, CCSD_BAKED_QM9_SIZE) ? 2 : 1;
}
static inline const void* r2_kernel(const ccsd_plan* pl) {
    if (r2_qm9(pl) == 2) return (const void*)k_r2<3, 1, true, false, 2>;
    if (r2_qm9(pl) == 1) return (const void*)k_r2<3, 1, true, false, 1>;
    if (r2_qm9(pl) == 3) return (const void*)k_r2<3, 0, false, false, 1>;
    const void* fn = nullptr;
#define R2_PTR(MT_, RS_, A_, G_) fn = (const void*)k_r2<MT_, RS_, A_, G_>
    R2_DISPATCH(pl, R2_PTR);
#undef R2_PTR
    return fn;
}

extern "C" const char* ccsd_last_error(void) { return g_last_error.c_str(); }

extern "C" void ccsd_rank2_dims(const ccsd_config_t* cfg, int32_t* E, int64_t* K) {
    int e = 0; int64_t k = 0;
    if (cfg) ccsd_dims(cfg, &e, &k);
    if (E) *E = e;
    if (K) *K = k;
}

extern "C" size_t ccsd_weight_count(const ccsd_config_t* cfg) {
    PlanD p; PlanBuilder pb;
    size_t n = ccsd_build_plan(cfg, &p, pb);
    if (pb.status != CCSD_OK) { set_err(pb.status, pb.err); return 0; }
    return n;
}

extern "C" int ccsd_debug_stamps(ccsd_plan_t* plan, void* dev_buffer) {
    if (!plan) return set_err(CCSD_ERR_INVALID, "NULL plan");
    plan->dbg = (long long*)dev_buffer;
    return CCSD_OK;
}
extern "C" int ccsd_profile_kernel(ccsd_plan_t* plan, int32_t kernel_id) {
    if (!plan) return set_err(CCSD_ERR_INVALID, "NULL plan");
    if (kernel_id < 0) plan->prof_mask = 0;
    else if (kernel_id < 8) plan->prof_mask |= 1u << kernel_id;
    for (int k = 0; k < 8; ++k) { plan->prof_used[k] = 0; plan->prof_calls[k] = 0; }
    return CCSD_OK;
}
extern "C" int ccsd_profile_stride(ccsd_plan_t* plan, int32_t stride) {
    if (!plan || stride < 1) return set_err(CCSD_ERR_INVALID, "bad argument");
    plan->prof_stride = stride;
    for (int k = 0; k < 8; ++k) plan->prof_calls[k] = 0;
    return CCSD_OK;
}
extern "C" int ccsd_profile_launches(ccsd_plan_t* plan, int32_t kernel_id, int64_t* launches) {
    if (!plan || !launches || kernel_id < 0 || kernel_id >= 8) return set_err(CCSD_ERR_INVALID, "bad argument");
    *launches = (int64_t)(plan->prof_calls[kernel_id] >> 1);       // two marks per launch
    return CCSD_OK;
}
extern "C" int ccsd_profile_read(ccsd_plan_t* plan, int32_t kernel_id, int64_t* launches, double* total_ms) {
    if (!plan || !launches || !total_ms || kernel_id < 0 || kernel_id >= 8) return set_err(CCSD_ERR_INVALID, "bad argument");
    *launches = 0; *total_ms = 0.0;
#ifndef CCSD_EMU
    for (size_t i = 0; i + 1 < plan->prof_used[kernel_id]; i += 2) {
        float ms = 0.f;
        RT_CHECK(hipEventSynchronize(plan->prof_ev[kernel_id][i + 1]));
        RT_CHECK(hipEventElapsedTime(&ms, plan->prof_ev[kernel_id][i], plan->prof_ev[kernel_id][i + 1]));
        *total_ms += ms; *launches += 1;
    }
#endif
    plan->prof_used[kernel_id] = 0;
    return CCSD_OK;
}

extern "C" void ccsd_plan_destroy(ccsd_plan_t* plan) {
    if (!plan) return;
#ifndef CCSD_EMU
    for (int k = 0; k < 8; ++k) for (hipEvent_t e : plan->prof_ev[k]) (void)hipEventDestroy(e);
#endif
    if (plan->d) (void)rt_free(plan->d);
    if (plan->w) (void)rt_free(plan->w);
    if (plan->wp) (void)rt_free(plan->wp);
    if (plan->hpairs) (void)rt_free(plan->hpairs);
    if (plan->edges) (void)rt_free(plan->edges);
    if (plan->cells) (void)rt_free(plan->cells);
    if (plan->init_off) (void)rt_free(plan->init_off);
    delete plan;
}

extern "C" int ccsd_plan_create(const ccsd_config_t* cfg, const float* weights, size_t n_weights,
                                const ccsd_step_coef_t* step_coef, ccsd_plan_t** out) {
    if (!cfg || !weights || !step_coef || !out) return set_err(CCSD_ERR_INVALID, "NULL argument");
    *out = nullptr;
    ccsd_plan* pl = new ccsd_plan();
    pl->cfg = *cfg;
    pl->opt_old_gemm_p = getenv("CCSD_OLD_GEMM_P") != nullptr;
    pl->opt_no_fused_apply = getenv("CCSD_NO_FUSED_APPLY") != nullptr;
    pl->opt_no_merge = getenv("CCSD_NO_MERGE") != nullptr;
    pl->opt_r2_masked = getenv("CCSD_NO_R2_MASKED") == nullptr;
    pl->opt_no_tiled_fuse = getenv("CCSD_NO_TILED_FUSE") != nullptr;
    pl->opt_no_h_full = getenv("CCSD_NO_H_FULL") != nullptr;
    if (const char* sp = getenv("CCSD_SPLIT_BF16")) pl->opt_split_bf16 = atoi(sp) == 3 ? 3 : 0;
    pl->opt_no_hp_full = getenv("CCSD_NO_HP_FULL") != nullptr;
    pl->opt_hp_full_norms = getenv("CCSD_HP_FULL_NORMS") != nullptr;
    if (const char* pr = getenv("CCSD_XA_PRIO")) pl->opt_xa_prio = atoi(pr);
    if (const char* sg = getenv("CCSD_XA_STAGGER")) sscanf(sg, "%d,%d", &pl->opt_xa_stagger_mask, &pl->opt_xa_stagger_sleep);
    if (const char* sg = getenv("CCSD_R2_STAGGER")) sscanf(sg, "%d,%d", &pl->opt_r2_stagger_mask, &pl->opt_r2_stagger_sleep);
    if (const char* xt = getenv("CCSD_XA_THREADS")) { const int v = atoi(xt); if (v >= 64 && v <= 1024 && v % 64 == 0) pl->opt_xa_threads = v; }
    PlanBuilder pb;
    pl->nweights = ccsd_build_plan(cfg, &pl->h, pb);
    if (pb.status != CCSD_OK) { delete pl; return set_err(pb.status, pb.err); }
    pl->h.geo_off = getenv("CCSD_NO_GEO") ? 1 : getenv("CCSD_NO_BAKE") ? 2 : 0;      // 2: geometry instances yes, baked-plan instances no
    pl->npacked = (size_t)pb.pcur;
    if (pl->nweights != n_weights) {
        delete pl;
        return set_err(CCSD_ERR_WEIGHTS, "weight blob has " + std::to_string(n_weights) + " floats, config needs " +
                                             std::to_string(pl->nweights));
    }
    if (cfg->predictor != CCSD_PRED_EULER && cfg->predictor != CCSD_PRED_REVERSE && cfg->predictor != CCSD_PRED_S4) { delete pl; return set_err(CCSD_ERR_UNSUPPORTED, "unknown predictor"); }
    if (cfg->corrector != CCSD_CORR_NONE && cfg->corrector != CCSD_CORR_LANGEVIN) { delete pl; return set_err(CCSD_ERR_UNSUPPORTED, "unknown corrector"); }
    if (cfg->diff_steps < 1 || cfg->n_corr_steps < 0) { delete pl; return set_err(CCSD_ERR_INVALID, "bad step counts"); }
    ccsd_fold_fnet(&pl->h, weights);
    pl->coef.assign(step_coef, step_coef + (size_t)cfg->diff_steps * 3);
    // enumeration tables (get_cells, cc_utils.py:72-94): edges = combinations(range(N),2) row-major;
    // cells = combinations(range(N),k) for k = d_min..d_max, lexicographic, as node bitmasks
    const int N = cfg->N, E = pl->h.E, K = pl->h.K;
    std::vector<unsigned char> edges((size_t)2 * E + 2);
    {
        int e = 0;
        for (int i = 0; i < N; ++i)
            for (int j = i + 1; j < N; ++j) { edges[2 * e] = (unsigned char)i; edges[2 * e + 1] = (unsigned char)j; ++e; }
    }
    std::vector<unsigned long long> cells((size_t)K + 1);
    if (cfg->is_cc) {
        size_t c = 0;
        std::vector<int> idx;
        for (int k = cfg->d_min; k <= cfg->d_max; ++k) {
            idx.resize(k);
            for (int i = 0; i < k; ++i) idx[i] = i;
            while (true) {
                unsigned long long m = 0;
                for (int i = 0; i < k; ++i) m |= 1ull << idx[i];
                cells[c++] = m;
                int i = k - 1;
                while (i >= 0 && idx[i] == N - k + i) --i;
                if (i < 0) break;
                ++idx[i];
                for (int j = i + 1; j < k; ++j) idx[j] = idx[j - 1] + 1;
            }
        }
        if ((int)c != K) { delete pl; return set_err(CCSD_ERR_INVALID, "cell enumeration mismatch"); }
    }
#define PC(expr) do { rtError_t _e = (expr); if (_e != RT_OK) { ccsd_plan_destroy(pl); return set_err(CCSD_ERR_RUNTIME, std::string(#expr) + ": " + rt_error_string(_e)); } } while (0)
    PC(rt_malloc((void**)&pl->d, sizeof(PlanD)));
    PC(rt_h2d(pl->d, &pl->h, sizeof(PlanD)));
    {
        const size_t wtotal = pl->h.f_blk >= 0 ? (size_t)pl->h.f_blk + CCSD_FBLK_FLOATS : n_weights;
        PC(rt_malloc((void**)&pl->w, wtotal * sizeof(float)));
        PC(rt_h2d(pl->w, weights, n_weights * sizeof(float)));
        if (pl->h.f_blk >= 0) {
            std::vector<float> blk(CCSD_FBLK_FLOATS);
            ccsd_pack_fnet_blocks(&pl->h, weights, blk.data());
            PC(rt_h2d(pl->w + pl->h.f_blk, blk.data(), blk.size() * sizeof(float)));
        }
    }
    {
        std::vector<float> packed(pl->npacked + 4, 0.f);
        ccsd_pack_mlp(pl->h.x_fin, weights, packed.data());
        for (int l = 0; l < pl->h.a_L; ++l) {
            ccsd_pack_mlp(pl->h.al[l].mlp, weights, packed.data()); ccsd_pack_mlp(pl->h.al[l].mc, weights, packed.data());
            ccsd_pack_qkv(pl->h.al[l], weights, packed.data()); ccsd_pack_mc(pl->h.al[l], weights, packed.data());
        }
        if (pl->h.x_gmh) for (int l = 0; l < pl->h.x_depth; ++l) {
            ccsd_pack_mlp(pl->h.gl[l].mlp, weights, packed.data()); ccsd_pack_mlp(pl->h.gl[l].mc, weights, packed.data());
            ccsd_pack_qkv(pl->h.gl[l], weights, packed.data()); ccsd_pack_mc(pl->h.gl[l], weights, packed.data());
        }
        if (pl->h.hb_L) ccsd_pack_mlp(pl->h.hb[0].mh, weights, packed.data());
        for (int l = 0; l < pl->h.hb_L; ++l) {   // transposed copies of the BaselineBlocks' weights
            const HodgeBaseD& h = pl->h.hb[l];
            for (int c = 0; c < h.cin; ++c) {
                const float* blk = weights + h.blk_base + (size_t)c * h.blk_stride;    // W1[hid][E] b1[hid] W2[E][hid] b2[E]
                const float* w2 = blk + h.hid * E + h.hid;
                for (int e = 0; e < E; ++e)
                    for (int hh = 0; hh < h.hid; ++hh) {
                        packed[(size_t)h.w2t + ((size_t)c * h.hid + hh) * E + e] = w2[e * h.hid + hh];
                        packed[(size_t)h.w1t + ((size_t)c * E + e) * h.hid + hh] = blk[hh * E + e];
                    }
            }
        }
        ccsd_pack_mlp(pl->h.a_fin, weights, packed.data());
        for (int l = 0; l < pl->h.h_L; ++l) {   // Wcat^T of the hodge projections for k_r2
            const HodgeLayerD& h = ccsd_hl(pl->h, l);
            const int Kp = (K + 31) & ~31;
            for (int k = 0; k < K; ++k)
                for (int n = 0; n < h.wc; ++n) packed[(size_t)h.wcatT + (size_t)n * Kp + k] = weights[(size_t)h.wcat + (size_t)k * h.wc + n];
        }
        PC(rt_malloc((void**)&pl->wp, packed.size() * sizeof(float)));
        PC(rt_h2d(pl->wp, packed.data(), packed.size() * sizeof(float)));
    }
    if (pl->h.h_L > 1) {
        if (E > 255) { ccsd_plan_destroy(pl); return set_err(CCSD_ERR_UNSUPPORTED, "dense hodge layer needs E <= 255"); }
        std::vector<unsigned char> hp;
        for (int e = 0; e < E; ++e)
            for (int e2 = e; e2 < E; ++e2) { hp.push_back((unsigned char)e); hp.push_back((unsigned char)e2); }
        PC(rt_malloc((void**)&pl->hpairs, hp.size()));
        PC(rt_h2d(pl->hpairs, hp.data(), hp.size()));
    }
    PC(rt_malloc((void**)&pl->edges, edges.size()));
    PC(rt_h2d(pl->edges, edges.data(), edges.size()));
    PC(rt_malloc((void**)&pl->cells, cells.size() * sizeof(unsigned long long)));
    PC(rt_h2d(pl->cells, cells.data(), cells.size() * sizeof(unsigned long long)));
    // fused rank-2 path: one complex's rank2 block (E x K) LDS-resident, E <= 64
    if (cfg->is_cc && E <= 64 && getenv("CCSD_NO_FUSED_R2") == nullptr) {
        const PlanD& p = pl->h;
        const int Kp4 = (K + 31) & ~31, Ep4 = (E + 3) & ~3;   // K zero-padded to whole 8-step MFMA batches
        int ldk = Kp4; while ((ldk & 31) != 8 && (ldk & 31) != 24) ldk += 4;   // conflict-free ds_read_b128 fragment reads (16 rows x 4 k-quads)
        int ldh = Ep4;                                                     // 16-byte aligned rows: phase 2 re-reads H's fragments per column tile as ds_read_b128
        const size_t fl = (size_t)E * ldk + (size_t)E * ldh + 64 * 2 + (size_t)p.a_cinit * E + 3 * N * N + 64 + (Kp4 + 3) / 4 + 4;
        const bool wc_ok = fnet_width(p) <= CCSD_FW;        // the fused kernel's per-element MLPs are padded to <= 16
        if (fl * 4 + 64 <= 160 * 1024 && wc_ok && r2_kernel(pl) != nullptr && p.f_cnum <= 2) {   // more Hodge powers: tiled kernels
            pl->fused_r2 = 1; pl->r2_ldk = ldk; pl->r2_ldh = ldh; pl->r2_lds = fl * 4;
        }
    }
    if (pl->h.h_L > 2) {
        bool affine_values = true;
        for (int l = 0; l + 1 < pl->h.h_L; ++l) affine_values = affine_values && ccsd_hl(pl->h, l).mval.n == 1;
        // (the folded route -- k_r2 hands over one consolidated projection, k_xa chains the M_j -- is built and tested for up to four layers)
        if (!affine_values || !pl->fused_r2 || pl->h.h_L > 4 || getenv("CCSD_HODGE_GENERAL") != nullptr) { pl->h_general = 1; pl->fused_r2 = 0; }
    }
    pl->ew1 = cfg->is_cc && !pl->fused_r2 && pl->h.f_affine && pl->h.f_cnum == 1 && cfg->predictor != CCSD_PRED_S4 &&
              getenv("CCSD_NO_EW1") == nullptr;
#ifndef CCSD_EMU
    if ((size_t)pl->h.xa_lds_floats * 4 > 64 * 1024) {
        PC(rt_set_max_dyn_smem(xa_kernel(pl->h), (size_t)pl->h.xa_lds_floats * 4));
        if (xa_variant(pl->h) == XA_PLAIN9 || xa_variant(pl->h) == XA_BAKED9)     // (its run-time-geometry twin serves launches with a diagnostic thread count)
            PC(rt_set_max_dyn_smem((const void*)k_xa<false, XA_PLAIN>, (size_t)pl->h.xa_lds_floats * 4));
    }
    if (pl->fused_r2 && pl->r2_lds > 64 * 1024) {
        PC(rt_set_max_dyn_smem(r2_kernel(pl), pl->r2_lds));
    }
    if (pl->h.is_cc && pl->h.E == 190 && pl->h.K == 1140) {      // k_hp_full: 66.6 KB of dynamic LDS
        const size_t lds = (size_t)(2 * 192 * H_LD + 2 * 16 * H_LD) * 4;
        PC(rt_set_max_dyn_smem((const void*)k_hp_full<190, 1140, 1>, lds));
        PC(rt_set_max_dyn_smem((const void*)k_hp_full<190, 1140, 2>, lds));
    }
#endif
#undef PC
    if (const char* path = getenv("CCSD_DUMP_PLAN")) {     // tools/bake_plan.py: the plan's architecture bytes as a C header
        const char* nm = getenv("CCSD_DUMP_PLAN_NAME");      // QM9 (default), CS
        if (!nm) nm = "QM9";
        std::vector<unsigned char> bytes(sizeof(PlanD));
        ccsd_plan_arch_bytes(pl->h, bytes.data());
        if (FILE* f = fopen(path, "w")) {
            fprintf(f, "// GENERATED by tools/bake_plan.py (do not edit): the PlanD of a shipped configuration at its bench batch as a compile-time\n"
                       "// constant (architecture bytes: ccsd_plan_arch_bytes, the weight-derived affine fold zeroed).  The baked kernel instances read\n"
                       "// their plan from it instead of from memory; the host selects them only for plans whose architecture bytes are equal.\n"
                       "#pragma once\n#define CCSD_BAKED_%s_SIZE %zu\n#define CCSD_BAKED_%s_A_L %d      /* AttentionLayers of ScoreNetworkA: unroll count */\n"
                       "alignas(16) static constexpr unsigned char CCSD_BAKED_%s_PLAN[CCSD_BAKED_%s_SIZE] = {", nm, sizeof(PlanD), nm, pl->h.a_L, nm, nm);
            for (size_t i = 0; i < bytes.size(); ++i) fprintf(f, "%s%u,", (i % 40) ? "" : "\n    ", (unsigned)bytes[i]);
            fprintf(f, "\n};\n");
            fclose(f);
        }
    }
    *out = pl;
    return CCSD_OK;
}

// (E, K) of the shipped geometries the general-path kernels have instances for: X(EC, KC) is expanded with the plan's pair as
// compile-time constants when it is one of them (loop bounds, row strides and divisions fold), with (0, 0) = run-time values otherwise
#define GEO_EK(p_, X) \
    do { \
        if ((p_).geo_off == 1) { X(0, 0); } \
        else if ((p_).E == 190 && (p_).K == 1140) { X(190, 1140); }    /* community_small (d_min 2 .. d_max) */ \
        else if ((p_).E == 703 && (p_).K == 8436) { X(703, 8436); }    /* N = 38 (zinc250k), the 5b substitute's cells */ \
        else if ((p_).E == 66 && (p_).K == 715) { X(66, 715); }        /* ENZYMES_small */ \
        else { X(0, 0); } \
    } while (0)

#define CCSD_P_SPLITS 8     /* most K slices k_gemm_p is split into when its row tiles cannot fill the chip */

// ---------------- workspace ----------------
struct Workspace {
    unsigned long long* offbits;
    unsigned char *mfr, *mfl;       // flag masks of every complex as byte tables (k_masktab): [B][Kp], [B][Ep]
    int Kp, Ep;
    float *H, *P0, *P1, *U1, *acoef, *net_x, *net_adj, *net_r, *norm2, *part, *sums, *chan, *zpart, *part2;
    float* psplit; size_t psplit_floats;   // K slices of the layer-1 projection (k_gemm_p with few row tiles)
    float *P0b, *P1b, *U1b;         // second set of hodge projections (merged k_r2 launch: the next norms pass's)
    int ntiles, nchunk;
    float *hgH, *hgR[2], *hgP[CCSD_MAXHL + CCSD_MAXHLX - 1];   // general hodge stack: dumped H^l, R_l (two alternating), P_l of the layers >= 1
    size_t hg_hstride;
    const float* hg_rank2;          // (the rank2 launch_p saw: launch_xa continues from it)
    int h_done;                     // H of this pass is already in w.H (k_hp_full produced it beside P_0): launch_h returns at once
    int p1_raw;     // who filled P1 last: k_r2 with the raw factors (1, see k_r2) or k_gemm_p with the finished projections (0)
    size_t bytes;
};
static Workspace carve_ws(const ccsd_plan* pl, int B, void* base) {
    const PlanD& p = pl->h;
    Workspace w{};
    size_t o = 0;
    auto take = [&](size_t nbytes) { size_t r = o; o += (nbytes + 255) / 256 * 256; return base ? (char*)base + r : (char*)nullptr; };
    w.offbits = (unsigned long long*)take((size_t)B * 8);
    const size_t E = p.E, K = p.K;
    w.Kp = p.is_cc ? (p.K + 3) & ~3 : 0; w.Ep = p.is_cc ? (p.E + 3) & ~3 : 0;
    w.mfr = (unsigned char*)take((size_t)B * w.Kp);
    w.mfl = (unsigned char*)take((size_t)B * w.Ep);
    w.H = (float*)take(p.is_cc ? (size_t)B * E * h_ld((int)E) * 4 * (p.f_cnum > 2 ? p.f_cnum - 1 : 1) : 0);   // H, H^2, ... (cnum > 2: one slab per power)
    w.P0 = (float*)take(p.h_L > 0 ? (size_t)B * E * p.hl[0].wc * 4 : 0);
    w.P1 = (float*)take(p.h_L > 1 ? (size_t)B * E * p.h_pw * 4 : 0);
    w.U1 = (float*)take(p.h_L > 1 ? (size_t)B * p.h_pw * 4 : 0);
    // K slices of k_gemm_p (always split: batch-invariant summation order; the step-wise score / norms calls of fused-rank-2 plans use k_gemm_p too)
    w.psplit_floats = p.h_L > 1 ? (size_t)CCSD_P_SPLITS * B * E * p.h_pw : 0;
    w.psplit = (float*)take(w.psplit_floats * 4);
    const bool two = pl->fused_r2 != 0;
    w.P0b = (float*)take(two && p.h_L > 0 ? (size_t)B * E * p.hl[0].wc * 4 : 0);
    w.P1b = (float*)take(two && p.h_L > 1 ? (size_t)B * E * p.h_pw * 4 : 0);
    w.U1b = (float*)take(two && p.h_L > 1 ? (size_t)B * p.h_pw * 4 : 0);
    w.acoef = (float*)take(p.h_L > 1 ? (size_t)B * p.a_cinit * E * 4 : 0);
    if (pl->h_general) {
        int cmax = 1;
        for (int l = 0; l + 1 < p.h_L; ++l) cmax = ccsd_hl(p, l).cout > cmax ? ccsd_hl(p, l).cout : cmax;
        w.hg_hstride = (size_t)cmax * E * E;
        w.hgH = (float*)take((size_t)B * w.hg_hstride * 4);
        w.hgR[0] = (float*)take((size_t)B * E * K * 4);
        w.hgR[1] = (float*)take((size_t)B * E * K * 4);
        for (int l = 1; l < p.h_L; ++l) w.hgP[l - 1] = (float*)take((size_t)B * E * ccsd_hl(p, l).wc * 4);
    }
    w.net_x = (float*)take((size_t)B * p.N * p.F * 4);
    w.net_adj = (float*)take((size_t)B * p.N * p.N * 4);
    w.net_r = (float*)take(p.is_cc ? (size_t)B * E * K * 4 : 0);
    w.norm2 = (float*)take((size_t)B * 4 * 4);
    w.ntiles = p.is_cc ? ((p.K + T_BN - 1) / T_BN) * ((p.E + T_BM - 1) / T_BM) : 0;
    w.nchunk = p.is_cc ? (int)((((size_t)p.E * p.K + 3) / 4 + CCSD_NN_CH - 1) / CCSD_NN_CH) : 0;   // k_noise_norm / k_ew1: chunks of flat groups per sample
    { int np = w.ntiles > w.nchunk ? w.ntiles : w.nchunk; if ((int)E > np) np = (int)E;      // (per-row partials of P0Fuse mode 3)
      w.part = (float*)take((size_t)B * (np ? np : 1) * 2 * 4); }
    w.zpart = (float*)take((size_t)B * ((size_t)w.nchunk > E ? (size_t)w.nchunk : E ? E : 1) * 4);   // chunk partials of k_noise_norm, or the row partials of P0Fuse mode 1
    w.part2 = (float*)take((size_t)B * 2 * 4);
    w.sums = (float*)take(64);
    w.chan = (float*)take(p.chan_global ? (size_t)B * p.chan_rows * p.N * p.N * 4 : 0);
    w.bytes = o;
    return w;
}
extern "C" size_t ccsd_workspace_bytes(const ccsd_plan_t* plan, int32_t B) {
    if (!plan || B < 1) return 0;
    return carve_ws(plan, B, nullptr).bytes;
}

static int grid_for(long long n, int block) {
    long long g = (n + block - 1) / block;
    if (g > 2048) g = 2048;   // grid-stride the rest
    if (g < 1) g = 1;
    return (int)g;
}

static int check_common(const ccsd_plan* pl, int B, const void* flags, const void* ws, size_t ws_bytes) {
    if (!pl) return set_err(CCSD_ERR_INVALID, "NULL plan");
    if (B < 1) return set_err(CCSD_ERR_INVALID, "B must be >= 1");
    if (!flags) return set_err(CCSD_ERR_INVALID, "NULL flags");
    if (!ws || ws_bytes < carve_ws(pl, B, nullptr).bytes) return set_err(CCSD_ERR_WORKSPACE, "workspace too small");
    return CCSD_OK;
}
static int check_state(const ccsd_plan* pl, const ccsd_state_t* s, const char* what) {
    if (!s || !s->x || !s->adj || (pl->h.is_cc && !s->rank2)) return set_err(CCSD_ERR_INVALID, std::string("NULL tensor in ") + what);
    return CCSD_OK;
}

static int launch_flagbits(const ccsd_plan* pl, int B, const float* flags, Workspace& w, void* stream) {
    CCSD_LAUNCH(k_flagbits, dim3(grid_for(B, 256)), dim3(CCSD_NTHREADS), 0, stream, flags, w.offbits, B, pl->h.N);
    LAUNCH_CHECK();
    if (pl->h.is_cc) {      // (consumers: k_ew1, k_langevin_apply, k_noise_norm; the fused rank-2 kernel builds its own masks in LDS)
        CCSD_LAUNCH(k_masktab, dim3(grid_for((long long)B * (w.Kp + w.Ep), 256)), dim3(CCSD_NTHREADS), 0, stream,
                    (const unsigned long long*)w.offbits, (const unsigned char*)pl->edges, (const unsigned long long*)pl->cells, w.mfr, w.mfl,
                    B, pl->h.E, pl->h.K, w.Kp, w.Ep);
        LAUNCH_CHECK();
    }
    return CCSD_OK;
}

static inline bool tiled_fuse_ok(const ccsd_plan* pl);
// one fused pass per half-step (k_hp_full) instead of k_gemm_p0<., ., 1 / 2> + k_gemm_h_full
static inline bool hp_full_ok(const ccsd_plan* pl, int B) {
    const PlanD& p = pl->h;
    return tiled_fuse_ok(pl) && p.E == 190 && p.K == 1140 && p.geo_off != 1 && B >= 256 && p.hl[0].wc <= 16 && p.f_cnum == 2 &&
           !pl->opt_no_h_full && !pl->opt_no_hp_full;
}
// H = F F^T (ScoreNetworkF) from `rank2`
static int launch_h(const ccsd_plan* pl, int B, const float* rank2, Workspace& w, void* stream) {
    const PlanD& p = pl->h;
    if (!p.is_cc || p.f_cnum < 2) return CCSD_OK;
    if (w.h_done) { w.h_done = 0; return CCSD_OK; }
    const int nth_ = (p.E + T_BM - 1) / T_BM;
    dim3 g(xcd_grid(B, nth_ * (nth_ + 1) / 2));
    prof_mark(const_cast<ccsd_plan*>(pl), KID_GEMM_H, stream);
#ifndef CCSD_EMU
    // community_small geometry, at least one complex per CU: one workgroup per complex, F streamed once (k_gemm_h_full; bit-identical)
    if (p.E == 190 && p.K == 1140 && p.geo_off != 1 && B >= 256 && !pl->opt_no_h_full) {
        if (pl->opt_split_bf16 == 3) hipLaunchKernelGGL((k_gemm_h_full<190, 1140, 2>), dim3(B), dim3(256), 0, (hipStream_t)stream, rank2, w.H, p.f_hmask);
        else hipLaunchKernelGGL((k_gemm_h_full<190, 1140>), dim3(B), dim3(256), 0, (hipStream_t)stream, rank2, w.H, p.f_hmask);
    } else
#endif
    {
#define H_GO(EC_, KC_) CCSD_LAUNCH((k_gemm_h<EC_, KC_>), g, dim3(CCSD_NTHREADS), 0, stream, rank2, w.H, p.E, p.K, p.f_hmask, B)
    GEO_EK(p, H_GO);
#undef H_GO
    }
    prof_mark(const_cast<ccsd_plan*>(pl), KID_GEMM_H, stream);
    LAUNCH_CHECK();
    for (int j = 2; j < p.f_cnum; ++j) {       // H^j = H^(j-1) . H  (pow_tensor_cc, cc_utils.py:972-977)
        const size_t slab = (size_t)B * p.E * h_ld(p.E);
        const int nt = (p.E + 15) / 16, per = CCSD_NTHREADS >= 64 ? CCSD_NTHREADS / 64 : 1;
        CCSD_LAUNCH(k_gemm_pow, dim3((nt * nt + per - 1) / per, 1, B), dim3(CCSD_NTHREADS), 0, stream,
                    (const float*)(w.H + (size_t)(j - 2) * slab), (const float*)w.H, w.H + (size_t)(j - 1) * slab, p.E);
        LAUNCH_CHECK();
    }
    return CCSD_OK;
}
struct RankEpi;
static int launch_r2(const ccsd_plan* pl, int B, const float* rank2, const float* adj, const float* flags, int want_p,
                     RankEpi& ep, NoiseArgs& na, Workspace& w, void* stream, const CorrFuse* cf, int merge_draw);
// hodge projections for ScoreNetworkA_CC from (adj, rank2)
// fuse (tiled path, h_L == 1; tiled_fuse_ok): the Langevin corrector's element-wise work on rank2 rides on the layer-0 projection
// pass -- mode 1: the noise norm of the corrector's draw per row (-> fuse->zrow), mode 2: the corrector apply (corrected rank2 -> fuse->f1,
// which the projection is then taken of).  See P0Fuse (ccsd_k_rank2.h).
static int launch_p(const ccsd_plan* pl, int B, const float* adj, const float* rank2, Workspace& w, void* stream, const P0Fuse* fuse = nullptr) {
    const PlanD& p = pl->h;
    if (p.h_L < 1) return CCSD_OK;
    w.hg_rank2 = rank2;
    if (p.h_L > 2 && !pl->h_general) {
        // more than two hodge layers: k_xa's general layer loop consumes the factors only the fused rank-2 kernel produces
        // (plan creation guarantees it exists); its ScoreNetworkF output lands in the net_r scratch, which every caller
        // of launch_p overwrites afterwards
        RankEpi ep{};
        ep.mode = MODE_SCORE; ep.sscale = 0.f; ep.out = w.net_r;
        NoiseArgs na0{};
        return launch_r2(pl, B, rank2, adj, nullptr, 1, ep, na0, w, stream, nullptr, -1);
    }
    w.p1_raw = 0;
    const int rows = B * p.E;
#ifndef CCSD_EMU
    // community_small geometry, at least one complex per CU, the corrector's work riding on the pass (modes 1 / 2): ONE kernel streams
    // the block once and leaves P_0, H and the noise norm / the corrected state (k_hp_full; P_0 and H bit-identical to the two-kernel
    // route); the launch_h that follows in the caller finds H done
    // (mode 1 -- the norms pass, whose only extra is the noise norm -- is slower fused: 370 us against 129 + 213, the eight waves of the one
    // workgroup a CU holds run in lockstep; CCSD_HP_FULL_NORMS turns it on for A/B)
    if (fuse && (fuse->mode == 2 || (fuse->mode == 1 && pl->opt_hp_full_norms)) && hp_full_ok(pl, B)) {
        const HodgeLayerD& h = p.hl[0];
        const float* WT = (const float*)pl->wp + h.wcatT;
        const size_t lds = (size_t)(2 * 192 * H_LD + 2 * 16 * H_LD) * 4;
        prof_mark(const_cast<ccsd_plan*>(pl), KID_GEMM_H, stream);
        if (fuse->mode == 1)
            hipLaunchKernelGGL((k_hp_full<190, 1140, 1>), dim3(B), dim3(512), lds, (hipStream_t)stream, rank2, WT, w.H, w.P0, h.wc, p.f_hmask, *fuse);
        else
            hipLaunchKernelGGL((k_hp_full<190, 1140, 2>), dim3(B), dim3(512), lds, (hipStream_t)stream, rank2, WT, w.H, w.P0, h.wc, p.f_hmask, *fuse);
        prof_mark(const_cast<ccsd_plan*>(pl), KID_GEMM_H, stream);
        LAUNCH_CHECK();
        w.h_done = p.f_cnum == 2;            // (more powers: k_gemm_pow needs launch_h's loop -- not this geometry's shipped network)
        return CCSD_OK;
    }
#endif
    {
        const HodgeLayerD& h = p.hl[0];
        dim3 g((h.wc + T_BN - 1) / T_BN, (rows + T_BM - 1) / T_BM, 1);
        prof_mark(const_cast<ccsd_plan*>(pl), KID_GEMM_P, stream);
        P0Fuse pf{};
        if (fuse) pf = *fuse;
#ifndef CCSD_EMU
        const int nt = (h.wc + 15) / 16, Kp = (p.K + 31) & ~31;
        if (nt <= 4 && !pl->opt_old_gemm_p) {     // narrow projections: no 64-column padding
            const dim3 g0((rows + T_BM - 1) / T_BM);
            const float* WT = (const float*)pl->wp + h.wcatT;
#define P0_GO(NT_, KC_, M_) hipLaunchKernelGGL((k_gemm_p0<NT_, KC_, M_>), g0, dim3(256), 0, (hipStream_t)stream, rank2, WT, w.P0, rows, p.K, Kp, h.wc, pf)
#define P0_MODES(NT_, KC_) do { if (pf.mode == 1) P0_GO(NT_, KC_, 1); else if (pf.mode == 2) P0_GO(NT_, KC_, 2); \
                                else if (pf.mode == 3) P0_GO(NT_, KC_, 3); else if (pf.mode == 4) P0_GO(NT_, KC_, 4); else P0_GO(NT_, KC_, 0); } while (0)
            switch (nt) {
                case 1:
                    if (p.K == 1140 && p.geo_off != 1) P0_MODES(1, 1140);
                    else if (p.K == 8436 && p.geo_off != 1) P0_MODES(1, 8436);
                    else P0_MODES(1, 0);
                    break;
                case 2: P0_MODES(2, 0); break;
                case 3: P0_MODES(3, 0); break;
                default: P0_MODES(4, 0); break;
            }
#undef P0_MODES
#undef P0_GO
        } else
#endif
        {
            const float* src = rank2;
            if (pf.mode) {      // (host emulation / wide projections: the corrector work as an element-wise pass of its own)
                CCSD_LAUNCH(k_p0_fuse_ew, dim3(grid_for(rows, 256)), dim3(CCSD_NTHREADS), 0, stream, rank2, pf, rows, p.K);
                LAUNCH_CHECK();
                if (pf.mode == 2 || pf.mode == 4) src = pf.f1;
            }
            CCSD_LAUNCH(k_gemm_p, g, dim3(CCSD_NTHREADS), 0, stream, src, (const float*)pl->w, w.P0, rows, p.E, p.K, h.wc,
                        h.wcat, 0, h.mval, h.cin, (const float*)nullptr, (const unsigned long long*)w.offbits,
                        (const unsigned char*)pl->edges, (const unsigned long long*)pl->cells, p.K);
        }
        prof_mark(const_cast<ccsd_plan*>(pl), KID_GEMM_P, stream);
        LAUNCH_CHECK();
    }
    if (p.h_L > 1) {
        CCSD_LAUNCH(k_edgecoef, dim3(B), dim3(CCSD_NTHREADS), (size_t)3 * p.N * p.N * 4, stream, adj, w.acoef, p.N, p.E,
                    p.a_cinit, (const unsigned char*)pl->edges);
        LAUNCH_CHECK();
        const HodgeLayerD& h0 = p.hl[0];
        const HodgeLayerD& h = p.hl[1];
        dim3 g((h.wc + T_BN - 1) / T_BN, (rows + T_BM - 1) / T_BM, 1);
        // K is ALWAYS split into the same slices (up to CCSD_P_SPLITS, one workgroup grid layer each, summed in a fixed order by
        // k_sum_splits): the slicing is a function of (K, T_BK) alone, so the summation order of a row of P_1 -- hence every score
        // downstream -- does not depend on the batch or shard size (a sharded run, a divide_batch chunk and the whole batch agree
        // per sample).  Small batches need the slices anyway to fill the chip (ENZYMES_small_CC at B = 64 has 66 row tiles).
        const int nslab = (p.K + T_BK - 1) / T_BK;
        int S = CCSD_P_SPLITS < nslab ? CCSD_P_SPLITS : nslab;
        if ((size_t)S * rows * h.wc > w.psplit_floats) S = 1;     // (cannot happen: carve_ws sizes the slices for every batch)
        const int kchunk = ((nslab + S - 1) / S) * T_BK;
        S = (p.K + kchunk - 1) / kchunk;
        g.z = S;
        float* P1 = pl->h_general ? w.hgP[0] : w.P1;
        CCSD_LAUNCH(k_gemm_p, g, dim3(CCSD_NTHREADS), 0, stream, rank2, (const float*)pl->w, S > 1 ? w.psplit : P1, rows, p.E, p.K, h.wc,
                    h.wcat, 1, h0.mval, h0.cin, (const float*)w.acoef, (const unsigned long long*)w.offbits,
                    (const unsigned char*)pl->edges, (const unsigned long long*)pl->cells, kchunk);
        LAUNCH_CHECK();
        if (S > 1) {
            const long long n = (long long)rows * h.wc;
            CCSD_LAUNCH(k_sum_splits, dim3(grid_for(n, 256)), dim3(CCSD_NTHREADS), 0, stream, (const float*)w.psplit, P1, n, S);
            LAUNCH_CHECK();
        }
        if (pl->h_general) {
            // R_1 = fl fr mlp_value_0(a_c o rank2), materialised for the layers behind it (launch_xa goes on from here)
            const int cw = p.E > 128 ? 32 : 64;
            CCSD_LAUNCH(k_hodge_value, dim3((p.K + cw - 1) / cw, B), dim3(CCSD_NTHREADS), (size_t)p.E * cw * 4, stream, rank2, (const float*)nullptr, 0,
                        (const float*)w.acoef, (const float*)pl->w, h0.mval, h0.cin, w.hgR[0], p.E, p.K, cw,
                        (const unsigned long long*)w.offbits, (const unsigned char*)pl->edges, (const unsigned long long*)pl->cells);
            LAUNCH_CHECK();
        }
    }
    return CCSD_OK;
}
static int launch_xa(const ccsd_plan* pl, int B, XaArgs& xa, NoiseArgs& na, Workspace& w, void* stream, bool set_b = false) {
    xa.P0 = set_b ? w.P0b : w.P0; xa.P1 = set_b ? w.P1b : w.P1; xa.U1 = set_b ? w.U1b : w.U1; xa.chan_ws = w.chan;
    xa.p1_raw = w.p1_raw; xa.dbg = pl->dbg ? pl->dbg + 32 : nullptr;
    xa.stagger_mask = pl->opt_xa_stagger_mask; xa.stagger_sleep = pl->opt_xa_stagger_sleep; xa.prio_mode = pl->opt_xa_prio;
    // Threads per graph.  256 (four waves) is right when the batch fills the chip -- 1024 graphs = four co-resident workgroups per CU --
    // and k_xa is bound by the latency of one graph's critical path either way; when the batch leaves a CU with one or two workgroups
    // (B <= 256 / <= 512) the same graph runs on sixteen / eight waves: every per-pair, per-tile and per-element loop of the kernel strides
    // by the workgroup's thread count (zinc250k B = 256: 419 -> 293 us per launch; community_small_CC B = 512: 418 -> 349 us; ENZYMES_small_CC
    // B = 64: 236 -> 180 us).  Only the instances compiled for 4 waves per SIMD without a fixed thread count take more than 256 (XA_4WAVES in
    // ccsd_k_xa.h); CCSD_XA_THREADS (read at plan creation) overrides the choice (diagnostic: 64 .. 1024).
    int xa_threads = pl->opt_xa_threads;
    const int v0 = xa_variant(pl->h);
    // most threads the variant's instance may be launched with (its __launch_bounds__, XA_4WAVES in ccsd_k_xa.h; the qm9 instances have
    // their 256 compiled in -- an override moves those plans to the run-time-geometry twin below)
    const bool fixed256 = v0 == XA_PLAIN9 || v0 == XA_BAKED9;
    const int max_threads = (pl->h.chan_global && v0 != XA_BAKED20 && v0 != XA_BAKED38) ? 256 : 1024;
    if (xa_threads == 0) xa_threads = (fixed256 || max_threads == 256) ? 256 : B <= 256 ? 1024 : B <= 512 ? 512 : 256;
    if (xa_threads > max_threads) xa_threads = max_threads;
    prof_mark(const_cast<ccsd_plan*>(pl), KID_XA, stream);
    xa.wp = pl->wp; xa.hpairs = pl->hpairs;
    const dim3 xblk(CCSD_NTHREADS == 1 ? 1 : xa_threads);
    const size_t xlds = (size_t)pl->h.xa_lds_floats * 4;
#define XA_GO(G_, V_, XA_, BLK_, LDS_, STR_) CCSD_LAUNCH((k_xa<G_, V_>), dim3(B), BLK_, LDS_, STR_, (const PlanD*)pl->d, (const float*)pl->w, \
                                                         (const unsigned char*)pl->edges, XA_, na)
    int variant = xa_variant(pl->h);
    if ((variant == XA_PLAIN9 || variant == XA_BAKED9) && xa_threads != 256) variant = XA_PLAIN;      // (they have their 256 threads compiled in)
    if (pl->h_general) {
        // general hodge stack: layer l >= 2 projects R_l = fl fr mlp_value_(l-1)(cat_c H^(l-1)_c R_(l-1)).  H^(l-1) is the dense output of
        // layer l - 2 inside k_xa: a launch that stops behind it dumps it, k_hodge_value forms R_l, k_gemm_p projects it -- then the next
        // layer, and at last the full launch with every P_l delivered.  (launch_p left P_0, P_1 and R_1.)
        const PlanD& p = pl->h;
        if (variant != XA_GEN) return set_err(CCSD_ERR_RUNTIME, "general hodge stack needs k_xa<., XA_GEN>");
        xa.pdirect = 1;
        for (int l = 1; l < p.h_L; ++l) xa.Pd[l - 1] = w.hgP[l - 1];
        const int rows = B * p.E;
        for (int l = 2; l < p.h_L; ++l) {
            XaArgs pre = xa;
            pre.hdump_layer = l - 1; pre.hdump = w.hgH; pre.hdump_stride = (int)w.hg_hstride;
            if (pl->h.chan_global) XA_GO(true, XA_GEN, pre, xblk, xlds, stream); else XA_GO(false, XA_GEN, pre, xblk, xlds, stream);
            LAUNCH_CHECK();
            const HodgeLayerD& hp = ccsd_hl(p, l - 1);
            const HodgeLayerD& h = ccsd_hl(p, l);
            float* Rl = w.hgR[(l - 1) & 1];
            const int cw = p.E > 128 ? 32 : 64;
            CCSD_LAUNCH(k_hodge_value, dim3((p.K + cw - 1) / cw, B), dim3(CCSD_NTHREADS), (size_t)p.E * cw * 4, stream, (const float*)w.hgR[(l - 2) & 1],
                        (const float*)w.hgH, (int)w.hg_hstride, (const float*)nullptr, (const float*)pl->w, hp.mval, hp.cin, Rl, p.E, p.K, cw,
                        (const unsigned long long*)w.offbits, (const unsigned char*)pl->edges, (const unsigned long long*)pl->cells);
            LAUNCH_CHECK();
            CCSD_LAUNCH(k_gemm_p, dim3((h.wc + T_BN - 1) / T_BN, (rows + T_BM - 1) / T_BM, 1), dim3(CCSD_NTHREADS), 0, stream, (const float*)Rl,
                        (const float*)pl->w, w.hgP[l - 1], rows, p.E, p.K, h.wc, h.wcat, 0, h.mval, h.cin, (const float*)nullptr,
                        (const unsigned long long*)w.offbits, (const unsigned char*)pl->edges, (const unsigned long long*)pl->cells, p.K);
            LAUNCH_CHECK();
        }
    }
    if (pl->h.chan_global) {
        if (variant == XA_HB) XA_GO(true, XA_HB, xa, xblk, xlds, stream);
        else if (variant == XA_GMH) XA_GO(true, XA_GMH, xa, xblk, xlds, stream);
        else if (variant == XA_GEN) XA_GO(true, XA_GEN, xa, xblk, xlds, stream);
        else if (variant == XA_PLAIN20) XA_GO(true, XA_PLAIN20, xa, xblk, xlds, stream);
        else if (variant == XA_BAKED20) XA_GO(true, XA_BAKED20, xa, xblk, xlds, stream);
        else if (variant == XA_PLAIN38) XA_GO(true, XA_PLAIN38, xa, xblk, xlds, stream);
        else if (variant == XA_BAKED38) XA_GO(true, XA_BAKED38, xa, xblk, xlds, stream);
        else XA_GO(true, XA_PLAIN, xa, xblk, xlds, stream);
    } else {
        if (variant == XA_HB) XA_GO(false, XA_HB, xa, xblk, xlds, stream);
        else if (variant == XA_GMH) XA_GO(false, XA_GMH, xa, xblk, xlds, stream);
        else if (variant == XA_GEN) XA_GO(false, XA_GEN, xa, xblk, xlds, stream);
        else if (variant == XA_PLAIN9) XA_GO(false, XA_PLAIN9, xa, xblk, xlds, stream);
        else if (variant == XA_BAKED9) XA_GO(false, XA_BAKED9, xa, xblk, xlds, stream);
        else if (variant == XA_BAKEDENZ) XA_GO(false, XA_BAKEDENZ, xa, xblk, xlds, stream);
        else XA_GO(false, XA_PLAIN, xa, xblk, xlds, stream);
    }
#undef XA_GO
    prof_mark(const_cast<ccsd_plan*>(pl), KID_XA, stream);
    LAUNCH_CHECK();
    return CCSD_OK;
}
static int launch_hf(const ccsd_plan* pl, int B, const float* rank2, RankEpi& ep, NoiseArgs& na, Workspace& w, void* stream) {
    const PlanD& p = pl->h;
    dim3 g(xcd_grid(B, ((p.K + T_BN - 1) / T_BN) * ((p.E + T_BM - 1) / T_BM)));
    prof_mark(const_cast<ccsd_plan*>(pl), KID_HF, stream);
#define HF_ARGS (const PlanD*)pl->d, (const float*)pl->w, rank2, (const float*)w.H, (const unsigned long long*)w.offbits, \
                (const unsigned char*)pl->edges, (const unsigned long long*)pl->cells, ep, na, B, (MaskTab{w.mfr, w.mfl, w.Kp, w.Ep})
    const int fw = fnet_width(p);
#define HF_AFF1(EC_, KC_) CCSD_LAUNCH((k_hf_score<true, 8, 1, EC_, KC_>), g, dim3(CCSD_NTHREADS), 0, stream, HF_ARGS)
#define HF_GEN1(EC_, KC_) CCSD_LAUNCH((k_hf_score<false, 8, 1, EC_, KC_>), g, dim3(CCSD_NTHREADS), 0, stream, HF_ARGS)
#define HF_GO(NP_) \
    do { \
        if (p.f_affine && NP_ == 1) GEO_EK(p, HF_AFF1); \
        else if (p.f_affine) CCSD_LAUNCH((k_hf_score<true, 8, NP_>), g, dim3(CCSD_NTHREADS), 0, stream, HF_ARGS); \
        else if (fw <= 8 && NP_ == 1) GEO_EK(p, HF_GEN1); \
        else if (fw <= 8) CCSD_LAUNCH((k_hf_score<false, 8, NP_>), g, dim3(CCSD_NTHREADS), 0, stream, HF_ARGS); \
        else if (fw <= CCSD_FW) CCSD_LAUNCH((k_hf_score<false, CCSD_FW, NP_>), g, dim3(CCSD_NTHREADS), 0, stream, HF_ARGS); \
        else CCSD_LAUNCH((k_hf_score<false, CCSD_FWMAX, NP_>), g, dim3(CCSD_NTHREADS), 0, stream, HF_ARGS); \
    } while (0)
    if (p.f_cnum <= 2) HF_GO(1); else HF_GO(CCSD_MAXCN - 1);
#undef HF_GO
#undef HF_AFF1
#undef HF_GEN1
#undef HF_ARGS
    prof_mark(const_cast<ccsd_plan*>(pl), KID_HF, stream);
    LAUNCH_CHECK();
    return CCSD_OK;
}

// element-wise ScoreNetworkF (k_ew1).  ep: mode / scalars / out / mean / part as for k_hf_score; `net_out` (NORMS, nullable): keep the
// raw score; `cf` (PRED, nullable): fused corrector apply, F1 goes to `f1`
static int launch_ew1(const ccsd_plan* pl, int B, const float* rank2, RankEpi& ep, NoiseArgs& na, Workspace& w, void* stream,
                      float* net_out = nullptr, const CorrFuse* cf = nullptr, float* f1 = nullptr) {
    const PlanD& p = pl->h;
    Ew1Args a{};
    a.r = rank2; a.out = ep.out; a.mean = ep.mean; a.f1 = f1; a.net_out = net_out; a.part = ep.part;
    a.mode = ep.mode; a.apply = (cf && cf->on) ? 1 : 0;
    a.sscale = ep.sscale; a.pa = ep.pa; a.pb = ep.pb; a.pc = ep.pc; a.alpha = p.f_alpha; a.gamma = p.f_gamma;
    if (a.apply) { a.sums = cf->sums; a.ss = cf->ss[2]; a.sde_alpha = cf->alpha[2]; a.snr = cf->snr; a.seps = cf->seps; a.draw_corr = cf->draw_r; }
    a.E = p.E; a.K = p.K; a.mt = MaskTab{w.mfr, w.mfl, w.Kp, w.Ep};
    prof_mark(const_cast<ccsd_plan*>(pl), KID_EW1, stream);
#define EW1_GO(EC_, KC_) CCSD_LAUNCH((k_ew1<EC_, KC_>), dim3(w.nchunk, B), dim3(CCSD_NTHREADS), 0, stream, a, na)
    GEO_EK(p, EW1_GO);
#undef EW1_GO
    prof_mark(const_cast<ccsd_plan*>(pl), KID_EW1, stream);
    LAUNCH_CHECK();
    return CCSD_OK;
}

// merge_draw >= 0: merged launch -- after this (predictor) pass the kernel runs the NEXT corrector's norms pass on the new block
// (draw index merge_draw; raw score -> w.net_r, partials -> w.part, projections -> the second buffer set)
static int launch_r2(const ccsd_plan* pl, int B, const float* rank2, const float* adj, const float* flags, int want_p,
                     RankEpi& ep, NoiseArgs& na, Workspace& w, void* stream, const CorrFuse* cf = nullptr, int merge_draw = -1) {
    R2Args ra{};
    if (cf) ra.cf = *cf;
    // launches of the sampler loop that carry the fused corrector apply work on states this library produced: masked (R2Args::masked)
    ra.masked = (cf && cf->on && pl->opt_r2_masked) ? 1 : 0;
    if (merge_draw >= 0) {
        ra.merge = 1; ra.draw_r2 = (unsigned)merge_draw;
        ra.P0b = w.P0b; ra.P1b = w.P1b; ra.U1b = w.U1b; ra.net2 = w.net_r; ra.part2 = w.part;
    }
    ra.rank2 = rank2; ra.adj = adj; ra.flags = flags; ra.offbits = w.offbits; ra.P0 = w.P0; ra.P1 = w.P1; ra.U1 = w.U1; ra.want_p = want_p;
    ra.ldk = pl->r2_ldk; ra.ldh = pl->r2_ldh; ra.dbg = pl->dbg; ra.wp = pl->wp;
    ra.stagger_mask = pl->opt_r2_stagger_mask; ra.stagger_sleep = pl->opt_r2_stagger_sleep;
    if (want_p && pl->h.h_L > 1) w.p1_raw = pl->h.hl[0].mval.n == 1;
    prof_mark(const_cast<ccsd_plan*>(pl), KID_R2, stream);
    const dim3 blk(CCSD_NTHREADS == 1 ? 1 : 512);
#define R2_GO(MT_, RS_, A_, G_) \
    CCSD_LAUNCH((k_r2<MT_, RS_, A_, G_>), dim3(B), blk, pl->r2_lds, stream, (const PlanD*)pl->d, (const float*)pl->w, \
                (const unsigned char*)pl->edges, (const unsigned long long*)pl->cells, ra, ep, na)
    const int qm9 = r2_qm9(pl);
    if (qm9 == 2) {
        CCSD_LAUNCH((k_r2<3, 1, true, false, 2>), dim3(B), blk, pl->r2_lds, stream, (const PlanD*)pl->d, (const float*)pl->w,
                    (const unsigned char*)pl->edges, (const unsigned long long*)pl->cells, ra, ep, na);
    } else if (qm9 == 1) {
        CCSD_LAUNCH((k_r2<3, 1, true, false, 1>), dim3(B), blk, pl->r2_lds, stream, (const PlanD*)pl->d, (const float*)pl->w,
                    (const unsigned char*)pl->edges, (const unsigned long long*)pl->cells, ra, ep, na);
    } else if (qm9 == 3) {
        CCSD_LAUNCH((k_r2<3, 0, false, false, 1>), dim3(B), blk, pl->r2_lds, stream, (const PlanD*)pl->d, (const float*)pl->w,
                    (const unsigned char*)pl->edges, (const unsigned long long*)pl->cells, ra, ep, na);
    } else
    R2_DISPATCH(pl, R2_GO);
#undef R2_GO
    prof_mark(const_cast<ccsd_plan*>(pl), KID_R2, stream);
    LAUNCH_CHECK();
    return CCSD_OK;
}

static unsigned int draw_base(const ccsd_plan* pl, int step, int phase) {
    const int per_step = pl->cfg.predictor == CCSD_PRED_S4 ? 3 : pl->cfg.n_corr_steps + 1;   // S4: three draws per target per step
    return 3u + (unsigned)((step * per_step + phase) * 3);
}
static NoiseArgs make_noise(const ccsd_noise_t* n, uint64_t seed, int64_t off, unsigned int base, int flat_r = 0) {
    NoiseArgs na{};
    if (n) { na.zx = n->zx; na.zadj = n->zadj; na.zr = n->zrank2; }
    na.seed = seed; na.b_off = off;
    na.draw_x = base; na.draw_adj = base + 1; na.draw_r = base + 2;
    na.flat_r = flat_r;
    return na;
}
// The Langevin corrector's rank2 draws are keyed by flat groups of four consecutive elements (NoiseArgs::flat_r): they are
// generated where rank2 streams through registers in 16-byte pieces (k_r2's block load, k_langevin_apply, k_noise_norm).
// Priors, predictor draws and the three draws of an S4 step keep the 4-row groups of the MFMA epilogues.
static inline int corrector_flat(const ccsd_plan* pl) { return pl->cfg.predictor != CCSD_PRED_S4 ? 1 : 0; }

// does ccsd_sampler_run fuse the Langevin corrector's apply pass into the predictor launches of this plan?
// merged k_r2 launches (predictor of step i + rank-2 side of the norms pass of step i + 1): the row-strip instantiation of the
// kernel (E = 33..36, affine ScoreNetworkF, linear mlp_value), pair-wise block load (K even, E K a multiple of 4)
static inline bool merge_ok(const ccsd_plan* pl) {
    if (!pl->fused_r2 || pl->opt_no_fused_apply || pl->opt_no_merge) return false;
    int mt, rs; bool aff, gen1;
    r2_shape(pl, &mt, &rs, &aff, &gen1);
    return mt == 3 && rs == 1 && aff && !gen1 && (pl->h.K & 1) == 0 && ((pl->h.E * pl->h.K) & 3) == 0;
}
// tiled rank-2 path (k_gemm_h / k_gemm_p0 / k_hf_score: community_small_CC) with ONE hodge layer: the corrector's rank2 work rides on
// the layer-0 projection pass (P0Fuse) -- flat-keyed corrector draws, K a multiple of 4 (a Philox group = one 16-byte piece of a row)
static inline bool tiled_fuse_ok(const ccsd_plan* pl) {
    return pl->cfg.is_cc && !pl->fused_r2 && !pl->ew1 && pl->h.h_L == 1 && (pl->h.K & 3) == 0 && pl->cfg.predictor != CCSD_PRED_S4 &&
           pl->cfg.corrector == CCSD_CORR_LANGEVIN && !pl->opt_no_tiled_fuse;
}
// element-wise ScoreNetworkF plans (k_ew1's: affine, cnum = 1) with ONE hodge layer: the whole rank-2 side of a half-step rides on the
// layer-0 projection pass (P0Fuse modes 3 / 4) -- one read of rank2 per norms pass, one read + one write per predictor pass
static inline bool ew1_fuse_ok(const ccsd_plan* pl) {
    return pl->ew1 && pl->h.h_L == 1 && (pl->h.K & 3) == 0 && pl->cfg.corrector == CCSD_CORR_LANGEVIN && !pl->opt_no_tiled_fuse;
}
// (k_r2 plans; k_ew1 plans whose hodge projections do not depend on the adjacency: one hodge layer; tiled plans with one hodge layer)
static inline bool fused_apply_ok(const ccsd_plan* pl) {
    return (pl->fused_r2 || (pl->ew1 && pl->h.h_L <= 1) || tiled_fuse_ok(pl)) && !pl->opt_no_fused_apply;
}
// every rank2 draw of a k_ew1 plan is keyed by flat groups (the kernel streams 16-byte pieces); otherwise only the corrector's
static inline int predictor_flat(const ccsd_plan* pl) { return pl->ew1 ? 1 : 0; }

// ---------------- API ----------------
extern "C" int ccsd_score(ccsd_plan_t* pl, int32_t target, int32_t B, const ccsd_state_t* in, const float* flags,
                          float sscale, float* out, void* workspace, size_t ws_bytes, void* stream) {
    int st = check_common(pl, B, flags, workspace, ws_bytes);
    if (st) return st;
    if ((st = check_state(pl, in, "in"))) return st;
    if (!out) return set_err(CCSD_ERR_INVALID, "NULL out");
    Workspace w = carve_ws(pl, B, workspace);
    if ((st = launch_flagbits(pl, B, flags, w, stream))) return st;
    NoiseArgs na{};
    if (target == CCSD_TARGET_X || target == CCSD_TARGET_ADJ) {
        XaArgs xa{};
        xa.xX = xa.xA = in->x; xa.adjX = xa.adjA = in->adj; xa.flags = flags;
        xa.do_x = target == CCSD_TARGET_X; xa.do_a = !xa.do_x; xa.mode = MODE_SCORE;
        xa.ss_x = xa.ss_a = sscale; xa.out_x = xa.out_a = out;
        if (xa.do_a && (st = launch_p(pl, B, in->adj, in->rank2, w, stream))) return st;
        return launch_xa(pl, B, xa, na, w, stream);
    }
    if (target == CCSD_TARGET_RANK2) {
        if (!pl->h.is_cc) return set_err(CCSD_ERR_INVALID, "rank2 score requested from a graph-only plan");
        RankEpi ep{};
        ep.mode = MODE_SCORE; ep.sscale = sscale; ep.out = out;
        if (pl->fused_r2) return launch_r2(pl, B, in->rank2, in->adj, flags, 0, ep, na, w, stream);
        if (pl->ew1) return launch_ew1(pl, B, in->rank2, ep, na, w, stream);
        if ((st = launch_h(pl, B, in->rank2, w, stream))) return st;
        return launch_hf(pl, B, in->rank2, ep, na, w, stream);
    }
    return set_err(CCSD_ERR_UNSUPPORTED, "Object not yet supported. Select from [x, adj, rank2].");
}

// masked draws of one draw base into `state` (k_init_state): the prior for base 0, the noise of a half-step otherwise
static int draws_to_state(ccsd_plan* pl, int32_t B, const float* flags, const ccsd_noise_t* raw, uint64_t seed, int64_t sample_offset,
                          unsigned int base, ccsd_state_t* state, void* stream, int flat_r = 0) {
    if (!pl || B < 1 || !flags) return set_err(CCSD_ERR_INVALID, "bad argument");
    int st = check_state(pl, state, "state");
    if (st) return st;
    const PlanD& p = pl->h;
    // this call takes no workspace: the off-bit table lives in a plan-owned buffer that only ever grows, so the call stays
    // asynchronous on `stream`.  A plan is reentrant per handle only and its calls belong on ONE stream (include/ccsd_hip.h):
    // when the buffer has to grow, the whole device is drained first, whatever stream an earlier call used
    if (pl->init_off_cap < (size_t)B) {
        if (pl->init_off) {
#ifndef CCSD_EMU
            (void)hipDeviceSynchronize();
#endif
            (void)rt_free(pl->init_off);
            pl->init_off = nullptr; pl->init_off_cap = 0;
        }
        RT_CHECK(rt_malloc((void**)&pl->init_off, (size_t)B * 8));
        pl->init_off_cap = (size_t)B;
    }
    unsigned long long* offbits = pl->init_off;
    CCSD_LAUNCH(k_flagbits, dim3(grid_for(B, 256)), dim3(CCSD_NTHREADS), 0, stream, flags, offbits, B, p.N);
    NoiseArgs na = make_noise(raw, seed, sample_offset, base, flat_r);
    const long long total = (long long)B * (p.N * p.F + p.N * p.N) +
                            (p.is_cc ? (flat_r ? (long long)B * (((long long)p.E * p.K + 3) / 4) : (long long)B * ((p.E + 3) / 4) * p.K) : 0);
    CCSD_LAUNCH(k_init_state, dim3(grid_for(total, 256)), dim3(CCSD_NTHREADS), 0, stream, state->x, state->adj, state->rank2,
                flags, na, (const unsigned long long*)offbits, (const unsigned char*)pl->edges,
                (const unsigned long long*)pl->cells, B, p.N, p.F, p.E, p.K, p.is_cc);
    LAUNCH_CHECK();
    return CCSD_OK;
}

extern "C" int ccsd_init_state(ccsd_plan_t* pl, int32_t B, const float* flags, const ccsd_noise_t* prior, uint64_t seed,
                               int64_t sample_offset, ccsd_state_t* state, void* stream) {
    return draws_to_state(pl, B, flags, prior, seed, sample_offset, 0, state, stream);
}

extern "C" int ccsd_noise_draws(ccsd_plan_t* pl, int32_t B, const float* flags, uint64_t seed, int64_t sample_offset, int32_t step,
                                int32_t phase, ccsd_state_t* out, void* stream) {
    if (!pl) return set_err(CCSD_ERR_INVALID, "NULL plan");
    const int per_step = pl->cfg.predictor == CCSD_PRED_S4 ? 3 : pl->cfg.n_corr_steps + 1;
    if (step < 0 || step >= pl->cfg.diff_steps || phase < 0 || phase >= per_step) return set_err(CCSD_ERR_INVALID, "step / phase out of range");
    const bool corr = pl->cfg.predictor != CCSD_PRED_S4 && phase < pl->cfg.n_corr_steps && pl->cfg.corrector == CCSD_CORR_LANGEVIN;
    return draws_to_state(pl, B, flags, nullptr, seed, sample_offset, draw_base(pl, step, phase), out, stream,
                          corr ? corrector_flat(pl) : predictor_flat(pl));
}

extern "C" int ccsd_plan_query(const ccsd_plan_t* pl, int32_t what, int64_t* value) {
    if (!pl || !value) return set_err(CCSD_ERR_INVALID, "NULL argument");
    switch (what) {
        case CCSD_QUERY_FUSED_R2: *value = pl->fused_r2; break;
        case CCSD_QUERY_XA_VARIANT: *value = xa_variant(pl->h); break;
        case CCSD_QUERY_R2_LDS_BYTES: *value = (int64_t)pl->r2_lds; break;
        case CCSD_QUERY_XA_LDS_BYTES: *value = (int64_t)pl->h.xa_lds_floats * 4; break;
        case CCSD_QUERY_FUSED_LOOP: {   // ccsd_sampler_run fuses the Langevin apply into the predictor launches
            const bool s4 = pl->cfg.predictor == CCSD_PRED_S4;
            *value = (!s4 && pl->cfg.corrector == CCSD_CORR_LANGEVIN && pl->cfg.n_corr_steps == 1 && fused_apply_ok(pl)) ? 1 : 0;
            break;
        }
        case CCSD_QUERY_MERGED_R2: *value = (pl->cfg.predictor != CCSD_PRED_S4 && pl->cfg.corrector == CCSD_CORR_LANGEVIN && pl->cfg.n_corr_steps == 1 &&
                                             fused_apply_ok(pl) && merge_ok(pl)) ? 1 : 0; break;
        case CCSD_QUERY_EW1: *value = pl->ew1; break;
        default: return set_err(CCSD_ERR_INVALID, "unknown query");
    }
    return CCSD_OK;
}

// phase 1 of the Langevin corrector.  `base` = pre-corrector state, `cur` = per-target current
// iterate (== base for the first inner step).
// r2_done: the rank-2 side of this norms pass (raw score, partials, hodge projections in the second buffer set) was already
// produced by the merged k_r2 launch of the previous predictor half-step
static int corrector_norms(ccsd_plan* pl, int B, int step, int it, const ccsd_state_t* base, const ccsd_state_t* cur,
                           const float* flags, const ccsd_noise_t* noise, uint64_t seed, int64_t off, float* sums,
                           Workspace& w, void* stream, bool keep_net = true, bool r2_done = false) {
    const PlanD& p = pl->h;
    int st;
    NoiseArgs na = make_noise(noise, seed, off, draw_base(pl, step, it), corrector_flat(pl));
    // A-net sees (x_0, adj_cur, rank2_0): hodge projections from the base rank2, edge coefficients from adj_cur.
    // When the rank2 iterate is still the base state the fused kernel serves both the A-net's projections
    // and ScoreNetworkF in one pass over rank2.
    const bool fused = pl->fused_r2 && p.is_cc && cur->rank2 == base->rank2;
    // tiled path, one hodge layer: the noise norm of the corrector's (flat-keyed, in-kernel) rank2 draw rides on the projection pass
    const bool zfuse = !fused && tiled_fuse_ok(pl) && na.flat_r && !na.zr;
    // element-wise ScoreNetworkF plans: raw score + both norms per row ride on it too (no k_ew1 launch in this pass)
    const bool e1fuse = !fused && ew1_fuse_ok(pl) && na.flat_r && !na.zr;
    int ntiles = w.ntiles;
    if (fused) {
        RankEpi ep{};
        ep.mode = MODE_NORMS; ep.out = w.net_r; ep.part = w.part;
        if (!r2_done && (st = launch_r2(pl, B, cur->rank2, cur->adj, flags, 1, ep, na, w, stream))) return st;
        ntiles = 1;
    } else {
        P0Fuse pf{};
        if (zfuse) {
            pf.mode = 1; pf.zrow = w.zpart; pf.seed = na.seed; pf.b_off = na.b_off; pf.draw = na.draw_r;
            pf.mt = MaskTab{w.mfr, w.mfl, w.Kp, w.Ep}; pf.E = p.E;
        }
        if (e1fuse && cur->rank2 == base->rank2) {
            pf.mode = 3; pf.zrow = w.part; pf.seed = na.seed; pf.b_off = na.b_off; pf.draw = na.draw_r;
            pf.mt = MaskTab{w.mfr, w.mfl, w.Kp, w.Ep}; pf.E = p.E; pf.alpha = p.f_alpha; pf.gamma = p.f_gamma;
            pf.net_out = keep_net ? w.net_r : nullptr;
        }
        if ((st = launch_p(pl, B, cur->adj, base->rank2, w, stream, pf.mode ? &pf : nullptr))) return st;
    }
    XaArgs xa{};
    xa.xX = cur->x; xa.adjX = base->adj;      // score_x(x_cur, adj_0)      solver.py:761
    xa.xA = base->x; xa.adjA = cur->adj;      // score_adj(x_0, adj_cur)    solver.py:775
    xa.flags = flags; xa.do_x = xa.do_a = 1; xa.mode = MODE_NORMS;
    xa.out_x = w.net_x; xa.out_a = w.net_adj; xa.norm2 = w.norm2;
    if ((st = launch_xa(pl, B, xa, na, w, stream, fused && r2_done))) return st;
    if (p.is_cc && !fused && pl->ew1 && e1fuse && cur->rank2 == base->rank2) {
        ntiles = p.E;                            // (per-row partials written by the projection pass above)
    } else if (p.is_cc && !fused && pl->ew1) {
        // element-wise ScoreNetworkF: one streaming pass gives both norm partials per (sample, chunk); the raw score is kept only
        // for a separate ccsd_corrector_apply (the fused loop recomputes it)
        RankEpi ep{};
        ep.mode = MODE_NORMS; ep.out = w.net_r; ep.part = w.part;
        if ((st = launch_ew1(pl, B, cur->rank2, ep, na, w, stream, keep_net ? w.net_r : nullptr))) return st;
        ntiles = w.nchunk;
    } else if (p.is_cc && !fused) {
        if ((st = launch_h(pl, B, cur->rank2, w, stream))) return st;
        RankEpi ep{};
        ep.mode = MODE_NORMS; ep.out = w.net_r; ep.part = w.part;
        if ((st = launch_hf(pl, B, cur->rank2, ep, na, w, stream))) return st;
    }
    const float* part = w.part;
    if (p.is_cc && !fused) {
        // tiled path: the noise norm of a flat-keyed Philox draw comes from its own (traffic-free) kernel; the per-tile partials of
        // k_hf_score and its chunk partials are reduced per sample first (one workgroup per sample), then over the batch
        const bool zk = na.flat_r && !na.zr && !pl->ew1;
        if (zk && !zfuse) {
#define NN_GO(EC_, KC_) CCSD_LAUNCH((k_noise_norm<EC_, KC_>), dim3(w.nchunk, B), dim3(CCSD_NTHREADS), 0, stream, na, (MaskTab{w.mfr, w.mfl, w.Kp, w.Ep}), p.E, p.K, w.zpart)
            GEO_EK(p, NN_GO);
#undef NN_GO
            LAUNCH_CHECK();
        }
        if (zk || ntiles > 8) {
            // (zpart: the chunk partials of k_noise_norm, or the E row partials of the fused projection pass)
            CCSD_LAUNCH(k_normpart, dim3(B), dim3(CCSD_NTHREADS), 0, stream, (const float*)w.part, ntiles, zk ? (const float*)w.zpart : (const float*)nullptr,
                        zfuse ? p.E : w.nchunk, w.part2);
            LAUNCH_CHECK();
            part = w.part2; ntiles = 1;
        }
    }
    CCSD_LAUNCH(k_normsum, dim3(1), dim3(CCSD_NTHREADS == 1 ? 1 : (B > 512 ? 1024 : B > 256 ? 512 : 256)), 0, stream, (const float*)w.norm2, part, B, ntiles,
                p.is_cc, sums);
    LAUNCH_CHECK();
    return CCSD_OK;
}
static int corrector_apply(ccsd_plan* pl, int B, int step, int it, const ccsd_state_t* cur, const float* flags,
                           const ccsd_noise_t* noise, uint64_t seed, int64_t off, const float* sums, ccsd_state_t* out,
                           Workspace& w, void* stream) {
    const PlanD& p = pl->h;
    NoiseArgs na = make_noise(noise, seed, off, draw_base(pl, step, it), corrector_flat(pl));
    LangArgs a{};
    a.x = cur->x; a.adj = cur->adj; a.r = cur->rank2;
    a.nx = w.net_x; a.nadj = w.net_adj; a.nr = w.net_r;
    a.ox = out->x; a.oadj = out->adj; a.orr = out->rank2;
    a.flags = flags; a.sums = sums;
    for (int t = 0; t < 3; ++t) {
        const ccsd_step_coef_t& c = pl->coef[(size_t)step * 3 + t];
        a.ss[t] = c.sscale; a.alpha[t] = c.alpha;
    }
    a.snr = p.snr; a.seps = p.seps;
    a.B = B; a.N = p.N; a.F = p.F; a.E = p.E; a.K = p.K; a.is_cc = p.is_cc;
    const long long total = (long long)B * (p.N * p.F + p.N * p.N) + (p.is_cc ? (long long)B * (((long long)p.E * p.K + 3) / 4) : 0);
    prof_mark(pl, KID_LANGEVIN, stream);
#define LA_GO(EC_, KC_) CCSD_LAUNCH((k_langevin_apply<EC_, KC_>), dim3(grid_for(total, 256)), dim3(CCSD_NTHREADS), 0, stream, a, na, (MaskTab{w.mfr, w.mfl, w.Kp, w.Ep}))
    GEO_EK(p, LA_GO);
#undef LA_GO
    prof_mark(pl, KID_LANGEVIN, stream);
    LAUNCH_CHECK();
    return CCSD_OK;
}
static int predictor(ccsd_plan* pl, int B, int step, const ccsd_state_t* in, const float* flags, const ccsd_noise_t* noise,
                     uint64_t seed, int64_t off, ccsd_state_t* out, ccsd_state_t* mean, Workspace& w, void* stream,
                     const float* fuse_sums = nullptr, bool merge_next = false) {
    const PlanD& p = pl->h;
    int st;
    NoiseArgs na = make_noise(noise, seed, off, draw_base(pl, step, pl->cfg.n_corr_steps), predictor_flat(pl));
    const ccsd_step_coef_t* c = &pl->coef[(size_t)step * 3];
    const bool fused = pl->fused_r2 && p.is_cc;
    const bool ew1 = pl->ew1 && p.is_cc && !fused;
    const float* r2_in = in->rank2;           // what the rank-2 kernels of the tiled path read (the corrected state when the apply is fused)
    CorrFuse cf{};
    if (fuse_sums) {   // the Langevin corrector's apply pass runs in the prologues of this half-step's kernels
        cf.on = 1; cf.net_x = w.net_x; cf.net_adj = w.net_adj; cf.net_r = w.net_r; cf.sums = fuse_sums;
        for (int t = 0; t < 3; ++t) { cf.ss[t] = c[t].sscale; cf.alpha[t] = c[t].alpha; }
        cf.snr = p.snr; cf.seps = p.seps;
        const unsigned int cb = draw_base(pl, step, 0);
        cf.draw_x = cb; cf.draw_adj = cb + 1; cf.draw_r = cb + 2;
    }
    if (fused) {
        RankEpi ep{};
        ep.mode = MODE_PRED; ep.pa = c[2].pa; ep.pb = c[2].pb; ep.pc = c[2].pc;
        ep.out = out->rank2; ep.mean = mean ? mean->rank2 : nullptr;
        // merged launch: the rank-2 side of the NEXT step's norms pass follows in the same launch (its corrector draw: rank2 slot of
        // draw_base(step + 1, 0))
        const int md = merge_next ? (int)draw_base(pl, step + 1, 0) + 2 : -1;
        if ((st = launch_r2(pl, B, in->rank2, in->adj, flags, 1, ep, na, w, stream, &cf, md))) return st;
    } else if (ew1 && cf.on && ew1_fuse_ok(pl) && !na.zr) {
        // element-wise ScoreNetworkF, one hodge layer: corrector apply + projection + predictor update in ONE pass over rank2
        P0Fuse pf{};
        pf.mode = 4; pf.seed = na.seed; pf.b_off = na.b_off; pf.draw = cf.draw_r; pf.draw_pred = na.draw_r;
        pf.mt = MaskTab{w.mfr, w.mfl, w.Kp, w.Ep}; pf.E = p.E; pf.cf = cf;
        pf.alpha = p.f_alpha; pf.gamma = p.f_gamma; pf.pa = c[2].pa; pf.pb = c[2].pb; pf.pc = c[2].pc;
        pf.out = out->rank2; pf.mean = mean ? mean->rank2 : nullptr;
        pf.f1 = w.net_r;                         // (host emulation only: its projection runs as a pass of its own over the corrected state)
        if ((st = launch_p(pl, B, in->adj, in->rank2, w, stream, &pf))) return st;
    } else if (ew1) {
        // element-wise ScoreNetworkF first: with the fused apply it produces the corrected rank2 (in the raw-score scratch, which
        // the fused loop does not fill) that the hodge projections of the A-network must see
        RankEpi ep{};
        ep.mode = MODE_PRED; ep.pa = c[2].pa; ep.pb = c[2].pb; ep.pc = c[2].pc;
        ep.out = out->rank2; ep.mean = mean ? mean->rank2 : nullptr;
        if ((st = launch_ew1(pl, B, in->rank2, ep, na, w, stream, nullptr, &cf, w.net_r))) return st;
        if ((st = launch_p(pl, B, in->adj, cf.on ? (const float*)w.net_r : in->rank2, w, stream))) return st;
    } else if (cf.on && tiled_fuse_ok(pl)) {
        // tiled path: the corrector apply rides on the projection pass -- corrected rank2 written in place over the raw scores it
        // consumes (w.net_r), P_0 taken of it; k_gemm_h / k_hf_score below read the corrected state from there
        P0Fuse pf{};
        pf.mode = 2; pf.net = w.net_r; pf.f1 = w.net_r; pf.seed = na.seed; pf.b_off = na.b_off; pf.draw = cf.draw_r;
        pf.mt = MaskTab{w.mfr, w.mfl, w.Kp, w.Ep}; pf.E = p.E; pf.cf = cf;
        if ((st = launch_p(pl, B, in->adj, in->rank2, w, stream, &pf))) return st;
        r2_in = w.net_r;
    } else if ((st = launch_p(pl, B, in->adj, in->rank2, w, stream))) return st;
    XaArgs xa{};
    xa.xX = xa.xA = in->x; xa.adjX = xa.adjA = in->adj; xa.flags = flags;
    xa.do_x = xa.do_a = 1; xa.mode = MODE_PRED;
    xa.pa_x = c[0].pa; xa.pb_x = c[0].pb; xa.pc_x = c[0].pc;
    xa.pa_a = c[1].pa; xa.pb_a = c[1].pb; xa.pc_a = c[1].pc;
    xa.out_x = out->x; xa.out_a = out->adj;
    xa.mean_x = mean ? mean->x : nullptr; xa.mean_a = mean ? mean->adj : nullptr;
    xa.cf = cf;
    if ((st = launch_xa(pl, B, xa, na, w, stream))) return st;
    if (p.is_cc && !fused && !ew1) {
        if ((st = launch_h(pl, B, r2_in, w, stream))) return st;
        RankEpi ep{};
        ep.mode = MODE_PRED; ep.pa = c[2].pa; ep.pb = c[2].pb; ep.pc = c[2].pc;
        ep.out = out->rank2; ep.mean = mean ? mean->rank2 : nullptr;
        if ((st = launch_hf(pl, B, r2_in, ep, na, w, stream))) return st;
    }
    return CCSD_OK;
}

// update half of one S4 step (k_s4_apply); the norms pass of the step is corrector_norms(step, 0, cur, cur)
static int s4_apply(ccsd_plan* pl, int B, int step, const ccsd_state_t* cur, const float* flags, const ccsd_noise_t* n1,
                    const ccsd_noise_t* n2, const ccsd_noise_t* n3, uint64_t seed, int64_t off, const float* sums,
                    ccsd_state_t* out, ccsd_state_t* mean, Workspace& w, void* stream) {
    const PlanD& p = pl->h;
    S4Args q{};
    LangArgs& a = q.a;
    a.x = cur->x; a.adj = cur->adj; a.r = cur->rank2;
    a.nx = w.net_x; a.nadj = w.net_adj; a.nr = w.net_r;
    a.ox = out->x; a.oadj = out->adj; a.orr = out->rank2;
    a.flags = flags; a.sums = sums;
    for (int t = 0; t < 3; ++t) {
        const ccsd_step_coef_t& c = pl->coef[(size_t)step * 3 + t];
        a.ss[t] = c.sscale; a.alpha[t] = c.alpha;
        q.m1[t] = c.m1; q.s1[t] = c.s1; q.d[t] = c.d; q.m2[t] = c.m2; q.s2[t] = c.s2;
    }
    a.snr = p.snr; a.seps = p.seps;
    a.B = B; a.N = p.N; a.F = p.F; a.E = p.E; a.K = p.K; a.is_cc = p.is_cc;
    q.mx = mean ? mean->x : nullptr; q.madj = mean ? mean->adj : nullptr; q.mr = mean ? mean->rank2 : nullptr;
    NoiseArgs na1 = make_noise(n1, seed, off, draw_base(pl, step, 0));
    NoiseArgs na2 = make_noise(n2, seed, off, draw_base(pl, step, 1));
    NoiseArgs na3 = make_noise(n3, seed, off, draw_base(pl, step, 2));
    const long long total = (long long)B * (p.N * p.F + p.N * p.N) + (p.is_cc ? (long long)B * ((p.E + 3) / 4) * p.K : 0);
    prof_mark(pl, KID_S4, stream);
    CCSD_LAUNCH(k_s4_apply, dim3(grid_for(total, 256)), dim3(CCSD_NTHREADS), 0, stream, q, na1, na2, na3,
                (const unsigned long long*)w.offbits, (const unsigned char*)pl->edges, (const unsigned long long*)pl->cells);
    prof_mark(pl, KID_S4, stream);
    LAUNCH_CHECK();
    return CCSD_OK;
}

static int check_step(const ccsd_plan* pl, int step) {
    if (step < 0 || step >= pl->cfg.diff_steps) return set_err(CCSD_ERR_INVALID, "step out of range");
    return CCSD_OK;
}

extern "C" int ccsd_corrector_norms(ccsd_plan_t* pl, int32_t B, int32_t step, int32_t corr_iter, const ccsd_state_t* base,
                                    const ccsd_state_t* cur, const float* flags, const ccsd_noise_t* noise, uint64_t seed,
                                    int64_t sample_offset, float* norm_sums, void* workspace, size_t ws_bytes, void* stream) {
    int st = check_common(pl, B, flags, workspace, ws_bytes);
    if (st || (st = check_step(pl, step)) || (st = check_state(pl, base, "base")) || (st = check_state(pl, cur, "cur"))) return st;
    if (!norm_sums) return set_err(CCSD_ERR_INVALID, "NULL norm_sums");
    Workspace w = carve_ws(pl, B, workspace);
    if ((st = launch_flagbits(pl, B, flags, w, stream))) return st;
    return corrector_norms(pl, B, step, corr_iter, base, cur, flags, noise, seed, sample_offset, norm_sums, w, stream);
}
extern "C" int ccsd_corrector_apply(ccsd_plan_t* pl, int32_t B, int32_t step, int32_t corr_iter, const ccsd_state_t* cur,
                                    const float* flags, const ccsd_noise_t* noise, uint64_t seed, int64_t sample_offset,
                                    const float* norm_sums, ccsd_state_t* out, void* workspace, size_t ws_bytes, void* stream) {
    int st = check_common(pl, B, flags, workspace, ws_bytes);
    if (st || (st = check_step(pl, step)) || (st = check_state(pl, cur, "cur")) || (st = check_state(pl, out, "out"))) return st;
    if (!norm_sums) return set_err(CCSD_ERR_INVALID, "NULL norm_sums");
    Workspace w = carve_ws(pl, B, workspace);
    if ((st = launch_flagbits(pl, B, flags, w, stream))) return st;
    return corrector_apply(pl, B, step, corr_iter, cur, flags, noise, seed, sample_offset, norm_sums, out, w, stream);
}
extern "C" int ccsd_predictor(ccsd_plan_t* pl, int32_t B, int32_t step, const ccsd_state_t* in, const float* flags,
                              const ccsd_noise_t* noise, uint64_t seed, int64_t sample_offset, ccsd_state_t* out,
                              ccsd_state_t* mean, void* workspace, size_t ws_bytes, void* stream) {
    int st = check_common(pl, B, flags, workspace, ws_bytes);
    if (!st && pl->cfg.predictor == CCSD_PRED_S4) return set_err(CCSD_ERR_INVALID, "S4 plans step with ccsd_corrector_norms + ccsd_s4_apply");
    if (st || (st = check_step(pl, step)) || (st = check_state(pl, in, "in")) || (st = check_state(pl, out, "out"))) return st;
    if (mean && (st = check_state(pl, mean, "mean"))) return st;
    if (in->x == out->x || in->adj == out->adj || (pl->h.is_cc && in->rank2 == out->rank2))
        return set_err(CCSD_ERR_INVALID, "predictor cannot run in place");
    Workspace w = carve_ws(pl, B, workspace);
    if ((st = launch_flagbits(pl, B, flags, w, stream))) return st;
    return predictor(pl, B, step, in, flags, noise, seed, sample_offset, out, mean, w, stream);
}

extern "C" int ccsd_s4_apply(ccsd_plan_t* pl, int32_t B, int32_t step, const ccsd_state_t* cur, const float* flags,
                             const ccsd_noise_t* noise1, const ccsd_noise_t* noise2, const ccsd_noise_t* noise3, uint64_t seed,
                             int64_t sample_offset, const float* norm_sums, ccsd_state_t* out, ccsd_state_t* mean,
                             void* workspace, size_t ws_bytes, void* stream) {
    int st = check_common(pl, B, flags, workspace, ws_bytes);
    if (st || (st = check_step(pl, step)) || (st = check_state(pl, cur, "cur")) || (st = check_state(pl, out, "out"))) return st;
    if (mean && (st = check_state(pl, mean, "mean"))) return st;
    if (pl->cfg.predictor != CCSD_PRED_S4) return set_err(CCSD_ERR_INVALID, "ccsd_s4_apply needs a plan created with CCSD_PRED_S4");
    if (!norm_sums) return set_err(CCSD_ERR_INVALID, "NULL norm_sums");
    Workspace w = carve_ws(pl, B, workspace);
    if ((st = launch_flagbits(pl, B, flags, w, stream))) return st;
    return s4_apply(pl, B, step, cur, flags, noise1, noise2, noise3, seed, sample_offset, norm_sums, out, mean, w, stream);
}

extern "C" int ccsd_sampler_run(ccsd_plan_t* pl, int32_t B, const float* flags, uint64_t seed, int64_t sample_offset,
                                int32_t first_step, int32_t last_step, ccsd_state_t* state, ccsd_state_t* scratch,
                                ccsd_state_t* result, float* traj, void* workspace, size_t ws_bytes, void* stream) {
    int st = check_common(pl, B, flags, workspace, ws_bytes);
    if (st || (st = check_state(pl, state, "state")) || (st = check_state(pl, scratch, "scratch")) ||
        (st = check_state(pl, result, "result"))) return st;
    if (first_step < 0 || last_step > pl->cfg.diff_steps || first_step >= last_step) return set_err(CCSD_ERR_INVALID, "bad step range");
    const bool s4 = pl->cfg.predictor == CCSD_PRED_S4;
    const bool lang = pl->cfg.corrector == CCSD_CORR_LANGEVIN && !s4;
    if (lang && pl->cfg.n_corr_steps != 1)
        return set_err(CCSD_ERR_UNSUPPORTED, "ccsd_sampler_run handles n_steps == 1; drive other values step by step");
    const PlanD& p = pl->h;
    Workspace w = carve_ws(pl, B, workspace);
    if ((st = launch_flagbits(pl, B, flags, w, stream))) return st;
    const size_t nx = (size_t)p.N * p.F, na = (size_t)p.N * p.N, nr = p.is_cc ? (size_t)p.E * p.K : 0;
    ccsd_state_t a = *state, b = *scratch;   // a = live buffer
    for (int step = first_step; step < last_step; ++step) {
        const bool lastone = step == last_step - 1;
        const bool want_mean = pl->cfg.denoise && (lastone || traj);
        if (s4) {   // scores + first draw + norm sums at the state, then the element-wise S4 update: a -> b, swap
            if ((st = corrector_norms(pl, B, step, 0, &a, &a, flags, nullptr, seed, sample_offset, w.sums, w, stream))) return st;
            if ((st = s4_apply(pl, B, step, &a, flags, nullptr, nullptr, nullptr, seed, sample_offset, w.sums, &b,
                               want_mean ? result : nullptr, w, stream))) return st;
            ccsd_state_t t = a; a = b; b = t;
        } else if (lang && fused_apply_ok(pl)) {
            // a -> [norms pass] ; [apply fused into the predictor kernels] -> b ; swap roles.  Merged k_r2 launches: the predictor's
            // k_r2 also runs the rank-2 side of the next step's norms pass on the block it has just produced (r2_done below)
            const bool merged = merge_ok(pl);
            const bool r2_done = merged && step > first_step;
            const bool merge_next = merged && !lastone;
            if ((st = corrector_norms(pl, B, step, 0, &a, &a, flags, nullptr, seed, sample_offset, w.sums, w, stream, /*keep_net=*/pl->fused_r2 != 0, r2_done))) return st;
            // u_1 = fr . Wcat_1 depends on the flags alone: the run's first (general) k_r2 launch has just written it; the masked launches
            // of the loop leave both copies alone (R2Args::masked), so the second buffer set gets its copy once
            if (merged && !r2_done && pl->opt_r2_masked && p.h_L > 1 && p.hl[0].mval.n == 1)
                RT_CHECK(rt_d2d_async(w.U1b, w.U1, (size_t)B * p.h_pw * 4, stream));
            if ((st = predictor(pl, B, step, &a, flags, nullptr, seed, sample_offset, &b, want_mean ? result : nullptr, w, stream, w.sums, merge_next))) return st;
            ccsd_state_t t = a; a = b; b = t;
        } else if (lang) {   // a -> (corrector) -> b -> (predictor) -> a
            if ((st = corrector_norms(pl, B, step, 0, &a, &a, flags, nullptr, seed, sample_offset, w.sums, w, stream))) return st;
            if ((st = corrector_apply(pl, B, step, 0, &a, flags, nullptr, seed, sample_offset, w.sums, &b, w, stream))) return st;
            if ((st = predictor(pl, B, step, &b, flags, nullptr, seed, sample_offset, &a, want_mean ? result : nullptr, w, stream))) return st;
        } else {      // a -> (predictor) -> b, then swap roles
            if ((st = predictor(pl, B, step, &a, flags, nullptr, seed, sample_offset, &b, want_mean ? result : nullptr, w, stream))) return st;
            ccsd_state_t t = a; a = b; b = t;
        }
        if (traj) {
            const ccsd_state_t* src = pl->cfg.denoise ? result : &a;
            float* slot = traj + (size_t)step * (nx + na + nr);
            RT_CHECK(rt_d2d_async(slot, src->x, nx * 4, stream));
            RT_CHECK(rt_d2d_async(slot + nx, src->adj, na * 4, stream));
            if (nr) RT_CHECK(rt_d2d_async(slot + nx + na, src->rank2, nr * 4, stream));
        }
    }
    if (a.x != state->x) {   // the live buffer ended up in `scratch`: bring the state home
        RT_CHECK(rt_d2d_async(state->x, a.x, (size_t)B * nx * 4, stream));
        RT_CHECK(rt_d2d_async(state->adj, a.adj, (size_t)B * na * 4, stream));
        if (nr) RT_CHECK(rt_d2d_async(state->rank2, a.rank2, (size_t)B * nr * 4, stream));
    }
    if (!pl->cfg.denoise) {
        RT_CHECK(rt_d2d_async(result->x, a.x, (size_t)B * nx * 4, stream));
        RT_CHECK(rt_d2d_async(result->adj, a.adj, (size_t)B * na * 4, stream));
        if (nr) RT_CHECK(rt_d2d_async(result->rank2, a.rank2, (size_t)B * nr * 4, stream));
    }
    return CCSD_OK;
}

extern "C" int ccsd_quantize(const float* in, int64_t n, float thr, int64_t* out, void* stream) {
    if (!in || !out || n < 0) return set_err(CCSD_ERR_INVALID, "bad argument");
    if (n == 0) return CCSD_OK;
    CCSD_LAUNCH(k_quantize, dim3(grid_for(n, 256)), dim3(CCSD_NTHREADS), 0, stream, in, (long long)n, thr, (long long*)out);
    LAUNCH_CHECK();
    return CCSD_OK;
}

extern "C" int ccsd_rank2_cells(const float* rank2, int32_t B, int32_t E, int64_t K, float thr, uint64_t* bits, int32_t* counts,
                                void* stream) {
    if (!rank2 || !bits || !counts || B < 1 || E < 1 || K < 1 || K > (1 << 24)) return set_err(CCSD_ERR_INVALID, "bad argument");
    CCSD_LAUNCH(k_rank2_cells, dim3(B), dim3(CCSD_NTHREADS), 0, stream, rank2, (int)E, (int)K, thr, (unsigned long long*)bits, (int*)counts);
    LAUNCH_CHECK();
    return CCSD_OK;
}

