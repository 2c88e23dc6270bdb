// ccsd_k_xa.h -- k_xa: ScoreNetworkX + ScoreNetworkA / ScoreNetworkA_CC for one graph per workgroup
// Part of the kernel source of libccsd_hip.so (see ccsd_kernels.h for the map).
#pragma once
#include "ccsd_baked_qm9.h"
#include "ccsd_baked_cs.h"
#include "ccsd_baked_z.h"
#include "ccsd_baked_enz.h"
#include "ccsd_rank2_common.h"

// ---------------------------------------------------------------------------------------------
// k_xa: ScoreNetworkX + ScoreNetworkA / ScoreNetworkA_CC for one graph per workgroup.
// ---------------------------------------------------------------------------------------------
#define XA_PLAIN 0
#define XA_HB 1
#define XA_GMH 2
#define XA_GEN 3          /* everything, selected at run time from the plan: both of the above together, conv = "MLP" */
#define XA_PLAIN9 4       /* XA_PLAIN with the qm9 geometry compiled in: N = 9, E = 36, F = 4, ldn = 16 (index arithmetic folds to constants) */
#define XA_BAKED9 7       /* XA_PLAIN9 with the WHOLE plan of the qm9_CC configuration (batch 1024) as a compile-time constant (ccsd_baked_qm9.h) */
#define XA_BAKED38 9      /* XA_PLAIN38 with the whole plan of the zinc250k configuration (graph-only, batch 256) baked (ccsd_baked_z.h) */
#define XA_BAKEDENZ 10    /* XA_GEN with the whole plan of the ENZYMES_small_CC configuration (S4 sampler, batch 64) baked (ccsd_baked_enz.h) */
#define XA_BAKED20 8      /* XA_PLAIN20 with the whole plan of the community_small_CC configuration (batch 512) baked (ccsd_baked_cs.h) */
#define XA_PLAIN20 5      /* XA_PLAIN, channel stack in HBM, community_small geometry: N = 20, E = 190, ldn = 24 */
#define XA_PLAIN38 6      /* XA_PLAIN, channel stack in HBM, zinc250k geometry: N = 38, E = 703, ldn = 40 */
// node count a variant has compiled in (0: run-time geometry); E = N (N - 1) / 2 and the node-row stride round_ld(N) follow
static constexpr int xa_geo_n(int var) { return (var == XA_PLAIN9 || var == XA_BAKED9) ? 9 : (var == XA_PLAIN20 || var == XA_BAKED20) ? 20 : (var == XA_PLAIN38 || var == XA_BAKED38) ? 38 : 0; }
// what a variant IS (which network features it carries), baked or not
static constexpr int xa_sem(int var) { return var == XA_BAKEDENZ ? XA_GEN : (var == XA_HB || var == XA_GMH || var == XA_GEN) ? var : XA_PLAIN; }
static constexpr int xa_geo_ld(int n) { return ((n + 7) / 8 * 8) % 32 == 0 ? (n + 7) / 8 * 8 + 8 : (n + 7) / 8 * 8; }   // == round_ld (ccsd_plan.h)
struct XaArgs {
    // inputs: the X-network and the A-network may see different (x, adj) when the Langevin
    // corrector runs more than one inner step (solver.py:759-784)
    const float* xX; const float* adjX;
    const float* xA; const float* adjA;
    const float* flags;
    const float* P0; const float* P1;     // hodge projections (B*E, wc_l)
    const float* U1; int p1_raw;          // p1_raw: P1 = (F o fr) Wcat_1 and U1 = fr Wcat_1 (B, wc_1); the kernel forms P_1 = fl (s P1 + b U1)
    int do_x, do_a;
    int mode;
    float ss_x, ss_a;                     // MODE_SCORE scaling
    float pa_x, pb_x, pc_x, pa_a, pb_a, pc_a;
    float* out_x; float* out_a;           // SCORE: scores; NORMS: raw nets; PRED: new state
    float* mean_x; float* mean_a;         // PRED, nullable
    float* norm2;                         // NORMS: [B][4] = |net_x|^2, |net_adj|^2, |z_x|^2, |z_adj|^2
    float* chan_ws;                       // GCH: [B][a_fdim][N*N] channel stack in the workspace
    const float* wp;                      // packed (zero-padded) chain-MLP weights
    const unsigned char* hpairs;          // (e, e2), e <= e2: unordered pairs of the dense hodge layer
    long long* dbg;
    CorrFuse cf;
    int prio_mode;                        // 0: priority = dispatch rank (default); 1: none; 2-4: diagnostic variants (CCSD_XA_PRIO)
    int stagger_mask, stagger_sleep;      // diagnostic (CCSD_XA_STAGGER): workgroups with (blockIdx.x & mask) != 0 start sleep x 64 cycles late
    // general hodge stack (h_L > 2 with a non-affine mlp_value; XA_GEN only, launch_xa drives it): the projections of the layers >= 1 come
    // materialised, Pd[l - 1] = R_l Wcat_l as [B * E][wc_l]; a launch with hdump_layer = s > 0 stops behind the dense hodge adjacency
    // H^s (the output of layer s - 1), writes it to hdump [B][hdump_stride] as [cout][E][E] and returns
    const float* Pd[CCSD_MAXHL + CCSD_MAXHLX - 1];
    int pdirect, hdump_layer, hdump_stride;
    float* hdump;
};

// fused Langevin corrector apply for x and adj held in LDS (same expressions as k_langevin_apply)
CCSD_DEV void corr_apply_xa(const CorrFuse& cf, const NoiseArgs& na, int b, int N, int F, float* s_x, float* s_adj,
                            const float* s_flags) {
    float c1x, c2x, c1a, c2a;
    corr_coef(cf, 0, &c1x, &c2x);
    corr_coef(cf, 1, &c1a, &c2a);
    NoiseArgs nc = na;
    nc.zx = nullptr; nc.zadj = nullptr; nc.draw_x = cf.draw_x; nc.draw_adj = cf.draw_adj;
    for (int t = threadIdx.x; t < N * F; t += blockDim.x) {
        const float z = raw_noise_x(nc, b, t, N * F) * s_flags[t / F];
        s_x[t] = fmaf(c2x, z, fmaf(c1x, cf.net_x[(size_t)b * N * F + t], s_x[t]));
    }
    for (int t = threadIdx.x; t < N * N; t += blockDim.x) {
        const int i = t / N, j = t % N;
        const float z = raw_noise_adj(nc, b, i, j, N) * s_flags[i] * s_flags[j];
        s_adj[t] = fmaf(c2a, z, fmaf(c1a, cf.net_adj[(size_t)b * N * N + t], s_adj[t]));
    }
}

// s_setprio with a run-time (wave-uniform) level 0..3
CCSD_DEV void wave_prio(int p) {
#ifndef CCSD_EMU
    if (p == 0) __builtin_amdgcn_s_setprio(0); else if (p == 1) __builtin_amdgcn_s_setprio(1);
    else if (p == 2) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(3);
#else
    (void)p;
#endif
}

// clamp(rowsum(A with unit diagonal), 1)^-1/2 for `nc` channels   (DenseGCNConv, layers.py:139-145)
CCSD_DEV void gcn_dinv(const float* a, float* dinv, int nc, int N) {
    const FastDiv dN(N);
    for (int t = threadIdx.x; t < nc * N; t += blockDim.x) {
        int c, i;
        dN.divmod(t, c, i);
        const float* r = a + c * N * N + i * N;
        float s = 0.f;
        for (int j = 0; j < N; ++j) s += (i == j) ? 1.f : r[j];
        dinv[t] = 1.0f / sqrtf(fmaxf(s, 1.f));
    }
}

// ScoreNetworkX (ScoreNetwork_X.py:102-132) by ONE wave, in four stages without a workgroup barrier inside them (PlanD::x_late):
// stage 0 = D^-1/2 + the first GCN layer, 1 = the other GCN layers, 2 = the head MLP chain, 3 = mask + epilogue.  k_xa runs them on
// its last wave inside the barrier intervals in which that wave has no tile -- the edge-MLP chains of AttentionLayers 0 and 1, the
// final MLP chain (E <= 48: three 16-pair tiles for four waves) and the A-network's epilogue -- so that the X-network costs the
// launch little more than its input load (barrier table: profiles/r02_b_k_xa_barrier_intervals.txt).  Inputs: the (corrected) x and
// adj in s_x / s_adj; everything else lives in the region at o_lx.  Returns this lane's (sum net^2, sum z^2) of the norms pass.
struct XLateOut { float n2, z2; };
CCSD_DEV void xlate_wave_sync() {
#ifndef CCSD_EMU
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");      // the wave's LDS writes before its (other lanes') LDS reads
    __builtin_amdgcn_wave_barrier();
#endif
}
template <bool NFIX = false>     // NFIX: the qm9 geometry as compile-time constants (k_xa<false, XA_PLAIN9>)
CCSD_DEV XLateOut xnet_late_stage(int stage, const PlanD& p, const float* __restrict__ w, const float* __restrict__ wp, float* sm,
                                  const XaArgs& xa, const NoiseArgs& na, int b) {
    XLateOut r{0.f, 0.f};
    const int N = NFIX ? 9 : p.N, F = NFIX ? 4 : p.F, ldn = NFIX ? 16 : p.ldn, H = p.x_nhid;
#ifdef CCSD_EMU
    const int lane = 0, wsz = 1;
#else
    // (opaque, as in mlp_chain_tile: otherwise the lane-dependent address arithmetic of every stage is hoisted to the head of the
    // enclosing region of the caller and spilled there)
    int lane = threadIdx.x & 63;
    asm volatile("" : "+v"(lane));
    const int wsz = 64;
#endif
    const float* s_flags = sm + p.o_flags;
    const float* s_x = sm + p.o_x;
    const float* s_adj = sm + p.o_adj;
    float* s_xcat = sm + p.o_lx;
    float* s_h1 = s_xcat + p.x_fdim * ldn;
    float* s_dinv = s_h1 + (F > 4 ? F : 4) * ldn;
    if (stage == 0) {
        for (int i = lane; i < N; i += wsz) {
            float sdeg = 0.f;
            for (int j = 0; j < N; ++j) sdeg += (i == j) ? 1.f : s_adj[i * N + j];
            s_dinv[i] = 1.0f / sqrtf(fmaxf(sdeg, 1.f));
        }
        for (int t = lane; t < N * F; t += wsz) { const int i = t / F, f = t - i * F; s_xcat[f * ldn + i] = s_x[t]; }
        xlate_wave_sync();
    }
    if (stage == 0 || stage == 1) {
        for (int l = stage ? 1 : 0; l < (stage ? p.x_depth : 1); ++l) {
            const int fin = l ? H : F;
            const float* src = s_xcat + (l ? (F + (l - 1) * H) : 0) * ldn;
            const float* W = w + p.x_gw[l];
            const float* B = w + p.x_gb[l];
            float* dst = s_xcat + (F + l * H) * ldn;
            for (int ct = 0; ct < (H + 15) >> 4; ++ct)
                gcn_tile_n<true>(src, ldn, fin, N, s_adj, s_dinv, 16 * ct, H,
                                 [&](int k, int col) { return W[k * H + col]; }, [&](int col) { return B[col]; },
                                 [&](int i, int col, float v) { dst[col * ldn + i] = tanh_f(v); });
            xlate_wave_sync();
        }
    } else if (stage == 2) {
        const MlpD& m = p.x_fin;
        auto epx = [&](int row, int f, float v) { s_h1[f * ldn + row] = v; };
        auto ident = [](int rr) { return rr; };
        for (int tile = 0; tile < (N + 15) >> 4; ++tile) {
            if (m.chain == 2) mlp_chain_tile<2, 3, 1>(m, wp, s_xcat, ldn, s_xcat, m.in, 16 * tile, N, ident, epx);
            else mlp_chain_tile<3, 6, 1>(m, wp, s_xcat, ldn, s_xcat, m.in, 16 * tile, N, ident, epx);
        }
        xlate_wave_sync();
    } else {
        for (int t = lane; t < N * F; t += wsz) {
            const int i = t / F, f = t - i * F;
            const float fl = s_flags[i];
            const float net = s_h1[f * ldn + i] * fl;                   // mask_x, graph_utils.py:37
            const size_t gi = (size_t)b * N * F + t;
            if (xa.mode == MODE_SCORE) {
                xa.out_x[gi] = xa.ss_x * net;
            } else {
                const float z = raw_noise_x(na, b, t, N * F) * fl;       // gen_noise(sym=False)
                if (xa.mode == MODE_NORMS) {
                    xa.out_x[gi] = net;
                    r.n2 = fmaf(net, net, r.n2);
                    r.z2 = fmaf(z, z, r.z2);
                } else {
                    const float mean = fmaf(xa.pa_x, s_x[t], xa.pb_x * net);
                    if (xa.mean_x) xa.mean_x[gi] = mean;
                    xa.out_x[gi] = fmaf(xa.pc_x, z, mean);
                }
            }
        }
    }
    return r;
}

// GCH: the channel stack (every AttentionLayer's adjacency channels, the final MLP's input) does not fit LDS
// (zinc250k, N = 38: 266 KB) and lives in a per-graph slab of the workspace instead; a workgroup's waves share one
// CU and its L1, so __syncthreads() orders those global accesses exactly like the LDS ones.
// Weights are read in place from L2.
// VAR: XA_PLAIN; XA_HB: the plan holds HodgeBaselineLayers (ScoreNetworkA_Base_CC); XA_GMH: the X-network is
// ScoreNetworkX_GMH; XA_GEN: both, and conv = "MLP" attention.  Separate instantiations keep those branches out of the register allocation of the headline variant.
template <bool GCH, int VAR>
// Launch bounds: 4 waves per SIMD (128 VGPRs) everywhere but in the run-time-plan large-graph variants (2: they need ~155 VGPRs).  The
// instances whose thread count is not compiled in may be launched with up to 1024 threads (launch_xa: more threads per graph when the
// batch leaves CUs with one or two workgroups -- every per-pair / per-tile loop of the kernel strides by the workgroup's thread count).
#define XA_4WAVES(G_, V_) (!(G_) || (V_) == XA_BAKED20 || (V_) == XA_BAKED38)
__global__ __launch_bounds__((VAR == XA_PLAIN9 || VAR == XA_BAKED9 || !XA_4WAVES(GCH, VAR)) ? 256 : 1024, XA_4WAVES(GCH, VAR) ? 4 : 2) void k_xa(const PlanD* __restrict__ plan, const float* __restrict__ w,
                                            const unsigned char* __restrict__ edges, XaArgs xa, NoiseArgs na) {
    CCSD_DYN_SMEM(sm);
    // XA_BAKED9: every plan field is a constant of the instance (the host selects it only for plans whose architecture bytes equal
    // the baked ones; the placeholder header of a tree without a bake leaves it reading the plan from memory like XA_PLAIN9)
    constexpr bool BAKED9 = VAR == XA_BAKED9 && CCSD_BAKED_QM9_SIZE == sizeof(PlanD);
    constexpr bool BAKED20 = VAR == XA_BAKED20 && CCSD_BAKED_CS_SIZE == sizeof(PlanD);
    constexpr bool BAKED38 = VAR == XA_BAKED38 && CCSD_BAKED_Z_SIZE == sizeof(PlanD);
    constexpr bool BAKEDENZ = VAR == XA_BAKEDENZ && CCSD_BAKED_ENZ_SIZE == sizeof(PlanD);
    constexpr bool BAKED = BAKED9 || BAKED20 || BAKED38 || BAKEDENZ;
    const PlanD& p = BAKED9 ? *reinterpret_cast<const PlanD*>(CCSD_BAKED_QM9_PLAN) : BAKED20 ? *reinterpret_cast<const PlanD*>(CCSD_BAKED_CS_PLAN)
                   : BAKED38 ? *reinterpret_cast<const PlanD*>(CCSD_BAKED_Z_PLAN) : BAKEDENZ ? *reinterpret_cast<const PlanD*>(CCSD_BAKED_ENZ_PLAN) : *plan;
    // unroll count of the AttentionLayer loop (ccsd_attn_stack.inc)
    constexpr int BAKED_UNROLL = BAKED9 ? CCSD_BAKED_QM9_A_L : BAKED20 ? CCSD_BAKED_CS_A_L : BAKED38 ? CCSD_BAKED_Z_A_L : BAKEDENZ ? CCSD_BAKED_ENZ_A_L : 1;
    constexpr int SEM = xa_sem(VAR);
    constexpr bool HB = SEM == XA_HB || SEM == XA_GEN, GMH = SEM == XA_GMH || SEM == XA_GEN, CONVMLP = SEM == XA_GEN;
    // XA_PLAIN9: a third of k_xa's vector instructions are 32-bit integer index arithmetic on strides the plan supplies at run
    // time (PMC, profiles/r03_c_phase_mix.txt); for the headline geometry they are compile-time constants (xa_variant() checks them)
    constexpr bool NFIX = VAR == XA_PLAIN9 || VAR == XA_BAKED9;          // everything fixed incl. F and the thread count
    constexpr int GN = xa_geo_n(VAR);                // node count compiled in (XA_PLAIN9 / XA_PLAIN20 / XA_PLAIN38), else 0
    const int N = GN ? GN : p.N, F = NFIX ? 4 : p.F, NN = N * N, E = GN ? GN * (GN - 1) / 2 : p.E, ldn = GN ? xa_geo_ld(GN) : p.ldn;
#ifndef CCSD_EMU
    if (NFIX) __builtin_assume(blockDim.x == 256);     // (launch_xa starts XA_PLAIN9 with 256 threads only)
#endif
    const int b = blockIdx.x, tid = threadIdx.x, nth = (NFIX && CCSD_NTHREADS != 1) ? 256 : blockDim.x;
    float* s_flags = sm + p.o_flags;
    float* s_x = sm + p.o_x;
    float* s_adj = sm + p.o_adj;
    float* s_dinv = sm + p.o_an;
    float* s_red = sm + (xa.do_a ? p.o_red : p.o_xcat);   // block reductions: a region that is idle at the end of the launch
    float* s_R = sm + p.o_c0;            // shared region: GCN scratch | MLP hidden activations | dense hodge layer
    const FastDiv dN(N), dNN(NN), dF(F), dE(E > 0 ? E : 1);
    const float* wp = xa.wp;
#ifdef CCSD_EMU
    const int wave_id = 0, n_waves = 1;
#else
    const int wave_id = wave_index(), n_waves = nth >> 6;
#endif

    stamp(xa.dbg, 0);
#if !defined(CCSD_EMU) && !defined(CCSD_BARRIER_PROF)
    if (xa.dbg && (tid & 63) == 0 && wave_id < 4) {      // diagnostic: where the hardware put this wave (HW_ID: SIMD, CU, SE, XCC)
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xa.dbg[(size_t)blockIdx.x * 64 + 26 + wave_id] = (long long)hw | ((long long)xcc << 32);
    }
#endif
    // Issue priority by dispatch age.  The workgroups b, b + 256, b + 512, b + 768 share a CU (tools/stamps.py reads HW_ID: 256 of
    // 256 CUs) and run in lockstep; the SIMDs arbitrate between their waves by priority, then AGE, so the youngest workgroup of a CU
    // loses every contended slot: measured lives 145 k / 145 k / 158 k / 177 k cycles, and the launch lasts as long as the slowest.
    // Raising the priority with the dispatch rank evens the four out.  (A hint: placement and timing never affect results.)
    const int prio_rank = (blockIdx.x >> 8) & 3;
    // prio_mode 0: the priority ROTATES with the phase -- rank r has level (r + phase) mod 4, so every workgroup is the favoured
    // one in a quarter of the phases and the four finish together; 2: static (level = rank); 1: none
    auto prio_phase = [&](int phase) {
        if (xa.prio_mode == 0) wave_prio((prio_rank + phase) & 3);
        else if (xa.prio_mode == 2 && phase == 0) wave_prio(prio_rank);
    };
    int prio_k = 1;
    prio_phase(0);
#ifndef CCSD_EMU
    if (xa.stagger_sleep && (blockIdx.x & xa.stagger_mask)) {
        for (int i = 0; i < xa.stagger_sleep; i += 100) __builtin_amdgcn_s_sleep(100);
    }
#endif
    for (int i = tid; i < N; i += nth) s_flags[i] = xa.flags[(size_t)b * N + i];
    float nx_net = 0.f, nx_z = 0.f, na_net = 0.f, na_z = 0.f;

    // ================= ScoreNetworkX (ScoreNetwork_X.py:102-132) =================
    // (see xnet_late_stage) both networks on the same inputs, plan permitting: only the input load + fused corrector stay here
    const bool x_late = SEM == XA_PLAIN && !GCH && p.x_late && xa.do_x && xa.do_a && xa.xA == xa.xX && xa.adjA == xa.adjX;
    if (x_late) {
        // Everything the launch reads from HBM at its start is requested in ONE batch and meets ONE barrier: the inputs, the raw
        // scores and norm sums of the fused corrector apply (same expressions as corr_apply_xa, the flags read from global memory
        // instead of waiting for their LDS copy), the edge table -- and the A-network's own staging (s_xcur, channel 0) is written
        // from the same registers.  (Was: load, barrier, apply, barrier, edge table, barrier, copy, barrier.)
        float c1x = 0.f, c2x = 0.f, c1a = 0.f, c2a = 0.f;
        NoiseArgs nc = na;
        const bool cfon = xa.cf.on != 0;
        if (cfon) {
            corr_coef(xa.cf, 0, &c1x, &c2x);
            corr_coef(xa.cf, 1, &c1a, &c2a);
            nc.zx = nullptr; nc.zadj = nullptr; nc.draw_x = xa.cf.draw_x; nc.draw_adj = xa.cf.draw_adj;
        }
        float* const xcur0 = sm + p.o_xcur;
        float* const chan0 = sm + p.o_chan;
        int* const edge0 = reinterpret_cast<int*>(sm + p.o_edge);
        const float* const fg = xa.flags + (size_t)b * N;
        for (int t = tid; t < N * F; t += nth) {
            float v = xa.xX[(size_t)b * N * F + t];
            int i, f;
            dF.divmod(t, i, f);
            if (cfon) {
                const float z = raw_noise_x(nc, b, t, N * F) * fg[i];
                v = fmaf(c2x, z, fmaf(c1x, xa.cf.net_x[(size_t)b * N * F + t], v));
            }
            s_x[t] = v;
            xcur0[f * ldn + i] = v;
        }
        for (int t = tid; t < NN; t += nth) {
            float v = xa.adjX[(size_t)b * NN + t];
            if (cfon) {
                const int i = t / N, j = t % N;
                const float z = raw_noise_adj(nc, b, i, j, N) * fg[i] * fg[j];
                v = fmaf(c2a, z, fmaf(c1a, xa.cf.net_adj[(size_t)b * NN + t], v));
            }
            s_adj[t] = v;
            chan0[t] = v;
        }
        for (int e = tid; e < E; e += nth) edge0[e] = ((int)edges[2 * e] << 8) | (int)edges[2 * e + 1];
        __syncthreads();
    } else
    if (xa.do_x) {
        float* s_xcat = sm + p.o_xcat;
        float* s_h1 = sm + p.o_h1;
        float* s_h2 = sm + p.o_h2;
        float* s_xw = s_R;
        const float* wx = w;
        for (int i = tid; i < N * F; i += nth) s_x[i] = xa.xX[(size_t)b * N * F + i];
        for (int i = tid; i < NN; i += nth) s_adj[i] = xa.adjX[(size_t)b * NN + i];
        if (xa.cf.on) { __syncthreads(); corr_apply_xa(xa.cf, na, b, N, F, s_x, s_adj, s_flags); }
        __syncthreads();
        const int H = p.x_nhid;
        if (GMH && p.x_gmh) {
            float* s_chan = GCH ? xa.chan_ws + (size_t)b * p.chan_rows * NN : sm + p.o_chan;
            // unordered pair e -> (i, j), i < j: the edge table, copied to LDS once (the global copy costs an L2 round trip
            // at the head of every per-pair phase)
            int* s_edge = reinterpret_cast<int*>(sm + p.o_edge);
            for (int e = tid; e < E; e += nth) s_edge[e] = ((int)edges[2 * e] << 8) | (int)edges[2 * e + 1];
            auto edge_i = [&](int e) { return s_edge[e] >> 8; };
            auto edge_j = [&](int e) { return s_edge[e] & 255; };
            auto pair_off = [&](int e) { const int v = s_edge[e]; return (v >> 8) * N + (v & 255); };
            float* s_att = sm + p.o_att;
            float* s_xcur = sm + p.o_xcur;
            float* s_xnext = sm + p.o_xnext;
            float* s_mch = sm + p.o_vcat;
            // ScoreNetworkX_GMH.forward_graph (ScoreNetwork_X.py:290-318): x_list = [x, tanh(AttentionLayer_k(...))...]
            for (int t = tid; t < N * F; t += nth) { int i, f; dF.divmod(t, i, f); s_xcat[f * ldn + i] = s_x[t]; s_xcur[f * ldn + i] = s_x[t]; }
            for (int i = tid; i < NN; i += nth) s_chan[i] = s_adj[i];
            __syncthreads();
            for (int c = 1; c < p.g_cinit; ++c) {                  // pow_tensor (graph_utils.py:285-292)
                for (int t = tid; t < NN; t += nth) {
                    int i, j;
                    dN.divmod(t, i, j);
                    float acc = 0.f;
                    for (int k = 0; k < N; ++k) acc = fmaf(s_chan[(c - 1) * NN + i * N + k], s_adj[k * N + j], acc);
                    s_chan[c * NN + t] = acc;
                }
                __syncthreads();
            }
            auto gmh_tap = [&](int l) {
                for (int t = tid; t < N * H; t += nth) {
                    int o, i;
                    dN.divmod(t, o, i);
                    const float v = tanh_f(s_xcur[o * ldn + i]);   // x = self.activation(x): feeds the next layer and x_list
                    s_xcur[o * ldn + i] = v;
                    s_xcat[(F + l * H + o) * ldn + i] = v;
                }
                __syncthreads();
            };
#define ATTN_LAYERS p.gl
#define ATTN_NL p.x_depth
#define ATTN_TAP(l) gmh_tap(l)
#define ATTN_MID(l) (void)0
#define ATTN_IDLE(l) (void)0
#include "ccsd_attn_stack.inc"
#undef ATTN_LAYERS
#undef ATTN_NL
#undef ATTN_TAP
#undef ATTN_MID
#undef ATTN_IDLE
        } else {
        gcn_dinv(s_adj, s_dinv, 1, N);
        for (int t = tid; t < N * F; t += nth) { int i, f; dF.divmod(t, i, f); s_xcat[f * ldn + i] = s_x[t]; }
        __syncthreads();
        for (int l = 0; l < p.x_depth; ++l) {
            const int fin = l ? H : F;
            const float* src = s_xcat + (l ? (F + (l - 1) * H) : 0) * ldn;
            const float* W = wx + p.x_gw[l];
            const float* B = wx + p.x_gb[l];
            float* dst = s_xcat + (F + l * H) * ldn;
            // tanh(DenseGCNConv(x, adj)) (ScoreNetwork_X.py:118-121): 16-column tiles over the waves
            for (int ct = wave_id; ct < (H + 15) >> 4; ct += n_waves)
                gcn_tile_n<true>(src, ldn, fin, N, s_adj, s_dinv, 16 * ct, H,
                           [&](int k, int col) { return W[k * H + col]; }, [&](int col) { return B[col]; },
                           [&](int i, int col, float v) { dst[col * ldn + i] = tanh_f(v); });
            __syncthreads();
        }
        }
        stamp(xa.dbg, 26);                                            // (X-network: conv layers done, head starts; overwrites the HW_ID diagnostic of wave 0)
        const MlpD& m = p.x_fin;
        if (m.chain) {
            auto epx = [&](int row, int f, float v) { s_h1[f * ldn + row] = v; };
            auto ident = [](int r) { return r; };
            if (m.chain == 2) mlp_chain<2, 3, 1>(m, wp, s_xcat, ldn, s_xcat, m.in, N, ident, epx);
            else mlp_chain<3, 6, 1>(m, wp, s_xcat, ldn, s_xcat, m.in, N, ident, epx);
        } else {
            block_linear<1>(s_h1, ldn, s_xcat, ldn, s_xcat, m.in, wx + m.w[0], wx + m.b[0], m.in, m.hid, N, mlp_wt(m, wp, 0));
            __syncthreads();
            block_linear<1>(s_h2, ldn, s_h1, ldn, s_h1, m.hid, wx + m.w[1], wx + m.b[1], m.hid, m.hid, N, mlp_wt(m, wp, 1));
            __syncthreads();
            block_linear<0>(s_h1, ldn, s_h2, ldn, s_h2, m.hid, wx + m.w[2], wx + m.b[2], m.hid, m.out, N, mlp_wt(m, wp, 2));
        }
        __syncthreads();
        stamp(xa.dbg, 27);                                            // (head done)
        for (int t = tid; t < N * F; t += nth) {
            int i, f;
            dF.divmod(t, i, f);
            const float fl = s_flags[i];
            const float net = s_h1[f * ldn + i] * fl;                   // mask_x, graph_utils.py:37
            const size_t gi = (size_t)b * N * F + t;
            if (xa.mode == MODE_SCORE) {
                xa.out_x[gi] = xa.ss_x * net;
            } else {
                const float z = raw_noise_x(na, b, t, N * F) * fl;       // gen_noise(sym=False)
                if (xa.mode == MODE_NORMS) {
                    xa.out_x[gi] = net;
                    nx_net = fmaf(net, net, nx_net);
                    nx_z = fmaf(z, z, nx_z);
                } else {
                    const float mean = fmaf(xa.pa_x, s_x[t], xa.pb_x * net);
                    if (xa.mean_x) xa.mean_x[gi] = mean;
                    xa.out_x[gi] = fmaf(xa.pc_x, z, mean);
                }
            }
        }
        __syncthreads();
    }

    stamp(xa.dbg, 1);
    // ================= ScoreNetworkA / ScoreNetworkA_CC =================
    if (xa.do_a) {
        float* s_chan = GCH ? xa.chan_ws + (size_t)b * p.chan_rows * NN : sm + p.o_chan;
        // unordered pair e -> (i, j), i < j: the edge table, copied to LDS once (the global copy costs an L2 round trip
        // at the head of every per-pair phase)
        int* s_edge = reinterpret_cast<int*>(sm + p.o_edge);
        if (!x_late) for (int e = tid; e < E; e += nth) s_edge[e] = ((int)edges[2 * e] << 8) | (int)edges[2 * e + 1];
        auto edge_i = [&](int e) { return s_edge[e] >> 8; };
        auto edge_j = [&](int e) { return s_edge[e] & 255; };
        auto pair_off = [&](int e) { const int v = s_edge[e]; return (v >> 8) * N + (v & 255); };
        float* s_att = sm + p.o_att;
        float* s_xcur = sm + p.o_xcur;
        float* s_xnext = sm + p.o_xnext;
        float* s_mch = sm + p.o_vcat;
        if (x_late) {
            // (staged, with the corrector applied, by the launch's first batch above; the barrier there covers it)
        } else if (xa.cf.on) {
            // fused corrector: the A-network sees the corrected (x, adj).  When the X-network phase of this launch has just
            // built them from the same inputs (predictor launches: xA == xX, adjA == adjX) they are still in LDS.
            const bool reuse = xa.do_x && xa.xA == xa.xX && xa.adjA == xa.adjX;
            __syncthreads();
            if (!reuse) {
                for (int i = tid; i < N * F; i += nth) s_x[i] = xa.xA[(size_t)b * N * F + i];
                for (int i = tid; i < NN; i += nth) s_adj[i] = xa.adjA[(size_t)b * NN + i];
                __syncthreads();
                corr_apply_xa(xa.cf, na, b, N, F, s_x, s_adj, s_flags);
                __syncthreads();
            }
            for (int t = tid; t < N * F; t += nth) { int i, f; dF.divmod(t, i, f); s_xcur[f * ldn + i] = s_x[t]; }
            for (int i = tid; i < NN; i += nth) s_chan[i] = s_adj[i];
        } else {
            for (int t = tid; t < N * F; t += nth) {
                const float v = xa.xA[(size_t)b * N * F + t];
                int i, f;
                dF.divmod(t, i, f);
                s_xcur[f * ldn + i] = v;
            }
            for (int i = tid; i < NN; i += nth) { const float v = xa.adjA[(size_t)b * NN + i]; s_adj[i] = v; s_chan[i] = v; }
        }
        if (!x_late) __syncthreads();
        // pow_tensor: channel c = channel(c-1) @ adj   (graph_utils.py:285-292)
        for (int c = 1; c < p.a_cinit; ++c) {
            for (int t = tid; t < NN; t += nth) {
                int i, j;
                dN.divmod(t, i, j);
                float acc = 0.f;
                for (int k = 0; k < N; ++k) acc = fmaf(s_chan[(c - 1) * NN + i * N + k], s_adj[k * N + j], acc);
                s_chan[c * NN + t] = acc;
            }
            __syncthreads();
        }
        // Hodge branch, early part: what depends only on the adjacency powers and the weights -- the mlp_attention weight blocks, the
        // diagonal hodge adjacency, P_1's per-edge factors -- is staged in regions of its own, by the threads t0, t0 + ts, ...:
        // by the last wave inside the third AttentionLayer's edge-MLP interval when that wave is idle there (the late
        // ScoreNetworkX's conditions + three layers), else by everybody right here
        auto hodge_early = [&](int t0, int ts) {
            float* s_hd = sm + p.o_hd;
            float* s_hw = sm + p.o_hw;
            float* s_p1c = s_hd + p.a_nch_hodge * E;
            const HodgeLayerD& h0 = p.hl[0];
            stage_mlp_blocks(p.hl[0].matt, w, s_hw, t0, ts);
            if (p.h_L > 1) stage_mlp_blocks(p.hl[1].matt, w, s_hw + p.hw_stride, t0, ts);
            if (SEM == XA_GEN) for (int l = 2; l < p.h_L; ++l) stage_mlp_blocks(ccsd_hl(p, l).matt, w, s_hw + l * p.hw_stride, t0, ts);
            // adj_to_hodgedual (cc_utils.py:1525-1536): diagonal hodge adjacency = upper triangle of the adjacency powers;
            // DenseHCNConv on a diagonal matrix (hodge_layers.py:185-193) is a row scaling
            for (int t = t0; t < p.a_cinit * E; t += ts) {
                int c, e;
                dE.divmod(t, c, e);
                s_hd[t] = s_chan[c * NN + pair_off(e)];
            }
            // P_1[e] = fl[e] (s[e] Q_1[e] + b u_1) with s[e] = sum_c w_c a_c[e] (linear mlp_value of the first layer,
            // hodge_attention.py:322-323; a_c = the diagonal hodge adjacency) when k_r2 delivered the raw factors Q_1, u_1:
            // per edge [fl s | fl b]
            if (p.h_L > 1 && xa.p1_raw) {
                const float mvb0 = w[h0.mval.b[0]];
                for (int e = t0; e < E; e += ts) {
                    float sc = 0.f;
                    for (int c = 0; c < h0.cin; ++c) sc = fmaf(w[h0.mval.w[0] + c], s_chan[c * NN + pair_off(e)], sc);
                    const float fl = s_flags[edge_i(e)] * s_flags[edge_j(e)];
                    s_p1c[e] = fl * sc; s_p1c[E + e] = fl * mvb0;
                }
            }
        };
        const bool hodge_on = SEM != XA_HB && p.h_L > 0;
        const bool hodge_idle = hodge_on && x_late && p.a_L >= 3;
        if (hodge_on && !hodge_idle) hodge_early(tid, nth);
#define ATTN_LAYERS p.al
#define ATTN_NL p.a_L
#define ATTN_TAP(l) prio_phase(prio_k++)
#define ATTN_MID(l) prio_phase(prio_k++)
#define ATTN_IDLE(l)                                                                                  \
    if (x_late && (l) < 2 && wave_id == n_waves - 1) (void)xnet_late_stage<NFIX>((l), p, w, wp, sm, xa, na, b);   \
    if (hodge_idle && (l) == 2 && wave_id == n_waves - 1) hodge_early(tid - wave_id * (nth / n_waves), nth / n_waves);
#include "ccsd_attn_stack.inc"
#undef ATTN_LAYERS
#undef ATTN_NL
#undef ATTN_TAP
#undef ATTN_MID
#undef ATTN_IDLE

        stamp(xa.dbg, 12);
        prio_phase(prio_k++);
        // ---- hodge branch of ScoreNetworkA_CC (ScoreNetwork_A_CC.py:295-316)
        if (SEM != XA_HB && p.h_L > 0) {
            float* s_hd = sm + p.o_hd;          // [hodge channel][E]: diagonals that reach the final MLP
            float* s_hq = sm + p.o_hq;          // [channel][E][2*adim]
            float* s_h1m = s_R;                 // [cout0][E][E] dense output of the first hodge layer
            const float kscale = (float)sqrt((double)p.K);  // hodge_attention.py:118,122: / sqrt(out_dim), out_dim = K
            const float rks = 1.0f / kscale;
            float* s_hw = sm + p.o_hw;         // zero-padded mlp_attention weight blocks of both hodge layers
            // (the mlp_attention weight blocks, the diagonal hodge adjacency and P_1's per-edge factors were staged behind pow_tensor)
            const HodgeLayerD& h0 = p.hl[0];
            const int qw0 = 2 * h0.adim;
            // row stride of the first layer's Q|K rows: odd (qw0 is even), so that the lanes of the dense pair loop -- consecutive
            // edges e2 -- read a Q|K element from 32 different banks instead of 4 (the general layer loop keeps the packed rows)
            const int ldq0 = SEM == XA_GEN ? qw0 : qw0 + 1;
            const FastDiv dqw0(qw0), dEqw0(E * qw0);
            const float* P0b = xa.P0 + (size_t)b * E * h0.wc;
            float* s_p1c = s_hd + p.a_nch_hodge * E;
            for (int t = tid; t < h0.cin * E * qw0; t += nth) {
                int c, r, e, d;
                dEqw0.divmod(t, c, r);
                dqw0.divmod(r, e, d);
                const float a = s_chan[c * NN + pair_off(e)];
                const float g = 1.0f / sqrtf(fmaxf(a, 1.f));
                s_hq[(c * E + e) * ldq0 + d] = fmaf(g * a * g, P0b[(size_t)e * h0.wc + c * qw0 + d], w[h0.bcat + c * qw0 + d]);
            }
            const bool w4_0 = mlp_maxw(h0.matt) <= 4 && h0.cin <= 4;
            __syncthreads();
            stamp(xa.dbg, 16);
            if (p.h_L == 1) {
                // only the diagonal is ever used (hodgedual_to_adj, cc_utils.py:1571)
                for (int e = tid; e < E; e += nth) {
                    float in[CCSD_SMALLW], out[CCSD_SMALLW];
#pragma unroll
                    for (int c = 0; c < CCSD_SMALLW; ++c) {
                        float sacc = 0.f;
                        if (c < h0.cin) {
                            const float* q = s_hq + (c * E + e) * ldq0;
                            sacc = attn_logits(q, q + h0.adim, h0.nchunk, h0.dsplit, rks) * (1.0f / (float)h0.nchunk);
                        }
                        in[c] = sacc;
                    }
                    small_mlp_lds<CCSD_SMALLW>(s_hw, h0.matt.n, in, out);   // mlp_attention -> mask -> tanh -> + transpose
                    const float fh = s_flags[edge_i(e)] * s_flags[edge_j(e)];
#pragma unroll
                    for (int o = 0; o < CCSD_SMALLW; ++o)
                        if (o < h0.cout) { const float tv = tanh_f(out[o] * fh * fh); s_hd[(p.a_cinit + o) * E + e] = tv + tv; }
                }
                __syncthreads();
            } else if (SEM != XA_GEN || p.h_L == 2) {
                // dense E x E attention of every channel, mlp_attention, mask, tanh, + transpose (hodge_attention.py:315-320):
                // one thread per unordered pair (e <= e2) from the pair table, both halves stored
                const HodgeLayerD& h1 = p.hl[1];
                const int qw1 = 2 * h1.adim;
                const float rnc0 = 1.0f / (float)h0.nchunk;
                const int npair = E * (E + 1) / 2;
                const float* P1b = xa.P1 + (size_t)b * E * h1.wc;   // [E][wc1] projections of the second layer (L2)
                auto p1_compose = [&](int e, float q, float u) {
                    return xa.p1_raw ? fmaf(s_p1c[e], q, s_p1c[E + e] * u) : q;
                };
                const int mtE = (E + 15) >> 4, ntq = (qw1 + 15) >> 4, ksE = (E + 3) >> 2;
#ifndef CCSD_EMU
                // The second layer's projection tasks are (channel, 16-column tile) x row tiles; with one (channel, column
                // tile) per wave its B operands are the same for every row tile: fetch them now, so the L2 latency hides
                // behind the dense attention below
                const bool pf_ok = h1.cin * ntq <= n_waves && ksE <= 16;
                float pfb[16], pfu = 0.f;          // (issued here, composed and consumed after the dense attention)
                const int pf_c = wave_id / ntq, pf_ct = wave_id % ntq;
                const int pf_l15 = tid & 15, pf_kq = (tid & 63) >> 4;
                if (pf_ok && wave_id < h1.cin * ntq) {
                    const int d = 16 * pf_ct + pf_l15, col = pf_c * qw1 + (d < qw1 ? d : qw1 - 1);
                    if (xa.p1_raw) pfu = xa.U1[(size_t)b * h1.wc + col];
#pragma unroll
                    for (int s0 = 0; s0 < 16; ++s0) {
                        const int k = 4 * s0 + pf_kq;
                        pfb[s0] = P1b[(k < E ? k : E - 1) * h1.wc + col];
                    }
                }
#endif
                // The loop is bound by the LATENCY of one thread's dependent chain (LDS reads -> dot products -> exp -> rcp ..., the same
                // 21 k cycles with one or with four workgroups on the CU), not by issue: with the channel count, the chunk count and the
                // chunk width known at compile time every logit of a pair is an independent chain in ONE basic block (the run-time
                // loops and the `c < cin` branches of the general form serialise them)
                auto dense_pairs = [&](auto CIN_, auto COUT_) {
                    constexpr int CIN = decltype(CIN_)::value, COUT = decltype(COUT_)::value;   // 0: general (run-time) form
                    if constexpr (CIN > 0) {
                        // nchunk == 2, dsplit == 2, adim == 4, one Linear in mlp_attention (checked by the caller): Q = row[0..3], K = row[4..7].
                        // PB pairs per thread and pass, all in one basic block: PB x CIN x 4 independent logit chains
                        constexpr int PB = 2;
                        for (int t0 = tid; t0 < npair; t0 += PB * nth) {
                            int pe[PB], pe2[PB];
#pragma unroll
                            for (int u = 0; u < PB; ++u) {
                                const int t = t0 + u * nth, tc = t < npair ? t : npair - 1;
                                const unsigned short pr = *reinterpret_cast<const unsigned short*>(xa.hpairs + 2 * tc);
                                pe[u] = pr & 0xff; pe2[u] = pr >> 8;
                            }
                            float outv[PB][COUT];
#pragma unroll
                            for (int u = 0; u < PB; ++u) {
                                float in[4] = {0.f, 0.f, 0.f, 0.f}, out[4];
#pragma unroll
                                for (int c = 0; c < CIN; ++c) {
                                    const float* q1 = s_hq + (c * E + pe[u]) * ldq0;
                                    const float* q2 = s_hq + (c * E + pe2[u]) * ldq0;
                                    float a1[8], a2[8];
#pragma unroll
                                    for (int j = 0; j < 8; ++j) { a1[j] = q1[j]; a2[j] = q2[j]; }
                                    // (same operation order as attn_logit_sum<2>: first product rounded, second fused)
                                    const float s1 = tanh_f(fmaf(a1[1], a2[5], a1[0] * a2[4]) * rks) + tanh_f(fmaf(a1[3], a2[7], a1[2] * a2[6]) * rks);
                                    const float s2 = tanh_f(fmaf(a2[1], a1[5], a2[0] * a1[4]) * rks) + tanh_f(fmaf(a2[3], a1[7], a2[2] * a1[6]) * rks);
                                    in[c] = (s1 * rnc0 + s2 * rnc0) * 0.5f;
                                }
                                small_mlp_lds<4>(s_hw, 1, in, out);
                                const float fh = s_flags[edge_i(pe[u])] * s_flags[edge_j(pe[u])];
                                const float fh2 = s_flags[edge_i(pe2[u])] * s_flags[edge_j(pe2[u])];
#pragma unroll
                                for (int o = 0; o < COUT; ++o) { const float tv = tanh_f(out[o] * fh * fh2); outv[u][o] = tv + tv; }   // symmetric inputs: h + h^T = 2h
                            }
#pragma unroll
                            for (int u = 0; u < PB; ++u)
                                if (t0 + u * nth < npair) {
                                    const int e = pe[u], e2 = pe2[u];
#pragma unroll
                                    for (int o = 0; o < COUT; ++o) {
                                        s_h1m[o * E * E + e * E + e2] = outv[u][o];
                                        s_h1m[o * E * E + e2 * E + e] = outv[u][o];
                                        if (e == e2) s_hd[(p.a_cinit + o) * E + e] = outv[u][o];
                                    }
                                }
                        }
                        return;
                    }
                    int pe_n = 0, pe2_n = 0;
                    if (tid < npair) { pe_n = xa.hpairs[2 * tid]; pe2_n = xa.hpairs[2 * tid + 1]; }
                    for (int t = tid; t < npair; t += nth) {
                        const int e = pe_n, e2 = pe2_n;
                        if (t + nth < npair) { pe_n = xa.hpairs[2 * (t + nth)]; pe2_n = xa.hpairs[2 * (t + nth) + 1]; }   // next pair: in flight
                        float in[CCSD_SMALLW], out[CCSD_SMALLW];
                        {
#pragma unroll
                            for (int c = 0; c < CCSD_SMALLW; ++c) {
                                float v = 0.f;
                                if (c < h0.cin) {
                                    const float* q1 = s_hq + (c * E + e) * ldq0;
                                    const float* q2 = s_hq + (c * E + e2) * ldq0;
                                    const float s1 = attn_logits(q1, q2 + h0.adim, h0.nchunk, h0.dsplit, rks);
                                    const float s2 = attn_logits(q2, q1 + h0.adim, h0.nchunk, h0.dsplit, rks);
                                    v = (s1 * rnc0 + s2 * rnc0) * 0.5f;
                                }
                                in[c] = v;
                            }
                            if (w4_0) small_mlp_lds<4>(s_hw, h0.matt.n, in, out); else small_mlp_lds<CCSD_SMALLW>(s_hw, h0.matt.n, in, out);
                        }
                        const float fh = s_flags[edge_i(e)] * s_flags[edge_j(e)];
                        const float fh2 = s_flags[edge_i(e2)] * s_flags[edge_j(e2)];
#pragma unroll
                        for (int o = 0; o < CCSD_SMALLW; ++o)
                            if (o < h0.cout) {
                                const float tv = tanh_f(out[o] * fh * fh2);   // inputs are exactly symmetric -> h + h^T = 2h
                                s_h1m[o * E * E + e * E + e2] = tv + tv;
                                s_h1m[o * E * E + e2 * E + e] = tv + tv;
                                if (e == e2) s_hd[(p.a_cinit + o) * E + e] = tv + tv;
                            }
                    }
                };
                if (SEM != XA_GEN && h0.cin == 2 && h0.cout == 4 && h0.adim == 4 && h0.nchunk == 2 && h0.dsplit == 2 && h0.matt.n == 1 && w4_0)
                    dense_pairs(std::integral_constant<int, 2>{}, std::integral_constant<int, 4>{});     // (the qm9_CC shape)
                else
                    dense_pairs(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
                __syncthreads();
                stamp(xa.dbg, 17);
                prio_phase(prio_k++);
                // second (last) HodgeAdjAttentionLayer: dense hodge adjacency, only the diagonal of its output
                float* s_deg = sm + p.o_deg;         // [cin1][E]
                for (int t = tid; t < h1.cin * E; t += nth) {
                    int c, e;
                    dE.divmod(t, c, e);
                    // degree = row sum; the matrix is symmetric, so walk the column: consecutive lanes hit consecutive banks
                    const float* Hc = s_h1m + (size_t)c * E * E + e;
                    float s0 = 0.f, s1 = 0.f;
                    int e2 = 0;
                    for (; e2 + 12 <= E; e2 += 12) {             // twelve LDS reads in flight per round (a 2-term body waits per pair)
                        float v[12];
#pragma unroll
                        for (int u = 0; u < 12; ++u) v[u] = Hc[(e2 + u) * E];
#pragma unroll
                        for (int u = 0; u < 12; u += 2) { s0 += v[u]; s1 += v[u + 1]; }
                    }
                    for (; e2 + 2 <= E; e2 += 2) { s0 += Hc[e2 * E]; s1 += Hc[(e2 + 1) * E]; }
                    if (e2 < E) s0 += Hc[e2 * E];
                    s_deg[t] = 1.0f / sqrtf(fmaxf(s0 + s1, 1.f));
                }
                __syncthreads();
                stamp(xa.dbg, 18);
                // Q|K of the dense layer on MFMA: per channel  Y = D H D P1_c  (hodge_layers.py:185-193), tile tasks
                // (channel, 16 rows of e) x 16 columns (2*adim <= 16 used) over the waves
                {
#ifndef CCSD_EMU
                    if (pf_ok) {
                        if (wave_id < h1.cin * ntq) {
                            typedef float f32x4 __attribute__((ext_vector_type(4)));
                            const float* Hc = s_h1m + (size_t)pf_c * E * E;
                            const float* dg = s_deg + pf_c * E;
                            const int d = 16 * pf_ct + pf_l15;
                            // (every LDS operand of a loop is requested before the first use: a uniform branch per step would put each
                            // read and its s_waitcnt into a basic block of its own -- 3.1 k cycles per row tile for nine MFMAs)
                            float bval[16], dgv[16], pc0[16], pc1[16];
#pragma unroll
                            for (int s0 = 0; s0 < 16; ++s0) {
                                const int k = 4 * s0 + pf_kq, kc = k < E ? k : E - 1;
                                dgv[s0] = dg[kc]; pc0[s0] = s_p1c[kc]; pc1[s0] = s_p1c[E + kc];
                            }
                            const bool raw = xa.p1_raw != 0;
#pragma unroll
                            for (int s0 = 0; s0 < 16; ++s0) {
                                const int k = 4 * s0 + pf_kq;
                                const float comp = raw ? fmaf(pc0[s0], pfb[s0], pc1[s0] * pfu) : pfb[s0];      // p1_compose
                                bval[s0] = (k < E && d < qw1) ? dgv[s0] * comp : 0.f;
                            }
                            const float bias = w[h1.bcat + pf_c * qw1 + (d < qw1 ? d : qw1 - 1)];
                            for (int rt = 0; rt < mtE; ++rt) {
                                const int e = 16 * rt + pf_l15, ec = e < E ? e : E - 1;
                                float hv[16], dgo[4];
#pragma unroll
                                for (int s0 = 0; s0 < 16; ++s0) {
                                    const int k = 4 * s0 + pf_kq;
                                    hv[s0] = Hc[ec * E + (k < E ? k : E - 1)];
                                }
#pragma unroll
                                for (int r = 0; r < 4; ++r) { const int eo = 16 * rt + 4 * pf_kq + r; dgo[r] = dg[eo < E ? eo : E - 1]; }
                                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                                for (int s0 = 0; s0 < 16; ++s0)
                                    if (s0 < ksE) {
                                        const int k = 4 * s0 + pf_kq;
                                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32((e < E && k < E) ? hv[s0] : 0.f, bval[s0], acc, 0, 0, 0);
                                    }
#pragma unroll
                                for (int r = 0; r < 4; ++r) {
                                    const int eo = 16 * rt + 4 * pf_kq + r;
                                    if (eo < E && d < qw1) s_hq[(pf_c * E + eo) * qw1 + d] = fmaf(dgo[r], acc[r], bias);
                                }
                            }
                        }
                    } else
#endif
                    for (int task = wave_id; task < h1.cin * mtE * ntq; task += n_waves) {
                        const int c = task / (mtE * ntq), rem = task % (mtE * ntq), rt = rem / ntq, ct = rem % ntq;
                        const float* Hc = s_h1m + (size_t)c * E * E;
                        const float* dg = s_deg + c * E;
                        wave_tile(16 * rt, 16 * ct, ksE,
                                  [&](int e, int k) { const float v = Hc[(e < E ? e : E - 1) * E + (k < E ? k : E - 1)]; return (e < E && k < E) ? v : 0.f; },
                                  [&](int k, int d) {
                                      const int kc = k < E ? k : E - 1, dc = d < qw1 ? d : qw1 - 1;
                                      const float v = dg[kc] * p1_compose(kc, P1b[kc * h1.wc + c * qw1 + dc],
                                                                          xa.p1_raw ? xa.U1[(size_t)b * h1.wc + c * qw1 + dc] : 0.f);
                                      return (k < E && d < qw1) ? v : 0.f;
                                  },
                                  [&](int e, int d, float acc) {
                                      if (e < E && d < qw1) s_hq[(c * E + e) * qw1 + d] = fmaf(dg[e], acc, w[h1.bcat + c * qw1 + d]);
                                  });
                    }
                }
                __syncthreads();
                stamp(xa.dbg, 19);
                const bool w4_1 = mlp_maxw(h1.matt) <= 4 && h1.cin <= 4;
#ifndef CCSD_EMU
                // one lane per (edge, channel), 8 lanes per edge: the channel's attention logit, then the edge's lane 0 collects the
                // eight and runs mlp_attention (one lane per edge walked the channels one after the other: 7.6 k cycles of a single
                // wave while the other three waited)
                const bool fast1 = h1.adim == 4 && h1.nchunk == 2 && h1.dsplit == 2;    // (the qm9_CC shape: both chunks in one block)
                // lanes per edge: 4 when the layer has at most four input channels (E = 36: 144 lanes, ONE pass of the 256 threads
                // instead of a second pass that only wave 0 walks), else 8
                const int lsh = h1.cin <= 4 ? 2 : 3, lpe = 1 << lsh;
                for (int t0 = tid; t0 < ((lpe * E + 63) & ~63); t0 += nth) {
                    const int e = t0 >> lsh, c = t0 & (lpe - 1);
                    float sacc = 0.f;
                    if (e < E && c < h1.cin) {
                        const float* q = s_hq + (c * E + e) * qw1;
                        if (fast1) {
                            const float4 qq = *reinterpret_cast<const float4*>(q), kk = *reinterpret_cast<const float4*>(q + 4);
                            sacc = (tanh_f(fmaf(qq.y, kk.y, qq.x * kk.x) * rks) + tanh_f(fmaf(qq.w, kk.w, qq.z * kk.z) * rks)) * 0.5f;
                        } else
                        sacc = attn_logits(q, q + h1.adim, h1.nchunk, h1.dsplit, rks) * (1.0f / (float)h1.nchunk);
                    }
                    float in[CCSD_SMALLW], out[CCSD_SMALLW];
#pragma unroll
                    for (int k = 0; k < CCSD_SMALLW; ++k) in[k] = k < lpe ? __shfl(sacc, ((tid & 63) & ~(lpe - 1)) + (k < lpe ? k : 0), 64) : 0.f;
                    if (c != 0 || e >= E) continue;
                    if (w4_1) small_mlp_lds<4>(s_hw + p.hw_stride, h1.matt.n, in, out); else small_mlp_lds<CCSD_SMALLW>(s_hw + p.hw_stride, h1.matt.n, in, out);
#else
                for (int e = tid; e < E; e += nth) {
                    float in[CCSD_SMALLW], out[CCSD_SMALLW];
                    for (int c = 0; c < CCSD_SMALLW; ++c) {
                        float sacc = 0.f;
                        if (c < h1.cin) {
                            const float* q = s_hq + (c * E + e) * qw1;
                            sacc = attn_logits(q, q + h1.adim, h1.nchunk, h1.dsplit, rks) * (1.0f / (float)h1.nchunk);
                        }
                        in[c] = sacc;
                    }
                    if (w4_1) small_mlp_lds<4>(s_hw + p.hw_stride, h1.matt.n, in, out); else small_mlp_lds<CCSD_SMALLW>(s_hw + p.hw_stride, h1.matt.n, in, out);
#endif
                    const float fh = s_flags[edge_i(e)] * s_flags[edge_j(e)];
#pragma unroll
                    for (int o = 0; o < CCSD_SMALLW; ++o)
                        if (o < h1.cout) { const float tv = tanh_f(out[o] * fh * fh); s_hd[(p.a_cinit + h0.cout + o) * E + e] = tv + tv; }
                }
                __syncthreads();
            } else if constexpr (SEM == XA_GEN) {
                // ---- three or four HodgeAdjAttentionLayers (num_linears_h == 1): the general layer loop.
                // Layer l >= 1 sees the dense hodge adjacency H^l [cin_l][E][E] (the previous layer's output) and the rank-2
                // features R_l = the l-fold mlp_value image of rank2 (hodge_attention.py:322-323):
                //   R_1 = fl fr (s o F + b_0),   R_{j+1} = fl fr (sum_c w^j_c H^j_c R_j + b_j)      (V_c = H_c R, :98)
                // Only P_l = R_l Wcat_l is ever needed.  With Q_l = (F o fr) Wcat_l and u_l = fr Wcat_l from k_r2:
                //   X_1 = fl (s Q_l + b_0 u_l),   X_{j+1} = fl (M_j X_j + b_j u_l),  M_j = sum_c w^j_c H^j_c,   P_l = X_l
                // -- a chain of l - 1 small (E x E)(E x wc_l) products on MFMA; the M_j of the layers passed stay in LDS.
                const int npair = E * (E + 1) / 2, mtE = (E + 15) >> 4, ksE = (E + 3) >> 2;
                float* bufA = s_h1m;
                float* bufB = sm + p.o_h2m;
                float* s_M = sm + p.o_hM;
                float* sX0 = sm + p.o_hX;
                float* s_deg = sm + p.o_deg;
                auto edge_fl = [&](int e) { return s_flags[edge_i(e)] * s_flags[edge_j(e)]; };
                // dense attention of a layer from its Q|K rows in s_hq -> Hout [cout][E][E] (+ diagonal -> s_hd)
                auto dense_att = [&](const HodgeLayerD& h, const float* hw, float* Hout, int hd0) {
                    const int qw = 2 * h.adim;
                    const float rnc = 1.0f / (float)h.nchunk;
                    for (int t = tid; t < npair; t += nth) {
                        const int e = xa.hpairs[2 * t], e2 = xa.hpairs[2 * t + 1];
                        float in[CCSD_SMALLW], out[CCSD_SMALLW];
#pragma unroll
                        for (int c = 0; c < CCSD_SMALLW; ++c) {
                            float v = 0.f;
                            if (c < h.cin) {
                                const float* q1 = s_hq + (c * E + e) * qw;
                                const float* q2 = s_hq + (c * E + e2) * qw;
                                const float s1 = attn_logits(q1, q2 + h.adim, h.nchunk, h.dsplit, rks);
                                const float s2 = attn_logits(q2, q1 + h.adim, h.nchunk, h.dsplit, rks);
                                v = (s1 * rnc + s2 * rnc) * 0.5f;
                            }
                            in[c] = v;
                        }
                        small_mlp_lds<CCSD_SMALLW>(hw, h.matt.n, in, out);
                        const float fh = edge_fl(e), fh2 = edge_fl(e2);
#pragma unroll
                        for (int o = 0; o < CCSD_SMALLW; ++o)
                            if (o < h.cout) {
                                const float tv = tanh_f(out[o] * fh * fh2);
                                Hout[o * E * E + e * E + e2] = tv + tv;
                                Hout[o * E * E + e2 * E + e] = tv + tv;
                                if (e == e2) s_hd[(hd0 + o) * E + e] = tv + tv;
                            }
                    }
                };
                dense_att(h0, s_hw, bufA, p.a_cinit);
                __syncthreads();
                auto dump_h = [&](const float* Hs, int cout) {
                    float* dst = xa.hdump + (size_t)b * xa.hdump_stride;
                    for (int t = tid; t < cout * E * E; t += nth) dst[t] = Hs[t];
                };
                if (xa.hdump_layer == 1) { dump_h(bufA, h0.cout); return; }
                float* Hin = bufA;
                float* Hnext = bufB;
                int hd0 = p.a_cinit + h0.cout;
                for (int l = 1; l < p.h_L; ++l) {
                    const HodgeLayerD& h = ccsd_hl(p, l);
                    const int qw = 2 * h.adim, wc = h.wc, ntw = (wc + 15) >> 4, ntq = (qw + 15) >> 4;
                    const float* Qb = xa.P1 + (size_t)b * E * p.h_pw + p.h_poff[l];     // rows of stride h_pw
                    const float* ub = xa.U1 + (size_t)b * p.h_pw + p.h_poff[l];
                    for (int t = tid; t < h.cin * E; t += nth) {                        // D^-1/2 of every input channel
                        int c, e;
                        dE.divmod(t, c, e);
                        const float* Hc = Hin + (size_t)c * E * E + e;                  // symmetric: walk the column
                        float sdeg = 0.f;
                        for (int e2 = 0; e2 < E; ++e2) sdeg += Hc[e2 * E];
                        s_deg[t] = 1.0f / sqrtf(fmaxf(sdeg, 1.f));
                    }
                    if (xa.pdirect) {                                                   // P_l as delivered (general hodge stack)
                        const float* Pl = xa.Pd[l - 1] + (size_t)b * E * wc;
                        for (int t = tid; t < E * wc; t += nth) sX0[t] = Pl[t];
                    }
                    if (!xa.pdirect && l + 1 < p.h_L)                                   // M_l, for the layers still to come
                        for (int t = tid; t < E * E; t += nth) {
                            float m = 0.f;
                            for (int c = 0; c < h.cin; ++c) m = fmaf(w[h.mval.w[0] + c], Hin[(size_t)c * E * E + t], m);
                            s_M[(l - 1) * E * E + t] = m;
                        }
                    if (!xa.pdirect)
                    for (int t = tid; t < E * wc; t += nth) {                           // X_1
                        const int e = t / wc, n = t - e * wc;
                        sX0[t] = fmaf(s_p1c[e], Qb[(size_t)e * p.h_pw + n], s_p1c[E + e] * ub[n]);
                    }
                    __syncthreads();
                    float* X = sX0;
                    float* Xn = sX0 + E * wc;
                    for (int j = 1; j < l && !xa.pdirect; ++j) {                        // X_{j+1} = fl (M_j X_j + b_j u_l)
                        const float* Mj = s_M + (j - 1) * E * E;
                        const float bj = w[ccsd_hl(p, j).mval.b[0]];
                        for (int task = wave_id; task < mtE * ntw; task += n_waves) {
                            const int rt = task / ntw, ct = task - rt * ntw;
                            wave_tile(16 * rt, 16 * ct, ksE,
                                      [&](int e, int k) { const float v = Mj[(e < E ? e : E - 1) * E + (k < E ? k : E - 1)]; return (e < E && k < E) ? v : 0.f; },
                                      [&](int k, int n) { const float v = X[(k < E ? k : E - 1) * wc + (n < wc ? n : wc - 1)]; return (k < E && n < wc) ? v : 0.f; },
                                      [&](int e, int n, float acc) { if (e < E && n < wc) Xn[e * wc + n] = edge_fl(e) * fmaf(bj, ub[n], acc); });
                        }
                        __syncthreads();
                        float* tx = X; X = Xn; Xn = tx;
                    }
                    // Q|K of the layer: per channel Y = D H D X_c + b  (hodge_layers.py:185-193)
                    for (int task = wave_id; task < h.cin * mtE * ntq; task += n_waves) {
                        const int c = task / (mtE * ntq), rem = task % (mtE * ntq), rt = rem / ntq, ct = rem % ntq;
                        const float* Hc = Hin + (size_t)c * E * E;
                        const float* dg = s_deg + c * E;
                        wave_tile(16 * rt, 16 * ct, ksE,
                                  [&](int e, int k) { const float v = Hc[(e < E ? e : E - 1) * E + (k < E ? k : E - 1)]; return (e < E && k < E) ? v : 0.f; },
                                  [&](int k, int d) {
                                      const int kc = k < E ? k : E - 1, dc = d < qw ? d : qw - 1;
                                      const float v = dg[kc] * X[kc * wc + c * qw + dc];
                                      return (k < E && d < qw) ? v : 0.f;
                                  },
                                  [&](int e, int d, float acc) {
                                      if (e < E && d < qw) s_hq[(c * E + e) * qw + d] = fmaf(dg[e], acc, w[h.bcat + c * qw + d]);
                                  });
                    }
                    __syncthreads();
                    const float* hw = s_hw + l * p.hw_stride;
                    if (l + 1 < p.h_L) {
                        dense_att(h, hw, Hnext, hd0);
                        if (xa.hdump_layer == l + 1) { __syncthreads(); dump_h(Hnext, h.cout); return; }
                        float* th = Hin; Hin = Hnext; Hnext = th;
                    } else {
                        // the last layer: only the diagonal of its output is used (hodgedual_to_adj, cc_utils.py:1571)
                        for (int e = tid; e < E; e += nth) {
                            float in[CCSD_SMALLW], out[CCSD_SMALLW];
#pragma unroll
                            for (int c = 0; c < CCSD_SMALLW; ++c) {
                                float sacc = 0.f;
                                if (c < h.cin) {
                                    const float* q = s_hq + (c * E + e) * qw;
                                    sacc = attn_logits(q, q + h.adim, h.nchunk, h.dsplit, rks) * (1.0f / (float)h.nchunk);
                                }
                                in[c] = sacc;
                            }
                            small_mlp_lds<CCSD_SMALLW>(hw, h.matt.n, in, out);
                            const float fh = edge_fl(e);
#pragma unroll
                            for (int o = 0; o < CCSD_SMALLW; ++o)
                                if (o < h.cout) { const float tv = tanh_f(out[o] * fh * fh); s_hd[(hd0 + o) * E + e] = tv + tv; }
                        }
                    }
                    __syncthreads();
                    hd0 += h.cout;
                }
            }
            stamp(xa.dbg, 20);
            // hodgedual_to_adj (cc_utils.py:1552-1588): scatter the diagonals behind the graph channels
            for (int t = tid; t < p.a_nch_hodge * E; t += nth) {
                int c, e;
                dE.divmod(t, c, e);
                const int i = edge_i(e), j = edge_j(e);
                const float v = s_hd[t];
                s_chan[(p.a_nch_graph + c) * NN + i * N + j] = v;
                s_chan[(p.a_nch_graph + c) * NN + j * N + i] = v;
            }
            __syncthreads();
        }

        // ---- hodge branch of ScoreNetworkA_Base_CC (ScoreNetwork_A_Base_CC.py:295-316): HodgeBaselineLayers
        // (hodge_layers.py:385-416) on the E x E hodge adjacency channels.  Their rank-2 outputs (bmm + mlp_rank2) never
        // reach the score and are not evaluated; of the last layer only the diagonal does (hodgedual_to_adj), so the dense
        // E x E output of the first layer is produced a chunk of rows at a time and consumed on the spot.
        if (HB && p.hb_L > 0) {
            float* s_hd = sm + p.o_hd;            // [hodge channel][E]: diagonals that reach the final MLP
            float* s_g = sm + p.o_hbg;            // [cin0][E][hid0]: hidden rows of the first layer's BaselineBlocks
            const HodgeBaseD& b0 = p.hb[0];
            const int hd0 = b0.hid;
            // mlp_hodge of both layers as zero-padded 16 x 16 blocks in LDS (broadcast reads instead of per-weight scalar loads)
            constexpr int HBS = CCSD_FW * CCSD_FW + CCSD_FW;
            float* s_mh = sm + p.o_hbw;
            stage_mlp_blocks_w<CCSD_FW>(b0.mh, w, s_mh);
            if (p.hb_L > 1) stage_mlp_blocks_w<CCSD_FW>(p.hb[1].mh, w, s_mh + CCSD_MAXLIN * HBS);
            const FastDiv dh0(hd0), dEh0(E * hd0);
            // adj_to_hodgedual: row e of input channel c is a_c[e] * onehot(e)  =>  hidden = elu(W1[:, e] * a_c[e] + b1)
            for (int t = tid; t < p.a_cinit * E; t += nth) {
                int c, e;
                dE.divmod(t, c, e);
                s_hd[t] = s_chan[c * NN + pair_off(e)];
            }
            for (int t = tid; t < b0.cin * E * hd0; t += nth) {
                int c, r, e, h;
                dEh0.divmod(t, c, r);
                dh0.divmod(r, e, h);
                const float* blk = w + b0.blk_base + c * b0.blk_stride;       // W1[hid][E] b1[hid] W2[E][hid] b2[E]
                const float a = s_chan[c * NN + pair_off(e)];
                s_g[t] = elu1(fmaf(blk[h * E + e], a, blk[hd0 * E + h]));
            }
            __syncthreads();
            stamp(xa.dbg, 16);
            // tanh(mlp_layer(H_c))[e][e2] from the hidden row of e (BaselineBlock.forward, hodge_layers.py:264)
            auto blockv = [&](int c, int e, int e2) {
                const float* blk = w + b0.blk_base + c * b0.blk_stride;
                const float* w2t = wp + b0.w2t + c * hd0 * E + e2;       // W2^T [hid][E]: lanes along e2 read consecutive floats
                const float* g = s_g + (c * E + e) * hd0;
                const float bias = blk[hd0 * E + hd0 + E * hd0 + e2];
                return tanh_f(dot_gl<true>(w2t, E, g, hd0) + bias);
            };
            // mlp_hodge over the symmetrised channels -> mask_hodge_adjs -> tanh -> + transpose, element (e, e2) of layer 0
            auto layer0 = [&](int e, int e2, float* out) {
                float in[CCSD_FW];
#pragma unroll
                for (int c = 0; c < CCSD_FW; ++c) in[c] = 0.f;
#pragma unroll
                for (int c = 0; c < CCSD_FW; ++c)
                    if (c < b0.cin) in[c] = e == e2 ? blockv(c, e, e) : (blockv(c, e, e2) + blockv(c, e2, e)) * 0.5f;
                small_mlp_ldsw<CCSD_FW>(s_mh, b0.mh.n, in, out);
                const float fh = s_flags[edge_i(e)] * s_flags[edge_j(e)] * s_flags[edge_i(e2)] * s_flags[edge_j(e2)];
#pragma unroll
                for (int o = 0; o < CCSD_FW; ++o) { const float tv = tanh_f(out[o] * fh); out[o] = tv + tv; }
            };
            if (p.hb_L == 1) {
                for (int e = tid; e < E; e += nth) {
                    float out[CCSD_FW];
                    layer0(e, e, out);
#pragma unroll
                    for (int o = 0; o < CCSD_FW; ++o)
                        if (o < b0.cout) s_hd[(p.a_cinit + o) * E + e] = out[o];
                }
                __syncthreads();
            } else {
                const HodgeBaseD& b1 = p.hb[1];
                const int hd1 = b1.hid, R = p.hb_rows;
                float* s_row = s_R;                           // [cout0][R][E]: rows r0 .. r0 + R of layer 0's output
                float* s_g2 = s_R + b0.cout * R * E;          // [R][cin1][hid1]: hidden rows of the second layer's blocks
                float* s_S = s_g2 + R * b1.cin * b1.hid;      // [cin0][R * E]: symmetrised block outputs = mlp_hodge's input rows
                const FastDiv dRE(R * E);
                float* s_d2 = sm + p.o_hbd;                   // [cin1][E]: diagonal of tanh(mlp_layer(H1_c))
                const FastDiv dch1(b1.cin * hd1);
                for (int r0 = 0; r0 < E; r0 += R) {
                    const int nr = (E - r0) < R ? (E - r0) : R;
                    if (r0 == 0) stamp(xa.dbg, 17);
                    if (b0.mh.chain) {
                        // (channel, pair) tasks fill mlp_hodge's input, then the MLP runs per 16-pair tile on MFMA
                        for (int t = tid; t < b0.cin * R * E; t += nth) {
                            int c, r, er, e2;
                            dRE.divmod(t, c, r);
                            dE.divmod(r, er, e2);
                            const int e = r0 + er;
                            if (er < nr) s_S[t] = e == e2 ? blockv(c, e, e) : (blockv(c, e, e2) + blockv(c, e2, e)) * 0.5f;
                        }
                        __syncthreads();
                        if (r0 == 0) stamp(xa.dbg, 18);
                        mlp_chain<1, 1, 1>(b0.mh, wp, s_S, R * E, s_S, b0.mh.in, nr * E, [](int r) { return r; },
                                           [&](int r, int o, float v) {
                                               int er, e2;
                                               dE.divmod(r, er, e2);
                                               const int e = r0 + er;
                                               const float fh = s_flags[edge_i(e)] * s_flags[edge_j(e)] * s_flags[edge_i(e2)] * s_flags[edge_j(e2)];
                                               const float tv = tanh_f(v * fh);
                                               s_row[(o * R + er) * E + e2] = tv + tv;
                                               if (e == e2) s_hd[(p.a_cinit + o) * E + e] = tv + tv;
                                           });
                    } else
                    for (int t = tid; t < nr * E; t += nth) {
                        int er, e2;
                        dE.divmod(t, er, e2);
                        const int e = r0 + er;
                        float out[CCSD_FW];
                        layer0(e, e2, out);
#pragma unroll
                        for (int o = 0; o < CCSD_FW; ++o)
                            if (o < b0.cout) {
                                s_row[(o * R + er) * E + e2] = out[o];
                                if (e == e2) s_hd[(p.a_cinit + o) * E + e] = out[o];
                            }
                    }
                    __syncthreads();
                    if (r0 == 0) stamp(xa.dbg, 19);
                    for (int t = tid; t < nr * b1.cin * hd1; t += nth) {
                        int er, r, c, h;
                        dch1.divmod(t, er, r);
                        c = r / hd1; h = r - c * hd1;
                        const float* blk = w + b1.blk_base + c * b1.blk_stride;
                        const float* w1t = wp + b1.w1t + c * E * hd1 + h;      // W1^T [E][hid]: lanes along h read consecutive floats
                        const float* row = s_row + (c * R + er) * E;
                        const float bias = blk[hd1 * E + h];
                        s_g2[t] = elu1(dot_gl<true>(w1t, hd1, row, E) + bias);
                    }
                    __syncthreads();
                    for (int t = tid; t < nr * b1.cin; t += nth) {
                        const int er = t / b1.cin, c = t - er * b1.cin, e = r0 + er;
                        const float* blk = w + b1.blk_base + c * b1.blk_stride;
                        const float* w2 = blk + hd1 * E + hd1 + e * hd1;
                        const float* g = s_g2 + t * hd1;
                        const float bias = blk[hd1 * E + hd1 + E * hd1 + e];
                        s_d2[c * E + e] = tanh_f(dot_gl<false>(w2, 1, g, hd1) + bias);
                    }
                    __syncthreads();
                    if (r0 == 0) stamp(xa.dbg, 20);
                }
                stamp(xa.dbg, 21);
                for (int e = tid; e < E; e += nth) {
                    float in[CCSD_FW], out[CCSD_FW];
#pragma unroll
                    for (int c = 0; c < CCSD_FW; ++c) in[c] = c < b1.cin ? s_d2[(c < b1.cin ? c : 0) * E + e] : 0.f;
                    small_mlp_ldsw<CCSD_FW>(s_mh + CCSD_MAXLIN * HBS, b1.mh.n, in, out);
                    const float fh = s_flags[edge_i(e)] * s_flags[edge_j(e)];
#pragma unroll
                    for (int o = 0; o < CCSD_FW; ++o)
                        if (o < b1.cout) { const float tv = tanh_f(out[o] * fh * fh); s_hd[(p.a_cinit + b0.cout + o) * E + e] = tv + tv; }
                }
                __syncthreads();
            }
            // hodgedual_to_adj (cc_utils.py:1552-1588): scatter the diagonals behind the graph channels
            for (int t = tid; t < p.a_nch_hodge * E; t += nth) {
                int c, e;
                dE.divmod(t, c, e);
                const int i = edge_i(e), j = edge_j(e);
                const float v = s_hd[t];
                s_chan[(p.a_nch_graph + c) * NN + i * N + j] = v;
                s_chan[(p.a_nch_graph + c) * NN + j * N + i] = v;
            }
            __syncthreads();
        }

        stamp(xa.dbg, 13);
        prio_phase(prio_k++);
        // ---- final MLP over every (i,j) on [graph channels | hodge channels]  (ScoreNetwork_A_CC.py:318-331)
        const MlpD& m = p.a_fin;
        const float* wf = w;
        const int fc = m.chain ? NN : p.pch, ldf = p.ldp;
        float* f0 = s_R;
        float* f1 = s_R + m.hid * ldf;
        for (int p0 = 0; p0 < NN; p0 += fc) {
            const int rows = (NN - p0) < fc ? (NN - p0) : fc;
            if (m.chain) {
                // symmetric input channels, masked diagonal: the E unordered pairs suffice (see the edge MLP above)
                auto epf = [&](int e, int f, float v) { (void)f; const int i = edge_i(e), j = edge_j(e); f0[i * N + j] = v; f0[j * N + i] = v; };
                // (the two widest shapes set the register demand of this whole region: the small-graph XA_PLAIN / XA_GMH variants leave them
                // to k_xa<false, XA_GEN> -- xa_variant() routes such plans there -- and run without them)
                constexpr bool WIDE = GCH || SEM == XA_GEN || SEM == XA_HB;
                if (m.chain == 3) mlp_chain<2, 4, 1>(m, wp, s_chan, NN, s_chan, m.in, E, pair_off, epf);
                else if (m.chain == 4) mlp_chain<3, 5, 1>(m, wp, s_chan, NN, s_chan, m.in, E, pair_off, epf);
                else if constexpr (WIDE) {
                    if (m.chain == 5) mlp_chain<3, 6, 1>(m, wp, s_chan, NN, s_chan, m.in, E, pair_off, epf);
                    else mlp_chain<4, 7, 1>(m, wp, s_chan, NN, s_chan, m.in, E, pair_off, epf);
                }
                if (x_late && wave_id == n_waves - 1) (void)xnet_late_stage<NFIX>(2, p, w, wp, sm, xa, na, b);
                stamp(xa.dbg, 11);
            } else {
                block_linear<1>(f0, ldf, s_chan + p0, NN, s_chan + p0, m.in, wf + m.w[0], wf + m.b[0], m.in, m.hid, rows, mlp_wt(m, wp, 0));
                __syncthreads();
                block_linear<1>(f1, ldf, f0, ldf, f0, m.hid, wf + m.w[1], wf + m.b[1], m.hid, m.hid, rows, mlp_wt(m, wp, 1));
                __syncthreads();
                block_linear<0>(f0, ldf, f1, ldf, f1, m.hid, wf + m.w[2], wf + m.b[2], m.hid, 1, rows, mlp_wt(m, wp, 2));
            }
            __syncthreads();
            stamp(xa.dbg, 15);
            int tid_e = tid;                       // opaque: the epilogue's per-thread index arithmetic stays here instead of being
#ifndef CCSD_EMU
            asm volatile("" : "+v"(tid_e));        // hoisted above the MLP (and spilled across it) as an invariant of the row-chunk loop
#endif
            for (int r = tid_e; r < rows; r += nth) {
                const int ij = p0 + r;
                int i, j;
                dN.divmod(ij, i, j);
                const float fm = s_flags[i] * s_flags[j];
                const float net = (i == j) ? 0.f : f0[r] * fm;         // * no-diag mask, then mask_adjs
                const size_t gi = (size_t)b * NN + ij;
                if (xa.mode == MODE_SCORE) {
                    xa.out_a[gi] = xa.ss_a * net;
                } else {
                    const float z = raw_noise_adj(na, b, i, j, N) * fm;   // gen_noise(sym=True), graph_utils.py:173-175
                    if (xa.mode == MODE_NORMS) {
                        xa.out_a[gi] = net;
                        na_net = fmaf(net, net, na_net);
                        na_z = fmaf(z, z, na_z);
                    } else {
                        const float mean = fmaf(xa.pa_a, s_adj[ij], xa.pb_a * net);
                        if (xa.mean_a) xa.mean_a[gi] = mean;
                        xa.out_a[gi] = fmaf(xa.pc_a, z, mean);
                    }
                }
            }
            if (x_late && wave_id == (n_waves > 1 ? n_waves - 2 : 0)) {      // (the wave with the least of the A-network's epilogue)
                const XLateOut xr = xnet_late_stage<NFIX>(3, p, w, wp, sm, xa, na, b);
                nx_net += xr.n2; nx_z += xr.z2;
            }
            __syncthreads();
        }
    }
    stamp(xa.dbg, 14);
#if defined(CCSD_BARRIER_PROF) && !defined(CCSD_EMU)
    if ((tid & 63) == 0 && blockIdx.x < 4096) {
        long long* r = g_bar + (size_t)blockIdx.x * 16 + (tid >> 6);
        if (xa.dbg) xa.dbg[(size_t)blockIdx.x * 64 + 26 + (tid >> 6)] = r[0];     // slots 26..29: barrier cycles of waves 0..3
        if (xa.dbg && tid == 0) xa.dbg[(size_t)blockIdx.x * 64 + 21] = r[8];      // slot 21: barriers passed
        // rows gridDim.x + 4 b + wave of the stamp buffer (the caller allocates gridDim.x + 256 rows): arrival clocks per barrier
        if (xa.dbg && blockIdx.x < 64 && (tid >> 6) < 4)
            for (int k = 0; k < 64; ++k)
                (xa.dbg - 32)[((size_t)gridDim.x + blockIdx.x * 4 + (tid >> 6)) * 64 + k] = k < r[8] ? g_arr[(blockIdx.x * 4 + (tid >> 6)) * 64 + k] : 0;
        if (xa.dbg && blockIdx.x == 0 && tid == 0) for (int k = 0; k < 16; ++k) (xa.dbg - 32)[((size_t)gridDim.x + 255) * 64 + k] = g_ct[k];
        r[0] = 0; r[8] = 0;
    }
#endif
    if (xa.mode == MODE_NORMS) {
        __syncthreads();
        float t4[4] = {nx_net, na_net, nx_z, na_z};
        block_sums<4>(t4, s_red);
        const float t0 = t4[0], t1 = t4[1], t2 = t4[2], t3 = t4[3];
        if (tid == 0) {
            float* o = xa.norm2 + (size_t)b * 4;
            if (xa.do_x) { o[0] = t0; o[2] = t2; }
            if (xa.do_a) { o[1] = t1; o[3] = t3; }
        }
    }
}
