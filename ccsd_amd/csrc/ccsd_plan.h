// ccsd_plan.h -- device-visible description of the three score networks (offsets into the flat
// weight blob, channel bookkeeping, LDS carve-up) and the host code that derives it from
// ccsd_config_t.  The canonical blob order written here is the contract with
// ccsd_amd/plan.py::pack_weights.
#pragma once
#include "../../include/ccsd_hip.h"
#include "ccsd_rt.h"

#define CCSD_MAXLIN 4     // linears per MLP
#define CCSD_MAXL 8       // attention layers / GCN depth
#define CCSD_MAXHL 2      // hodge layers in the hot part of the plan (PlanD::hl); CCSD_MAXHLX more behind it (PlanD::hlx)
#define CCSD_MAXHLX 6
#define CCSD_MAXFL 4      // HodgeNetworkLayers in ScoreNetworkF
#define CCSD_MAXCN 4      // channels [F, HF, H^2 F, H^3 F] of ScoreNetworkF's input (cnum, cc_utils.py:961-979)
#define CCSD_SMALLW 8     // widest per-thread MLP in the hodge branch
#define CCSD_FW 16        // widest per-thread MLP in ScoreNetworkF's general path (fused kernel; hodge baseline mlp_hodge)
#define CCSD_FWMAX 32     // ... in the tiled k_hf_score path

struct MlpD {
    int n, in, hid, out;
    int w[CCSD_MAXLIN], b[CCSD_MAXLIN];
    // register-resident chain (mlp_chain_tile): chain = index into CCSD_CHAIN_SHAPES (0: not chained, block_linear
    // path); offsets of the zero-padded copies Wp[16*to][16*ti], bp[16*to] of every linear in the packed weight buffer,
    // padded to the chosen shape's tile counts.
    // chain == 0 and pb[0] == CCSD_MLP_WT: pw[i] = offset of the TRANSPOSED zero-padded copy Wt_i[in][pad16(out)] of linear i in the packed
    // buffer -- block_linear's B operands: the 16 lanes of a load read 64 consecutive bytes instead of one word from each of 16 rows of W
    // (ScoreNetworkX's 107 -> 214 -> 214 -> 11 head on community_small: 174 k -> 81 k cycles of k_xa; PlanBuilder::transposed)
    int chain;
    int pw[CCSD_MAXLIN], pb[CCSD_MAXLIN];
};
// (input, hidden, output) widths in 16-feature tiles the chain is instantiated for (index 0 = none)
#define CCSD_NSHAPES 7
static const int CCSD_CHAIN_SHAPES[CCSD_NSHAPES][3] = {{0, 0, 0}, {1, 1, 1}, {2, 3, 1}, {2, 4, 1}, {3, 5, 1}, {3, 6, 1}, {4, 7, 1}};
#define CCSD_CHAIN_EDGE 0x02u      /* shapes allowed per call site (bit = shape index) */
#define CCSD_CHAIN_XFIN 0x24u
#define CCSD_CHAIN_AFIN 0x78u
static inline __host__ __device__ int pad16(int v) { return (v + 15) & ~15; }
#define CCSD_MLP_WT 0x5754
// dims of linear i of an MlpD
static inline __host__ __device__ int mlp_in(const MlpD& m, int i) { return i == 0 ? m.in : m.hid; }
static inline __host__ __device__ int mlp_out(const MlpD& m, int i) { return i == m.n - 1 ? m.out : m.hid; }

struct AttnLayerD {
    int cin, cout, fin, adim, fout, dsplit, nchunk;
    int ci0, co0;             // first input / output channel inside the channel stack
    int attn_base, attn_stride;  // per-channel block: Wq[fin][ad] bq Wk bk Wv[fin][fo] bv
    int conv_mlp;                // conv="MLP": Q / K blocks are W1[2ad][fin] b1[2ad] W2[ad][2ad] b2[ad] (attention.py:168-178)
    int w_lo, w_hi;              // blob range of this layer's weights (attention blocks, mlp, multi_channel)
    MlpD mlp, mc;
    // packed buffer (conv = "GCN"): per input channel [fin][cp] weights + [cp] biases with the Q | K | V columns side by side,
    // zero padded to cp = pad16(2 adim + fout) -- the B operand / bias of gcn_tile_multi without column-part arithmetic
    int qkvp, cp;
    // packed buffer: multi_channel's first Linear W0[hid][cin*fout] as MFMA A fragments, [hidden tile][channel][k-step][lane]
    // with lane = 16 (k & 3) + (unit & 15) and every channel's fout columns zero padded to mcs = ceil(fout / 4) k-steps:
    // one coalesced 256-byte load per k-step (ccsd_pack_mc)
    int mcp, mcs;
};
struct HodgeLayerD {
    int cin, cout, adim, dsplit, nchunk, wc;   // wc = cin*2*adim columns of Wcat
    int wcat, bcat;                             // Wcat[K][wc], bcat[wc]
    int wcatT;                                  // packed buffer: Wcat^T [pad16(wc)][Kp], zero padded (k_r2 float4 B operands)
    MlpD mval, matt;
};

// HodgeBaselineLayer of ScoreNetworkA_Base_CC (hodge_layers.py:273-416).  Per input channel a BaselineBlock: row-wise
// MLP E -> hid -> E (block = W1[hid][E] b1[hid] W2[E][hid] b2[E]); mlp_hodge mixes the channels per (e, e').  The layer's
// rank-2 output (bmm + mlp_rank2) never reaches the score (ScoreNetwork_A_Base_CC.py:303-321) and is not evaluated.
struct HodgeBaseD {
    int cin, cout, hid;
    int blk_base, blk_stride;
    int w2t, w1t;            // packed buffer: per channel W2^T [hid][E] / W1^T [E][hid] (coalesced reads along e / along hid)
    MlpD mh;
};

struct PlanD {
    int N, F, E, K, is_cc;
    float snr, seps;
    // ScoreNetworkX
    int x_depth, x_nhid, x_fdim, x_wlo, x_whi;
    int x_gw[CCSD_MAXL], x_gb[CCSD_MAXL];
    MlpD x_fin;
    // ScoreNetworkA(_CC)
    int a_L, a_cinit, a_is_cc, a_nch_graph, a_nch_hodge, a_fdim;
    AttnLayerD al[CCSD_MAXL];
    int h_L;
    HodgeLayerD hl[CCSD_MAXHL];
    MlpD a_fin;
    // ScoreNetworkF
    int f_L, f_cnum, f_hmask, f_fdim, f_affine;
    float f_alpha, f_beta, f_gamma;
    MlpD fl[CCSD_MAXFL];
    MlpD f_fin;
    // k_xa LDS carve-up (float offsets) and strides
    int ldn, ldp, pch;          // node-row stride; final-MLP chunk: stride, pairs per chunk
    int cg, pchp, ldpp, pw_pair; // attention channels per group; edge-MLP chunk: pairs, stride, widest hidden layer
    int o_flags, o_x, o_adj, o_an, o_xw, o_qkv, o_tmp, o_xcat, o_h1, o_h2, o_chan, o_att, o_xcur, o_xnext,
        o_vcat, o_c0, o_c1, o_acoef, o_hq, o_hatt, o_h1m, o_hd, o_red;
    int xa_lds_floats;
    int o_wst, wst_floats;      // weight staging buffer (0 floats: weights are read in place)
    int o_hw, hw_stride;        // zero-padded hodge mlp_attention weight blocks (stride between the two layers)
    int o_deg;                  // degree scratch of the dense hodge layer
    int o_edge;                 // LDS copy of the edge table (E ints)
    int x_lds_floats;           // LDS of an X-network-only launch (the ScoreNetworkX phase's regions)
    int chan_global;            // 1: the channel stack [a_fdim][N*N] lives in the HBM workspace (large graphs), not in LDS
    // ---- variants off the headline path, kept behind the fields every launch reads (scalar-cache footprint)
    int chan_rows;               // rows of the channel stack: max(a_fdim, g_nch)
    int f_blk;                   // ScoreNetworkF's general path from zero-padded 8x8 blocks behind the weight blob (-1: none)
    int hb_L;                     // ScoreNetworkA_Base_CC: HodgeBaselineLayers (0 otherwise)
    HodgeBaseD hb[CCSD_MAXHL];
    int o_hbw;                    // k_xa LDS: mlp_hodge weight blocks of both layers (2 * CCSD_MAXLIN * (16*16+16) floats)
    int o_hbg, o_hbd, hb_rows;    // k_xa LDS: hidden rows of layer 0 [cin][E][hid]; diagonals of layer 1's blocks [cin][E]; rows per chunk
    // HodgeAdjAttentionLayers 2.. (num_layers_h > 2; k_xa<., XA_GEN>), and the layout of the projections k_r2 hands over for
    // layers >= 1: P1 = [E][h_pw] with layer l's wc_l columns at h_poff[l] (each earlier layer padded to 16), U1 = [h_pw]
    HodgeLayerD hlx[CCSD_MAXHLX];
    int h_pw, h_poff[CCSD_MAXHL + CCSD_MAXHLX];
    int geo_off;                  // 1: CCSD_NO_GEO set when the plan was created -- never the compile-time-geometry instances; 2: CCSD_NO_BAKE -- never the baked-plan ones (A/B, parity tests)
    int x_late, o_lx;             // ScoreNetworkX in the idle wave of the A-network's MLP-chain intervals (k_xa); its own LDS region
    int o_h2m, o_hM, o_hX;        // k_xa LDS (h_L > 2): second dense hodge buffer; M_j = sum_c w_c H_c of layers 1..h_L-2; two [E][wc] buffers
    // ScoreNetworkX_GMH (x_gmh = 1): x_depth AttentionLayers gl[] on g_cinit adjacency powers, g_nch channels in all
    int x_gmh, g_cinit, g_nch;
    AttnLayerD gl[CCSD_MAXL];
    float f_betas[CCSD_MAXCN];    // affine ScoreNetworkF with cnum > 2: coefficient of H^j F, j = 1 .. cnum - 1 (f_betas[1] == f_beta)
};
static inline __host__ __device__ const HodgeLayerD& ccsd_hl(const PlanD& p, int l) { return l < CCSD_MAXHL ? p.hl[l] : p.hlx[l - CCSD_MAXHL]; }
static inline HodgeLayerD& ccsd_hl_mut(PlanD& p, int l) { return l < CCSD_MAXHL ? p.hl[l] : p.hlx[l - CCSD_MAXHL]; }

#ifndef CCSD_DEVICE_ONLY
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>

// The plan without its weight-derived fields (the affine fold of ScoreNetworkF) and without the sampler's run-time scalars (snr,
// scale_eps: the kernels receive them as launch arguments -- LangArgs / CorrFuse -- and never read them from the plan): what the
// ARCHITECTURE alone determines (network hyper-parameters, geometry, the LDS layout the batch bucket selects).  A kernel instance
// with a BAKED plan (ccsd_baked_*.h) serves exactly the plans whose architecture bytes equal the baked ones -- whatever snr /
// scale_eps / predictor the sampling configuration names.
static inline void ccsd_plan_arch_bytes(const PlanD& p, unsigned char* out) {
    PlanD q;
    memcpy(&q, &p, sizeof(PlanD));
    q.snr = q.seps = 0.f;
    q.f_alpha = q.f_beta = q.f_gamma = 0.f;
    for (int j = 0; j < CCSD_MAXCN; ++j) q.f_betas[j] = 0.f;
    memcpy(out, &q, sizeof(PlanD));
}

static inline int64_t ccsd_comb(int n, int k) {
    if (k < 0 || k > n) return 0;
    long double r = 1;
    for (int i = 1; i <= k; ++i) r = r * (n - k + i) / i;
    return (int64_t)(r + 0.5L);
}
static inline void ccsd_dims(const ccsd_config_t* c, int* E, int64_t* K) {
    *E = c->N * (c->N - 1) / 2;
    int64_t k = 0;
    if (c->is_cc)
        for (int d = c->d_min; d <= c->d_max; ++d) k += ccsd_comb(c->N, d);
    *K = k;
}

struct PlanBuilder {
    int cur = 0;
    int pcur = 0;     // packed (chain) weight buffer
    // reserve padded copies of an MLP's linears for mlp_chain_tile
    void chainify(MlpD& m, unsigned allowed);
    void transposed(MlpD& m) {          // (non-chained MLPs that run through block_linear; CCSD_NO_MLP_WT: diagnostic)
        if (m.chain || getenv("CCSD_NO_MLP_WT")) return;
        for (int i = 0; i < m.n; ++i) { m.pw[i] = pcur; pcur += (i == 0 ? m.in : m.hid) * (((i == m.n - 1 ? m.out : m.hid) + 15) & ~15); }
        m.pb[0] = CCSD_MLP_WT;
    }
    std::string err;
    int status = CCSD_OK;
    int take(int64_t n) {
        int o = cur;
        cur += (int)n;
        return o;
    }
    void fail(int st, const std::string& m) {
        if (status == CCSD_OK) { status = st; err = m; }
    }
    MlpD mlp(int n, int in, int hid, int out) {
        MlpD m{};
        m.n = n; m.in = in; m.hid = hid; m.out = out;
        if (n < 1 || n > CCSD_MAXLIN) { fail(CCSD_ERR_UNSUPPORTED, "MLP with more than 4 (or fewer than 1) linears"); m.n = 1; }
        for (int i = 0; i < m.n; ++i) {
            m.w[i] = take((int64_t)mlp_out(m, i) * mlp_in(m, i));
            m.b[i] = take(mlp_out(m, i));
        }
        return m;
    }
};

inline void PlanBuilder::chainify(MlpD& m, unsigned allowed) {
    m.chain = 0;
    if (getenv("CCSD_NO_CHAIN")) return;
    const int ni = pad16(m.in) / 16, nh = m.n > 1 ? pad16(m.hid) / 16 : 1, no = pad16(m.out) / 16;
    for (int sidx = 1; sidx < CCSD_NSHAPES && !m.chain; ++sidx) {
        const int* sh = CCSD_CHAIN_SHAPES[sidx];
        if (((allowed >> sidx) & 1u) && sh[0] >= ni && sh[1] >= nh && sh[2] >= no) m.chain = sidx;
    }
    if (!m.chain) return;
    const int* sh = CCSD_CHAIN_SHAPES[m.chain];
    for (int i = 0; i < m.n; ++i) {
        const int ip = 16 * (i == 0 ? sh[0] : sh[1]), op = 16 * (i == m.n - 1 ? sh[2] : sh[1]);
        m.pw[i] = pcur; pcur += op * ip;
        m.pb[i] = pcur; pcur += op;
    }
}
// Position of W[o][k] inside a chain linear's packed block: MFMA A-fragment order [out tile][in tile][lane][4] with
// lane = 16 ((k & 15) >> 2) + (o & 15) -- the float4 a lane feeds to four consecutive k-steps; one wave-level load is 1 KB
// contiguous (eight full cache lines; the row-major copy cost sixteen half-used ones per load)
static inline __host__ __device__ int ccsd_chain_widx(int o, int k, int ip) {
    return (((o >> 4) * (ip >> 4) + (k >> 4)) * 64 + (((k & 15) >> 2) << 4) + (o & 15)) * 4 + (k & 3);
}
// zero-padded copies of a chain MLP's linears (torch layout [out][in]) into the packed buffer
static inline void ccsd_pack_mlp(const MlpD& m, const float* w, float* packed) {
    if (!m.chain && m.pb[0] == CCSD_MLP_WT) {       // transposed copies (block_linear)
        for (int i = 0; i < m.n; ++i) {
            const int in = mlp_in(m, i), out = mlp_out(m, i), op = pad16(out);
            for (int k = 0; k < in; ++k)
                for (int o = 0; o < out; ++o) packed[(size_t)m.pw[i] + (size_t)k * op + o] = w[(size_t)m.w[i] + (size_t)o * in + k];
        }
    }
    if (!m.chain) return;
    const int* sh = CCSD_CHAIN_SHAPES[m.chain];
    for (int i = 0; i < m.n; ++i) {
        const int in = i == 0 ? m.in : m.hid, out = i == m.n - 1 ? m.out : m.hid, ip = 16 * (i == 0 ? sh[0] : sh[1]);
        for (int o = 0; o < out; ++o) {
            for (int k = 0; k < in; ++k) packed[m.pw[i] + ccsd_chain_widx(o, k, ip)] = w[m.w[i] + o * in + k];
            packed[m.pb[i] + o] = w[m.b[i] + o];
        }
    }
}

// multi_channel's first Linear in MFMA A-fragment order, zero padded (AttnLayerD::mcp)
static inline void ccsd_pack_mc(const AttnLayerD& a, const float* w, float* packed) {
    const int hid = a.mc.hid, nht = pad16(hid) >> 4;
    for (int ht = 0; ht < nht; ++ht)
        for (int c = 0; c < a.cin; ++c)
            for (int st = 0; st < a.mcs; ++st)
                for (int lane = 0; lane < 64; ++lane) {
                    const int hh = 16 * ht + (lane & 15), o = 4 * st + (lane >> 4);
                    packed[a.mcp + (size_t)((ht * a.cin + c) * a.mcs + st) * 64 + lane] =
                        (hh < hid && o < a.fout) ? w[a.mc.w[0] + (size_t)hh * a.mc.in + c * a.fout + o] : 0.f;
                }
}
// Q | K | V weights of a GCN-conv AttentionLayer side by side, zero padded (AttnLayerD::qkvp)
static inline void ccsd_pack_qkv(const AttnLayerD& a, const float* w, float* packed) {
    if (a.conv_mlp) return;
    const int ad = a.adim, fo = a.fout, fi = a.fin, cp = a.cp;
    for (int c = 0; c < a.cin; ++c) {
        const float* blk = w + a.attn_base + (size_t)c * a.attn_stride;      // Wq[fi][ad] bq[ad] Wk[fi][ad] bk[ad] Wv[fi][fo] bv[fo]
        float* dst = packed + a.qkvp + (size_t)c * (fi * cp + cp);
        for (int part = 0; part < 3; ++part) {
            const int ow = part == 2 ? fo : ad, c0 = part * ad;
            const float* wp_ = blk + part * (fi * ad + ad);
            for (int k = 0; k < fi; ++k)
                for (int o = 0; o < ow; ++o) dst[k * cp + c0 + o] = wp_[k * ow + o];
            for (int o = 0; o < ow; ++o) dst[fi * cp + c0 + o] = wp_[fi * ow + o];
        }
    }
}

static inline int round_ld(int rows) {  // node-row stride of the feature-major LDS arrays: multiple of 8, never a multiple of 32
    int r = (rows + 7) / 8 * 8;          // (== 16 mod 32 is conflict-free for the MFMA A-fragment reads, 8 / 24 mod 32 two-way)
    if (r % 32 == 0) r += 8;
    return r;
}

// Fills `p` (except the affine fold, which needs the weights) and returns the blob size.
static inline size_t ccsd_build_plan(const ccsd_config_t* c, PlanD* p, PlanBuilder& pb) {
    memset(p, 0, sizeof(*p));
    if (!c || c->abi_version != CCSD_ABI_VERSION) { pb.fail(CCSD_ERR_INVALID, "abi_version mismatch"); return 0; }
    if (c->N < 2 || c->N > 64 || c->F < 1) { pb.fail(CCSD_ERR_UNSUPPORTED, "need 2 <= N <= 64 and F >= 1"); return 0; }
    int E; int64_t K64;
    ccsd_dims(c, &E, &K64);
    if (c->is_cc && (c->d_min < 1 || c->d_max < c->d_min || c->d_max > c->N)) { pb.fail(CCSD_ERR_INVALID, "bad d_min/d_max"); return 0; }
    if (K64 > (1 << 24)) { pb.fail(CCSD_ERR_UNSUPPORTED, "rank-2 width K too large"); return 0; }
    p->N = c->N; p->F = c->F; p->E = E; p->K = (int)K64; p->is_cc = c->is_cc;
    p->snr = c->snr; p->seps = c->scale_eps;
    const int N = c->N, F = c->F, K = (int)K64;
    // ---- ScoreNetworkX
    if (c->x_depth < 1 || c->x_depth > CCSD_MAXL) { pb.fail(CCSD_ERR_UNSUPPORTED, "x_depth out of range"); return 0; }
    p->x_depth = c->x_depth; p->x_nhid = c->x_nhid; p->x_fdim = F + c->x_depth * c->x_nhid;
    // one AttentionLayer (attention.py:203-304) of a stack: dims, blob ranges, chained edge MLP
    auto attn_layer = [&](AttnLayerD& a, bool first, bool last, int c_init, int c_hid, int c_final, int nhid, int adim,
                          int heads, int num_linears, int conv_mlp, int& ch) -> bool {
        a.cin = first ? c_init : c_hid;
        a.cout = last ? c_final : c_hid;
        a.fin = first ? F : nhid;
        a.adim = first ? nhid : adim;
        a.fout = nhid;
        a.dsplit = a.adim / heads;
        if (a.dsplit < 1 || a.adim % a.dsplit) { pb.fail(CCSD_ERR_INVALID, "attn_dim not divisible into head chunks"); return false; }
        a.nchunk = a.adim / a.dsplit;
        a.ci0 = ch - a.cin; a.co0 = ch; ch += a.cout;
        a.conv_mlp = conv_mlp ? 1 : 0;
        a.attn_stride = (a.conv_mlp ? 2 * (2 * a.adim * a.fin + 2 * a.adim + a.adim * 2 * a.adim + a.adim) : 2 * (a.fin * a.adim + a.adim)) +
                        a.fin * a.fout + a.fout;
        a.attn_base = pb.take((int64_t)a.cin * a.attn_stride);
        a.w_lo = a.attn_base;
        a.cp = pad16(2 * a.adim + a.fout);
        a.qkvp = pb.pcur;
        if (!a.conv_mlp) pb.pcur += a.cin * (a.fin * a.cp + a.cp);
        const int hid = 2 * (a.cin > a.cout ? a.cin : a.cout);
        a.mlp = pb.mlp(num_linears, 2 * a.cin, hid, a.cout);
        pb.chainify(a.mlp, CCSD_CHAIN_EDGE);
        pb.transposed(a.mlp);
        a.mc = pb.mlp(2, a.cin * a.fout, hid, a.fout);
        pb.transposed(a.mc);
        a.w_hi = pb.cur;
        a.mcs = (a.fout + 3) >> 2;
        a.mcp = pb.pcur;
        pb.pcur += (pad16(hid) >> 4) * a.cin * a.mcs * 64;
        return true;
    };
    p->x_gmh = c->x_gmh ? 1 : 0; p->g_cinit = 0; p->g_nch = 0;
    if (p->x_gmh) {
        if (c->x_num_heads < 1 || c->x_c_init < 1) { pb.fail(CCSD_ERR_INVALID, "bad heads/c_init (ScoreNetworkX_GMH)"); return 0; }
        p->g_cinit = c->x_c_init;
        int gch = c->x_c_init;
        for (int l = 0; l < c->x_depth; ++l)
            if (!attn_layer(p->gl[l], l == 0, l == c->x_depth - 1 && l != 0, c->x_c_init, c->x_c_hid, c->x_c_final, c->x_nhid,
                            c->x_adim, c->x_num_heads, c->x_num_linears, c->x_conv_mlp, gch)) return 0;
        p->g_nch = gch;
        p->x_gw[0] = p->gl[0].attn_base;
    } else
    for (int l = 0; l < c->x_depth; ++l) {
        p->x_gw[l] = pb.take((int64_t)(l ? c->x_nhid : F) * c->x_nhid);
        p->x_gb[l] = pb.take(c->x_nhid);
    }
    p->x_wlo = p->x_gw[0];
    p->x_fin = pb.mlp(3, p->x_fdim, 2 * p->x_fdim, F);
    pb.chainify(p->x_fin, CCSD_CHAIN_XFIN);
    pb.transposed(p->x_fin);
    p->x_whi = pb.cur;
    // ---- ScoreNetworkA graph branch
    if (c->a_num_layers < 1 || c->a_num_layers > CCSD_MAXL) { pb.fail(CCSD_ERR_UNSUPPORTED, "a_num_layers out of range"); return 0; }
    if (c->a_num_heads < 1 || c->a_c_init < 1) { pb.fail(CCSD_ERR_INVALID, "bad heads/c_init"); return 0; }
    p->a_L = c->a_num_layers; p->a_cinit = c->a_c_init; p->a_is_cc = c->a_is_cc_net;
    int ch = c->a_c_init;
    for (int l = 0; l < p->a_L; ++l) {
        if (!attn_layer(p->al[l], l == 0, l == p->a_L - 1 && l != 0, c->a_c_init, c->a_c_hid, c->a_c_final, c->a_nhid, c->a_adim,
                        c->a_num_heads, c->a_num_linears, c->a_conv_mlp, ch)) return 0;
    }
    p->a_nch_graph = ch;
    int fdim = c->a_c_hid * (p->a_L - 1) + c->a_c_final + c->a_c_init;
    if (fdim != ch) { pb.fail(CCSD_ERR_INVALID, "ScoreNetworkA channel count inconsistent (num_layers==1 needs c_hid==c_final)"); return 0; }
    // ---- hodge branch
    p->h_L = 0; p->a_nch_hodge = 0; p->hb_L = 0;
    if (c->a_is_cc_net == 2) {
        if (!c->is_cc) { pb.fail(CCSD_ERR_INVALID, "ScoreNetworkA_Base_CC is only for combinatorial complexes"); return 0; }
        if (c->h_num_layers < 1 || c->h_num_layers > 2) {
            pb.fail(CCSD_ERR_UNSUPPORTED, "HIP path supports 1 or 2 HodgeBaselineLayers"); return 0; }
        p->hb_L = c->h_num_layers;
        int hch = c->a_c_init;
        for (int l = 0; l < p->hb_L; ++l) {
            HodgeBaseD& h = p->hb[l];
            const bool first = (l == 0), last = (l == p->hb_L - 1) && !first;
            h.cin = first ? c->a_c_init : c->h_c_hid;
            h.cout = last ? c->h_c_final : c->h_c_hid;
            h.hid = first ? c->h_nhid : c->h_adim;
            if (h.hid < 1) { pb.fail(CCSD_ERR_INVALID, "HodgeBaselineLayer hidden width < 1"); return 0; }
            h.blk_stride = 2 * h.hid * E + h.hid + E;
            h.blk_base = pb.take((int64_t)h.cin * h.blk_stride);
            h.w2t = pb.pcur; pb.pcur += h.cin * h.hid * E;
            h.w1t = pb.pcur; pb.pcur += h.cin * h.hid * E;
            const int hid = 2 * (h.cin > h.cout ? h.cin : h.cout);
            h.mh = pb.mlp(c->h_num_linears, h.cin, hid, h.cout);
            if (first) pb.chainify(h.mh, CCSD_CHAIN_EDGE);       // the dense first layer evaluates it per (e, e') pair on MFMA
            if (h.cin > CCSD_FW || (c->h_num_linears > 1 && hid > CCSD_FW) || h.cout > CCSD_FW) {
                pb.fail(CCSD_ERR_UNSUPPORTED, "HodgeBaselineLayer mlp_hodge wider than 16"); return 0; }
            hch += h.cout;
        }
        p->a_nch_hodge = hch;
        int hf = c->h_c_hid * (p->hb_L - 1) + c->h_c_final + c->a_c_init;
        if (hf != hch) { pb.fail(CCSD_ERR_INVALID, "hodge channel count inconsistent (num_layers_h==1 needs c_hid_h==c_final_h)"); return 0; }
    } else if (c->a_is_cc_net) {
        if (!c->is_cc) { pb.fail(CCSD_ERR_INVALID, "ScoreNetworkA_CC is only for combinatorial complexes"); return 0; }
        if (c->h_num_layers < 1 || c->h_num_layers > CCSD_MAXHL + CCSD_MAXHLX) {
            pb.fail(CCSD_ERR_UNSUPPORTED, "HIP path supports 1 to 8 HodgeAdjAttentionLayers"); return 0; }
        p->h_L = c->h_num_layers;
        int hch = c->a_c_init;
        for (int l = 0; l < p->h_L; ++l) {
            HodgeLayerD& h = ccsd_hl_mut(*p, l);
            const bool first = (l == 0), last = (l == p->h_L - 1) && !first;
            h.cin = first ? c->a_c_init : c->h_c_hid;
            h.cout = last ? c->h_c_final : c->h_c_hid;
            h.adim = first ? c->h_nhid : c->h_adim;
            h.dsplit = h.adim / c->h_num_heads;
            if (h.dsplit < 1 || h.adim % h.dsplit) { pb.fail(CCSD_ERR_INVALID, "hodge attn_dim not divisible into head chunks"); return 0; }
            h.nchunk = h.adim / h.dsplit;
            h.wc = h.cin * 2 * h.adim;
            h.wcat = pb.take((int64_t)K * h.wc);
            h.bcat = pb.take(h.wc);
            h.wcatT = pb.pcur; pb.pcur += pad16(h.wc) * ((K + 31) & ~31);
            const int hid = 2 * (h.cin > h.cout ? h.cin : h.cout);
            h.mval = pb.mlp(c->h_num_linears, h.cin, hid, 1);
            h.matt = pb.mlp(c->h_num_linears, h.cin, hid, h.cout);
            if (h.cin > CCSD_SMALLW || hid > CCSD_SMALLW || h.cout > CCSD_SMALLW) {
                pb.fail(CCSD_ERR_UNSUPPORTED, "hodge MLP wider than 8"); return 0; }
            hch += h.cout;
        }
        p->a_nch_hodge = hch;
        int hf = c->h_c_hid * (p->h_L - 1) + c->h_c_final + c->a_c_init;
        if (hf != hch) { pb.fail(CCSD_ERR_INVALID, "hodge channel count inconsistent (num_layers_h==1 needs c_hid_h==c_final_h)"); return 0; }
        // the transposed copies Wcat_l^T of layers >= 1 are consecutive [pad16(wc_l)][Kp] blocks: k_r2 treats them as ONE
        // projection of h_pw columns
        p->h_pw = 0;
        for (int l = 1; l < p->h_L; ++l) {
            p->h_poff[l] = p->h_pw;
            if (ccsd_hl(*p, l).wcatT != ccsd_hl(*p, 1).wcatT + p->h_pw * ((K + 31) & ~31)) { pb.fail(CCSD_ERR_RUNTIME, "Wcat^T blocks not consecutive"); return 0; }
            p->h_pw += l == p->h_L - 1 ? ccsd_hl(*p, l).wc : pad16(ccsd_hl(*p, l).wc);
        }
    }
    p->a_fdim = p->a_nch_graph + p->a_nch_hodge;
    p->a_fin = pb.mlp(3, p->a_fdim, 2 * p->a_fdim, 1);
    pb.chainify(p->a_fin, CCSD_CHAIN_AFIN);
    pb.transposed(p->a_fin);
    // ---- ScoreNetworkF
    if (c->is_cc) {
        if (c->f_num_layers < 1 || c->f_num_layers > CCSD_MAXFL) { pb.fail(CCSD_ERR_UNSUPPORTED, "f_num_layers out of range"); return 0; }
        if (c->f_cnum < 1 || c->f_cnum > CCSD_MAXCN) { pb.fail(CCSD_ERR_UNSUPPORTED, "HIP path supports cnum in 1..4"); return 0; }
        p->f_L = c->f_num_layers; p->f_cnum = c->f_cnum; p->f_hmask = c->f_use_hodge_mask;
        int fch = c->f_cnum;
        for (int l = 0; l < p->f_L; ++l) {
            const bool first = (l == 0), last = (l == p->f_L - 1) && !first;
            const int cin = first ? c->f_cnum : c->f_c_hid, cout = last ? c->f_c_final : c->f_c_hid;
            p->fl[l] = pb.mlp(c->f_num_linears, cin, c->f_nhid, cout);
            if (cin > CCSD_FWMAX || cout > CCSD_FWMAX || c->f_nhid > CCSD_FWMAX) { pb.fail(CCSD_ERR_UNSUPPORTED, "ScoreNetworkF layer wider than 32"); return 0; }
            fch += cout;
        }
        p->f_fdim = c->f_c_hid * (p->f_L - 1) + c->f_c_final + c->f_cnum;
        if (p->f_fdim != fch) { pb.fail(CCSD_ERR_INVALID, "ScoreNetworkF channel count inconsistent"); return 0; }
        p->f_fin = pb.mlp(c->f_num_layers_mlp, p->f_fdim, 2 * p->f_fdim, 1);
        if (p->f_fdim > CCSD_FWMAX || (c->f_num_layers_mlp > 1 && 2 * p->f_fdim > CCSD_FWMAX)) {
            pb.fail(CCSD_ERR_UNSUPPORTED, "ScoreNetworkF final MLP wider than 32"); return 0; }
        p->f_affine = (c->f_num_linears == 1 && c->f_num_layers_mlp == 1) ? 1 : 0;
    }
    const size_t nweights = (size_t)pb.cur;
    // ScoreNetworkF, general (non-affine) path with every layer width <= 8 and a single-Linear head: zero-padded [8][8] + [8]
    // blocks (CCSD_HWBLK floats per Linear) and the head's weights in 8-wide segments, appended to the device copy of the blob
    p->f_blk = -1;
    if (c->is_cc && !p->f_affine && p->f_fin.n == 1 && p->f_cnum <= 8) {
        bool ok = true;
        for (int l = 0; l < p->f_L; ++l) ok = ok && p->fl[l].in <= 8 && p->fl[l].out <= 8 && (p->fl[l].n == 1 || p->fl[l].hid <= 8);
        if (ok) p->f_blk = (int)((nweights + 15) & ~(size_t)15);
    }

    // ---- k_xa LDS carve-up
    const int NN = N * N;
    p->ldn = round_ld(N);
    int fmaxA = F > c->a_nhid ? F : c->a_nhid;
    int colmax = 0, mchid = 0, cinmax = 0;
    int pw_pair = 1;   // widest hidden activation of the per-layer edge MLPs
    bool pair_chained_all = true;
    auto layer_max = [&](const AttnLayerD& a) {
        int cols = 2 * a.adim + a.fout; if (cols > colmax) colmax = cols;
        if (a.mc.hid > mchid) mchid = a.mc.hid;
        if (a.cin > cinmax) cinmax = a.cin;
        if (a.mlp.n > 1 && a.mlp.hid > pw_pair) pw_pair = a.mlp.hid;
        pair_chained_all = pair_chained_all && a.mlp.chain != 0;
    };
    for (int l = 0; l < p->a_L; ++l) layer_max(p->al[l]);
    if (p->x_gmh) {
        for (int l = 0; l < p->x_depth; ++l) layer_max(p->gl[l]);
        if (c->x_nhid > fmaxA) fmaxA = c->x_nhid;
    }
    p->chan_rows = p->a_fdim > p->g_nch ? p->a_fdim : p->g_nch;
    const int pw_fin = 2 * p->a_fdim;
    p->pw_pair = pw_pair;
    const int NNpad = (NN + 15) / 16 * 16;
    int hq_floats = 0, h1m_floats = 0;
    if (p->h_L) {
        // (+1: the first layer's rows have an odd stride in k_xa -- bank-conflict-free reads in the dense pair loop)
        for (int l = 0; l < p->h_L; ++l) { int v = ccsd_hl(*p, l).cin * E * (2 * ccsd_hl(*p, l).adim + (l == 0 ? 1 : 0)); if (v > hq_floats) hq_floats = v; }
        for (int l = 0; l + 1 < p->h_L; ++l) { int v = ccsd_hl(*p, l).cout * E * E; if (v > h1m_floats) h1m_floats = v; }
        if (E > NN) { pb.fail(CCSD_ERR_UNSUPPORTED, "E > N*N"); return 0; }
    }
    // weight staging: largest section (X-network, one AttentionLayer, final MLP); only when it is small
    int wst = 0;   // (weights are read in place from L2; the LDS-staged variant is gone)
    // Candidate LDS budgets, best first: 4 workgroups/CU reading weights from L2 (32 waves/CU hide the latency of the
    // ~45 barrier-separated phases best); 2/CU with the weights staged in LDS; 3/CU; then whatever fits in one CU.
    // Within a budget take the largest channel group / chunk sizes.  CCSD_XA_PASS=<n> skips the first n candidates.
    const int wst_full = wst;
    const int NCAND = 4;
    const int budgets_b[NCAND] = {40960, 53 * 1024, 79 * 1024, 152 * 1024};
    const int stage_on[NCAND] = {0, 0, 0, 0};
    int hw_n = 1;
    for (int l = 0; l < p->h_L; ++l) if (ccsd_hl(*p, l).matt.n > hw_n) hw_n = ccsd_hl(*p, l).matt.n;
    auto ld_of = [](int rows) { int r = (rows + 15) / 16 * 16; if (r % 32 == 0) r += 8; return r; };   // 2-way conflicts at worst
    int best_total = -1;
    const char* skip = getenv("CCSD_XA_PASS");
    // Large graphs (zinc250k: 46 channels x 38 x 38 = 266 KB): second round of candidates with the channel stack in the
    // HBM workspace (L2-resident, one slab per graph) and only the per-layer working set in LDS.  CCSD_XA_GCH=1 forces it.
    // Start at the budget that keeps ceil(batch / #CUs) workgroups co-resident per CU (batch_hint; unknown = 4 per CU, the
    // qm9_CC B = 1024 case), then grow.  Within a budget the channel stack in LDS is tried before the HBM variant: with the
    // residency the batch needs, more workgroups per CU buy nothing and the HBM stack costs.
    const int gch_first = getenv("CCSD_XA_GCH") ? 1 : 0;
    int need = c->batch_hint > 0 ? (c->batch_hint + 255) / 256 : 4;
    need = need < 1 ? 1 : need > 4 ? 4 : need;
    for (int pass = skip ? atoi(skip) : 4 - need; pass < NCAND && best_total < 0; ++pass)
    for (int gch = gch_first; gch < 2 && best_total < 0; ++gch) {
        if (stage_on[pass] && (wst_full == 0 || gch)) continue;
        p->chan_global = gch;
        wst = stage_on[pass] ? wst_full : 0;
        const int budget = budgets_b[pass] / 4;
        for (int cg = cinmax; cg >= 1 && best_total < 0; --cg) {
            int o = 0;
            auto carve = [&](int n) { int r = o; o += (n + 3) / 4 * 4; return r; };
            p->o_flags = carve(N);
            p->o_x = carve(N * F);
            p->o_adj = carve(NN);
            p->o_an = carve(gch ? cinmax * N : cg * N > N ? cg * N : N);    // D^-1/2 per channel of the group (HBM stack: of the layer)
            p->o_edge = carve(E);
            const int phase0 = o;
            p->o_xcat = carve(p->x_fdim * p->ldn);                          // X-network phase ...
            // hidden activations of the head: in registers when it is chained (then h1 only receives the F outputs)
            p->o_h1 = carve(p->x_fin.chain ? (F > 4 ? F : 4) * p->ldn : 2 * p->x_fdim * p->ldn);
            p->o_h2 = carve(p->x_fin.chain ? 4 : 2 * p->x_fdim * p->ldn);
            const int xphase_end = o;
            p->x_lds_floats = xphase_end + 64;                              // (GMH: the whole layout, set below)
            // (ScoreNetworkX_GMH runs the attention-stack machinery itself: its regions stay clear of the A-network phase's)
            if (!p->x_gmh) o = phase0;                                      // ... aliased by the A-network phase
            p->o_chan = gch ? 0 : carve(p->chan_rows * NN);
            p->o_tmp = o;                                                   // (raw attention scratch: gone, symmetrisation is fused)
            p->o_att = carve(cinmax * NN);                                  // attention of every input channel
            p->o_xcur = carve(fmaxA * p->ldn);
            p->o_xnext = carve(fmaxA * p->ldn);
            const int o_after_xnext = o;
            p->o_vcat = carve(mchid * p->ldn);                              // hidden layer of multi_channel
            p->o_deg = p->o_vcat;
            if (p->h_L) {
                // the hodge branch runs after the attention stack: its Q|K scratch reuses [raw attention | attention]
                // the hodge branch runs after the attention stack: its Q|K scratch reuses [attention | node features |
                // multi_channel hidden] (all dead by then); the degree scratch then needs its own slot
                if (hq_floats <= o - p->o_att) {
                    p->o_hq = p->o_att;
                    if (p->h_L > 1 && (hq_floats > o_after_xnext - p->o_att || p->hl[1].cin * E > mchid * p->ldn))
                        p->o_deg = carve(p->hl[1].cin * E);
                } else {
                    p->o_hq = carve(hq_floats);
                    if (p->h_L > 1 && p->hl[1].cin * E > mchid * p->ldn) p->o_deg = carve(p->hl[1].cin * E);
                }
                p->o_hd = carve(p->a_nch_hodge * E + (p->h_L > 1 ? 2 * E : 0));   // + [s fl | b fl] of P_1's composition (k_xa)
                p->o_hw = carve((p->h_L > 2 ? p->h_L : 2) * hw_n * 72);
                p->hw_stride = hw_n * 72;
                if (p->h_L > 2) {
                    int degmax = 0, wcmax = 0;
                    for (int l = 1; l < p->h_L; ++l) {
                        if (ccsd_hl(*p, l).cin * E > degmax) degmax = ccsd_hl(*p, l).cin * E;
                        if (ccsd_hl(*p, l).wc > wcmax) wcmax = ccsd_hl(*p, l).wc;
                    }
                    p->o_deg = carve(degmax);
                    p->o_h2m = carve(h1m_floats);
                    p->o_hM = carve((p->h_L - 2) * E * E);
                    p->o_hX = carve(2 * E * wcmax);
                }
            }
            int hb_rmin = 0;
            if (p->hb_L) {
                // the baseline hodge branch runs after the attention stack too
                int g = p->hb[0].cin * E * p->hb[0].hid;
                p->o_hbg = (g <= o - p->o_att) ? p->o_att : carve(g);
                p->o_hbd = carve(p->hb_L > 1 ? p->hb[1].cin * E : 4);
                p->o_hbw = carve(2 * CCSD_MAXLIN * (CCSD_FW * CCSD_FW + CCSD_FW));
                p->o_hd = carve(p->a_nch_hodge * E);
                // per row of the chunk: the symmetrised block outputs (mlp_hodge's input), layer 0's output, the next layer's hidden row
                if (p->hb_L > 1) hb_rmin = p->hb[0].cin * E + p->hb[0].cout * (E + p->hb[1].hid);
            }
            if (xphase_end > o) o = xphase_end;
            p->o_wst = carve(wst);
            p->wst_floats = wst;
            // shared region R: GCN scratch of a channel group | hidden activations of the MLPs | dense hodge layer
            // [channel][node][Q | K | V] of a group (the GCN's x W intermediate lives in registers)
            int rmin = cg * N * colmax;
            if (N * c->x_nhid > rmin) rmin = N * c->x_nhid;
            if (h1m_floats > rmin) rmin = h1m_floats;
            if (NN > rmin) rmin = NN;                                       // raw output of the chained final MLP
            if (hb_rmin > rmin) rmin = hb_rmin;
            if (o + rmin > budget) continue;
            const bool pair_chained = pair_chained_all;
            const bool fin_chained = p->a_fin.chain != 0;
            if (!fin_chained && o + 2 * pw_fin * 16 > budget) continue;
            int pch = 16, pchp = 16;
            while (!fin_chained && pch + 16 <= NNpad && o + 2 * pw_fin * ld_of(pch + 16) <= budget) pch += 16;
            while (!pair_chained && pchp + 16 <= NNpad && o + 2 * pw_pair * ld_of(pchp + 16) <= budget) pchp += 16;
            int r = rmin;
            if (!fin_chained && 2 * pw_fin * ld_of(pch) > r) r = 2 * pw_fin * ld_of(pch);
            if (!pair_chained && 2 * pw_pair * ld_of(pchp) > r) r = 2 * pw_pair * ld_of(pchp);
            p->cg = cg; p->pch = pch; p->ldp = ld_of(pch); p->pchp = pchp; p->ldpp = ld_of(pchp);
            p->o_c0 = carve(r > 64 ? r : 64); p->o_c1 = p->o_c0;
            if (hb_rmin) { p->hb_rows = (r > 64 ? r : 64) / hb_rmin; if (p->hb_rows > E) p->hb_rows = E; }
            p->o_red = p->o_c0;                                              // block reductions run when R is idle
            p->xa_lds_floats = o;
            if (p->x_gmh) p->x_lds_floats = o;
            best_total = o;
        }
    }
    // ScoreNetworkX late (k_xa): one wave runs it in stages inside the A-network's MLP-chain intervals, where that wave has
    // no tile.  Needs: the plain networks, chained MLPs, two AttentionLayers, at most three 16-pair tiles (E <= 48), node
    // tiles of 16, and room for its own [x_fdim + max(F, 4)][ldn] + 16 floats inside the budget the layout already fits.
    p->x_late = 0; p->o_lx = 0;
    if (best_total > 0 && !p->x_gmh && !p->hb_L && !p->chan_global && p->x_fin.chain && p->a_fin.chain && p->a_L >= 2 && N <= 16 && E <= 48 &&
        p->al[0].mlp.chain && p->al[1].mlp.chain && getenv("CCSD_NO_XLATE") == nullptr) {
        const int lx = (p->x_fdim + (F > 4 ? F : 4)) * p->ldn + 16;
        int cap = 160 * 1024 / 4;
        for (int q = NCAND - 1; q >= 0; --q) if (best_total * 4 <= budgets_b[q]) cap = budgets_b[q] / 4;
        if (best_total + lx <= cap) { p->x_late = 1; p->o_lx = best_total; best_total += lx; p->xa_lds_floats = best_total; if (p->x_gmh) p->x_lds_floats = best_total; }
    }
    if (getenv("CCSD_VERBOSE"))
        fprintf(stderr, "[ccsd] k_xa LDS %d B (cg=%d pch=%d/%d pchp=%d/%d stage=%d floats, channel stack in %s)\n", best_total * 4, p->cg, p->pch, p->ldp, p->pchp, p->ldpp, p->wst_floats, p->chan_global ? "HBM" : "LDS");
    if (best_total < 0 || (size_t)best_total * 4 > 160 * 1024)
        pb.fail(CCSD_ERR_UNSUPPORTED, "graph-network working set exceeds the 160 KB LDS of a CU");
    return nweights;
}

// ScoreNetworkF block layout behind the weight blob (PlanD::f_blk): Linear q of layer l at (l * CCSD_MAXLIN + q) * 72
// ([8][8] weights, row = output, + [8] biases); then the head: (CCSD_MAXFL + 1) segments of 8 weights (segment 0: the cnum
// input channels, segment l + 1: the outputs of layer l) + 8 floats whose first is the bias.
#define CCSD_FBLK_HEAD (CCSD_MAXFL * CCSD_MAXLIN * 72)
#define CCSD_FBLK_FLOATS (CCSD_FBLK_HEAD + (CCSD_MAXFL + 1) * 8 + 8)
static inline void ccsd_pack_fnet_blocks(const PlanD* p, const float* w, float* dst) {
    for (int i = 0; i < CCSD_FBLK_FLOATS; ++i) dst[i] = 0.f;
    int co0 = p->f_cnum;
    for (int i = 0; i < p->f_cnum; ++i) dst[CCSD_FBLK_HEAD + i] = w[p->f_fin.w[0] + i];
    for (int l = 0; l < p->f_L; ++l) {
        const MlpD& m = p->fl[l];
        for (int q = 0; q < m.n; ++q) {
            float* blk = dst + (l * CCSD_MAXLIN + q) * 72;
            const int ni = mlp_in(m, q), no = mlp_out(m, q);
            for (int o = 0; o < no; ++o) {
                for (int i = 0; i < ni; ++i) blk[o * 8 + i] = w[m.w[q] + o * ni + i];
                blk[64 + o] = w[m.b[q] + o];
            }
        }
        for (int i = 0; i < m.out; ++i) dst[CCSD_FBLK_HEAD + (l + 1) * 8 + i] = w[p->f_fin.w[0] + co0 + i];
        co0 += m.out;
    }
    dst[CCSD_FBLK_HEAD + (CCSD_MAXFL + 1) * 8] = w[p->f_fin.b[0]];
}

// Fold ScoreNetworkF into  score = mask * (alpha*F + beta*(H F) + gamma)  when every MLP in it is a
// single Linear (num_linears == 1 and num_layers_mlp == 1: true for every shipped CC checkpoint but
// ENZYMES).  The intermediate mask_rank2 factors are 0/1 and the final output is masked again, so
// the fold is exact in real arithmetic.
static inline void ccsd_fold_fnet(PlanD* p, const float* w) {
    if (!p->is_cc || !p->f_affine) return;
    const int fd = p->f_fdim, cn = p->f_cnum, nb = cn + 1;
    std::vector<double> A((size_t)fd * nb, 0.0);  // channel j = sum_i A[j][i] * H^i F  (i < cnum)  + A[j][cnum]
    for (int j = 0; j < cn; ++j) A[(size_t)j * nb + j] = 1.0;
    int ci0 = 0, co0 = cn;
    for (int l = 0; l < p->f_L; ++l) {
        const MlpD& m = p->fl[l];
        for (int oo = 0; oo < m.out; ++oo) {
            std::vector<double> acc(nb, 0.0);
            acc[cn] = (double)w[m.b[0] + oo];
            for (int i = 0; i < m.in; ++i)
                for (int t = 0; t < nb; ++t) acc[t] += (double)w[m.w[0] + oo * m.in + i] * A[(size_t)(ci0 + i) * nb + t];
            for (int t = 0; t < nb; ++t) A[(size_t)(co0 + oo) * nb + t] = acc[t];
        }
        ci0 = co0; co0 += m.out;
    }
    std::vector<double> r(nb, 0.0);
    r[cn] = (double)w[p->f_fin.b[0]];
    for (int j = 0; j < fd; ++j)
        for (int t = 0; t < nb; ++t) r[t] += (double)w[p->f_fin.w[0] + j] * A[(size_t)j * nb + t];
    p->f_alpha = (float)r[0]; p->f_beta = cn > 1 ? (float)r[1] : 0.f; p->f_gamma = (float)r[cn];
    for (int j = 0; j < CCSD_MAXCN; ++j) p->f_betas[j] = (j >= 1 && j < cn) ? (float)r[j] : 0.f;
}
#endif  // CCSD_DEVICE_ONLY
