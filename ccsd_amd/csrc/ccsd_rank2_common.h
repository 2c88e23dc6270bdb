// ccsd_rank2_common.h -- helpers shared by the rank-2 kernels and the graph-network kernel: flag masks from the off-bit table,
// epilogue modes, ScoreNetworkF per element, wave-level MFMA tile loops, the fused Langevin-apply coefficients.
// Part of the kernel source of libccsd_hip.so (see ccsd_kernels.h for the map).
#pragma once
#include "ccsd_dev.h"

CCSD_DEV float edge_on(unsigned long long off, const unsigned char* __restrict__ edges, int e) {
    return ((off >> edges[2 * e]) | (off >> edges[2 * e + 1])) & 1ull ? 0.f : 1.f;
}
CCSD_DEV float cell_on(unsigned long long off, const unsigned long long* __restrict__ cells, int k) {
    return (cells[k] & off) ? 0.f : 1.f;
}

// The masks flags_left[e] * flags_right[k] of the four consecutive elements of flat group (e, k) .. of a sample's (E, K) block from
// the byte tables of k_masktab (rows mfr + b Kp, mfl + b Ep): with K a multiple of 4 the group lies inside one row and k is a
// multiple of 4 -- one 32-bit word of mfr and one byte of mfl; otherwise byte by byte across the row end.
// row stride of the Hodge Laplacian buffers H, H^2, ... in the workspace (k_gemm_h / k_gemm_h_full / k_gemm_pow write, k_hf_score reads):
// E rounded up to whole 16-byte groups, so that k_hf_score stages its rows of H with 16-byte loads (E = 190 -> 192; the pad columns
// are never written and never used: the loader masks them)
static inline __host__ __device__ int h_ld(int E) { return (E + 3) & ~3; }
struct MaskTab { const unsigned char* mfr; const unsigned char* mfl; int Kp, Ep; };
CCSD_DEV void group_masks(const MaskTab& mt, int b, int E, int K, int e, int k, float* m) {
    const unsigned char* fr = mt.mfr + (size_t)b * mt.Kp;
    const unsigned char* fl = mt.mfl + (size_t)b * mt.Ep;
    if ((K & 3) == 0) {
        const unsigned f4 = *reinterpret_cast<const unsigned*>(fr + k);
        const float l = (float)fl[e < E ? e : E - 1];
        m[0] = l * (float)(f4 & 0xffu); m[1] = l * (float)((f4 >> 8) & 0xffu); m[2] = l * (float)((f4 >> 16) & 0xffu); m[3] = l * (float)(f4 >> 24);
    } else {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            m[s] = (float)fl[e < E ? e : E - 1] * (float)fr[k];      // (beyond the block only in a ragged last group: never used)
            if (++k == K) { k = 0; ++e; }
        }
    }
}

enum { MODE_SCORE = 0, MODE_NORMS = 1, MODE_PRED = 2 };

struct RankEpi {
    int mode;
    float sscale;            // MODE_SCORE: out = sscale * net
    float pa, pb, pc;        // MODE_PRED
    float* out;              // SCORE: score; NORMS: raw net output (kept for the apply pass); PRED: new state
    float* mean;             // PRED: nullable
    float* part;             // NORMS: [B][ntiles][2] partial sums of net^2 and z^2
};

// FW: width the per-element MLPs are padded to (8 when every layer of the network fits, else CCSD_FW = 16).
// hf[j] = (H^(j+1) F) at this element, j < cnum - 1 (pow_tensor_cc, cc_utils.py:961-979).
template <bool AFFINE, int FW = CCSD_FW>
CCSD_DEV float fnet_element(const PlanD& p, const float* __restrict__ w, float f, const float* hf, float m) {
    if (AFFINE) {
        float t = fmaf(p.f_alpha, f, p.f_gamma);
#pragma unroll
        for (int j = 1; j < CCSD_MAXCN; ++j)
            if (j < p.f_cnum) t = fmaf(p.f_betas[j], hf[j - 1], t);
        return m * t;
    }
    if (p.f_blk >= 0) {
        // every layer <= 8 wide, single-Linear head: zero-padded blocks behind the weight blob (ccsd_pack_fnet_blocks), read
        // with wide scalar loads; each layer's output stays in its own registers and the head is accumulated segment by
        // segment in concat order (no dynamic register indexing, padded lanes contribute exact zeros)
        const float* fb = w + p.f_blk;
        const float* hd = fb + CCSD_FBLK_HEAD;
        float prev[8] = {f, p.f_cnum > 1 ? hf[0] : 0.f, p.f_cnum > 2 ? hf[1] : 0.f, p.f_cnum > 3 ? hf[2] : 0.f, 0.f, 0.f, 0.f, 0.f};
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) acc = fmaf(prev[i], hd[i], acc);
#pragma unroll
        for (int l = 0; l < CCSD_MAXFL; ++l)
            if (l < p.f_L) {
                float o8[8];
                small_mlp_lds<8>(fb + l * CCSD_MAXLIN * CCSD_HWBLK, p.fl[l].n, prev, o8);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    prev[i] = m * o8[i];                             // mask_rank2 after every layer (hodge_layers.py:90)
                    acc = fmaf(prev[i], hd[(l + 1) * 8 + i], acc);
                }
            }
        return m * (acc + hd[(CCSD_MAXFL + 1) * 8]);
    }
    // general path: channels [F, HF] -> L x (MLP, mask) -> concat -> final MLP -> mask
    float ch[FW];
#pragma unroll
    for (int i = 0; i < FW; ++i) ch[i] = 0.f;
    ch[0] = f;
#pragma unroll
    for (int j = 1; j < CCSD_MAXCN; ++j)
        if (j < p.f_cnum) ch[j] = hf[j - 1];
    int ci0 = 0, co0 = p.f_cnum;
    for (int l = 0; l < p.f_L; ++l) {
        float in[FW], out[FW];
#pragma unroll
        for (int i = 0; i < FW; ++i) {
            float v = 0.f;
#pragma unroll
            for (int j = 0; j < FW; ++j)
                if (j == ci0 + i) v = ch[j];
            in[i] = v;
        }
        small_mlp<FW>(p.fl[l], w, in, out);
        const int no = p.fl[l].out;
#pragma unroll
        for (int j = 0; j < FW; ++j)
#pragma unroll
            for (int i = 0; i < FW; ++i)
                if (i < no && j == co0 + i) ch[j] = m * out[i];   // mask_rank2 after every layer (hodge_layers.py:90)
        ci0 = co0; co0 += no;
    }
    float out[FW];
    small_mlp<FW>(p.f_fin, w, ch, out);
    return m * out[0];
}

// ---------------------------------------------------------------------------------------------
// k_r2: the whole rank-2 side of one joint score evaluation for ONE complex per workgroup, with the
// complex's rank2 block resident in LDS (E x K fp32 = 67 KB for qm9_CC): one HBM read and one HBM
// write of rank2 per half-step.  Used when E <= 64 and the block fits (ccsd_plan::fused_r2).
//   phase 0  load F -> LDS (row stride ldk == 2 mod 32: conflict-free MFMA fragment reads), cell masks,
//            adjacency powers' upper triangle (adj_to_hodgedual inputs)
//   phase 1  MFMA tiles over the full K:  H = F F^T (upper-triangle tiles, mirrored),
//            P_0 = F Wcat_0,  P_1 = rank2' Wcat_1  (hodge projections for k_xa, written to HBM)
//   phase 2  per 16-column tile: (H F) on MFMA, ScoreNetworkF element-wise, epilogue in place in LDS
//   phase 3  coalesced LDS -> HBM copy of the result
// Same arithmetic as k_gemm_h / k_gemm_p / k_hf_score (those remain the general path).
// ---------------------------------------------------------------------------------------------
// D = k-steps whose operand loads are issued together, ahead of the MFMAs that consume them (one L2 / LDS round trip per D steps)
template <int D = 4, class LA, class LB, class EP>
CCSD_DEV void wave_tile(int m0, int n0, int ks, LA la, LB lb, EP ep) {
#ifdef CCSD_EMU
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
            float acc = 0.f;
            for (int k = 0; k < 4 * ks; ++k) acc = fmaf(la(m0 + i, k), lb(k, n0 + j), acc);
            ep(m0 + i, n0 + j, acc);
        }
#else
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x & 63, l15 = lane & 15, kq = lane >> 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < ks; s += D) {
        float a[D], bv[D];
#pragma unroll
        for (int u = 0; u < D; ++u) {
            const int sc = s + u < ks ? s + u : ks - 1;       // tail: a harmless reload of the last step
            a[u] = la(m0 + l15, 4 * sc + kq);
            bv[u] = lb(4 * sc + kq, n0 + l15);
        }
#pragma unroll
        for (int u = 0; u < D; ++u)
            if (s + u < ks) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], bv[u], acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) ep(m0 + 4 * kq + r, n0 + l15, acc[r]);
#endif
}

// All `mt` (<= 4) 16-row tiles of one 16-column block: accumulate every tile first, run the epilogue
// afterwards (the epilogue may overwrite the B operand in place).
template <class LA, class LB, class EP>
CCSD_DEV void wave_coltile(int n0, int mt, int ks, LA la, LB lb, EP ep4) {
#ifdef CCSD_EMU
    float acc[64][16];
    for (int i = 0; i < 16 * mt; ++i)
        for (int j = 0; j < 16; ++j) {
            float a = 0.f;
            for (int k = 0; k < 4 * ks; ++k) a = fmaf(la(i, k), lb(k, n0 + j), a);
            acc[i][j] = a;
        }
    for (int i = 0; i < 16 * mt; i += 4)
        for (int j = 0; j < 16; ++j) {
            const float v[4] = {acc[i][j], acc[i + 1][j], acc[i + 2][j], acc[i + 3][j]};
            ep4(i, n0 + j, v);
        }
#else
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x & 63, l15 = lane & 15, kq = lane >> 4;
    f32x4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    int s = 0;
    for (; s + 4 <= ks; s += 4) {      // four k-steps: all operand loads first (B may come from L2), then the MFMAs
        float bv[4], a[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) bv[u] = lb(4 * (s + u) + kq, n0 + l15);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int t = 0; t < 4; ++t) a[u][t] = t < mt ? la(16 * t + l15, 4 * (s + u) + kq) : 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int t = 0; t < 4; ++t)
                if (t < mt) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][t], bv[u], acc[t], 0, 0, 0);
    }
    for (; s < ks; ++s) {
        const float bv = lb(4 * s + kq, n0 + l15);
#pragma unroll
        for (int t = 0; t < 4; ++t)
            if (t < mt) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(la(16 * t + l15, 4 * s + kq), bv, acc[t], 0, 0, 0);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
        if (t < mt) {
            const float v[4] = {acc[t][0], acc[t][1], acc[t][2], acc[t][3]};
            ep4(16 * t + 4 * kq, n0 + l15, v);
        }
#endif
}

CCSD_DEV float raw_noise_r1(const NoiseArgs& na, int b, int e, int k, int E, int K) {
    if (na.zr) return na.zr[((size_t)b * E + e) * K + k];
    float n[4];
    philox_normal4(na.seed, na.draw_r, na.b_off + b, (unsigned)((e >> 2) * K + k), n);
    const int s = e & 3;
    return s == 0 ? n[0] : s == 1 ? n[1] : s == 2 ? n[2] : n[3];
}

// Langevin corrector apply fused into the predictor kernels (ccsd_sampler_run): v <- v + step*score + sqrt(2 step)*z*scale_eps
// with step from the batch norm sums (solver.py:767-769, 781-783, 797-801); same arithmetic as k_langevin_apply.
struct CorrFuse {
    int on;
    const float* net_x; const float* net_adj; const float* net_r;   // raw network outputs kept by the NORMS pass
    const float* sums;
    float ss[3], alpha[3];
    float snr, seps;
    unsigned int draw_x, draw_adj, draw_r;                            // corrector draw indices (predictor ones are in NoiseArgs)
};
CCSD_DEV void corr_coef(const CorrFuse& cf, int t, float* c1, float* c2) {
    const float gn = fabsf(cf.ss[t]) * cf.sums[t], zn = cf.sums[3 + t];
    const float q = cf.snr * zn / gn;
    const float step = q * q * 2.f * cf.alpha[t];
    *c1 = step * cf.ss[t];
    *c2 = sqrtf(step * 2.f) * cf.seps;
}

