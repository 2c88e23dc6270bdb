// ccsd_kernels.h -- hand-written gfx950 kernels for the CCSD predictor-corrector sampling path.
//
// Data layout in HBM (all fp32, contiguous, batch-major; same as the reference tensors):
//   x (B,N,F)   adj (B,N,N)   rank2 (B,E,K)   flags (B,N) of exact 0/1
//   H (B,E,E) = F F^T (diag zeroed)            P_l (B*E, wc_l) = hodge Q|K projections of layer l
//
// Kernels
//   k_flagbits      flags -> per-sample bitmask of switched-off nodes
//   k_gemm_h        H = F F^T                      (64x64 MFMA f32 16x16x4 tiles, LDS staged)
//   k_gemm_p        P_l = A_l(F) . Wcat_l          (same tile engine; A_1 = on-the-fly rank2' of hodge layer 0)
//   k_edgecoef      triu entries of adj powers     (inputs of the hodge branch)
//   k_hf_score      (H F) + ScoreNetworkF epilogue + fused predictor update / Langevin norms
//   k_xa            ScoreNetworkX + ScoreNetworkA(_CC): one workgroup per graph, everything LDS-resident,
//                   all per-(i,j) MLPs on MFMA through block_linear
//   k_normsum / k_langevin_apply / k_init_state / k_quantize
//
// Reference file:line citations sit next to each restated formula.
#pragma once
#include "ccsd_plan.h"
#include <type_traits>

// ---------------------------------------------------------------------------------------------
// small device helpers
// ---------------------------------------------------------------------------------------------
// Diagnostic cycle stamps (ccsd_debug_stamps): thread 0 of every workgroup writes the shader clock at phase
// boundaries into a caller buffer [workgroup][64] (k_r2: slots 0.., k_xa: slots 32..).  NULL (the default) compiles to a uniform branch not taken.
CCSD_DEV void stamp(long long* dbg, int slot) {
#ifndef CCSD_EMU
    if (dbg && threadIdx.x == 0) dbg[(size_t)blockIdx.x * 64 + slot] = (long long)__builtin_readcyclecounter();
#else
    (void)dbg; (void)slot;
#endif
}
// exp(x) through the hardware base-2 exponential (v_exp_f32, ~1 ulp)
CCSD_DEV float fast_exp(float x) {
#ifdef CCSD_EMU
    return exp2f(x * 1.4426950408889634f);
#else
    return __builtin_amdgcn_exp2f(x * 1.4426950408889634f);
#endif
}
CCSD_DEV float fast_rcp(float x) {
#ifdef CCSD_EMU
    return 1.0f / x;
#else
    return __builtin_amdgcn_rcpf(x);
#endif
}
// tanh(x) = 1 - 2 / (e^{2x} + 1): branch-free, five instructions, ~1e-7 ABSOLUTE error (the cancellation near 0 costs
// relative accuracy there, which nothing downstream needs: every consumer is compared at 1e-4 of the tensor's scale).
// e^{2x} -> +inf gives 1, -> 0 gives -1.
CCSD_DEV float tanh_f(float x) {
    const float e = fast_exp(2.0f * x);
    return fmaf(-2.0f, fast_rcp(e + 1.0f), 1.0f);
}
// F.elu, alpha = 1: x > 0 ? x : e^x - 1 (branch-free select; ~6e-8 absolute error on the negative side)
CCSD_DEV float elu1(float v) {
    const float ex = fast_exp(v) - 1.0f;
    return v > 0.f ? v : ex;
}
CCSD_DEV float elu1_sel(float v) { return elu1(v); }
// t / d and t % d for 0 <= t < 2^22 and small d without the ~40-instruction integer division:
// (t + 0.5) * (1/d) is never within 0.5/d of an integer, far more than the fp32 rounding of the product.
struct FastDiv {
    int d; float inv;
    CCSD_DEV explicit FastDiv(int dd) : d(dd), inv(1.0f / (float)dd) {}
    CCSD_DEV int div(int t) const { return (int)(((float)t + 0.5f) * inv); }
    CCSD_DEV void divmod(int t, int& q, int& r) const { q = div(t); r = t - q * d; }
};

template <bool V> struct BoolTag { static constexpr bool v = V; };

struct NoiseArgs {
    const float* zx;
    const float* zadj;
    const float* zr;
    unsigned long long seed;
    unsigned int draw_x, draw_adj, draw_r;
    long long b_off;
};

// Philox4x32-10 (Salmon et al. 2011), counter = (group, sample, draw, 0), key = seed
CCSD_DEV void philox4(unsigned int c0, unsigned int c1, unsigned int c2, unsigned int c3, unsigned int k0,
                      unsigned int k1, unsigned int* out) {
#ifndef CCSD_EMU
    // keep the key schedule on the scalar ALU of every call: hoisted out of the callers' loops its 20 round keys
    // exhaust the SGPRs and come back as v_readlane spill traffic inside the hot loops
    asm volatile("" : "+s"(k0), "+s"(k1));
#endif
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0;
        const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c2;
        const unsigned int n0 = (unsigned int)(p1 >> 32) ^ c1 ^ k0;       // (gfx950 has no three-operand xor: two v_xor_b32)
        const unsigned int n2 = (unsigned int)(p0 >> 32) ^ c3 ^ k1;
        const unsigned int n1 = (unsigned int)p1;
        const unsigned int n3 = (unsigned int)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// four standard normals of group `g` of sample `b` of draw `draw` (Box-Muller on two uniform pairs)
CCSD_DEV void philox_normal4(unsigned long long seed, unsigned int draw, long long b, unsigned int g, float* n) {
    unsigned int r[4];
    philox4(g, (unsigned int)b, draw, (unsigned int)((unsigned long long)b >> 32), (unsigned int)seed,
            (unsigned int)(seed >> 32), r);
    const float inv24 = 1.0f / 16777216.0f;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const float u1 = (float)((r[2 * h] >> 8) + 1u) * inv24;    // (0, 1]
        const float u2 = (float)(r[2 * h + 1] >> 8) * inv24;       // [0, 1)
#ifdef CCSD_EMU
        const float rad = sqrtf(-2.0f * logf(u1));
        n[2 * h] = rad * cosf(6.283185307179586f * u2);
        n[2 * h + 1] = rad * sinf(6.283185307179586f * u2);
#else
        // v_log_f32 is log2, v_sin/v_cos take their argument in revolutions: no range reduction needed
        const float rad = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));
        n[2 * h] = rad * __builtin_amdgcn_cosf(u2);
        n[2 * h + 1] = rad * __builtin_amdgcn_sinf(u2);
#endif
    }
}
CCSD_DEV float philox_normal1(unsigned long long seed, unsigned int draw, long long b, unsigned int idx) {
    float n[4];
    philox_normal4(seed, draw, b, idx >> 2, n);
    const unsigned int s = idx & 3u;
    return s == 0 ? n[0] : s == 1 ? n[1] : s == 2 ? n[2] : n[3];
}

// raw draws (before triu/sym/masks), gen_noise graph_utils.py:171
CCSD_DEV float raw_noise_x(const NoiseArgs& na, int b, int idx, int per_sample) {
    return na.zx ? na.zx[(size_t)b * per_sample + idx] : philox_normal1(na.seed, na.draw_x, na.b_off + b, (unsigned)idx);
}
// symmetric noise: z.triu(1) + transpose -> element (i,j) takes the raw draw at (min,max); diag = 0
CCSD_DEV float raw_noise_adj(const NoiseArgs& na, int b, int i, int j, int N) {
    if (i == j) return 0.f;
    const int lo = i < j ? i : j, hi = i < j ? j : i;
    const int idx = lo * N + hi;
    return na.zadj ? na.zadj[(size_t)b * N * N + idx] : philox_normal1(na.seed, na.draw_adj, na.b_off + b, (unsigned)idx);
}
// rank2 noise for the four consecutive edge rows 4*eg .. 4*eg+3 at column k (one Philox group)
CCSD_DEV void raw_noise_r4(const NoiseArgs& na, int b, int eg, int k, int E, int K, float* n) {
    if (na.zr) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int e = 4 * eg + s;
            n[s] = e < E ? na.zr[((size_t)b * E + e) * K + k] : 0.f;
        }
    } else {
        philox_normal4(na.seed, na.draw_r, na.b_off + b, (unsigned)(eg * K + k), n);
    }
}

// block-wide sum; result valid in every thread.  `red` = 64 floats of LDS.
CCSD_DEV float block_sum(float v, float* red) {
#ifdef CCSD_EMU
    (void)red;
    return v;
#else
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float t = 0.f;
    for (int w = 0; w < nw; ++w) t += red[w];
    __syncthreads();
    return t;
#endif
}

// per-thread MLP over at most W features (fully unrolled, predicated: stays in registers).
// Restates layers.py:260-275 for the tiny channel-mixing MLPs (hodge branch, ScoreNetworkF).
template <int W>
CCSD_DEV void small_mlp(const MlpD& m, const float* __restrict__ w, const float* in, float* out) {
    float a[W], t[W];
#pragma unroll
    for (int i = 0; i < W; ++i) a[i] = in[i];
    for (int l = 0; l < m.n; ++l) {
        const int ni = mlp_in(m, l), no = mlp_out(m, l);
        const float* wl = w + m.w[l];
        const float* bl = w + m.b[l];
#pragma unroll
        for (int o = 0; o < W; ++o) {
            float acc = 0.f;
            if (o < no) {
                acc = 0.f;
#pragma unroll
                for (int i = 0; i < W; ++i)
                    if (i < ni) acc = fmaf(a[i], wl[o * ni + i], acc);
                acc += bl[o];
                if (l < m.n - 1) acc = elu1(acc);
            }
            t[o] = acc;
        }
#pragma unroll
        for (int i = 0; i < W; ++i) a[i] = t[i];
    }
#pragma unroll
    for (int i = 0; i < W; ++i) out[i] = a[i];
}

// The hodge branch's channel-mixing MLPs (mlp_attention) evaluated from zero-padded 8x8 weight blocks staged in
// LDS: block q of an MLP = [8][8] weights (row = output) + [8] biases (72 floats).  All lanes read the same
// addresses (LDS broadcast); padded rows/columns contribute exact zeros.
#define CCSD_HWBLK 72
CCSD_DEV void stage_mlp_blocks(const MlpD& m, const float* __restrict__ w, float* blk) {
    for (int t = threadIdx.x; t < m.n * CCSD_HWBLK; t += blockDim.x) {
        const int q = t / CCSD_HWBLK, r = t % CCSD_HWBLK;
        const int ni = mlp_in(m, q), no = mlp_out(m, q);
        float v = 0.f;
        if (r < 64) { const int o = r >> 3, i = r & 7; if (o < no && i < ni) v = w[m.w[q] + o * ni + i]; }
        else { const int o = r - 64; if (o < no) v = w[m.b[q] + o]; }
        blk[t] = v;
    }
}
// sum_i wg[i (* ws)] * xs[i], i < n (n >= 1), accumulated in index order; wg in global memory, xs in LDS.  The weight loads go out
// eight at a time ahead of the FMAs: one L2 round trip per 8 terms instead of one per term (a counted loop with a global load
// feeding each FMA serialises on the load latency).
template <bool STRIDED>   // STRIDED: term i of wg sits at wg[i * ws] (a transposed copy read along its other index)
CCSD_DEV float dot_gl(const float* __restrict__ wg, int ws, const float* xs, int n) {
    float acc = 0.f;
    for (int i0 = 0; i0 < n; i0 += 8) {
        float wv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) wv[j] = wg[(i0 + j < n ? i0 + j : n - 1) * (STRIDED ? ws : 1)];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float xv = xs[i0 + j < n ? i0 + j : n - 1];
            if (i0 + j < n) acc = fmaf(xv, wv[j], acc);
        }
    }
    return acc;
}

// The same with W x W blocks (block q = [W][W] weights, row = output, + [W] biases), for MLPs up to W wide.
template <int W>
CCSD_DEV void stage_mlp_blocks_w(const MlpD& m, const float* __restrict__ w, float* blk) {
    constexpr int BS = W * W + W;
    for (int t = threadIdx.x; t < m.n * BS; t += blockDim.x) {
        const int q = t / BS, r = t % BS;
        const int ni = mlp_in(m, q), no = mlp_out(m, q);
        float v = 0.f;
        if (r < W * W) { const int o = r / W, i = r % W; if (o < no && i < ni) v = w[m.w[q] + o * ni + i]; }
        else { const int o = r - W * W; if (o < no) v = w[m.b[q] + o]; }
        blk[t] = v;
    }
}
template <int W>
CCSD_DEV void small_mlp_ldsw(const float* blk, int nlin, const float* in, float* out) {
    constexpr int BS = W * W + W;
    float a[W], t[W];
#pragma unroll
    for (int i = 0; i < W; ++i) a[i] = in[i];
    for (int l = 0; l < nlin; ++l) {
        const float* wb = blk + l * BS;
        const bool act = l < nlin - 1;
#pragma unroll
        for (int o = 0; o < W; ++o) {
            float acc = wb[W * W + o];
#pragma unroll
            for (int i = 0; i < W; ++i) acc = fmaf(a[i], wb[o * W + i], acc);
            t[o] = act ? elu1(acc) : acc;
        }
#pragma unroll
        for (int i = 0; i < W; ++i) a[i] = t[i];
    }
#pragma unroll
    for (int i = 0; i < W; ++i) out[i] = a[i];
}
template <int W>   // W = 4 when every width of the MLP is <= 4 (the shipped hodge branches), else CCSD_SMALLW
CCSD_DEV void small_mlp_lds(const float* blk, int nlin, const float* in, float* out) {
    float a[W], t[W];
#pragma unroll
    for (int i = 0; i < W; ++i) a[i] = in[i];
    for (int l = 0; l < nlin; ++l) {
        const float* wb = blk + l * CCSD_HWBLK;
        const bool act = l < nlin - 1;
#pragma unroll
        for (int o = 0; o < W; ++o) {
            float acc = wb[64 + o];
#pragma unroll
            for (int i = 0; i < W; ++i) acc = fmaf(a[i], wb[o * 8 + i], acc);
            t[o] = act ? elu1(acc) : acc;
        }
#pragma unroll
        for (int i = 0; i < W; ++i) a[i] = t[i];
    }
#pragma unroll
    for (int i = 0; i < W; ++i) out[i] = a[i];
}
CCSD_DEV int mlp_maxw(const MlpD& m) {
    int wd = m.in > m.out ? m.in : m.out;
    if (m.n > 1 && m.hid > wd) wd = m.hid;
    return wd;
}

// ---------------------------------------------------------------------------------------------
// block_linear: Y[o][m] = act( sum_k X[k][m] * W[o][k] + b[o] )  for m < rows, o < out.
// X, Y: LDS, feature-major (row stride ldx / ldy; strides == 16 mod 32 give conflict-free fragment
// reads).  The input features may come from two arrays: k < ksplit from X, the rest from X2 (the
// [attention | adjacency] concatenation of attention.py:295-297 is never materialised).
// W: global, torch Linear layout [out][in].  One MFMA f32 16x16x4 output tile per task, tasks
// round-robin over the waves of the workgroup.  The accumulation is a k-ordered fmaf chain, the
// same as the emulation loop below.
// ---------------------------------------------------------------------------------------------
template <int ACT>  // 0 none, 1 ELU
CCSD_DEV void block_linear(float* Y, int ldy, const float* X, int ldx, const float* X2, int ksplit,
                           const float* __restrict__ W, const float* __restrict__ bias, int in, int out, int rows) {
#ifdef CCSD_EMU
    for (int o = 0; o < out; ++o)
        for (int m = 0; m < rows; ++m) {
            float acc = 0.f;
            for (int k = 0; k < in; ++k) acc = fmaf(k < ksplit ? X[k * ldx + m] : X2[(k - ksplit) * ldx + m], W[o * in + k], acc);
            acc += bias[o];
            Y[o * ldy + m] = ACT ? elu1(acc) : acc;
        }
#else
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    const int mt = (rows + 15) >> 4, nt = (out + 15) >> 4, ks = (in + 3) >> 2;
    const int l15 = lane & 15, kq = lane >> 4;
    for (int task = wave; task < mt * nt; task += nw) {
        const int m0 = (task % mt) << 4, n0 = (task / mt) << 4;
        const int bn = n0 + l15;
        const int am = (m0 + l15 < rows) ? m0 + l15 : rows - 1;      // clamp: rows beyond `rows` are never stored
        const float* wr = W + (size_t)(bn < out ? bn : 0) * in;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const float nok = bn < out ? 1.f : 0.f;
        for (int s0 = 0; s0 < ks; s0 += 4) {   // weights come from L2: issue the loads of four k-steps before the MFMAs.
            float a[4], bv[4];                 // Loads are unconditional (clamped addresses); out-of-range lanes get a zero weight.
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = 4 * (s0 + u) + kq;
                const int kc = k < in ? k : in - 1;
                const int off = kc < ksplit ? kc * ldx : (kc - ksplit) * ldx;
                const float* xb = kc < ksplit ? X : X2;
                a[u] = xb[off + am];
                bv[u] = wr[kc] * (k < in ? nok : 0.f);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], bv[u], acc, 0, 0, 0);
        }
        if (bn < out) {
            const float bb = bias[bn];
            float* yr = Y + bn * ldy;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + 4 * kq + r;
                const float v = acc[r] + bb;
                if (m < rows) yr[m] = ACT ? elu1(v) : v;
            }
        }
    }
#endif
}


// ---------------------------------------------------------------------------------------------
// mlp_chain_tile: a whole MLP (layers.py:260-275; up to CCSD_MAXLIN linears, ELU between) for ONE tile of 16
// rows, by ONE wave, with every activation held in registers -- no LDS round trip and no workgroup barrier
// between the linears.  Transposed formulation Y^T = W . X^T on v_mfma_f32_16x16x4_f32:
//   A operand = W (row = output feature 16*to + l15), B operand = X^T (column = row p0 + l15 of the tile),
//   accumulator element r of lane (l15, kq) = feature 16*to + 4*kq + r of row p0 + l15.
// The MFMA's k slot `kq` of step (t, j) is assigned to input feature 16*t + 4*kq + j: exactly the feature the lane
// already holds in register j of the previous layer's accumulator tile t, so the next linear's B operands ARE
// the previous accumulators; the matching A operands W[.][16t + 4kq .. +3] are one aligned float4 of the
// zero-padded copy Wp[pad16(out)][pad16(in)] (ccsd_pack_mlp).  The first linear's input comes from LDS (or the
// HBM channel stack), feature-major, optionally as two segments [X (k < ksplit) | X2].
// epi(row, feature, value) is called for the valid outputs.  Tile counts are compile-time (CHAIN_SHAPES below).
// ---------------------------------------------------------------------------------------------
#ifndef CCSD_EMU
typedef float chain_f32x4 __attribute__((ext_vector_type(4)));
// one linear of the chain: TI input tiles (registers) -> TO output tiles, straight-line code
template <int TI, int TO>
CCSD_DEV void chain_layer(const float* __restrict__ W, const float* __restrict__ Bv, int ip, bool act,
                          const chain_f32x4* in, chain_f32x4* out) {
#pragma unroll
    for (int to = 0; to < TO; ++to) {
        const float* Wr = W + (size_t)(16 * to) * ip;
        float4 wv[TI];
#pragma unroll
        for (int t = 0; t < TI; ++t) wv[t] = *reinterpret_cast<const float4*>(Wr + 16 * t);
        const float4 bb = *reinterpret_cast<const float4*>(Bv + 16 * to);
        chain_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < TI; ++t) {
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[t].x, in[t][0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[t].y, in[t][1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[t].z, in[t][2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[t].w, in[t][3], acc, 0, 0, 0);
        }
        acc[0] += bb.x; acc[1] += bb.y; acc[2] += bb.z; acc[3] += bb.w;
        if (act) { acc[0] = elu1_sel(acc[0]); acc[1] = elu1_sel(acc[1]); acc[2] = elu1_sel(acc[2]); acc[3] = elu1_sel(acc[3]); }
        out[to] = acc;
    }
}
#endif

// NI / NH / NO: input / hidden / output width in 16-feature tiles (compile time: the code is branch-free)
template <int NI, int NH, int NO, class ROWOFF, class EPI>
CCSD_DEV void mlp_chain_tile(const MlpD& m, const float* __restrict__ wp, const float* X, int ldx, const float* X2,
                             int ksplit, int p0, int rows, ROWOFF rowoff, EPI epi) {
#ifdef CCSD_EMU
    constexpr int MAXW = 16 * (NI > NH ? (NI > NO ? NI : NO) : (NH > NO ? NH : NO));
    for (int rr = 0; rr < 16; ++rr) {
        const int row = p0 + rr;
        if (row >= rows) break;
        float a[MAXW], t[MAXW];
        for (int k = 0; k < MAXW; ++k) a[k] = 0.f;
        const int roff = rowoff(row);
        for (int k = 0; k < m.in; ++k) a[k] = k < ksplit ? X[k * ldx + roff] : X2[(k - ksplit) * ldx + roff];
        for (int i = 0; i < m.n; ++i) {
            const int ip = 16 * (i == 0 ? NI : NH), op = 16 * (i == m.n - 1 ? NO : NH);
            const float* W = wp + m.pw[i];
            const float* Bv = wp + m.pb[i];
            for (int o = 0; o < op; ++o) {
                float acc = 0.f;
                for (int k = 0; k < ip; ++k) acc = fmaf(W[o * ip + k], a[k], acc);
                acc += Bv[o];
                t[o] = (i < m.n - 1) ? elu1(acc) : acc;
            }
            for (int o = 0; o < op; ++o) a[o] = t[o];
        }
        for (int f = 0; f < m.out; ++f) epi(row, f, a[f]);
    }
#else
    // The lane id is made opaque here: otherwise the per-lane index / address arithmetic of EVERY shape instantiated in a
    // kernel is hoisted above the shape dispatch and spilled to scratch (42 MB of spill writes per k_xa launch, PMC).
    int lane = threadIdx.x & 63;
    asm volatile("" : "+v"(lane));
    const int l15 = lane & 15, kq = lane >> 4;
    const int prow = rowoff((p0 + l15 < rows) ? p0 + l15 : rows - 1);      // clamped: rows beyond `rows` are never stored
    const int in = m.in;
    chain_f32x4 xin[NI], h0[NH], h1[NH], yo[NO];
#pragma unroll
    for (int t = 0; t < NI; ++t) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = 16 * t + 4 * kq + j;
            const int kc = k < in ? k : in - 1;                    // padded features meet zero weights: any finite value
            const float* src = kc < ksplit ? X + kc * ldx : X2 + (kc - ksplit) * ldx;
            xin[t][j] = src[prow];
        }
    }
    const size_t lo = (size_t)l15;
    if (m.n == 1) {
        chain_layer<NI, NO>(wp + m.pw[0] + lo * (16 * NI) + 4 * kq, wp + m.pb[0] + 4 * kq, 16 * NI, false, xin, yo);
    } else {
        chain_layer<NI, NH>(wp + m.pw[0] + lo * (16 * NI) + 4 * kq, wp + m.pb[0] + 4 * kq, 16 * NI, true, xin, h0);
        for (int i = 1; i < m.n - 1; ++i) {
            chain_layer<NH, NH>(wp + m.pw[i] + lo * (16 * NH) + 4 * kq, wp + m.pb[i] + 4 * kq, 16 * NH, true, h0, h1);
#pragma unroll
            for (int t = 0; t < NH; ++t) h0[t] = h1[t];
        }
        const int il = m.n - 1;
        if (NO == 1 && m.out == 1) {
            // a single output feature: 16 of 16 MFMA rows would be padding -- dot product on the VALU instead; the lane
            // holds features 16t + 4kq + r of its row, the four kq groups are summed with two cross-lane adds
            const float* W3 = wp + m.pw[il] + 4 * kq;
            float d = 0.f;
#pragma unroll
            for (int t = 0; t < NH; ++t) {
                const float4 wv = *reinterpret_cast<const float4*>(W3 + 16 * t);
                d = fmaf(wv.x, h0[t][0], d); d = fmaf(wv.y, h0[t][1], d); d = fmaf(wv.z, h0[t][2], d); d = fmaf(wv.w, h0[t][3], d);
            }
            d += __shfl_xor(d, 16, 64);
            d += __shfl_xor(d, 32, 64);
            d += wp[m.pb[il]];
            if (kq == 0 && p0 + l15 < rows) epi(p0 + l15, 0, d);
            return;
        }
        chain_layer<NH, NO>(wp + m.pw[il] + lo * (16 * NH) + 4 * kq, wp + m.pb[il] + 4 * kq, 16 * NH, false, h0, yo);
    }
    const bool rok = p0 + l15 < rows;
#pragma unroll
    for (int to = 0; to < NO; ++to)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int f = 16 * to + 4 * kq + r;
            if (f < m.out && rok) epi(p0 + l15, f, yo[to][r]);
        }
#endif
}
// all 16-row tiles of `rows`, round-robin over the waves of the workgroup
template <int NI, int NH, int NO, class ROWOFF, class EPI>
CCSD_DEV void mlp_chain(const MlpD& m, const float* __restrict__ wp, const float* X, int ldx, const float* X2, int ksplit,
                        int rows, ROWOFF rowoff, EPI epi) {
#ifdef CCSD_EMU
    const int wave = 0, nw = 1;
#else
    const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
#endif
    for (int tile = wave; tile < (rows + 15) >> 4; tile += nw) mlp_chain_tile<NI, NH, NO>(m, wp, X, ldx, X2, ksplit, 16 * tile, rows, rowoff, epi);
}

// ---------------------------------------------------------------------------------------------
// gcn_tile: one 16-column tile of a DenseGCNConv (layers.py:139-158) for ALL nodes of one graph, by one wave:
//   out[i][col] = dinv_i * sum_j A'_ij * ( dinv_j * sum_k x[j][k] W[k][col] ) + b[col],   A' = A with unit diagonal.
// Both products run on v_mfma_f32_16x16x4_f32 and the intermediate x W never leaves registers: the first
// product's accumulator element r of lane (l15, kq) is (node 16*tn + 4*kq + r, column l15) -- with the second
// product's k slot kq of step (tn, j) assigned to node 16*tn + 4*kq + j it IS that product's B operand.
// xT: LDS, feature-major [k][ldn].  A: [N][N] (LDS or the HBM channel stack).  wf(k, col) / bf(col): weight / bias.
// NTN = ceil(N / 16) node tiles (compile time).
// ---------------------------------------------------------------------------------------------
// BATCH: load the weights of four k-steps ahead of their MFMAs (large graphs; off in the small-graph instantiations, whose
// code is then exactly the plain loop).
template <int NTN, bool BATCH, class WF, class BF, class OUT>
CCSD_DEV void gcn_tile(const float* xT, int ldn, int fin, int N, const float* A, const float* dinv, int col0, int ncols,
                       WF wf, BF bf, OUT out) {
#ifdef CCSD_EMU
    for (int cc = 0; cc < 16; ++cc) {
        const int col = col0 + cc;
        if (col >= ncols) break;
        float xw[16 * NTN];
        for (int j = 0; j < N; ++j) {
            float acc = 0.f;
            for (int k = 0; k < fin; ++k) acc = fmaf(xT[k * ldn + j], wf(k, col), acc);
            xw[j] = acc * dinv[j];
        }
        for (int i = 0; i < N; ++i) {
            float acc = 0.f;
            for (int j = 0; j < N; ++j) acc = fmaf((i == j) ? 1.f : A[i * N + j], xw[j], acc);
            out(i, col, fmaf(acc, dinv[i], bf(col)));
        }
    }
#else
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    int lane = threadIdx.x & 63;
    asm volatile("" : "+v"(lane));          // keeps each instantiation's index arithmetic inside it (see mlp_chain_tile)
    const int l15 = lane & 15, kq = lane >> 4;
    const int col = col0 + l15;
    const bool cok = col < ncols;
    const int colc = cok ? col : ncols - 1;
    f32x4 xw[NTN];
#pragma unroll
    for (int tn = 0; tn < NTN; ++tn) xw[tn] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int ks = (fin + 3) >> 2;
    if (NTN == 1 || !BATCH) {       // small graphs (fin of a few k-steps): the plain loop is as fast and lighter on registers
        for (int s0 = 0; s0 < ks; ++s0) {
            const int k = 4 * s0 + kq, kc = k < fin ? k : fin - 1;
            const float bw = wf(kc, colc);
            const float bv = (k < fin && cok) ? bw : 0.f;
#pragma unroll
            for (int tn = 0; tn < NTN; ++tn) {
                const int j = 16 * tn + l15;
                const float av = xT[kc * ldn + (j < N ? j : N - 1)];
                xw[tn] = __builtin_amdgcn_mfma_f32_16x16x4f32(j < N ? av : 0.f, bv, xw[tn], 0, 0, 0);
            }
        }
    } else
    // the weights of four k-steps are loaded back to back ahead of their MFMAs: one L2 round trip per four steps instead of one
    // per step (the loads are the critical path of a task)
    for (int s00 = 0; s00 < ks; s00 += 4) {
        float bw[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = 4 * (s00 + u) + kq;
            bw[u] = wf(k < fin ? k : fin - 1, colc);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (s00 + u < ks) {
                const int k = 4 * (s00 + u) + kq, kc = k < fin ? k : fin - 1;
                const float bv = (k < fin && cok) ? bw[u] : 0.f;
#pragma unroll
                for (int tn = 0; tn < NTN; ++tn) {
                    const int j = 16 * tn + l15;
                    const float av = xT[kc * ldn + (j < N ? j : N - 1)];
                    xw[tn] = __builtin_amdgcn_mfma_f32_16x16x4f32(j < N ? av : 0.f, bv, xw[tn], 0, 0, 0);
                }
            }
        }
    }
#pragma unroll
    for (int tn = 0; tn < NTN; ++tn)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = 16 * tn + 4 * kq + r;
            const float dj = dinv[j < N ? j : N - 1];
            xw[tn][r] = j < N ? xw[tn][r] * dj : 0.f;
        }
    const float bb = bf(colc);
#pragma unroll
    for (int ti = 0; ti < NTN; ++ti) {
        const int i = 16 * ti + l15, ic = i < N ? i : N - 1;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tn = 0; tn < NTN; ++tn)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int j = 16 * tn + 4 * kq + jj, jc = j < N ? j : N - 1;
                const float a0 = A[ic * N + jc];
                const float av = (i < N && j < N) ? (i == j ? 1.f : a0) : 0.f;
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, xw[tn][jj], acc, 0, 0, 0);
            }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int io = 16 * ti + 4 * kq + r;
            if (io < N && cok) out(io, col, fmaf(acc[r], dinv[io], bb));
        }
    }
#endif
}
template <bool BATCH, class WF, class BF, class OUT>
CCSD_DEV void gcn_tile_n(const float* xT, int ldn, int fin, int N, const float* A, const float* dinv, int col0, int ncols,
                         WF wf, BF bf, OUT out) {
    if (N <= 16) gcn_tile<1, BATCH>(xT, ldn, fin, N, A, dinv, col0, ncols, wf, bf, out);
    else if (N <= 32) gcn_tile<2, BATCH>(xT, ldn, fin, N, A, dinv, col0, ncols, wf, bf, out);
    else if (N <= 48) gcn_tile<3, BATCH>(xT, ldn, fin, N, A, dinv, col0, ncols, wf, bf, out);
    else gcn_tile<4, BATCH>(xT, ldn, fin, N, A, dinv, col0, ncols, wf, bf, out);
}

// ---------------------------------------------------------------------------------------------
// 64x64 output tile engine for the rank-2 contractions.  LDS slabs As[BK][TLD] (k-major, m fast)
// and Bs[BK][TLD] (k-major, n fast); 4 waves as 2x2, each wave 32x32 = 2x2 MFMA 16x16x4 tiles.
// ---------------------------------------------------------------------------------------------
#define T_BM 64
#define T_BN 64
#define T_BK 32
#define T_LD 80  // 64 + 16: lanes l and l+16 (next k) land on disjoint banks

struct TileAcc {
#ifdef CCSD_EMU
    float a[T_BM][T_BN];
#else
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    f32x4 a[2][2];
#endif
};
CCSD_DEV void tile_zero(TileAcc& t) {
#ifdef CCSD_EMU
    for (int i = 0; i < T_BM; ++i)
        for (int j = 0; j < T_BN; ++j) t.a[i][j] = 0.f;
#else
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) t.a[i][j] = (TileAcc::f32x4){0.f, 0.f, 0.f, 0.f};
#endif
}
CCSD_DEV void tile_mma(TileAcc& t, const float* As, const float* Bs) {
#ifdef CCSD_EMU
    for (int i = 0; i < T_BM; ++i)
        for (int j = 0; j < T_BN; ++j) {
            float acc = t.a[i][j];
            for (int k = 0; k < T_BK; ++k) acc = fmaf(As[k * T_LD + i], Bs[k * T_LD + j], acc);
            t.a[i][j] = acc;
        }
#else
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32, l15 = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int s = 0; s < T_BK / 4; ++s) {
        const float* ar = As + (4 * s + kq) * T_LD + wm + l15;
        const float* br = Bs + (4 * s + kq) * T_LD + wn + l15;
        const float a0 = ar[0], a1 = ar[16], b0 = br[0], b1 = br[16];
        t.a[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, t.a[0][0], 0, 0, 0);
        t.a[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b1, t.a[0][1], 0, 0, 0);
        t.a[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b0, t.a[1][0], 0, 0, 0);
        t.a[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, t.a[1][1], 0, 0, 0);
    }
#endif
}
// visit the accumulator in groups of four consecutive rows: f(m_local (multiple of 4), n_local, v[4])
template <class Fn>
CCSD_DEV void tile_foreach4(TileAcc& t, Fn f) {
#ifdef CCSD_EMU
    for (int i = 0; i < T_BM; i += 4)
        for (int j = 0; j < T_BN; ++j) {
            float v[4] = {t.a[i][j], t.a[i + 1][j], t.a[i + 2][j], t.a[i + 3][j]};
            f(i, j, v);
        }
#else
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32, l15 = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float v[4] = {t.a[i][j][0], t.a[i][j][1], t.a[i][j][2], t.a[i][j][3]};
            f(wm + 16 * i + 4 * kq, wn + 16 * j + l15, v);
        }
#endif
}

// ---------------------------------------------------------------------------------------------
// k_flagbits: offbits[b] has bit n set iff flags[b][n] == 0  (get_rank2_flags tests `flags == 0`,
// cc_utils.py:549)
// ---------------------------------------------------------------------------------------------
__global__ void k_flagbits(const float* __restrict__ flags, unsigned long long* __restrict__ offbits, int B, int N) {
    for (int b = blockIdx.x * blockDim.x + threadIdx.x; b < B; b += gridDim.x * blockDim.x) {
        unsigned long long m = 0;
        for (int n = 0; n < N; ++n)
            if (flags[(size_t)b * N + n] == 0.f) m |= 1ull << n;
        offbits[b] = m;
    }
}
CCSD_DEV float edge_on(unsigned long long off, const unsigned char* __restrict__ edges, int e) {
    return ((off >> edges[2 * e]) | (off >> edges[2 * e + 1])) & 1ull ? 0.f : 1.f;
}
CCSD_DEV float cell_on(unsigned long long off, const unsigned long long* __restrict__ cells, int k) {
    return (cells[k] & off) ? 0.f : 1.f;
}

// ---------------------------------------------------------------------------------------------
// k_gemm_h: H[b] = (F[b] F[b]^T) * hodge_mask           hodge_laplacian + mask, cc_utils.py:929, 964-969
// grid (ceil(E/64), ceil(E/64), B)
// ---------------------------------------------------------------------------------------------
#define H_BK 32   // k per slab (two 16-wide MFMA k blocks)
#define H_LD 40   // LDS row stride in floats: 16-byte aligned rows, == 8 mod 32 -> conflict-free ds_read_b128 fragments
__global__ __launch_bounds__(256) void k_gemm_h(const float* __restrict__ rank2, float* __restrict__ H, int E, int K,
                                                int zero_diag) {
    const int b = blockIdx.z, m0 = blockIdx.y * T_BM, n0 = blockIdx.x * T_BN;
    if (blockIdx.x < blockIdx.y) return;        // H is symmetric: upper-triangle tiles only, mirrored on store
    const float* Fb = rank2 + (size_t)b * E * K;
    TileAcc acc;
    tile_zero(acc);
#ifdef CCSD_EMU
    static float As[T_BK * T_LD], Bs[T_BK * T_LD];
    for (int k0 = 0; k0 < K; k0 += T_BK) {
        for (int idx = threadIdx.x; idx < T_BM * T_BK; idx += blockDim.x) {
            const int r = idx / T_BK, kk = idx % T_BK, k = k0 + kk;
            const int ra = m0 + r, rb = n0 + r;
            As[kk * T_LD + r] = (ra < E && k < K) ? Fb[(size_t)ra * K + k] : 0.f;
            Bs[kk * T_LD + r] = (rb < E && k < K) ? Fb[(size_t)rb * K + k] : 0.f;
        }
        tile_mma(acc, As, Bs);
    }
#else
    // Both operands are rows of F, contiguous along the contraction index: the slabs are straight row copies
    // (As[row][k], 16-byte vectors, no transposition), and with the MFMA k slot kq of step j of a 16-wide block assigned
    // to k = 16*t + 4*kq + j a lane's four-step fragment is one ds_read_b128.  The next slab's global loads are issued
    // before the MFMAs of the current one.
    __shared__ __align__(16) float As[T_BM * H_LD];
    __shared__ __align__(16) float Bs[T_BN * H_LD];
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l15 = lane & 15, kq = lane >> 4;
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
    const bool diag = blockIdx.x == blockIdx.y, vec = (K & 3) == 0;
    // thread -> (row, 4-float column group) of the 64 x 32 slab: two groups per thread and matrix
    const int r0 = tid >> 3, c4 = (tid & 7) * 4;
    auto ldg = [&](int row, int k) -> float4 {
        const int rc = row < E ? row : E - 1;
        const float* src = Fb + (size_t)rc * K;
        float4 v;
        if (vec && k + 3 < K) v = *reinterpret_cast<const float4*>(src + k);
        else {
            v.x = k < K ? src[k] : 0.f; v.y = k + 1 < K ? src[k + 1] : 0.f;
            v.z = k + 2 < K ? src[k + 2] : 0.f; v.w = k + 3 < K ? src[k + 3] : 0.f;
        }
        if (row >= E) v = make_float4(0.f, 0.f, 0.f, 0.f);
        return v;
    };
    float4 ra[2], rb[2];
    ra[0] = ldg(m0 + r0, c4); ra[1] = ldg(m0 + r0 + 32, c4);
    if (!diag) { rb[0] = ldg(n0 + r0, c4); rb[1] = ldg(n0 + r0 + 32, c4); }
    const float* Bp = diag ? As : Bs;
    for (int k0 = 0; k0 < K; k0 += H_BK) {
        __syncthreads();                                   // the previous slab's MFMAs are done reading LDS
        *reinterpret_cast<float4*>(As + r0 * H_LD + c4) = ra[0];
        *reinterpret_cast<float4*>(As + (r0 + 32) * H_LD + c4) = ra[1];
        if (!diag) {
            *reinterpret_cast<float4*>(Bs + r0 * H_LD + c4) = rb[0];
            *reinterpret_cast<float4*>(Bs + (r0 + 32) * H_LD + c4) = rb[1];
        }
        __syncthreads();
        if (k0 + H_BK < K) {                               // next slab: in flight during the MFMAs
            ra[0] = ldg(m0 + r0, k0 + H_BK + c4); ra[1] = ldg(m0 + r0 + 32, k0 + H_BK + c4);
            if (!diag) { rb[0] = ldg(n0 + r0, k0 + H_BK + c4); rb[1] = ldg(n0 + r0 + 32, k0 + H_BK + c4); }
        }
#pragma unroll
        for (int t = 0; t < H_BK / 16; ++t) {
            const float4 a0 = *reinterpret_cast<const float4*>(As + (wm + l15) * H_LD + 16 * t + 4 * kq);
            const float4 a1 = *reinterpret_cast<const float4*>(As + (wm + 16 + l15) * H_LD + 16 * t + 4 * kq);
            const float4 b0 = *reinterpret_cast<const float4*>(Bp + (wn + l15) * H_LD + 16 * t + 4 * kq);
            const float4 b1 = *reinterpret_cast<const float4*>(Bp + (wn + 16 + l15) * H_LD + 16 * t + 4 * kq);
            const float av0[4] = {a0.x, a0.y, a0.z, a0.w}, av1[4] = {a1.x, a1.y, a1.z, a1.w};
            const float bv0[4] = {b0.x, b0.y, b0.z, b0.w}, bv1[4] = {b1.x, b1.y, b1.z, b1.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc.a[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av0[j], bv0[j], acc.a[0][0], 0, 0, 0);
                acc.a[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av0[j], bv1[j], acc.a[0][1], 0, 0, 0);
                acc.a[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av1[j], bv0[j], acc.a[1][0], 0, 0, 0);
                acc.a[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av1[j], bv1[j], acc.a[1][1], 0, 0, 0);
            }
        }
    }
#endif
    float* Hb = H + (size_t)b * E * E;
    tile_foreach4(acc, [&](int ml, int nl, const float* v) {
        const int n = n0 + nl;
        if (n >= E) return;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int m = m0 + ml + s;
            if (m < E) {
                const float hv = (zero_diag && m == n) ? 0.f : v[s];
                Hb[(size_t)m * E + n] = hv;
                if (blockIdx.x != blockIdx.y) Hb[(size_t)n * E + m] = hv;
            }
        }
    });
}

// ---------------------------------------------------------------------------------------------
// k_gemm_p: P[r][c] = sum_k A(r,k) * Wcat[k][c]   over the flattened rows r = b*E + e.
// layer 0: A = rank2 as given                               (DenseHCNConv out = rank2 @ W, hodge_layers.py:185)
// layer 1: A = rank2' = mask_rank2(mlp_value(stack_c a_c[e]*rank2[e,k]))   (hodge_attention.py:107,322-323
//          with the layer-0 hodge adjacency diagonal, cc_utils.py:1536) -- produced on the fly, never stored.
// grid (ceil(wc/64), ceil(B*E/64), 1)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_gemm_p(const float* __restrict__ rank2, const float* __restrict__ W,
                                                float* __restrict__ P, int rows, int E, int K, int wc, int wcat_off,
                                                int layer, MlpD mval, int cin, const float* __restrict__ acoef,
                                                const unsigned long long* __restrict__ offbits,
                                                const unsigned char* __restrict__ edges,
                                                const unsigned long long* __restrict__ cells) {
    __shared__ float As[T_BK * T_LD];
    __shared__ float Bs[T_BK * T_LD];
    __shared__ float s_mv[CCSD_MAXLIN * CCSD_HWBLK];   // mlp_value as zero-padded 8x8 blocks (LDS broadcast reads)
    const int m0 = blockIdx.y * T_BM, n0 = blockIdx.x * T_BN;
    const float* Wc = W + wcat_off;
    TileAcc acc;
    tile_zero(acc);
    if (layer == 1) { stage_mlp_blocks(mval, W, s_mv); __syncthreads(); }
    for (int k0 = 0; k0 < K; k0 += T_BK) {
        for (int idx = threadIdx.x; idx < T_BM * T_BK; idx += blockDim.x) {
            const int r = idx / T_BK, kk = idx % T_BK, k = k0 + kk, row = m0 + r;
            float v = 0.f;
            if (row < rows && k < K) {
                v = rank2[(size_t)row * K + k];
                if (layer == 1) {
                    const int b = row / E, e = row % E;
                    const unsigned long long off = offbits[b];
                    float in[CCSD_SMALLW], out[CCSD_SMALLW];
#pragma unroll
                    for (int c = 0; c < CCSD_SMALLW; ++c) in[c] = c < cin ? acoef[((size_t)b * cin + c) * E + e] * v : 0.f;
                    small_mlp_lds<CCSD_SMALLW>(s_mv, mval.n, in, out);
                    v = edge_on(off, edges, e) * out[0] * cell_on(off, cells, k);
                }
            }
            As[kk * T_LD + r] = v;
        }
        for (int idx = threadIdx.x; idx < T_BK * T_BN; idx += blockDim.x) {
            const int kk = idx / T_BN, c = idx % T_BN, k = k0 + kk, col = n0 + c;
            Bs[kk * T_LD + c] = (k < K && col < wc) ? Wc[(size_t)k * wc + col] : 0.f;
        }
        __syncthreads();
        tile_mma(acc, As, Bs);
        __syncthreads();
    }
    tile_foreach4(acc, [&](int ml, int nl, const float* v) {
        const int n = n0 + nl;
        if (n >= wc) return;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int m = m0 + ml + s;
            if (m < rows) P[(size_t)m * wc + n] = v[s];
        }
    });
}

// ---------------------------------------------------------------------------------------------
// k_gemm_p0: layer-0 hodge projections  P_0[r][c] = sum_k rank2[r][k] * Wcat_0[k][c]  (DenseHCNConv out = rank2 @ W,
// hodge_layers.py:185) over the flattened rows r = b*E + e, for narrow outputs (wc <= 64: 16 columns for every shipped
// network).  One workgroup = 64 rows x all pad16(wc) columns: both operands are row copies (rank2 rows, rows of the
// transposed packed weights Wcat^T[col][Kp]) read back as ds_read_b128 permuted-k fragments; wave w owns rows
// 16w..16w+15 and every 16-column tile, so no MFMA is spent on the 64-column padding of the general tile engine.
// ---------------------------------------------------------------------------------------------
#ifndef CCSD_EMU
template <int NT>
__global__ __launch_bounds__(256) void k_gemm_p0(const float* __restrict__ rank2, const float* __restrict__ WT, float* __restrict__ P,
                                                 int rows, int K, int Kp, int wc) {
    __shared__ __align__(16) float As[T_BM * H_LD];
    __shared__ __align__(16) float Bs[16 * NT * H_LD];
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l15 = lane & 15, kq = lane >> 4;
    const int m0 = blockIdx.x * T_BM;
    const bool vec = (K & 3) == 0;
    const int r0 = tid >> 3, c4 = (tid & 7) * 4;           // (row, 4-float column group) of a 64 x 32 slab; rows r0, r0 + 32
    auto lda = [&](int row, int k) -> float4 {
        const float* src = rank2 + (size_t)(row < rows ? row : rows - 1) * K;
        float4 v;
        if (vec && k + 3 < K) v = *reinterpret_cast<const float4*>(src + k);
        else {
            v.x = k < K ? src[k] : 0.f; v.y = k + 1 < K ? src[k + 1] : 0.f;
            v.z = k + 2 < K ? src[k + 2] : 0.f; v.w = k + 3 < K ? src[k + 3] : 0.f;
        }
        return v;
    };
    f32x4 acc[NT];
#pragma unroll
    for (int c = 0; c < NT; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float4 ra[2], rb[(NT + 1) / 2];
    auto load_slab = [&](int k0) {
        ra[0] = lda(m0 + r0, k0 + c4); ra[1] = lda(m0 + r0 + 32, k0 + c4);
#pragma unroll
        for (int u = 0; u < (NT + 1) / 2; ++u) {           // 16*NT weight rows x 8 float4: tid + 256u < 128*NT
            const int idx = tid + 256 * u, wr = idx >> 3;
            rb[u] = wr < 16 * NT ? *reinterpret_cast<const float4*>(WT + (size_t)wr * Kp + k0 + (idx & 7) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    load_slab(0);
    for (int k0 = 0; k0 < Kp; k0 += H_BK) {
        __syncthreads();
        *reinterpret_cast<float4*>(As + r0 * H_LD + c4) = ra[0];
        *reinterpret_cast<float4*>(As + (r0 + 32) * H_LD + c4) = ra[1];
#pragma unroll
        for (int u = 0; u < (NT + 1) / 2; ++u) {
            const int idx = tid + 256 * u, wr = idx >> 3;
            if (wr < 16 * NT) *reinterpret_cast<float4*>(Bs + wr * H_LD + (idx & 7) * 4) = rb[u];
        }
        __syncthreads();
        if (k0 + H_BK < Kp) load_slab(k0 + H_BK);
#pragma unroll
        for (int t = 0; t < H_BK / 16; ++t) {
            const float4 a = *reinterpret_cast<const float4*>(As + (16 * wave + l15) * H_LD + 16 * t + 4 * kq);
#pragma unroll
            for (int c = 0; c < NT; ++c) {
                const float4 bq = *reinterpret_cast<const float4*>(Bs + (16 * c + l15) * H_LD + 16 * t + 4 * kq);
                acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, bq.x, acc[c], 0, 0, 0);
                acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, bq.y, acc[c], 0, 0, 0);
                acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, bq.z, acc[c], 0, 0, 0);
                acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, bq.w, acc[c], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int c = 0; c < NT; ++c) {
        const int n = 16 * c + l15;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + 16 * wave + 4 * kq + r;
            if (m < rows && n < wc) P[(size_t)m * wc + n] = acc[c][r];
        }
    }
}
#endif

// ---------------------------------------------------------------------------------------------
// k_edgecoef: acoef[b][c][e] = (adj^(c+1))[i_e][j_e]      pow_tensor + adj_to_hodgedual,
// graph_utils.py:285-292, cc_utils.py:1525-1536.  One workgroup per graph; LDS: 3*N*N floats.
// ---------------------------------------------------------------------------------------------
__global__ void k_edgecoef(const float* __restrict__ adj, float* __restrict__ acoef, int N, int E, int cinit,
                           const unsigned char* __restrict__ edges) {
    CCSD_DYN_SMEM(sm);
    float* A = sm;
    float* P0 = sm + N * N;
    float* P1 = sm + 2 * N * N;
    const int b = blockIdx.x, NN = N * N;
    for (int i = threadIdx.x; i < NN; i += blockDim.x) { A[i] = adj[(size_t)b * NN + i]; P0[i] = A[i]; }
    __syncthreads();
    for (int c = 0; c < cinit; ++c) {
        for (int e = threadIdx.x; e < E; e += blockDim.x)
            acoef[((size_t)b * cinit + c) * E + e] = P0[edges[2 * e] * N + edges[2 * e + 1]];
        if (c + 1 < cinit) {
            for (int i = threadIdx.x; i < NN; i += blockDim.x) {
                const int r = i / N, cc = i % N;
                float acc = 0.f;
                for (int k = 0; k < N; ++k) acc = fmaf(P0[r * N + k], A[k * N + cc], acc);
                P1[i] = acc;
            }
            __syncthreads();
            float* t = P0; P0 = P1; P1 = t;
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// k_hf_score: ScoreNetworkF.  Tile (edge rows m0.., cell columns n0..) of  H.F  on MFMA, then per
// element the channel MLP stack of ScoreNetwork_F.py:198-217 and one of three fused epilogues.
// grid (ceil(K/64), ceil(E/64), B)
// ---------------------------------------------------------------------------------------------
enum { MODE_SCORE = 0, MODE_NORMS = 1, MODE_PRED = 2 };

struct RankEpi {
    int mode;
    float sscale;            // MODE_SCORE: out = sscale * net
    float pa, pb, pc;        // MODE_PRED
    float* out;              // SCORE: score; NORMS: raw net output (kept for the apply pass); PRED: new state
    float* mean;             // PRED: nullable
    float* part;             // NORMS: [B][ntiles][2] partial sums of net^2 and z^2
};

// FW: width the per-element MLPs are padded to (8 when every layer of the network fits, else CCSD_FW = 16)
template <bool AFFINE, int FW = CCSD_FW>
CCSD_DEV float fnet_element(const PlanD& p, const float* __restrict__ w, float f, float hf, float m) {
    if (AFFINE) return m * fmaf(p.f_alpha, f, fmaf(p.f_beta, hf, p.f_gamma));
    if (p.f_blk >= 0) {
        // every layer <= 8 wide, single-Linear head: zero-padded blocks behind the weight blob (ccsd_pack_fnet_blocks), read
        // with wide scalar loads; each layer's output stays in its own registers and the head is accumulated segment by
        // segment in concat order (no dynamic register indexing, padded lanes contribute exact zeros)
        const float* fb = w + p.f_blk;
        const float* hd = fb + CCSD_FBLK_HEAD;
        float prev[8] = {f, p.f_cnum == 2 ? hf : 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
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
    if (p.f_cnum == 2) ch[1] = hf;
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

template <bool AFFINE, int FW>
__global__ __launch_bounds__(256) void k_hf_score(const PlanD* __restrict__ plan, const float* __restrict__ w,
                                                  const float* __restrict__ rank2, const float* __restrict__ H,
                                                  const unsigned long long* __restrict__ offbits,
                                                  const unsigned char* __restrict__ edges,
                                                  const unsigned long long* __restrict__ cells, RankEpi ep,
                                                  NoiseArgs na) {
    __shared__ float red[64];
    const PlanD& p = *plan;
    const int E = p.E, K = p.K;
    const int b = blockIdx.z, m0 = blockIdx.y * T_BM, n0 = blockIdx.x * T_BN;
    const float* Fb = rank2 + (size_t)b * E * K;
    const float* Hb = H + (size_t)b * E * E;
    TileAcc acc;
    tile_zero(acc);
#ifdef CCSD_EMU
    static float As[T_BK * T_LD], Bs[T_BK * T_LD];
    if (p.f_cnum == 2) {
        for (int k0 = 0; k0 < E; k0 += T_BK) {
            for (int idx = threadIdx.x; idx < T_BM * T_BK; idx += blockDim.x) {
                const int r = idx / T_BK, kk = idx % T_BK, k = k0 + kk, row = m0 + r;
                As[kk * T_LD + r] = (row < E && k < E) ? Hb[(size_t)row * E + k] : 0.f;
            }
            for (int idx = threadIdx.x; idx < T_BK * T_BN; idx += blockDim.x) {
                const int kk = idx / T_BN, c = idx % T_BN, k = k0 + kk, col = n0 + c;
                Bs[kk * T_LD + c] = (k < E && col < K) ? Fb[(size_t)k * K + col] : 0.f;
            }
            tile_mma(acc, As, Bs);
        }
    }
#else
    // (H F) tile: A = rows of H (contraction index contiguous: row-copy slab As[row][k], one ds_read_b128 per 16-wide k
    // block with the permuted k slots k = 16t + 4kq + j); B = rows of F, k-major slab Bs[k][col] read with the same
    // permutation (row stride 68: 4 * 68 == 16 mod 32 keeps the four kq groups on disjoint banks).  Next slab's global
    // loads are issued before the MFMAs of the current one.
    constexpr int BLD = 68;
    __shared__ __align__(16) float As[T_BM * H_LD];
    __shared__ __align__(16) float Bs[H_BK * BLD];
    if (p.f_cnum == 2) {
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l15 = lane & 15, kq = lane >> 4;
        const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
        const bool vec = (K & 3) == 0;
        // A slab 64 x 32: thread -> (row ar + 8u, column ak), u < 8 (scalar: E is not 16-byte friendly in general)
        const int ar = tid >> 5, ak = tid & 31;
        // B slab 32 x 64: thread -> (k row bk + 16u, 4-float column group bc4), u < 2
        const int bk = tid >> 4, bc4 = (tid & 15) * 4;
        float ra[8];
        float4 rb[2];
        auto load_slab = [&](int k0) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int row = m0 + ar + 8 * u, k = k0 + ak;
                const float v = Hb[(size_t)(row < E ? row : E - 1) * E + (k < E ? k : E - 1)];
                ra[u] = (row < E && k < E) ? v : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int k = k0 + bk + 16 * u, col = n0 + bc4;
                const float* src = Fb + (size_t)(k < E ? k : E - 1) * K;
                float4 v;
                if (vec && col + 3 < K) v = *reinterpret_cast<const float4*>(src + col);
                else {
                    v.x = col < K ? src[col] : 0.f; v.y = col + 1 < K ? src[col + 1] : 0.f;
                    v.z = col + 2 < K ? src[col + 2] : 0.f; v.w = col + 3 < K ? src[col + 3] : 0.f;
                }
                if (k >= E) v = make_float4(0.f, 0.f, 0.f, 0.f);
                rb[u] = v;
            }
        };
        load_slab(0);
        for (int k0 = 0; k0 < E; k0 += H_BK) {
            __syncthreads();
#pragma unroll
            for (int u = 0; u < 8; ++u) As[(ar + 8 * u) * H_LD + ak] = ra[u];
#pragma unroll
            for (int u = 0; u < 2; ++u) *reinterpret_cast<float4*>(Bs + (bk + 16 * u) * BLD + bc4) = rb[u];
            __syncthreads();
            if (k0 + H_BK < E) load_slab(k0 + H_BK);
#pragma unroll
            for (int t = 0; t < H_BK / 16; ++t) {
                const float4 a0 = *reinterpret_cast<const float4*>(As + (wm + l15) * H_LD + 16 * t + 4 * kq);
                const float4 a1 = *reinterpret_cast<const float4*>(As + (wm + 16 + l15) * H_LD + 16 * t + 4 * kq);
                const float av0[4] = {a0.x, a0.y, a0.z, a0.w}, av1[4] = {a1.x, a1.y, a1.z, a1.w};
                float bv0[4], bv1[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float* br = Bs + (16 * t + 4 * kq + j) * BLD + wn + l15;
                    bv0[j] = br[0]; bv1[j] = br[16];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc.a[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av0[j], bv0[j], acc.a[0][0], 0, 0, 0);
                    acc.a[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av0[j], bv1[j], acc.a[0][1], 0, 0, 0);
                    acc.a[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av1[j], bv0[j], acc.a[1][0], 0, 0, 0);
                    acc.a[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av1[j], bv1[j], acc.a[1][1], 0, 0, 0);
                }
            }
        }
    }
#endif
    const unsigned long long off = offbits[b];
    float s_net = 0.f, s_z = 0.f;
    tile_foreach4(acc, [&](int ml, int nl, const float* hf) {
        const int k = n0 + nl, e0 = m0 + ml;
        if (k >= K || e0 >= E) return;
        const float fr = cell_on(off, cells, k);
        float z[4] = {0.f, 0.f, 0.f, 0.f};
        if (ep.mode != MODE_SCORE) raw_noise_r4(na, b, e0 >> 2, k, E, K, z);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int e = e0 + s;
            if (e >= E) continue;
            const size_t gi = ((size_t)b * E + e) * K + k;
            const float f = Fb[(size_t)e * K + k];
            const float m = edge_on(off, edges, e) * fr;          // flags_left * flags_right, cc_utils.py:590
            const float net = fnet_element<AFFINE, FW>(p, w, f, hf[s], m);
            const float zz = z[s] * m;                            // gen_noise_rank2, cc_utils.py:613-615
            if (ep.mode == MODE_SCORE) {
                ep.out[gi] = ep.sscale * net;
            } else if (ep.mode == MODE_NORMS) {
                ep.out[gi] = net;
                s_net = fmaf(net, net, s_net);
                s_z = fmaf(zz, zz, s_z);
            } else {
                const float mean = fmaf(ep.pa, f, ep.pb * net);   // v_mean = pa*v + pb*net
                if (ep.mean) ep.mean[gi] = mean;
                ep.out[gi] = fmaf(ep.pc, zz, mean);
            }
        }
    });
    if (ep.mode == MODE_NORMS) {
        const float tn = block_sum(s_net, red);
        const float tz = block_sum(s_z, red);
        if (threadIdx.x == 0) {
            const int tile = blockIdx.y * gridDim.x + blockIdx.x, nt = gridDim.x * gridDim.y;
            ep.part[((size_t)b * nt + tile) * 2 + 0] = tn;
            ep.part[((size_t)b * nt + tile) * 2 + 1] = tz;
        }
    }
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
template <class LA, class LB, class EP>
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
    int s = 0;
    for (; s + 4 <= ks; s += 4) {          // issue the 8 operand loads of four k-steps before the MFMAs consume them
        float a[4], bv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { a[u] = la(m0 + l15, 4 * (s + u) + kq); bv[u] = lb(4 * (s + u) + kq, n0 + l15); }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], bv[u], acc, 0, 0, 0);
    }
    for (; s < ks; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(la(m0 + l15, 4 * s + kq), lb(4 * s + kq, n0 + l15), acc, 0, 0, 0);
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

struct R2Args {
    const float* rank2; const float* adj; const float* flags;
    const unsigned long long* offbits;     // per-sample bitmask of switched-off nodes
    float* P0; float* P1;
    int want_p;            // write the hodge projections (the A-network will run on the same state)
    int ldk, ldh;
    long long* dbg;
    const float* wp;       // packed buffer (Wcat^T of the hodge projections)
    CorrFuse cf;
};

// MT = ceil(E / 16) row tiles (1..4); RS (affine phase 2 only): plain MFMA steps covering E mod 16 behind the MT - 1 full
// 16-wide blocks of the contraction index (0: the last block is taken whole, zero padded -- E mod 16 == 0 or > 12);
// AFFINE: ScoreNetworkF folds to alpha F + beta HF + gamma; GEN1: general (non-affine) mlp_value in the hodge branch.
// Compile-time so that the common variant carries no general-path code.
template <int MT, int RS, bool AFFINE, bool GEN1>
__global__ __launch_bounds__(512, 4) void k_r2(const PlanD* __restrict__ plan, const float* __restrict__ w,
                                            const unsigned char* __restrict__ edges,
                                            const unsigned long long* __restrict__ cells, R2Args ra, RankEpi ep,
                                            NoiseArgs na) {
    CCSD_DYN_SMEM(sm);
    const PlanD& p = *plan;
    const int E = p.E, K = p.K, N = p.N, NN = N * N, ldk = ra.ldk, ldh = ra.ldh;
    const int b = blockIdx.x, tid = threadIdx.x, nth = blockDim.x;
    const int Kp4 = (K + 31) & ~31, Ep4 = (E + 3) & ~3;   // K is zero-padded to whole 8-step batches in LDS
    float* sF = sm;                        // [E][ldk]
    float* sH = sF + E * ldk;              // [E][ldh]
    float* sFl = sH + E * ldh;             // [64]  flags_left (edge masks)
    float* sRow = sFl + 64;                // [64]  per-row scale of rank2' (linear mlp_value)
    float* sAco = sRow + 64;               // [cinit][E] adjacency powers' upper triangle
    float* sAdj = sAco + p.a_cinit * E;    // 3 x [N*N] scratch for the powers
    float* sRed = sAdj + 3 * NN;           // [64]
    unsigned char* sFrb = reinterpret_cast<unsigned char*>(sRed + 64);   // [Kp4] flags_right (cell masks) as bytes
    __shared__ unsigned long long s_off;
#ifndef CCSD_EMU
    __shared__ int s_hdone;              // H-tile tasks finished (phase 1 -> 2 hand-over)
#endif
    const float* Fg = ra.rank2 + (size_t)b * E * K;
    const FastDiv dK(K);

    // ---- phase 0: rank2 block -> LDS; masks; adjacency powers
    stamp(ra.dbg, 0);
    if (tid == 0) {
        s_off = ra.offbits[b];                // switched-off nodes (k_flagbits): one load instead of a serial walk over the flags
#ifndef CCSD_EMU
        s_hdone = 0;
#endif
    }
    // With the fused Langevin apply (predictor launches of ccsd_sampler_run) the raw scores of the norms pass are loaded
    // alongside and F + c1*net goes to LDS in the same pass (same fma as k_langevin_apply; the noise term follows below).
    float c1f = 0.f, c2f = 0.f;
    if (ra.cf.on) corr_coef(ra.cf, 2, &c1f, &c2f);
    const float* Ng = ra.cf.on ? ra.cf.net_r + (size_t)b * E * K : Fg;
    if (((E * K) & 3) == 0) {
        // the block is 16-byte aligned and a multiple of 16 bytes: batches of four float4 loads in flight per thread
        const float4* F4 = reinterpret_cast<const float4*>(Fg);
        const float4* N4 = reinterpret_cast<const float4*>(Ng);
        const int n4 = (E * K) >> 2;
        for (int base = tid; base < n4; base += 4 * nth) {
            float4 v[4], nv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int i4 = base + u * nth; v[u] = F4[i4 < n4 ? i4 : n4 - 1]; }
            if (ra.cf.on) {
#pragma unroll
                for (int u = 0; u < 4; ++u) { const int i4 = base + u * nth; nv[u] = N4[i4 < n4 ? i4 : n4 - 1]; }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    v[u].x = fmaf(c1f, nv[u].x, v[u].x); v[u].y = fmaf(c1f, nv[u].y, v[u].y);
                    v[u].z = fmaf(c1f, nv[u].z, v[u].z); v[u].w = fmaf(c1f, nv[u].w, v[u].w);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i4 = base + u * nth;
                if (i4 < n4) {
                    int e, k;
                    dK.divmod(4 * i4, e, k);
                    const float vv[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        sF[e * ldk + k] = vv[q];
                        if (++k == K) { k = 0; ++e; }
                    }
                }
            }
        }
    } else {
        for (int t = tid; t < E * K; t += nth) {
            int e, k;
            dK.divmod(t, e, k);
            sF[e * ldk + k] = ra.cf.on ? fmaf(c1f, Ng[t], Fg[t]) : Fg[t];
        }
    }
    for (int t = tid; t < E * (Kp4 - K); t += nth) { const int e = t / (Kp4 - K), k = K + t % (Kp4 - K); sF[e * ldk + k] = 0.f; }
    const int hodge2 = (p.h_L > 1) && ra.want_p;
    if (hodge2) {
        float c1a = 0.f, c2a = 0.f;
        if (ra.cf.on) corr_coef(ra.cf, 1, &c1a, &c2a);
        for (int i = tid; i < NN; i += nth) {
            float v = ra.adj[(size_t)b * NN + i];
            if (ra.cf.on) {   // the A-network of the predictor sees the corrected adjacency
                NoiseArgs nc = na;
                nc.zadj = nullptr; nc.draw_adj = ra.cf.draw_adj;
                const int ii = i / N, jj = i % N;
                const float z = raw_noise_adj(nc, b, ii, jj, N) * ra.flags[(size_t)b * N + ii] * ra.flags[(size_t)b * N + jj];
                v = fmaf(c2a, z, fmaf(c1a, ra.cf.net_adj[(size_t)b * NN + i], v));
            }
            sAdj[i] = v; sAdj[NN + i] = v;
        }
    }
    __syncthreads();
    const unsigned long long off = s_off;
    for (int k = tid; k < Kp4; k += nth) sFrb[k] = (k < K && !(cells[k] & off)) ? 1 : 0;
    for (int e = tid; e < 64; e += nth) sFl[e] = e < E ? edge_on(off, edges, e) : 0.f;
    if (hodge2) {
        // acoef[c][e] = (adj^(c+1))[i_e][j_e]   (pow_tensor + adj_to_hodgedual, graph_utils.py:285-292, cc_utils.py:1525-1536)
        float* A = sAdj; float* P0_ = sAdj + NN; float* P1_ = sAdj + 2 * NN;
        for (int c = 0; c < p.a_cinit; ++c) {
            for (int e = tid; e < E; e += nth) sAco[c * E + e] = P0_[edges[2 * e] * N + edges[2 * e + 1]];
            if (c + 1 < p.a_cinit) {
                for (int i = tid; i < NN; i += nth) {
                    const int r = i / N, cc = i % N;
                    float acc = 0.f;
                    for (int kk = 0; kk < N; ++kk) acc = fmaf(P0_[r * N + kk], A[kk * N + cc], acc);
                    P1_[i] = acc;
                }
                __syncthreads();
                float* t2 = P0_; P0_ = P1_; P1_ = t2;
            }
            __syncthreads();
        }
    }
    __syncthreads();
    if (ra.cf.on) {
        // fused Langevin corrector apply on the LDS-resident block, noise term: F <- (F + c1*net) + c2*z (masked)
        const float c2 = c2f;
        NoiseArgs nc = na;
        nc.zr = nullptr; nc.draw_r = ra.cf.draw_r;
        const int egn = (E + 3) >> 2;
        for (int t = tid; t < egn * K; t += nth) {
            int eg, k;
            dK.divmod(t, eg, k);
            float z[4];
            raw_noise_r4(nc, b, eg, k, E, K, z);
            const float fr = (float)sFrb[k];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int e = 4 * eg + r;
                if (e < E) {
                    // same expression as k_langevin_apply: fma(c2, z*fl*fr, fma(c1, net, v))
                    const float zz = z[r] * sFl[e] * fr;
                    sF[e * ldk + k] = fmaf(c2, zz, sF[e * ldk + k]);
                }
            }
        }
        __syncthreads();
    }
    stamp(ra.dbg, 1);
    const HodgeLayerD& h0 = p.hl[0];
    const HodgeLayerD& h1 = p.hl[1];
    const bool doP0 = ra.want_p && p.h_L > 0, doP1 = hodge2;
    const bool lin1 = doP1 && h0.mval.n == 1;      // rank2' affine in rank2: fold it around the GEMM
    const int wc0 = doP0 ? h0.wc : 0, wc1 = doP1 ? h1.wc : 0;
    if (lin1) {
        // rank2'[e,k] = fl[e] fr[k] (sum_c w_c a_c[e] F[e,k] + b)  ->  P_1[e,:] = fl[e] (s[e] ((F.fr) W_1)[e,:] + b (fr W_1))
        for (int e = tid; e < E; e += nth) {
            float sc = 0.f;
            for (int c = 0; c < h0.cin; ++c) sc = fmaf(w[h0.mval.w[0] + c], sAco[c * E + e], sc);
            sRow[e] = sc;
        }
        __syncthreads();
    }

    // ---- phase 1: H = F F^T (upper-triangle tiles, mirrored), P_0 = F Wcat_0, P_1 = rank2' Wcat_1.
    // One 16x16 output tile over the full K per task; a wave runs two tasks interleaved (independent MFMA
    // chains).  Operands of eight k-steps are fetched at once; the weight fragments, which come from L2,
    // are double-buffered in registers one batch ahead.  No atomics: results are bitwise reproducible.
    stamp(ra.dbg, 2);
    const int ks = Kp4 >> 2;
    int nHtasks = 0;
    const int nH = p.f_cnum == 2 ? MT * (MT + 1) / 2 : 0;
    const int nt0 = doP0 ? (wc0 + 15) >> 4 : 0, nt1 = doP1 ? (wc1 + 15) >> 4 : 0;
    const int ntask = nH + MT * nt0 + MT * nt1;
#ifdef CCSD_EMU
    (void)ks; (void)ntask;
    for (int m = 0; m < E; ++m) {
        for (int n = 0; n < E; ++n) {
            float acc = 0.f;
            if (p.f_cnum == 2) for (int kk = 0; kk < K; ++kk) acc = fmaf(sF[m * ldk + kk], sF[n * ldk + kk], acc);
            sH[m * ldh + n] = (p.f_hmask && m == n) ? 0.f : acc;
        }
        for (int n = 0; n < wc0; ++n) {
            float acc = 0.f;
            for (int kk = 0; kk < K; ++kk) acc = fmaf(sF[m * ldk + kk], w[h0.wcat + (size_t)kk * wc0 + n], acc);
            ra.P0[((size_t)b * E + m) * wc0 + n] = acc;
        }
        for (int n = 0; n < wc1; ++n) {
            float acc = 0.f, un = 0.f;
            for (int kk = 0; kk < K; ++kk) {
                const float frk = (float)sFrb[kk], wv = w[h1.wcat + (size_t)kk * wc1 + n];
                float a;
                if (lin1) a = sF[m * ldk + kk] * frk;
                else {
                    float in[CCSD_SMALLW], out[CCSD_SMALLW];
                    for (int c = 0; c < CCSD_SMALLW; ++c) in[c] = c < h0.cin ? sAco[c * E + m] * sF[m * ldk + kk] : 0.f;
                    small_mlp<CCSD_SMALLW>(h0.mval, w, in, out);
                    a = sFl[m] * out[0] * frk;
                }
                acc = fmaf(a, wv, acc);
                un = fmaf(frk, wv, un);
            }
            ra.P1[((size_t)b * E + m) * wc1 + n] = lin1 ? sFl[m] * fmaf(sRow[m], acc, w[h0.mval.b[0]] * un) : acc;
        }
    }
#endif
#ifndef CCSD_EMU
    // Tile tasks: one 16x16 output tile over the full K per task.  Task list: the H tiles (upper triangle, row-major), the
    // P_0 tiles (row tile major), the P_1 tiles.  The MFMA k slot kq of step j of a 16-wide k block is assigned to
    // k = 16*blk + 4*kq + j, so a lane's A (and, for H, B) values of four steps are ONE aligned ds_read_b128 of F, and its
    // weight values one float4 of the transposed copy Wcat^T[col][Kp].  Rows / columns beyond E / wc read clamped (valid)
    // addresses and are never stored; k >= K meets the zero padding of F.  Even and odd k blocks accumulate into two
    // independent MFMA chains (a dependent f32 16x16x4 MFMA waits 40 cycles, an independent one issues after 32).  Operands of
    // four k blocks are kept in flight (the weights come from L2: ~500+ cycles); a slot is refilled only after the MFMAs that
    // read it have been issued, so the load lands in the same registers.  No atomics: results are bitwise reproducible.
    typedef float r2_f32x4 __attribute__((ext_vector_type(4)));
    const int nblk = Kp4 >> 4;
    const float* WT0 = ra.wp + h0.wcatT;
    const float* WT1 = ra.wp + h1.wcatT;
    const bool hmask = p.f_hmask != 0;
    const float mval_b0 = lin1 ? w[h0.mval.b[0]] : 0.f;       // fetched before the k loops
    auto run_tile = [&](int t) {
        const int lane = tid & 63, l15 = lane & 15, kq = lane >> 4;
        int type, i, c;                                       // 0: H(i, c >= i); 1: P_0(i, c); 2: P_1(i, c)
        if (t < nH) {
            type = 0; i = 0;
            int rem = t;
            while (rem >= MT - i) { rem -= MT - i; ++i; }
            c = i + rem;
        } else if (t < nH + MT * nt0) {
            type = 1; i = (t - nH) / nt0; c = (t - nH) % nt0;
        } else {
            type = 2; i = (t - nH - MT * nt0) / nt1; c = (t - nH - MT * nt0) % nt1;
        }
        const int ra_ = 16 * i + l15;
        const float* pa = sF + (ra_ < E ? ra_ : E - 1) * ldk + 4 * kq;
        int offB = 0;                                         // H tiles: B rows of F in LDS
        const float* wtp = WT0;                               // P tiles: column of Wcat^T in global memory
        if (type == 0) {
            const int rb_ = 16 * c + l15;
            offB = (rb_ < E ? rb_ : E - 1) * ldk + 4 * kq;
        } else {
            const int wcn = type == 1 ? wc0 : wc1, n = 16 * c + l15;
            wtp = (type == 1 ? WT0 : WT1) + (size_t)(n < wcn ? n : wcn - 1) * Kp4 + 4 * kq;
        }
        r2_f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        float upart = 0.f;
        // the k loop, specialised on where the B operand lives (LDS / global) and on the masked-A kind
        auto kloop = [&](auto LB, auto K1) {
            constexpr bool lb = decltype(LB)::v, k1 = decltype(K1)::v;
            auto ldB = [&](int blk) -> float4 {
                if (lb) return *reinterpret_cast<const float4*>(sF + offB + 16 * blk);
                return *reinterpret_cast<const float4*>(wtp + 16 * blk);
            };
            constexpr int D = 4;
            float4 ab[D], bb[D];
            unsigned int fb[D];                                   // cell-mask bytes of the block (masked-A kind only)
#pragma unroll
            for (int u = 0; u < D; ++u) {
                const int bl = u < nblk ? u : nblk - 1;
                ab[u] = *reinterpret_cast<const float4*>(pa + 16 * bl);
                bb[u] = ldB(bl);
                fb[u] = k1 ? *reinterpret_cast<const unsigned int*>(sFrb + 16 * bl + 4 * kq) : 0u;
            }
            auto block = [&](int u, int blk, bool refill) {
                float4 a4 = ab[u];
                const float4 b4 = bb[u];
                if (k1) {
                    const unsigned int f4 = fb[u];
                    const float fr0 = (float)(f4 & 0xffu), fr1 = (float)((f4 >> 8) & 0xffu), fr2 = (float)((f4 >> 16) & 0xffu),
                                fr3 = (float)(f4 >> 24);
                    if (GEN1) {   // general mlp_value: rank2' element-wise on the fly (hodge_attention.py:322-323)
                        const int r = 16 * i + l15, e = r < E ? r : E - 1;
                        float fv[4] = {a4.x, a4.y, a4.z, a4.w};
                        const float frv[4] = {fr0, fr1, fr2, fr3};
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            float iin[CCSD_SMALLW], out[CCSD_SMALLW];
#pragma unroll
                            for (int cc = 0; cc < CCSD_SMALLW; ++cc) iin[cc] = cc < h0.cin ? sAco[cc * E + e] * fv[j] : 0.f;
                            small_mlp<CCSD_SMALLW>(h0.mval, w, iin, out);
                            fv[j] = sFl[e] * out[0] * frv[j];
                        }
                        a4 = make_float4(fv[0], fv[1], fv[2], fv[3]);
                    } else {
                        a4.x *= fr0; a4.y *= fr1; a4.z *= fr2; a4.w *= fr3;
                        upart = fmaf(fr0, b4.x, fmaf(fr1, b4.y, fmaf(fr2, b4.z, fmaf(fr3, b4.w, upart))));
                    }
                }
                r2_f32x4& acc = (u & 1) ? acc1 : acc0;            // block parity == slot parity (D even, block counter a multiple of D)
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, b4.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, b4.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, b4.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, b4.w, acc, 0, 0, 0);
                if (refill) {
                    __builtin_amdgcn_sched_barrier(0);
                    const int bl = blk + D < nblk ? blk + D : nblk - 1;   // clamped: a harmless reload at the tail
                    ab[u] = *reinterpret_cast<const float4*>(pa + 16 * bl);
                    bb[u] = ldB(bl);
                    if (k1) fb[u] = *reinterpret_cast<const unsigned int*>(sFrb + 16 * bl + 4 * kq);
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
            int blk0 = 0;
            for (; blk0 + D <= nblk; blk0 += D) {      // branch-free body: the waits at the loop head stay counted
#pragma unroll
                for (int u = 0; u < D; ++u) block(u, blk0 + u, true);
            }
#pragma unroll
            for (int u = 0; u < D - 1; ++u)
                if (blk0 + u < nblk) block(u, blk0 + u, false);
        };
        if (type == 2) kloop(BoolTag<false>{}, BoolTag<true>{});
        else if (type == 0) kloop(BoolTag<true>{}, BoolTag<false>{});
        else kloop(BoolTag<false>{}, BoolTag<false>{});
        const r2_f32x4 acc = acc0 + acc1;
        const int n = 16 * c + l15;
        const int mb = 16 * i + 4 * kq;
        if (type == 0) {
            if (n < E) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = mb + r;
                    if (m < E) {
                        const float hv = (hmask && m == n) ? 0.f : acc[r];   // hodge_mask zeroes the diagonal (cc_utils.py:964-969)
                        sH[m * ldh + n] = hv;
                        sH[n * ldh + m] = hv;
                    }
                }
            }
            // this task wrote an H tile: publish (a wave's LDS operations complete in order)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) atomicAdd(&s_hdone, 1);
        } else if (type == 1) {
            if (n < wc0) {
                float* dst = ra.P0 + ((size_t)b * E + mb) * wc0 + n;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (mb + r < E) dst[(size_t)r * wc0] = acc[r];
            }
        } else {
            float un = upart;                      // fr . Wcat_1 column: reduce the four k residue classes
            un += __shfl_xor(un, 16, 64);
            un += __shfl_xor(un, 32, 64);
            if (n < wc1) {
                // rank2'[e,k] = fl[e] fr[k] (s[e] F[e,k] + b)  ->  P_1[e,:] = fl[e] (s[e] ((F.fr) W_1)[e,:] + b (fr W_1))
                float* dst = ra.P1 + ((size_t)b * E + mb) * wc1 + n;
                const float bu = mval_b0 * un;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = mb + r, mc = m < E ? m : E - 1;
                    const float v = GEN1 ? acc[r] : sFl[mc] * fmaf(sRow[mc], acc[r], bu);
                    if (m < E) dst[(size_t)r * wc1] = v;
                }
            }
        }
    };
    // Phase 1 = the H tiles plus as many projection tiles as it takes to give every wave the same number of tasks.  The other
    // projection tiles (they depend on nothing but F) are run by the waves BETWEEN their column tiles of phase 2 (affine
    // path): phase 2's epilogue is pure VALU work (Philox, Box-Muller, masks, update), the projection tiles pure MFMA work, and
    // the partner waves of a SIMD then feed different pipes instead of queueing for the same one phase after phase.
    nHtasks = nH;
    const int nw1 = nth >> 6, wave1 = tid >> 6;
    int n1 = ntask;
    if (AFFINE) {
        n1 = ((nH + nw1 - 1) / nw1) * nw1;
        if (n1 > ntask) n1 = ntask;
    }
    for (int t = wave1; t < n1; t += nw1) {
        run_tile(t);
        if (t == 0) stamp(ra.dbg, 6);
    }

    stamp(ra.dbg, 7);
    // No workgroup barrier here: phase 2 only READS the rank-2 block (its results go straight to HBM), so a wave may start
    // it as soon as H is complete -- the waves with the lighter phase-1 tasks do not wait for the projection tasks.
    // Every wave of the workgroup is resident and runs its phase-1 tasks unconditionally, so the count is always reached:
    // the wait has no give-up path into phase 2 (an incomplete H would mean silently wrong scores).  The guard only turns a
    // broken invariant (never observed; ~10 s of polling) into a loud kernel abort instead of an endless spin.
    if (nHtasks > 0) {
        unsigned spins = 0;
        while (__hip_atomic_load(&s_hdone, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < nHtasks) {
            __builtin_amdgcn_s_sleep(8);
            if (++spins == (1u << 27)) __builtin_trap();
        }
    }
#endif

    // ---- phase 2: (H F) per 16-column tile, ScoreNetworkF element-wise, epilogue straight to HBM.
    // H's A-fragments live in registers for the whole phase.
    stamp(ra.dbg, 3);
    float s_net = 0.f, s_z = 0.f;
    const int ntn = (K + 15) >> 4, ksE = Ep4 >> 2;
    auto epi4 = [&](int e0, int k, const float* hf) {
        if (e0 >= E || k >= K) return;
        float z[4] = {0.f, 0.f, 0.f, 0.f};
        if (ep.mode != MODE_SCORE) raw_noise_r4(na, b, e0 >> 2, k, E, K, z);   // one Philox group = 4 edge rows
        const float fr = (float)sFrb[k];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int e = e0 + r;
            if (e >= E) continue;
            const float f = sF[e * ldk + k];
            const float m = sFl[e] * fr;                         // flags_left * flags_right, cc_utils.py:590
            const float net = fnet_element<AFFINE>(p, w, f, hf[r], m);
            const size_t gi = ((size_t)b * E + e) * K + k;
            if (ep.mode == MODE_SCORE) {
                ep.out[gi] = ep.sscale * net;
            } else {
                const float zz = z[r] * m;                       // gen_noise_rank2, cc_utils.py:613-615
                if (ep.mode == MODE_NORMS) {
                    ep.out[gi] = net;
                    s_net = fmaf(net, net, s_net);
                    s_z = fmaf(zz, zz, s_z);
                } else {
                    const float mean = fmaf(ep.pa, f, ep.pb * net);
                    if (ep.mean) ep.mean[gi] = mean;
                    ep.out[gi] = fmaf(ep.pc, zz, mean);
                }
            }
        }
    };
#ifdef CCSD_EMU
    for (int tn = 0; tn < ntn; ++tn) {
        float hfv[64][16];
        for (int e = 0; e < E; ++e)
            for (int j = 0; j < 16; ++j) {
                const int k = 16 * tn + j;
                float acc = 0.f;
                if (p.f_cnum == 2 && k < K)
                    for (int e2 = 0; e2 < E; ++e2) acc = fmaf(sH[e * ldh + e2], sF[e2 * ldk + k], acc);
                hfv[e][j] = acc;
            }
        for (int e0 = 0; e0 < E; e0 += 4)
            for (int j = 0; j < 16; ++j) {
                const float v[4] = {hfv[e0][j], e0 + 1 < E ? hfv[e0 + 1][j] : 0.f, e0 + 2 < E ? hfv[e0 + 2][j] : 0.f, e0 + 3 < E ? hfv[e0 + 3][j] : 0.f};
                epi4(e0, 16 * tn + j, v);
            }
    }
#else
    if constexpr (AFFINE) {
        // Affine ScoreNetworkF (every shipped CC checkpoint but ENZYMES): net = fl[e] fr[k] (alpha f + beta (H F) + gamma).
        // The tile loop is specialised per epilogue mode and noise source (no per-element mode branches) and organised so that
        // the epilogue needs no address arithmetic of its own:
        //  * contraction index in the PERMUTED slot order for the TF full 16-wide blocks (k slot kq of step j of block t <->
        //    e' = 16 t + 4 kq + j): the lane's B operands of block t ARE F[16 t + 4 kq + r][n], r = 0..3, i.e. the F values of its
        //    own accumulator rows of row tile t -- the epilogue reads f from registers; the remainder of E (E mod 16 <= 12)
        //    follows in RS plain steps (e' = 16 TF + 4 s + kq): no MFMA step is spent on padding of the contraction index;
        //  * H's A-fragments are re-read from LDS per column tile (16-byte aligned rows: one ds_read_b128 per (row tile, block))
        //    instead of living in 36 registers for the whole phase;
        //  * masks: fl of the lane's four rows is one ds_read_b128 of sFl, fr one byte per column tile;
        //  * HBM: uniform base pointer + one per-lane 32-bit element offset, advanced by uniform row / tile strides.
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        const int wave = tid >> 6, nw = nth >> 6, lane = tid & 63, l15 = lane & 15, kq = lane >> 4;
        constexpr int TF = RS ? MT - 1 : MT;               // full blocks of the contraction index
        const bool padblk = RS == 0 && (E & 15) != 0;      // the last full block reaches beyond E: its A values are zeroed
        const int ar0 = (l15 < E ? l15 : E - 1) * ldh + 4 * kq;   // A: row l15 of row tile 0, slot group kq
        const int arL = ((16 * (MT - 1) + l15 < E) ? 16 * (MT - 1) + l15 : E - 1) * ldh + 4 * kq;   // ... of the last row tile (clamped)
        const int brow = 4 * kq * ldk;                     // B: row 4 kq of block 0; block t, step j: + (16 t + j) ldk
        const unsigned vo = (unsigned)(4 * kq * K + l15);  // element offset of (row 4 kq, column l15) inside the complex's block
        const bool cn2 = p.f_cnum == 2;
        auto coltile = [&](auto MODE_, auto INJ_, int tn) {
            constexpr int MODE = decltype(MODE_)::value;   // 0 score, 1 norms, 2 predictor, 3 predictor + mean output
            constexpr bool INJ = decltype(INJ_)::value;    // host-supplied raw draws instead of Philox
            // net' = s * net with s = sscale (score), 1 (norms), pb (predictor): folded into the three affine constants
            const float s_ = MODE == 0 ? ep.sscale : MODE == 1 ? 1.f : ep.pb;
            const float sa = s_ * p.f_alpha, sb = s_ * p.f_beta, sg = s_ * p.f_gamma;
            const float pa = ep.pa, pc = ep.pc;
            float* const outp = ep.out + (size_t)b * E * K;
            float* const meanp = MODE == 3 ? ep.mean + (size_t)b * E * K : nullptr;
            const float* const zrp = INJ ? na.zr + (size_t)b * E * K : nullptr;
            {
                const int n = 16 * tn + l15;
                const bool nin = n < K;
                const int nc = nin ? n : K - 1;
                // B operands: bv[4 t + j] = F[16 t + 4 kq + j][n] (permuted blocks), bvr[s] = F[16 TF + 4 s + kq][n] (remainder)
                float bv[TF ? 4 * TF : 1], bvr[RS ? RS : 1];
                const float* fb = sF + brow + nc;
#pragma unroll
                for (int t = 0; t < TF; ++t)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int c = 16 * t + 4 * kq + j;
                        bv[4 * t + j] = (t < MT - 1 || !padblk) ? fb[(16 * t + j) * ldk] : sF[(c < E ? c : E - 1) * ldk + nc];
                    }
#pragma unroll
                for (int s0 = 0; s0 < RS; ++s0) {
                    const int c = 16 * TF + 4 * s0 + kq;
                    bvr[s0] = sF[(c < E ? c : E - 1) * ldk + nc];
                }
                f32x4 acc[MT];
#pragma unroll
                for (int i = 0; i < MT; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (cn2) {
                    int ao = ar0, aL = arL;
                    asm volatile("" : "+v"(ao), "+v"(aL));  // opaque per tile: the loop-invariant A loads must not be hoisted into registers
#pragma unroll
                    for (int t = 0; t < TF; ++t) {
                        float4 a4[MT];
#pragma unroll
                        for (int i = 0; i < MT; ++i)
                            a4[i] = *reinterpret_cast<const float4*>(sH + (i < MT - 1 ? ao + 16 * i * ldh : aL) + 16 * t);
                        if (t == MT - 1 && padblk) {       // columns 16 t + 4 kq + j >= E: whatever was read, the operand is zero
                            const int c0 = 16 * t + 4 * kq;
#pragma unroll
                            for (int i = 0; i < MT; ++i) {
                                a4[i].x = c0 < E ? a4[i].x : 0.f; a4[i].y = c0 + 1 < E ? a4[i].y : 0.f;
                                a4[i].z = c0 + 2 < E ? a4[i].z : 0.f; a4[i].w = c0 + 3 < E ? a4[i].w : 0.f;
                            }
                        }
#pragma unroll
                        for (int i = 0; i < MT; ++i) {
                            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[i].x, bv[4 * t + 0], acc[i], 0, 0, 0);
                            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[i].y, bv[4 * t + 1], acc[i], 0, 0, 0);
                            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[i].z, bv[4 * t + 2], acc[i], 0, 0, 0);
                            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[i].w, bv[4 * t + 3], acc[i], 0, 0, 0);
                        }
                    }
#pragma unroll
                    for (int s0 = 0; s0 < RS; ++s0) {
                        const int c = 16 * TF + 4 * s0 + kq;   // contraction index of this lane's slot
                        const int cc = (c < E ? c : E - 1) - 4 * kq;
#pragma unroll
                        for (int i = 0; i < MT; ++i) {
                            const float av = sH[(i < MT - 1 ? ao + 16 * i * ldh : aL) + cc];
                            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(c < E ? av : 0.f, bvr[s0], acc[i], 0, 0, 0);
                        }
                    }
                }
                if (nin) {                                 // false only for the padding columns of the last column tile
                    const float fr = (float)sFrb[n];
                    unsigned gi = vo + 16u * (unsigned)tn;
#pragma unroll
                    for (int i = 0; i < MT; ++i) {
                        const bool last = i == MT - 1;     // only the last row tile can reach beyond E
                        const int e0 = 16 * i + 4 * kq;
                        if (!last || e0 < E) {
                            float z[4] = {0.f, 0.f, 0.f, 0.f};
                            if (MODE != 0) {
                                if (INJ) {
#pragma unroll
                                    for (int r = 0; r < 4; ++r) z[r] = (!last || e0 + r < E) ? zrp[gi + (unsigned)(r * K)] : 0.f;
                                } else {
                                    philox_normal4(na.seed, na.draw_r, na.b_off + b, (unsigned)((4 * i + kq) * K + n), z);   // one Philox group = 4 edge rows
                                }
                            }
                            const float4 fl4 = *reinterpret_cast<const float4*>(sFl + e0);   // sFl: 64 entries, zero beyond E
                            const float flv[4] = {fl4.x, fl4.y, fl4.z, fl4.w};
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int e = e0 + r;
                                const float f = i < TF ? bv[4 * (i < TF ? i : 0) + r] : sF[(e < E ? e : E - 1) * ldk + n];
                                const float m = flv[r] * fr;                     // flags_left * flags_right, cc_utils.py:590
                                const float net = m * fmaf(sb, acc[i][r], fmaf(sa, f, sg));
                                const unsigned g = gi + (unsigned)(r * K);
                                if (!last || e < E) {
                                    if (MODE == 0) {
                                        outp[g] = net;
                                    } else {
                                        const float zz = z[r] * m;               // gen_noise_rank2, cc_utils.py:613-615
                                        if (MODE == 1) {
                                            outp[g] = net;
                                            s_net = fmaf(net, net, s_net);
                                            s_z = fmaf(zz, zz, s_z);
                                        } else {
                                            const float mean = fmaf(pa, f, net); // v_mean = pa v + pb net (pb folded into net)
                                            if (MODE == 3) meanp[g] = mean;
                                            outp[g] = fmaf(pc, zz, mean);
                                        }
                                    }
                                }
                            }
                        }
                        gi += 16u * (unsigned)K;
                    }
                }
            }
        };
        typedef std::integral_constant<bool, false> NoInj;
        typedef std::integral_constant<bool, true> Inj;
        const bool inj = na.zr != nullptr && ep.mode != MODE_SCORE;
        const int cmode = ep.mode == MODE_SCORE ? 0 : ep.mode == MODE_NORMS ? 1 : ep.mean == nullptr ? 2 : 3;
        // Static schedule of a wave: its column tiles tn = wave, wave + nw, ... with its projection tiles (task n1 + wave, + nw,
        // ...) in between -- before the first column tile for the waves of the lower half, after the second one for the upper
        // half, so that the two waves a SIMD holds are in MFMA-bound and VALU-bound code at different times.  (Static, hence
        // the per-thread accumulation order of the Langevin norms is fixed and runs are bitwise reproducible.)
        int pt = n1 + wave;
        const int pslot = wave < (nw >> 1) ? 0 : 2;
        int cnt = 0;
        for (int tn = wave; tn < ntn; tn += nw, ++cnt) {
            if (cnt == pslot && pt < ntask) { run_tile(pt); pt += nw; }
            switch (cmode * 2 + (inj ? 1 : 0)) {
                case 0: case 1: coltile(std::integral_constant<int, 0>{}, NoInj{}, tn); break;
                case 2: coltile(std::integral_constant<int, 1>{}, NoInj{}, tn); break;
                case 3: coltile(std::integral_constant<int, 1>{}, Inj{}, tn); break;
                case 4: coltile(std::integral_constant<int, 2>{}, NoInj{}, tn); break;
                case 5: coltile(std::integral_constant<int, 2>{}, Inj{}, tn); break;
                case 6: coltile(std::integral_constant<int, 3>{}, NoInj{}, tn); break;
                default: coltile(std::integral_constant<int, 3>{}, Inj{}, tn); break;
            }
        }
        for (; pt < ntask; pt += nw) run_tile(pt);
    } else {
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        const int wave = tid >> 6, nw = nth >> 6, lane = tid & 63, l15 = lane & 15, kq = lane >> 4;
        constexpr int KSE = 4 * MT;                        // ceil(16*MT / 4) k-steps cover E <= 16*MT
        float hA[MT][KSE];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int s0 = 0; s0 < KSE; ++s0) {
                const int r = 16 * i + l15, c = 4 * s0 + kq;
                const float v = sH[(r < E ? r : E - 1) * ldh + (c < E ? c : E - 1)];
                hA[i][s0] = (r < E && c < E) ? v : 0.f;
            }
        // (static tile -> wave assignment: the per-thread accumulation order of the Langevin norms stays fixed, runs are
        // bitwise reproducible)
        for (int tn = wave; tn < ntn; tn += nw) {
            const int n = 16 * tn + l15;
            const bool nin = n < K;
            f32x4 acc[MT];
#pragma unroll
            for (int i = 0; i < MT; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (p.f_cnum == 2) {
                float bv[KSE];
#pragma unroll
                for (int s0 = 0; s0 < KSE; ++s0) {
                    const int kk = 4 * s0 + kq;
                    const float v = sF[(kk < E ? kk : E - 1) * ldk + (nin ? n : K - 1)];
                    bv[s0] = (kk < E && nin) ? v : 0.f;
                }
#pragma unroll
                for (int s0 = 0; s0 < KSE; ++s0)
                    if (s0 < ksE) {
#pragma unroll
                        for (int i = 0; i < MT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(hA[i][s0], bv[s0], acc[i], 0, 0, 0);
                    }
            }
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const float v[4] = {acc[i][0], acc[i][1], acc[i][2], acc[i][3]};
                epi4(16 * i + 4 * kq, n, v);
            }
        }
    }
#endif
    // (no phase 3: the epilogue wrote the results to HBM)
    stamp(ra.dbg, 4);
    if (ep.mode == MODE_NORMS) {
        const float tn_ = block_sum(s_net, sRed);
        const float tz_ = block_sum(s_z, sRed);
        if (tid == 0) { ep.part[(size_t)b * 2 + 0] = tn_; ep.part[(size_t)b * 2 + 1] = tz_; }
    }
    stamp(ra.dbg, 5);
}

// ---------------------------------------------------------------------------------------------
// k_xa: ScoreNetworkX + ScoreNetworkA / ScoreNetworkA_CC for one graph per workgroup.
// ---------------------------------------------------------------------------------------------
#define XA_PLAIN 0
#define XA_HB 1
#define XA_GMH 2
#define XA_GEN 3          /* everything, selected at run time from the plan: both of the above together, conv = "MLP" */
struct XaArgs {
    // inputs: the X-network and the A-network may see different (x, adj) when the Langevin
    // corrector runs more than one inner step (solver.py:759-784)
    const float* xX; const float* adjX;
    const float* xA; const float* adjA;
    const float* flags;
    const float* P0; const float* P1;     // hodge projections (B*E, wc_l)
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

// GCH: the channel stack (every AttentionLayer's adjacency channels, the final MLP's input) does not fit LDS
// (zinc250k, N = 38: 266 KB) and lives in a per-graph slab of the workspace instead; a workgroup's waves share one
// CU and its L1, so __syncthreads() orders those global accesses exactly like the LDS ones.
// Weights are read in place from L2.
// VAR: XA_PLAIN; XA_HB: the plan holds HodgeBaselineLayers (ScoreNetworkA_Base_CC); XA_GMH: the X-network is
// ScoreNetworkX_GMH; XA_GEN: both, and conv = "MLP" attention.  Separate instantiations keep those branches out of the register allocation of the headline variant.
template <bool GCH, int VAR>
__global__ __launch_bounds__(256, GCH ? 2 : 4) void k_xa(const PlanD* __restrict__ plan, const float* __restrict__ w,
                                            const unsigned char* __restrict__ edges, XaArgs xa, NoiseArgs na) {
    CCSD_DYN_SMEM(sm);
    const PlanD& p = *plan;
    constexpr bool HB = VAR == XA_HB || VAR == XA_GEN, GMH = VAR == XA_GMH || VAR == XA_GEN, CONVMLP = VAR == XA_GEN;
    const int N = p.N, F = p.F, NN = N * N, E = p.E, ldn = p.ldn;
    const int b = blockIdx.x, tid = threadIdx.x, nth = blockDim.x;
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
    const int wave_id = tid >> 6, n_waves = nth >> 6;
#endif

    stamp(xa.dbg, 0);
    for (int i = tid; i < N; i += nth) s_flags[i] = xa.flags[(size_t)b * N + i];
    float nx_net = 0.f, nx_z = 0.f, na_net = 0.f, na_z = 0.f;

    // ================= ScoreNetworkX (ScoreNetwork_X.py:102-132) =================
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
#include "ccsd_attn_stack.inc"
#undef ATTN_LAYERS
#undef ATTN_NL
#undef ATTN_TAP
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
                gcn_tile_n<GCH>(src, ldn, fin, N, s_adj, s_dinv, 16 * ct, H,
                           [&](int k, int col) { return W[k * H + col]; }, [&](int col) { return B[col]; },
                           [&](int i, int col, float v) { dst[col * ldn + i] = tanh_f(v); });
            __syncthreads();
        }
        }
        const MlpD& m = p.x_fin;
        if (m.chain) {
            auto epx = [&](int row, int f, float v) { s_h1[f * ldn + row] = v; };
            auto ident = [](int r) { return r; };
            if (m.chain == 2) mlp_chain<2, 3, 1>(m, wp, s_xcat, ldn, s_xcat, m.in, N, ident, epx);
            else mlp_chain<3, 6, 1>(m, wp, s_xcat, ldn, s_xcat, m.in, N, ident, epx);
        } else {
            block_linear<1>(s_h1, ldn, s_xcat, ldn, s_xcat, m.in, wx + m.w[0], wx + m.b[0], m.in, m.hid, N);
            __syncthreads();
            block_linear<1>(s_h2, ldn, s_h1, ldn, s_h1, m.hid, wx + m.w[1], wx + m.b[1], m.hid, m.hid, N);
            __syncthreads();
            block_linear<0>(s_h1, ldn, s_h2, ldn, s_h2, m.hid, wx + m.w[2], wx + m.b[2], m.hid, m.out, N);
        }
        __syncthreads();
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
        for (int e = tid; e < E; e += nth) s_edge[e] = ((int)edges[2 * e] << 8) | (int)edges[2 * e + 1];
        auto edge_i = [&](int e) { return s_edge[e] >> 8; };
        auto edge_j = [&](int e) { return s_edge[e] & 255; };
        auto pair_off = [&](int e) { const int v = s_edge[e]; return (v >> 8) * N + (v & 255); };
        float* s_att = sm + p.o_att;
        float* s_xcur = sm + p.o_xcur;
        float* s_xnext = sm + p.o_xnext;
        float* s_mch = sm + p.o_vcat;
        if (xa.cf.on) {
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
        __syncthreads();
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
#define ATTN_LAYERS p.al
#define ATTN_NL p.a_L
#define ATTN_TAP(l) (void)0
#include "ccsd_attn_stack.inc"
#undef ATTN_LAYERS
#undef ATTN_NL
#undef ATTN_TAP

        stamp(xa.dbg, 12);
        // ---- hodge branch of ScoreNetworkA_CC (ScoreNetwork_A_CC.py:295-316)
        if (VAR != XA_HB && p.h_L > 0) {
            float* s_hd = sm + p.o_hd;          // [hodge channel][E]: diagonals that reach the final MLP
            float* s_hq = sm + p.o_hq;          // [channel][E][2*adim]
            float* s_h1m = s_R;                 // [cout0][E][E] dense output of the first hodge layer
            const float kscale = (float)sqrt((double)p.K);  // hodge_attention.py:118,122: / sqrt(out_dim), out_dim = K
            const float rks = 1.0f / kscale;
            float* s_hw = sm + p.o_hw;         // zero-padded mlp_attention weight blocks of both hodge layers
            stage_mlp_blocks(p.hl[0].matt, w, s_hw);
            if (p.h_L > 1) stage_mlp_blocks(p.hl[1].matt, w, s_hw + p.hw_stride);
            const HodgeLayerD& h0 = p.hl[0];
            const int qw0 = 2 * h0.adim;
            const FastDiv dqw0(qw0), dEqw0(E * qw0);
            const float* P0b = xa.P0 + (size_t)b * E * h0.wc;
            // adj_to_hodgedual (cc_utils.py:1525-1536): diagonal hodge adjacency = upper triangle of the adjacency powers;
            // DenseHCNConv on a diagonal matrix (hodge_layers.py:185-193) is a row scaling
            for (int t = tid; t < p.a_cinit * E; t += nth) {
                int c, e;
                dE.divmod(t, c, e);
                s_hd[t] = s_chan[c * NN + pair_off(e)];
            }
            for (int t = tid; t < h0.cin * E * qw0; t += nth) {
                int c, r, e, d;
                dEqw0.divmod(t, c, r);
                dqw0.divmod(r, e, d);
                const float a = s_chan[c * NN + pair_off(e)];
                const float g = 1.0f / sqrtf(fmaxf(a, 1.f));
                s_hq[t] = fmaf(g * a * g, P0b[(size_t)e * h0.wc + c * qw0 + d], w[h0.bcat + c * qw0 + d]);
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
                            const float* q = s_hq + (c * E + e) * qw0;
                            for (int hh = 0; hh < h0.nchunk; ++hh) {
                                float d = 0.f;
                                for (int u = 0; u < h0.dsplit; ++u) d = fmaf(q[hh * h0.dsplit + u], q[h0.adim + hh * h0.dsplit + u], d);
                                sacc += tanh_f(d * rks);
                            }
                            sacc *= 1.0f / (float)h0.nchunk;
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
            } else {
                // dense E x E attention of every channel, mlp_attention, mask, tanh, + transpose (hodge_attention.py:315-320):
                // one thread per unordered pair (e <= e2) from the pair table, both halves stored
                const HodgeLayerD& h1 = p.hl[1];
                const int qw1 = 2 * h1.adim;
                const float rnc0 = 1.0f / (float)h0.nchunk;
                const int npair = E * (E + 1) / 2;
                const float* P1b = xa.P1 + (size_t)b * E * h1.wc;   // [E][wc1] projections of the second layer (L2)
                const int mtE = (E + 15) >> 4, ntq = (qw1 + 15) >> 4, ksE = (E + 3) >> 2;
#ifndef CCSD_EMU
                // The second layer's projection tasks are (channel, 16-column tile) x row tiles; with one (channel, column
                // tile) per wave its B operands are the same for every row tile: fetch them now, so the L2 latency hides
                // behind the dense attention below
                const bool pf_ok = h1.cin * ntq <= n_waves && ksE <= 16;
                float pfb[16];
                const int pf_c = wave_id / ntq, pf_ct = wave_id % ntq;
                const int pf_l15 = tid & 15, pf_kq = (tid & 63) >> 4;
                if (pf_ok && wave_id < h1.cin * ntq) {
#pragma unroll
                    for (int s0 = 0; s0 < 16; ++s0) {
                        const int k = 4 * s0 + pf_kq, d = 16 * pf_ct + pf_l15;
                        pfb[s0] = P1b[(k < E ? k : E - 1) * h1.wc + pf_c * qw1 + (d < qw1 ? d : qw1 - 1)];
                    }
                }
#endif
                int pe_n = 0, pe2_n = 0;
                if (tid < npair) { pe_n = xa.hpairs[2 * tid]; pe2_n = xa.hpairs[2 * tid + 1]; }
                for (int t = tid; t < npair; t += nth) {
                    const int e = pe_n, e2 = pe2_n;
                    if (t + nth < npair) { pe_n = xa.hpairs[2 * (t + nth)]; pe2_n = xa.hpairs[2 * (t + nth) + 1]; }   // next pair: in flight
                    float in[CCSD_SMALLW], out[CCSD_SMALLW];
#pragma unroll
                    for (int c = 0; c < CCSD_SMALLW; ++c) {
                        float v = 0.f;
                        if (c < h0.cin) {
                            const float* q1 = s_hq + (c * E + e) * qw0;
                            const float* q2 = s_hq + (c * E + e2) * qw0;
                            float s1 = 0.f, s2 = 0.f;
                            for (int hh = 0; hh < h0.nchunk; ++hh) {
                                float d1 = 0.f, d2 = 0.f;
                                for (int u = 0; u < h0.dsplit; ++u) {
                                    const int oq = hh * h0.dsplit + u, ok = h0.adim + oq;
                                    d1 = fmaf(q1[oq], q2[ok], d1);
                                    d2 = fmaf(q2[oq], q1[ok], d2);
                                }
                                s1 += tanh_f(d1 * rks);
                                s2 += tanh_f(d2 * rks);
                            }
                            v = (s1 * rnc0 + s2 * rnc0) * 0.5f;
                        }
                        in[c] = v;
                    }
                    if (w4_0) small_mlp_lds<4>(s_hw, h0.matt.n, in, out); else small_mlp_lds<CCSD_SMALLW>(s_hw, h0.matt.n, in, out);
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
                __syncthreads();
                stamp(xa.dbg, 17);
                // second (last) HodgeAdjAttentionLayer: dense hodge adjacency, only the diagonal of its output
                float* s_deg = sm + p.o_deg;         // [cin1][E]
                for (int t = tid; t < h1.cin * E; t += nth) {
                    int c, e;
                    dE.divmod(t, c, e);
                    // degree = row sum; the matrix is symmetric, so walk the column: consecutive lanes hit consecutive banks
                    const float* Hc = s_h1m + (size_t)c * E * E + e;
                    float s0 = 0.f, s1 = 0.f;
                    int e2 = 0;
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
                            float bval[16];
#pragma unroll
                            for (int s0 = 0; s0 < 16; ++s0) {
                                const int k = 4 * s0 + pf_kq;
                                bval[s0] = (k < E && d < qw1) ? dg[k < E ? k : E - 1] * pfb[s0] : 0.f;
                            }
                            const float bias = w[h1.bcat + pf_c * qw1 + (d < qw1 ? d : qw1 - 1)];
                            for (int rt = 0; rt < mtE; ++rt) {
                                const int e = 16 * rt + pf_l15, ec = e < E ? e : E - 1;
                                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                                for (int s0 = 0; s0 < 16; ++s0)
                                    if (s0 < ksE) {
                                        const int k = 4 * s0 + pf_kq;
                                        const float hv = Hc[ec * E + (k < E ? k : E - 1)];
                                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32((e < E && k < E) ? hv : 0.f, bval[s0], acc, 0, 0, 0);
                                    }
#pragma unroll
                                for (int r = 0; r < 4; ++r) {
                                    const int eo = 16 * rt + 4 * pf_kq + r;
                                    if (eo < E && d < qw1) s_hq[(pf_c * E + eo) * qw1 + d] = fmaf(dg[eo], acc[r], bias);
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
                                      const float v = dg[kc] * P1b[kc * h1.wc + c * qw1 + dc];
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
                for (int e = tid; e < E; e += nth) {
                    float in[CCSD_SMALLW], out[CCSD_SMALLW];
#pragma unroll
                    for (int c = 0; c < CCSD_SMALLW; ++c) {
                        float sacc = 0.f;
                        if (c < h1.cin) {
                            const float* q = s_hq + (c * E + e) * qw1;
                            for (int hh = 0; hh < h1.nchunk; ++hh) {
                                float d = 0.f;
                                for (int u = 0; u < h1.dsplit; ++u) d = fmaf(q[hh * h1.dsplit + u], q[h1.adim + hh * h1.dsplit + u], d);
                                sacc += tanh_f(d * rks);
                            }
                            sacc *= 1.0f / (float)h1.nchunk;
                        }
                        in[c] = sacc;
                    }
                    if (w4_1) small_mlp_lds<4>(s_hw + p.hw_stride, h1.matt.n, in, out); else small_mlp_lds<CCSD_SMALLW>(s_hw + p.hw_stride, h1.matt.n, in, out);
                    const float fh = s_flags[edge_i(e)] * s_flags[edge_j(e)];
#pragma unroll
                    for (int o = 0; o < CCSD_SMALLW; ++o)
                        if (o < h1.cout) { const float tv = tanh_f(out[o] * fh * fh); s_hd[(p.a_cinit + h0.cout + o) * E + e] = tv + tv; }
                }
                __syncthreads();
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
                if (m.chain == 3) mlp_chain<2, 4, 1>(m, wp, s_chan, NN, s_chan, m.in, E, pair_off, epf);
                else if (m.chain == 4) mlp_chain<3, 5, 1>(m, wp, s_chan, NN, s_chan, m.in, E, pair_off, epf);
                else if (m.chain == 5) mlp_chain<3, 6, 1>(m, wp, s_chan, NN, s_chan, m.in, E, pair_off, epf);
                else mlp_chain<4, 7, 1>(m, wp, s_chan, NN, s_chan, m.in, E, pair_off, epf);
                stamp(xa.dbg, 11);
            } else {
                block_linear<1>(f0, ldf, s_chan + p0, NN, s_chan + p0, m.in, wf + m.w[0], wf + m.b[0], m.in, m.hid, rows);
                __syncthreads();
                block_linear<1>(f1, ldf, f0, ldf, f0, m.hid, wf + m.w[1], wf + m.b[1], m.hid, m.hid, rows);
                __syncthreads();
                block_linear<0>(f0, ldf, f1, ldf, f1, m.hid, wf + m.w[2], wf + m.b[2], m.hid, 1, rows);
            }
            __syncthreads();
            stamp(xa.dbg, 15);
            for (int r = tid; r < rows; r += nth) {
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
            __syncthreads();
        }
    }
    stamp(xa.dbg, 14);
    if (xa.mode == MODE_NORMS) {
        __syncthreads();
        const float t0 = block_sum(nx_net, s_red), t1 = block_sum(na_net, s_red);
        const float t2 = block_sum(nx_z, s_red), t3 = block_sum(na_z, s_red);
        if (tid == 0) {
            float* o = xa.norm2 + (size_t)b * 4;
            if (xa.do_x) { o[0] = t0; o[2] = t2; }
            if (xa.do_a) { o[1] = t1; o[3] = t3; }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// k_normsum: sums[0..5] = sum_b sqrt(|net_x|^2), |net_adj|, |net_rank2|, |z_x|, |z_adj|, |z_rank2|
// (torch.norm(...).mean() numerators, solver.py:763-767).  One workgroup, deterministic order.
// ---------------------------------------------------------------------------------------------
__global__ void k_normsum(const float* __restrict__ norm2, const float* __restrict__ part, int B, int ntiles,
                          int is_cc, float* __restrict__ sums) {
    __shared__ float red[64];
    float acc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
        acc[0] += sqrtf(norm2[b * 4 + 0]);
        acc[1] += sqrtf(norm2[b * 4 + 1]);
        acc[3] += sqrtf(norm2[b * 4 + 2]);
        acc[4] += sqrtf(norm2[b * 4 + 3]);
        if (is_cc) {
            float sn = 0.f, sz = 0.f;
            for (int t = 0; t < ntiles; ++t) {
                sn += part[((size_t)b * ntiles + t) * 2 + 0];
                sz += part[((size_t)b * ntiles + t) * 2 + 1];
            }
            acc[2] += sqrtf(sn);
            acc[5] += sqrtf(sz);
        }
    }
    for (int i = 0; i < 6; ++i) {
        const float t = block_sum(acc[i], red);
        if (threadIdx.x == 0) sums[i] = t;
    }
}

// ---------------------------------------------------------------------------------------------
// k_langevin_apply: step = (snr * zn / gn)^2 * 2 * alpha; v_mean = v + step*score;
// v = v_mean + sqrt(2 step) * z * scale_eps          (solver.py:767-769, 781-783, 797-801)
// score = sscale * net, so gn = |sscale| * sum|net| and step*score = step*sscale*net.
// grid-stride over the three tensors of the whole batch.
// ---------------------------------------------------------------------------------------------
struct LangArgs {
    const float* x; const float* adj; const float* r;          // state in
    const float* nx; const float* nadj; const float* nr;        // raw network outputs kept by the NORMS pass
    float* ox; float* oadj; float* orr;                          // state out
    const float* flags;
    const float* sums;
    float ss[3], alpha[3];
    float snr, seps;
    int B, N, F, E, K, is_cc;
};
CCSD_DEV void langevin_coef(const LangArgs& a, int t, float* c1, float* c2) {
    const float gn = fabsf(a.ss[t]) * a.sums[t], zn = a.sums[3 + t];
    const float q = a.snr * zn / gn;
    const float step = q * q * 2.f * a.alpha[t];
    *c1 = step * a.ss[t];
    *c2 = sqrtf(step * 2.f) * a.seps;
}
__global__ void k_langevin_apply(LangArgs a, NoiseArgs na, const unsigned long long* __restrict__ offbits,
                                 const unsigned char* __restrict__ edges, const unsigned long long* __restrict__ cells) {
    const long long nxe = (long long)a.B * a.N * a.F, nae = (long long)a.B * a.N * a.N;
    const long long nre = a.is_cc ? (long long)a.B * ((a.E + 3) / 4) * a.K : 0;  // one thread per 4-edge group x column
    const long long total = nxe + nae + nre;
    float c1x, c2x, c1a, c2a, c1r = 0.f, c2r = 0.f;
    langevin_coef(a, 0, &c1x, &c2x);
    langevin_coef(a, 1, &c1a, &c2a);
    if (a.is_cc) langevin_coef(a, 2, &c1r, &c2r);
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        if (t < nxe) {
            const int per = a.N * a.F, b = (int)(t / per), idx = (int)(t % per), i = idx / a.F;
            const float z = raw_noise_x(na, b, idx, per) * a.flags[(size_t)b * a.N + i];
            a.ox[t] = fmaf(c2x, z, fmaf(c1x, a.nx[t], a.x[t]));
        } else if (t < nxe + nae) {
            const long long u = t - nxe;
            const int per = a.N * a.N, b = (int)(u / per), ij = (int)(u % per), i = ij / a.N, j = ij % a.N;
            const float z = raw_noise_adj(na, b, i, j, a.N) * a.flags[(size_t)b * a.N + i] * a.flags[(size_t)b * a.N + j];
            a.oadj[u] = fmaf(c2a, z, fmaf(c1a, a.nadj[u], a.adj[u]));
        } else {
            const long long u = t - nxe - nae;
            const int eg_n = (a.E + 3) / 4;
            const int k = (int)(u % a.K), eg = (int)((u / a.K) % eg_n), b = (int)(u / ((long long)a.K * eg_n));
            float z[4];
            raw_noise_r4(na, b, eg, k, a.E, a.K, z);
            const unsigned long long off = offbits[b];
            const float fr = cell_on(off, cells, k);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int e = 4 * eg + s;
                if (e >= a.E) continue;
                const size_t gi = ((size_t)b * a.E + e) * a.K + k;
                const float zz = z[s] * edge_on(off, edges, e) * fr;
                a.orr[gi] = fmaf(c2r, zz, fmaf(c1r, a.nr[gi], a.r[gi]));
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// k_s4_apply: the update half of one S4_solver step (solver.py:1296-1352 graph, 1446-1529 CC), element-wise:
//   v1 = v + step*score + sqrt(2 step)*z1*scale_eps        Langevin-style correction with the step's score
//   v2 = m1*v1 + s1*z2                                     sde.transition(v1, t, dt/2)
//   v3 = v2 + d*net                                        + Sdrift*dt, Sdrift = -g(t)^2 * score
//   mean = m2*v3 ;  v = mean + s2*z3                       sde.transition(v3, t + dt/2, dt/2)
// Same indexing and masks as k_langevin_apply; three independent draws per element.
// ---------------------------------------------------------------------------------------------
struct S4Args {
    LangArgs a;                       // state in/out, raw nets, flags, norm sums, Langevin scalars
    float m1[3], s1[3], d[3], m2[3], s2[3];
    float* mx; float* madj; float* mr;   // means (nullable)
};
CCSD_DEV float s4_chain(float v, float net, float z1, float z2, float z3, float c1, float c2, const S4Args& q, int t, float* mean) {
    const float v1 = fmaf(c2, z1, fmaf(c1, net, v));
    const float v2 = fmaf(q.s1[t], z2, q.m1[t] * v1);
    const float v3 = fmaf(q.d[t], net, v2);
    const float mu = q.m2[t] * v3;
    *mean = mu;
    return fmaf(q.s2[t], z3, mu);
}
__global__ void k_s4_apply(S4Args q, NoiseArgs n1, NoiseArgs n2, NoiseArgs n3, const unsigned long long* __restrict__ offbits,
                           const unsigned char* __restrict__ edges, const unsigned long long* __restrict__ cells) {
    const LangArgs& a = q.a;
    const long long nxe = (long long)a.B * a.N * a.F, nae = (long long)a.B * a.N * a.N;
    const long long nre = a.is_cc ? (long long)a.B * ((a.E + 3) / 4) * a.K : 0;
    const long long total = nxe + nae + nre;
    float c1x, c2x, c1a, c2a, c1r = 0.f, c2r = 0.f;
    langevin_coef(a, 0, &c1x, &c2x);
    langevin_coef(a, 1, &c1a, &c2a);
    if (a.is_cc) langevin_coef(a, 2, &c1r, &c2r);
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        float mu;
        if (t < nxe) {
            const int per = a.N * a.F, b = (int)(t / per), idx = (int)(t % per), i = idx / a.F;
            const float fl = a.flags[(size_t)b * a.N + i];
            a.ox[t] = s4_chain(a.x[t], a.nx[t], raw_noise_x(n1, b, idx, per) * fl, raw_noise_x(n2, b, idx, per) * fl,
                               raw_noise_x(n3, b, idx, per) * fl, c1x, c2x, q, 0, &mu);
            if (q.mx) q.mx[t] = mu;
        } else if (t < nxe + nae) {
            const long long u = t - nxe;
            const int per = a.N * a.N, b = (int)(u / per), ij = (int)(u % per), i = ij / a.N, j = ij % a.N;
            const float fl = a.flags[(size_t)b * a.N + i] * a.flags[(size_t)b * a.N + j];
            a.oadj[u] = s4_chain(a.adj[u], a.nadj[u], raw_noise_adj(n1, b, i, j, a.N) * fl, raw_noise_adj(n2, b, i, j, a.N) * fl,
                                 raw_noise_adj(n3, b, i, j, a.N) * fl, c1a, c2a, q, 1, &mu);
            if (q.madj) q.madj[u] = mu;
        } else {
            const long long u = t - nxe - nae;
            const int eg_n = (a.E + 3) / 4;
            const int k = (int)(u % a.K), eg = (int)((u / a.K) % eg_n), b = (int)(u / ((long long)a.K * eg_n));
            float z1[4], z2[4], z3[4];
            raw_noise_r4(n1, b, eg, k, a.E, a.K, z1);
            raw_noise_r4(n2, b, eg, k, a.E, a.K, z2);
            raw_noise_r4(n3, b, eg, k, a.E, a.K, z3);
            const unsigned long long off = offbits[b];
            const float fr = cell_on(off, cells, k);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int e = 4 * eg + s;
                if (e >= a.E) continue;
                const size_t gi = ((size_t)b * a.E + e) * a.K + k;
                const float m = edge_on(off, edges, e) * fr;
                a.orr[gi] = s4_chain(a.r[gi], a.nr[gi], z1[s] * m, z2[s] * m, z3[s] * m, c1r, c2r, q, 2, &mu);
                if (q.mr) q.mr[gi] = mu;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// k_init_state: masked prior (solver.py:1111-1118; sde.py:436,448-449).  Same indexing as above.
// ---------------------------------------------------------------------------------------------
__global__ void k_init_state(float* x, float* adj, float* r, const float* __restrict__ flags, NoiseArgs na,
                             const unsigned long long* __restrict__ offbits, const unsigned char* __restrict__ edges,
                             const unsigned long long* __restrict__ cells, int B, int N, int F, int E, int K, int is_cc) {
    const long long nxe = (long long)B * N * F, nae = (long long)B * N * N;
    const int eg_n = (E + 3) / 4;
    const long long nre = is_cc ? (long long)B * eg_n * K : 0;
    const long long total = nxe + nae + nre;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        if (t < nxe) {
            const int per = N * F, b = (int)(t / per), idx = (int)(t % per);
            x[t] = raw_noise_x(na, b, idx, per) * flags[(size_t)b * N + idx / F];
        } else if (t < nxe + nae) {
            const long long u = t - nxe;
            const int per = N * N, b = (int)(u / per), ij = (int)(u % per), i = ij / N, j = ij % N;
            adj[u] = raw_noise_adj(na, b, i, j, N) * flags[(size_t)b * N + i] * flags[(size_t)b * N + j];
        } else {
            const long long u = t - nxe - nae;
            const int k = (int)(u % K), eg = (int)((u / K) % eg_n), b = (int)(u / ((long long)K * eg_n));
            float z[4];
            raw_noise_r4(na, b, eg, k, E, K, z);
            const unsigned long long off = offbits[b];
            const float fr = cell_on(off, cells, k);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int e = 4 * eg + s;
                if (e < E) r[((size_t)b * E + e) * K + k] = edge_on(off, edges, e) * z[s] * fr;
            }
        }
    }
}

// quantize / quantize_mol (graph_utils.py:191, 209-213)
__global__ void k_quantize(const float* __restrict__ in, long long n, float thr, long long* __restrict__ out) {
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long long)gridDim.x * blockDim.x) {
        const float v = in[t];
        long long q;
        if (thr >= 0.f) q = v < thr ? 0 : 1;
        else q = v >= 2.5f ? 3 : v >= 1.5f ? 2 : v >= 0.5f ? 1 : 0;
        out[t] = q;
    }
}

// ---------------------------------------------------------------------------------------------
// k_rank2_cells: sparse form of the quantised rank-2 incidence matrix -- the input cc_from_incidence needs
// (cc_utils.py:243-262: column k holds a rank-2 cell iff any of its entries is non-zero after quantize()).
// bits[b][k / 64] bit (k % 64) = any_e( rank2[b][e][k] >= thr );  counts[b] = number of set bits.
// One workgroup per complex; a wave covers 64 consecutive columns per pass (coalesced rows), its ballot is the word.
// ---------------------------------------------------------------------------------------------
__global__ void k_rank2_cells(const float* __restrict__ rank2, int E, int K, float thr, unsigned long long* __restrict__ bits,
                              int* __restrict__ counts) {
    const int b = blockIdx.x, W = (K + 63) >> 6;
    const float* Fb = rank2 + (size_t)b * E * K;
#ifdef CCSD_EMU
    int total = 0;
    for (int wd = 0; wd < W; ++wd) {
        unsigned long long m = 0;
        for (int q = 0; q < 64; ++q) {
            const int k = 64 * wd + q;
            bool any = false;
            if (k < K)
                for (int e = 0; e < E; ++e) any = any || Fb[(size_t)e * K + k] >= thr;
            if (any) { m |= 1ull << q; ++total; }
        }
        bits[(size_t)b * W + wd] = m;
    }
    counts[b] = total;
#else
    __shared__ int s_cnt;
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    int mine = 0;
    for (int wd = wave; wd < W; wd += nw) {
        const int k = 64 * wd + lane, kc = k < K ? k : K - 1;
        bool any = false;
        for (int e = 0; e < E; ++e) any = any || Fb[(size_t)e * K + kc] >= thr;
        const unsigned long long m = __ballot(any && k < K);
        if (lane == 0) { bits[(size_t)b * W + wd] = m; mine += __popcll(m); }
    }
    if (lane == 0 && mine) atomicAdd(&s_cnt, mine);
    __syncthreads();
    if (threadIdx.x == 0) counts[b] = s_cnt;
#endif
}

