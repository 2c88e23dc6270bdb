// ccsd_dev.h -- device helpers shared by every kernel: stamps, fast math, Philox4x32-10 + Box-Muller, raw-noise accessors,
// per-thread MLPs, block_linear / mlp_chain_tile / gcn_tile (MFMA f32 16x16x4 building blocks), the 64x64 tile engine.
// Part of the kernel source of libccsd_hip.so (see ccsd_kernels.h for the map).
#pragma once
#include "ccsd_plan.h"
#include <type_traits>

#if defined(CCSD_BARRIER_PROF) && !defined(CCSD_EMU)
// Diagnostic build only (tools/dev/barrier_prof.sh; never defined by __graft_entry__.build()): every __syncthreads() of the
// unit adds, per wave, the cycles the wave spent inside it to g_bar[workgroup][wave] and counts it in g_bar[..][8 + wave];
// k_xa copies its row to stamp slots 40.. at its end.  Waves that arrive early show up as barrier time: busy = life - barrier.
__device__ long long g_bar[4096 * 16];
__device__ long long g_arr[64 * 4 * 64];      // [workgroup < 64][wave < 4][barrier < 64]: arrival clock
__device__ inline void ccsd_real_sync() { __syncthreads(); }
__device__ inline void ccsd_prof_sync() {
    const long long t0 = (long long)__builtin_readcyclecounter();
    ccsd_real_sync();
    const long long t1 = (long long)__builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0 && blockIdx.x < 4096) {
        long long* r = g_bar + (size_t)blockIdx.x * 16 + (threadIdx.x >> 6);
        if (blockIdx.x < 64 && r[8] < 64 && (threadIdx.x >> 6) < 4) g_arr[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 64 + r[8]] = t0;
        r[0] += t1 - t0; r[8] += 1;
    }
}
__device__ long long g_ct[16];
#define CCSD_CT(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_ct[i] = (long long)__builtin_readcyclecounter(); } while (0)
#define CCSD_CTW(i, w) do { if (blockIdx.x == 0 && threadIdx.x == 64 * (w)) g_ct[i] = (long long)__builtin_readcyclecounter(); } while (0)
#define __syncthreads() ccsd_prof_sync()
#else
#define CCSD_CT(i) do {} while (0)
#define CCSD_CTW(i, w) do {} while (0)
#endif
// ---------------------------------------------------------------------------------------------
// small device helpers
// ---------------------------------------------------------------------------------------------
// Index of the calling wave in its workgroup, as a value the compiler KNOWS to be wave-uniform.  hipcc treats threadIdx.x >> 6 as
// divergent (the launch bounds give no y / z extents), so every `for (task = wave; ...)` loop, its index arithmetic and the
// weight pointers derived from it would run on the vector ALU under exec masks; v_readfirstlane moves it to an SGPR once.
CCSD_DEV int wave_index() {
#ifdef CCSD_EMU
    return 0;
#else
    return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
#endif
}
// Diagnostic cycle stamps (ccsd_debug_stamps): thread 0 of every workgroup writes the shader clock at phase
// boundaries into a caller buffer [workgroup][64] (k_r2: slots 0.., k_xa: slots 32..).  NULL (the default) compiles to a uniform branch not taken.
CCSD_DEV void stamp(long long* dbg, int slot) {
#ifndef CCSD_EMU
#ifdef CCSD_STOP_DIAG
    // Diagnostic build only (tools/dev/phase_mix.sh): the whole grid ends at the stamp whose slot + 1 the host wrote behind the stamp rows, so
    // that per-dispatch PMC counters of launches stopped at successive stamps difference into a per-phase instruction mix.
    // VALID STOP SLOTS are the stamps EVERY thread of the workgroup executes (the lists XA_STOPS / R2_STOPS in tools/dev/phase_mix.py):
    // a stamp inside a `t == 0` / single-wave branch (k_r2 slot 6) would end that wave alone -- its siblings run on, the differenced
    // counters are wrong, and a wave stopped ahead of k_r2's s_hdone hand-over leaves the others spinning into the trap.  Never list one.
    if (dbg && dbg[((size_t)gridDim.x + 254) * 64 + 63] == slot + 1) __builtin_amdgcn_endpgm();
#endif
    if (dbg && threadIdx.x == 0) dbg[(size_t)blockIdx.x * 64 + slot] = (long long)__builtin_readcyclecounter();
    // slot 0 / the kernel's last slot also leave the constant 100 MHz counter two slots from the end of the kernel's half-row:
    // shader clock = cycles / real time (tools/stamps.py)
    if (dbg && threadIdx.x == 0 && slot == 0) dbg[(size_t)blockIdx.x * 64 + 30] = (long long)__builtin_amdgcn_s_memrealtime();
    if (dbg && threadIdx.x == 0 && (slot == 5 || slot == 14)) dbg[(size_t)blockIdx.x * 64 + 31] = (long long)__builtin_amdgcn_s_memrealtime();
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

#ifndef CCSD_EMU
// gfx950 lane swaps between the four 16-lane rows r0..r3 of a wave (one VALU instruction each):
//   lane_swap16(a, b): a <- [a.r0, b.r0, a.r2, b.r2], b <- [a.r1, b.r1, a.r3, b.r3]   (odd rows of a <-> even rows of b)
//   lane_swap32(a, b): a <- [a.r0, a.r1, b.r0, b.r1], b <- [a.r2, a.r3, b.r2, b.r3]   (upper half of a <-> lower half of b)
// Inline asm on purpose: hipcc (ROCm 7.2) compiles `r[0] + r[1]` of __builtin_amdgcn_permlane{16,32}_swap as `r[0] + r[0]`
// (tools/ubench/permlane_swap.hip shows both forms).  s_nop 1: the instruction may not read a VGPR a VALU wrote in the two
// preceding slots (the compiler emits the same padding for the builtin).
CCSD_DEV void lane_swap16(float& a, float& b) { asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
CCSD_DEV void lane_swap32(float& a, float& b) { asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
// sum over the four rows, result in every row
CCSD_DEV float rows_allsum(float x) {
    float y = x;
    lane_swap16(x, y);
    x += y; y = x;
    lane_swap32(x, y);
    return x + y;
}
#endif

struct NoiseArgs {
    const float* zx;
    const float* zadj;
    const float* zr;
    unsigned long long seed;
    unsigned int draw_x, draw_adj, draw_r;
    long long b_off;
    // element -> Philox group map of the rank2 draw: 0: group (e >> 2, k) = the four edge rows 4 (e >> 2) .. + 3 of column k (the
    // layout of an MFMA accumulator: predictor draws, priors, S4 draws); 1: group t >> 2 of the flattened index t = e K + k = four
    // CONSECUTIVE elements of a row (the layout of a 16-byte load: the Langevin corrector's draws, generated where rank2 streams
    // through registers -- k_r2's block load, k_langevin_apply, k_noise_norm)
    int flat_r;
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

// rank2 noise of flat group g: elements 4 g .. 4 g + 3 of the sample's flattened (E, K) block (NoiseArgs::flat_r == 1)
CCSD_DEV void raw_noise_rflat4(const NoiseArgs& na, int b, int g, int EK, float* n) {
    if (na.zr) {
#pragma unroll
        for (int s = 0; s < 4; ++s) n[s] = 4 * g + s < EK ? na.zr[(size_t)b * EK + 4 * g + s] : 0.f;
    } else {
        philox_normal4(na.seed, na.draw_r, na.b_off + b, (unsigned)g, n);
    }
}

// sum over head chunks of tanh(q_h . k_h * scale): the attention logits of one direction of one pair (attention.py:111-129,
// hodge_attention.py:108-124).  The chunk width DS = attn_dim // num_heads is a compile-time constant for the common widths
// (inner loop unrolled: no per-term loop overhead), any other width takes the counted loop.
template <int DS>
CCSD_DEV float attn_logit_sum(const float* q, const float* k, int nchunk, int dsplit, float scale) {
    float s = 0.f;
    for (int h = 0; h < nchunk; ++h) {
        float d = 0.f;
        if (DS > 0) {
#pragma unroll
            for (int u = 0; u < DS; ++u) d = fmaf(q[h * DS + u], k[h * DS + u], d);
        } else {
            for (int u = 0; u < dsplit; ++u) d = fmaf(q[h * dsplit + u], k[h * dsplit + u], d);
        }
        s += tanh_f(d * scale);
    }
    return s;
}
CCSD_DEV float attn_logits(const float* q, const float* k, int nchunk, int dsplit, float scale) {
    if (dsplit == 2) return attn_logit_sum<2>(q, k, nchunk, dsplit, scale);      // qm9_CC (10 // 4), the hodge layers (4 // 2)
    return attn_logit_sum<0>(q, k, nchunk, dsplit, scale);
}

// Sum of the values two ADJACENT work items (t even, t + 1) hold, valid in both on the GPU: the partner is the neighbouring
// lane.  In the host emulation (one "thread" walks all items in order) the partner is the previous iteration: `stash` carries
// its value, and only the odd item sees the sum -- callers store from the odd item.
CCSD_DEV float pair_sum(float v, int t, float& stash) {
#ifdef CCSD_EMU
    if (!(t & 1)) { stash = v; return v; }
    return stash + v;
#else
    (void)t; (void)stash;
    return v + __shfl_xor(v, 1, 64);
#endif
}

// block-wide sum; result valid in every thread.  `red` = 64 floats of LDS.
CCSD_DEV float block_sum(float v, float* red) {
#ifdef CCSD_EMU
    (void)red;
    return v;
#else
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    const int wave = wave_index(), lane = threadIdx.x & 63, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float t = 0.f;
    for (int w = 0; w < nw; ++w) t += red[w];
    __syncthreads();
    return t;
#endif
}

// NV block-wide sums at once (one pair of barriers instead of NV triples); each value in block_sum's order, so bit-identical to NV
// block_sum calls.  Results valid in every thread.  `red` = 16 * NV floats of LDS.
template <int NV>
CCSD_DEV void block_sums(float (&v)[NV], float* red) {
#ifdef CCSD_EMU
    (void)v; (void)red;
#else
#pragma unroll
    for (int i = 0; i < NV; ++i) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v[i] += __shfl_xor(v[i], o, 64);
    }
    const int wave = wave_index(), lane = threadIdx.x & 63, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) red[wave * NV + i] = v[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        float t = 0.f;
        for (int w = 0; w < nw; ++w) t += red[w * NV + i];
        v[i] = t;
    }
    __syncthreads();
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
CCSD_DEV void stage_mlp_blocks(const MlpD& m, const float* __restrict__ w, float* blk, int t0, int ts) {
    for (int t = t0; t < m.n * CCSD_HWBLK; t += ts) {
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
// (mlp_wt: the transposed copy of linear i of a non-chained MLP in the packed buffer, or nullptr -- MlpD, ccsd_plan.h)
CCSD_DEV const float* mlp_wt(const MlpD& m, const float* wp, int i) { return (!m.chain && m.pb[0] == CCSD_MLP_WT) ? wp + m.pw[i] : nullptr; }
template <int ACT>  // 0 none, 1 ELU
// Wt (optional): the transposed zero-padded copy Wt[in][pad16(out)] of W -- the 16 lanes of a B-operand load then read 64 consecutive
// bytes instead of one word from each of 16 rows of W (same values, same k order: bit-identical)
CCSD_DEV void block_linear(float* Y, int ldy, const float* X, int ldx, const float* X2, int ksplit,
                           const float* __restrict__ W, const float* __restrict__ bias, int in, int out, int rows,
                           const float* __restrict__ Wt = nullptr) {
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
    const int wave = wave_index(), lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    const int mt = (rows + 15) >> 4, nt = (out + 15) >> 4, ks = (in + 3) >> 2;
    const int l15 = lane & 15, kq = lane >> 4;
    for (int task = wave; task < mt * nt; task += nw) {
        const int m0 = (task % mt) << 4, n0 = (task / mt) << 4;
        const int bn = n0 + l15;
        const int am = (m0 + l15 < rows) ? m0 + l15 : rows - 1;      // clamp: rows beyond `rows` are never stored
        const float* wr = W + (size_t)(bn < out ? bn : 0) * in;
        const int outp = (out + 15) & ~15;
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
                bv[u] = (Wt ? Wt[kc * outp + bn] : wr[kc]) * (k < in ? nok : 0.f);        // (Wt: bn < outp always; its pad columns are zero)
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
// one linear of the chain: TI input tiles (registers) -> TO output tiles, straight-line code; W = the packed block + 4 * lane
// (ccsd_chain_widx: [out tile][in tile][lane][4])
template <int TI, int TO>
CCSD_DEV void chain_layer(const float* __restrict__ W, const float* __restrict__ Bv, bool act,
                          const chain_f32x4* in, chain_f32x4* out) {
#pragma unroll
    for (int to = 0; to < TO; ++to) {
        float4 wv[TI];
#pragma unroll
        for (int t = 0; t < TI; ++t) wv[t] = *reinterpret_cast<const float4*>(W + (size_t)(to * TI + t) * 256);
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
                for (int k = 0; k < ip; ++k) acc = fmaf(W[ccsd_chain_widx(o, k, ip)], a[k], acc);
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
    CCSD_CT(0);
    const int l15 = lane & 15, kq = lane >> 4;
    const int prow = rowoff((p0 + l15 < rows) ? p0 + l15 : rows - 1);      // clamped: rows beyond `rows` are never stored
    const int in = m.in;
    if constexpr (NI == 1 && NH == 1 && NO == 1) {
        // The 16-wide MLP (the AttentionLayers' edge MLP): one float4 of weights and one of biases per linear and lane.  All of them
        // are requested before the inputs are gathered -- otherwise every linear waits for its own plan-field scalar load and L2
        // round trip (6.2 k cycles per tile for twelve MFMAs; barrier table, profiles/r02_b_k_xa_barrier_intervals.txt).
        if (m.n <= 3) {
            float4 wq[3], bq[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int ii = i < m.n ? i : m.n - 1;
                wq[i] = *reinterpret_cast<const float4*>(wp + m.pw[ii] + 4 * lane);
                bq[i] = *reinterpret_cast<const float4*>(wp + m.pb[ii] + 4 * kq);
            }
            chain_f32x4 a;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = 4 * kq + j;
                const int kc = k < in ? k : in - 1;
                const float* src = kc < ksplit ? X + kc * ldx : X2 + (kc - ksplit) * ldx;
                a[j] = src[prow];
            }
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                if (i < m.n) {
                    chain_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[i].x, a[0], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[i].y, a[1], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[i].z, a[2], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[i].w, a[3], acc, 0, 0, 0);
                    acc[0] += bq[i].x; acc[1] += bq[i].y; acc[2] += bq[i].z; acc[3] += bq[i].w;
                    if (i < m.n - 1) { acc[0] = elu1_sel(acc[0]); acc[1] = elu1_sel(acc[1]); acc[2] = elu1_sel(acc[2]); acc[3] = elu1_sel(acc[3]); }
                    a = acc;
                }
            }
            const bool rok1 = p0 + l15 < rows;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int f = 4 * kq + r;
                if (f < m.out && rok1) epi(p0 + l15, f, a[r]);
            }
            return;
        }
    }
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
    CCSD_CT(1);
    const int l4 = 4 * lane;
    if (m.n == 1) {
        chain_layer<NI, NO>(wp + m.pw[0] + l4, wp + m.pb[0] + 4 * kq, false, xin, yo);
    } else {
        chain_layer<NI, NH>(wp + m.pw[0] + l4, wp + m.pb[0] + 4 * kq, true, xin, h0);
        CCSD_CT(2);
        for (int i = 1; i < m.n - 1; ++i) {
            chain_layer<NH, NH>(wp + m.pw[i] + l4, wp + m.pb[i] + 4 * kq, true, h0, h1);
#pragma unroll
            for (int t = 0; t < NH; ++t) h0[t] = h1[t];
        }
        CCSD_CT(3);
        const int il = m.n - 1;
        if (NO == 1 && m.out == 1) {
            // a single output feature: 16 of 16 MFMA rows would be padding -- dot product on the VALU instead; the lane
            // holds features 16t + 4kq + r of its row, the four kq groups are summed with two cross-lane adds
            const float* W3 = wp + m.pw[il] + 64 * kq;      // row 0 of the packed block: lane 16 kq of in tile t
            float d = 0.f;
#pragma unroll
            for (int t = 0; t < NH; ++t) {
                const float4 wv = *reinterpret_cast<const float4*>(W3 + 256 * t);
                d = fmaf(wv.x, h0[t][0], d); d = fmaf(wv.y, h0[t][1], d); d = fmaf(wv.z, h0[t][2], d); d = fmaf(wv.w, h0[t][3], d);
            }
            d += __shfl_xor(d, 16, 64);
            d += __shfl_xor(d, 32, 64);
            d += wp[m.pb[il]];
            CCSD_CT(4);
            if (kq == 0 && p0 + l15 < rows) epi(p0 + l15, 0, d);
            CCSD_CT(5);
            return;
        }
        chain_layer<NH, NO>(wp + m.pw[il] + l4, wp + m.pb[il] + 4 * kq, false, h0, yo);
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
    const int wave = wave_index(), nw = blockDim.x >> 6;
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
    if (!BATCH) {                    // plain loop (kept for reference / diagnostics)
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

// gcn_tile_multi: NC adjacent 16-column tiles of a DenseGCNConv for all nodes of one graph by one wave -- the same two
// products as gcn_tile, with everything that does not depend on the column tile done once: the x fragments of the first
// product and the masked, unit-diagonal adjacency fragments of the second are shared by the NC accumulators.  Wp: packed
// weights [fin][cp] followed by [cp] biases (AttnLayerD::qkvp), zero padded: no column bounds logic on the loads.
template <int NTN, int NC, class OUT>
CCSD_DEV void gcn_tile_multi(const float* xT, int ldn, int fin, int N, const float* A, const float* dinv, int col0, int ncols,
                             const float* __restrict__ Wp, int cp, OUT out) {
#ifdef CCSD_EMU
    for (int cc = 0; cc < 16 * NC; ++cc) {
        const int col = col0 + cc;
        if (col >= ncols) break;
        float xw[16 * NTN];
        for (int j = 0; j < N; ++j) {
            float acc = 0.f;
            for (int k = 0; k < fin; ++k) acc = fmaf(xT[k * ldn + j], Wp[k * cp + col], acc);
            xw[j] = acc * dinv[j];
        }
        for (int i = 0; i < N; ++i) {
            float acc = 0.f;
            for (int j = 0; j < N; ++j) acc = fmaf((i == j) ? 1.f : A[i * N + j], xw[j], acc);
            out(i, col, fmaf(acc, dinv[i], Wp[fin * cp + col]));
        }
    }
#else
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    int lane = threadIdx.x & 63;
    asm volatile("" : "+v"(lane));          // keeps each instantiation's index arithmetic inside it (see mlp_chain_tile)
    const int l15 = lane & 15, kq = lane >> 4;
    const float* wcol = Wp + col0 + l15;    // column col0 + 16 c + l15 of the packed block (always inside the padding)
    f32x4 xw[NC][NTN];
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int tn = 0; tn < NTN; ++tn) xw[c][tn] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int ks = (fin + 3) >> 2;
    for (int s00 = 0; s00 < ks; s00 += 4) {  // weights of four k-steps fetched together: one L2 round trip per four steps
        float bw[4][NC];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = 4 * (s00 + u) + kq, kc = k < fin ? k : fin - 1;
#pragma unroll
            for (int c = 0; c < NC; ++c) bw[u][c] = wcol[kc * cp + 16 * c];
        }
        // (the x fragments of the four steps are requested before the first MFMA: inside the per-step branch every LDS read would
        // sit in a basic block of its own and be waited for on the spot)
        float avs[4][NTN];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = 4 * (s00 + u) + kq, kc = k < fin ? k : fin - 1;
#pragma unroll
            for (int tn = 0; tn < NTN; ++tn) {
                const int j = 16 * tn + l15;
                const float a0 = xT[kc * ldn + (j < N ? j : N - 1)];
                avs[u][tn] = (j < N && k < fin) ? a0 : 0.f;              // rows beyond N / k beyond fin contribute nothing
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (s00 + u < ks) {
#pragma unroll
                for (int tn = 0; tn < NTN; ++tn) {
#pragma unroll
                    for (int c = 0; c < NC; ++c) xw[c][tn] = __builtin_amdgcn_mfma_f32_16x16x4f32(avs[u][tn], bw[u][c], xw[c][tn], 0, 0, 0);
                }
            }
    }
#pragma unroll
    for (int tn = 0; tn < NTN; ++tn)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = 16 * tn + 4 * kq + r;
            const float dj = j < N ? dinv[j < N ? j : N - 1] : 0.f;
#pragma unroll
            for (int c = 0; c < NC; ++c) xw[c][tn][r] *= dj;
        }
    float bb[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) bb[c] = wcol[fin * cp + 16 * c];
#pragma unroll
    for (int ti = 0; ti < NTN; ++ti) {
        const int i = 16 * ti + l15, ic = i < N ? i : N - 1;
        f32x4 acc[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
        float afr[NTN][4], dio[4];
#pragma unroll
        for (int tn = 0; tn < NTN; ++tn)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int j = 16 * tn + 4 * kq + jj, jc = j < N ? j : N - 1;
                const float a0 = A[ic * N + jc];
                afr[tn][jj] = (i < N && j < N) ? (i == j ? 1.f : a0) : 0.f;
            }
#pragma unroll
        for (int r = 0; r < 4; ++r) { const int io = 16 * ti + 4 * kq + r; dio[r] = dinv[io < N ? io : N - 1]; }
#pragma unroll
        for (int tn = 0; tn < NTN; ++tn)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
#pragma unroll
                for (int c = 0; c < NC; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(afr[tn][jj], xw[c][tn][jj], acc[c], 0, 0, 0);
            }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int io = 16 * ti + 4 * kq + r;
            if (io < N) {
#pragma unroll
                for (int c = 0; c < NC; ++c)
                    if (col0 + 16 * c + l15 < ncols) out(io, col0 + 16 * c + l15, fmaf(acc[c][r], dio[r], bb[c]));
            }
        }
    }
#endif
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
    const int wave = wave_index(), lane = threadIdx.x & 63;
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
    const int wave = wave_index(), lane = threadIdx.x & 63;
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
// the same over NP accumulators that share their geometry: f(m_local, n_local, v[NP][4])
template <int NP, class Fn>
CCSD_DEV void tile_foreach4n(TileAcc* t, Fn f) {
#ifdef CCSD_EMU
    for (int i = 0; i < T_BM; i += 4)
        for (int j = 0; j < T_BN; ++j) {
            float v[NP][4];
            for (int q = 0; q < NP; ++q)
                for (int r = 0; r < 4; ++r) v[q][r] = t[q].a[i + r][j];
            f(i, j, v);
        }
#else
    const int wave = wave_index(), lane = threadIdx.x & 63;
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32, l15 = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float v[NP][4];
#pragma unroll
            for (int q = 0; q < NP; ++q)
#pragma unroll
                for (int r = 0; r < 4; ++r) v[q][r] = t[q].a[i][j][r];
            f(wm + 16 * i + 4 * kq, wn + 16 * j + l15, v);
        }
#endif
}
