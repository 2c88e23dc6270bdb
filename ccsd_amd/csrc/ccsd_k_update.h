// ccsd_k_update.h -- k_normsum, k_langevin_apply, k_s4_apply, k_init_state, k_quantize, k_rank2_cells
// Part of the kernel source of libccsd_hip.so (see ccsd_kernels.h for the map).
#pragma once
#include "ccsd_rank2_common.h"

// ---------------------------------------------------------------------------------------------
// k_normsum: sums[0..5] = sum_b sqrt(|net_x|^2), |net_adj|, |net_rank2|, |z_x|, |z_adj|, |z_rank2|
// (torch.norm(...).mean() numerators, solver.py:763-767).  One workgroup, deterministic order.
// ---------------------------------------------------------------------------------------------
__global__ void k_normsum(const float* __restrict__ norm2, const float* __restrict__ part, int B, int ntiles,
                          int is_cc, float* __restrict__ sums) {
    // up to 1024 threads: one sample per thread and one L2 round trip for B <= 1024; the six sums are reduced together (wave
    // butterflies, one barrier, a fixed-order pass over the waves) -- the launch sits between the norms pass and the predictor
    // kernels of every PC step, so its latency is on the step's critical path.  (Many tiles per sample: k_normpart first.)
    __shared__ float red[16 * 6];
    float acc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
        const float4 n4 = *reinterpret_cast<const float4*>(norm2 + (size_t)b * 4);
        acc[0] += sqrtf(n4.x);
        acc[1] += sqrtf(n4.y);
        acc[3] += sqrtf(n4.z);
        acc[4] += sqrtf(n4.w);
        if (is_cc) {
            float sn = 0.f, sz = 0.f;
            for (int t = 0; t < ntiles; ++t) {
                const float* p2 = part + ((size_t)b * ntiles + t) * 2;
                sn += p2[0];
                sz += p2[1];
            }
            acc[2] += sqrtf(sn);
            acc[5] += sqrtf(sz);
        }
    }
#ifdef CCSD_EMU
    for (int i = 0; i < 6; ++i) sums[i] = acc[i];
#else
    const int wave = wave_index(), lane = threadIdx.x & 63, nw = (blockDim.x + 63) >> 6;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        float v = acc[i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if (lane == 0) red[wave * 6 + i] = v;
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        float t = 0.f;
        for (int w = 0; w < nw; ++w) t += red[w * 6 + threadIdx.x];
        sums[threadIdx.x] = t;
    }
#endif
}

// k_normpart: first level of the norm reduction when a sample has many partials (tiled rank-2 path: one pair per 64 x 64 tile of
// k_hf_score, plus the chunk partials of k_noise_norm): out[b] = { sum_t part[b][t][0], sum_t part[b][t][1] + sum_c zpart[b][c] }.
// One workgroup per sample, fixed summation order (deterministic).  k_normsum then runs with ntiles = 1 on `out`.
__global__ void k_normpart(const float* __restrict__ part, int ntiles, const float* __restrict__ zpart, int nchunk,
                           float* __restrict__ out) {
    __shared__ float red[64];
    const int b = blockIdx.x;
    float sn = 0.f, sz = 0.f;
    for (int t = threadIdx.x; t < ntiles; t += blockDim.x) {
        const float* p2 = part + ((size_t)b * ntiles + t) * 2;
        sn += p2[0];
        sz += p2[1];
    }
    if (zpart)
        for (int c = threadIdx.x; c < nchunk; c += blockDim.x) sz += zpart[(size_t)b * nchunk + c];
    const float tn = block_sum(sn, red);
    const float tz = block_sum(sz, red);
    if (threadIdx.x == 0) { out[(size_t)b * 2 + 0] = tn; out[(size_t)b * 2 + 1] = tz; }
}

// k_noise_norm: sum of squares of the masked rank2 noise of a corrector draw, per sample and chunk (tiled rank-2 path, Philox
// noise: the draw is keyed by flat groups, NoiseArgs::flat_r, so the MFMA-layout epilogue of k_hf_score cannot produce it
// cheaply; pure arithmetic, no rank2 traffic).  zpart[b][chunk] = sum over the chunk's groups of (z fl fr)^2
// (gen_noise_rank2 + torch.norm, cc_utils.py:613-615, solver.py:793-797).  grid (nchunk, B); CH groups per workgroup.
#define CCSD_NN_CH 4096
template <int EC = 0, int KC = 0>       // E, K as compile-time constants (0: the arguments): GEO_EK in ccsd_api.h
__global__ void k_noise_norm(NoiseArgs na, MaskTab mt, int E_, int K_, float* __restrict__ zpart) {
    __shared__ float red[64];
    const int E = EC ? EC : E_, K = KC ? KC : K_;
    const int b = blockIdx.y, EK = E * K, ng = (EK + 3) >> 2;
    const int g0 = blockIdx.x * CCSD_NN_CH, g1 = g0 + CCSD_NN_CH < ng ? g0 + CCSD_NN_CH : ng;
    const FastDiv dK(K);
    float acc = 0.f;
    for (int g = g0 + threadIdx.x; g < g1; g += blockDim.x) {
        float z[4];
        raw_noise_rflat4(na, b, g, EK, z);
        int e, k;
        dK.divmod(4 * g, e, k);
        float m[4];
        group_masks(mt, b, E, K, e, k, m);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (4 * g + s < EK) {
                const float zz = z[s] * m[s];
                acc = fmaf(zz, zz, acc);
            }
        }
    }
    const float t = block_sum(acc, red);
    if (threadIdx.x == 0) zpart[(size_t)b * gridDim.x + blockIdx.x] = t;
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
template <int EC = 0, int KC = 0>       // E, K as compile-time constants (0: LangArgs'): GEO_EK in ccsd_api.h
__global__ void k_langevin_apply(LangArgs a, NoiseArgs na, MaskTab mt) {
    if (EC) { a.E = EC; a.K = KC; }
    const long long nxe = (long long)a.B * a.N * a.F, nae = (long long)a.B * a.N * a.N;
    // rank2: one thread per flat Philox group = four consecutive elements of the sample's (E, K) block (NoiseArgs::flat_r): with
    // E K a multiple of 4 every group is one aligned 16-byte load of the state, one of the raw score and one 16-byte store
    const int EK = a.E * a.K, ng = (EK + 3) >> 2;
    const long long nre = a.is_cc ? (long long)a.B * ng : 0;
    const long long total = nxe + nae + nre;
    const bool vec = (EK & 3) == 0;
    float c1x, c2x, c1a, c2a, c1r = 0.f, c2r = 0.f;
    langevin_coef(a, 0, &c1x, &c2x);
    langevin_coef(a, 1, &c1a, &c2a);
    if (a.is_cc) langevin_coef(a, 2, &c1r, &c2r);
    const FastDiv dK(a.K > 0 ? a.K : 1);
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        if (t < nre) {                                           // (the rank2 groups come first: the bulk of the work, aligned)
            const int b = (int)(t / ng), g = (int)(t - (long long)b * ng);
            const size_t base = (size_t)b * EK + 4 * (size_t)g;
            float v[4], nv[4], z[4];
            if (vec) {
                const float4 v4 = *reinterpret_cast<const float4*>(a.r + base), n4 = *reinterpret_cast<const float4*>(a.nr + base);
                v[0] = v4.x; v[1] = v4.y; v[2] = v4.z; v[3] = v4.w;
                nv[0] = n4.x; nv[1] = n4.y; nv[2] = n4.z; nv[3] = n4.w;
            } else {
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const bool in = 4 * g + s < EK;
                    v[s] = in ? a.r[base + s] : 0.f;
                    nv[s] = in ? a.nr[base + s] : 0.f;
                }
            }
            raw_noise_rflat4(na, b, g, EK, z);
            int e, k;
            dK.divmod(4 * g, e, k);
            float o[4], m[4];
            group_masks(mt, b, a.E, a.K, e, k, m);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const float zz = z[s] * m[s];
                o[s] = fmaf(c2r, zz, fmaf(c1r, nv[s], v[s]));
            }
            if (vec) *reinterpret_cast<float4*>(a.orr + base) = make_float4(o[0], o[1], o[2], o[3]);
            else {
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    if (4 * g + s < EK) a.orr[base + s] = o[s];
            }
        } else if (t < nre + nxe) {
            const long long u = t - nre;
            const int per = a.N * a.F, b = (int)(u / per), idx = (int)(u % per), i = idx / a.F;
            const float z = raw_noise_x(na, b, idx, per) * a.flags[(size_t)b * a.N + i];
            a.ox[u] = fmaf(c2x, z, fmaf(c1x, a.nx[u], a.x[u]));
        } else {
            const long long u = t - nre - nxe;
            const int per = a.N * a.N, b = (int)(u / per), ij = (int)(u % per), i = ij / a.N, j = ij % a.N;
            const float z = raw_noise_adj(na, b, i, j, a.N) * a.flags[(size_t)b * a.N + i] * a.flags[(size_t)b * a.N + j];
            a.oadj[u] = fmaf(c2a, z, fmaf(c1a, a.nadj[u], a.adj[u]));
        }
    }
}

// ---------------------------------------------------------------------------------------------
// k_ew1: the whole rank-2 side of a half-step when ScoreNetworkF is affine with cnum = 1 -- net = fl fr (alpha F + gamma), no
// Hodge Laplacian term (ScoreNetwork_F.py:175-217 with cnum = 1: pow_tensor_cc returns [F]; the N = 38 substitute of
// zinc250k_CC) -- so the score is element-wise and rank2 streams through once, 16 bytes per lane, in flat Philox groups:
//   SCORE  out = sscale net
//   NORMS  partial sums of net^2 and (z fl fr)^2 per (sample, chunk); the raw score is stored only on request (step-wise API:
//          ccsd_corrector_apply reads it; ccsd_sampler_run recomputes it in the apply below instead of a 2 x E K x 4 B round trip)
//   PRED   [fused Langevin corrector apply  F1 = fma(c2, z fl fr, fma(c1, net(F), F)), written to `f1`: the hodge projection
//          GEMM of the A-network reads it]  then  mean = pa F1 + pb net(F1),  out = mean + pc z' fl fr   (solver.py:797-801, 429-457)
// Same expressions as k_hf_score / k_langevin_apply.  grid (chunks of CCSD_NN_CH groups, B).
// ---------------------------------------------------------------------------------------------
struct Ew1Args {
    const float* r; float* out; float* mean; float* f1; float* net_out; float* part;
    int mode, apply;
    float sscale, pa, pb, pc, alpha, gamma;
    const float* sums; float ss, sde_alpha, snr, seps; unsigned int draw_corr;
    int E, K;
    MaskTab mt;                       // mask byte tables (k_masktab)
};
template <int EC = 0, int KC = 0>       // E, K as compile-time constants (0: Ew1Args'): GEO_EK in ccsd_api.h
__global__ void k_ew1(Ew1Args a, NoiseArgs na) {
    __shared__ float red[64];
    if (EC) { a.E = EC; a.K = KC; }
    const int b = blockIdx.y, EK = a.E * a.K, ng = (EK + 3) >> 2;
    const int g0 = blockIdx.x * CCSD_NN_CH, g1 = g0 + CCSD_NN_CH < ng ? g0 + CCSD_NN_CH : ng;
    const FastDiv dK(a.K);
    const bool vec = (EK & 3) == 0;
    float c1 = 0.f, c2 = 0.f;
    if (a.mode == MODE_PRED && a.apply) {
        const float gn = fabsf(a.ss) * a.sums[2], zn = a.sums[5];          // (corr_coef / langevin_coef for the rank2 target)
        const float q = a.snr * zn / gn;
        const float step = q * q * 2.f * a.sde_alpha;
        c1 = step * a.ss;
        c2 = sqrtf(step * 2.f) * a.seps;
    }
    NoiseArgs nc = na;                                                      // corrector draw (NORMS: the launch's own draw)
    if (a.mode == MODE_PRED) { nc.zr = nullptr; nc.draw_r = a.draw_corr; }
    float s_net = 0.f, s_z = 0.f;
    for (int g = g0 + threadIdx.x; g < g1; g += blockDim.x) {
        const size_t base = (size_t)b * EK + 4 * (size_t)g;
        float v[4];
        if (vec) { const float4 v4 = *reinterpret_cast<const float4*>(a.r + base); v[0] = v4.x; v[1] = v4.y; v[2] = v4.z; v[3] = v4.w; }
        else {
#pragma unroll
            for (int s = 0; s < 4; ++s) v[s] = 4 * g + s < EK ? a.r[base + s] : 0.f;
        }
        int e, k;
        dK.divmod(4 * g, e, k);
        float m[4];                                                            // flags_left * flags_right, cc_utils.py:590
        group_masks(a.mt, b, a.E, a.K, e, k, m);
        float o[4], mu[4], w1[4], nt[4];
        float zc[4] = {0.f, 0.f, 0.f, 0.f}, zp[4] = {0.f, 0.f, 0.f, 0.f};
        if (a.mode == MODE_NORMS || (a.mode == MODE_PRED && a.apply)) raw_noise_rflat4(nc, b, g, EK, zc);
        if (a.mode == MODE_PRED) raw_noise_rflat4(na, b, g, EK, zp);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            float f = v[s];
            float net = m[s] * fmaf(a.alpha, f, a.gamma);                     // fnet_element<AFFINE>, cnum = 1
            if (a.mode == MODE_SCORE) {
                o[s] = a.sscale * net;
            } else if (a.mode == MODE_NORMS) {
                const float zz = zc[s] * m[s];
                nt[s] = net;
                s_net = fmaf(net, net, s_net);
                s_z = fmaf(zz, zz, s_z);
            } else {
                if (a.apply) {
                    f = fmaf(c2, zc[s] * m[s], fmaf(c1, net, f));             // k_langevin_apply
                    w1[s] = f;
                    net = m[s] * fmaf(a.alpha, f, a.gamma);
                }
                const float mean = fmaf(a.pa, f, a.pb * net);                 // v_mean = pa*v + pb*net (k_hf_score, MODE_PRED)
                mu[s] = mean;
                o[s] = fmaf(a.pc, zp[s] * m[s], mean);
            }
        }
        auto st4 = [&](float* dst, const float* val) {
            if (vec) *reinterpret_cast<float4*>(dst + base) = make_float4(val[0], val[1], val[2], val[3]);
            else {
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    if (4 * g + s < EK) dst[base + s] = val[s];
            }
        };
        if (a.mode == MODE_NORMS) { if (a.net_out) st4(a.net_out, nt); }
        else {
            st4(a.out, o);
            if (a.mode == MODE_PRED) {
                if (a.apply) st4(a.f1, w1);
                if (a.mean) st4(a.mean, mu);
            }
        }
    }
    if (a.mode == MODE_NORMS) {
        const float tn = block_sum(s_net, red);
        const float tz = block_sum(s_z, red);
        if (threadIdx.x == 0) {
            a.part[((size_t)b * gridDim.x + blockIdx.x) * 2 + 0] = tn;
            a.part[((size_t)b * gridDim.x + blockIdx.x) * 2 + 1] = tz;
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
    const int eg_n = (E + 3) / 4, EK = E * K, ng = (EK + 3) >> 2;
    const long long nre = is_cc ? (na.flat_r ? (long long)B * ng : (long long)B * eg_n * K) : 0;
    const long long total = nxe + nae + nre;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        if (t >= nxe + nae && na.flat_r) {            // a corrector draw: flat groups (ccsd_noise_draws)
            const long long u = t - nxe - nae;
            const int b = (int)(u / ng), g = (int)(u - (long long)b * ng);
            float z[4];
            raw_noise_rflat4(na, b, g, EK, z);
            const unsigned long long off = offbits[b];
            int e = (4 * g) / K, k = (4 * g) % K;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                if (4 * g + s < EK) r[(size_t)b * EK + 4 * g + s] = edge_on(off, edges, e) * z[s] * cell_on(off, cells, k);
                if (++k == K) { k = 0; ++e; }
            }
        } else if (t < nxe) {
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
    const int wave = wave_index(), lane = threadIdx.x & 63, nw = blockDim.x >> 6;
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

