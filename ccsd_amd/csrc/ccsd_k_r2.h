// ccsd_k_r2.h -- k_r2: the fused rank-2 kernel (one complex per workgroup, rank2 block LDS-resident)
// Part of the kernel source of libccsd_hip.so (see ccsd_kernels.h for the map).
#pragma once
#include "ccsd_baked_qm9.h"
#include "ccsd_rank2_common.h"

#ifndef CCSD_R2_LB
#define CCSD_R2_LB 4       // float4 loads in flight per thread in the block load (without the raw scores alongside)
#endif
struct R2Args {
    const float* rank2; const float* adj; const float* flags;
    const unsigned long long* offbits;     // per-sample bitmask of switched-off nodes
    float* P0; float* P1;
    float* U1;             // [B][wc_1]: fr . Wcat_1 of the complex (linear mlp_value: P1 then holds the raw (F o fr) Wcat_1)
    int want_p;            // write the hodge projections (the A-network will run on the same state)
    // The caller guarantees that the rank2 state is MASKED (fl F fr == F bit for bit: every state ccsd_init_state or an update epilogue
    // of this library produced).  The sampler loop's launches (ccsd_sampler_run) set it: the layer-1 projection factor Q_1 = (F o fr) Wcat_1
    // is then a plain F Wcat_1 tile (no cell-mask conversion / product per element in its loader), and u_1 = fr . Wcat_1 -- a function of
    // the flags alone -- is not recomputed: U1 / U1b keep what the run's first (general) launch wrote.
    int masked;
    int stagger_mask, stagger_sleep;   // workgroups with (blockIdx.x & mask) != 0 start `sleep` x 64 cycles late (see launch_r2)
    // Merged launch (ccsd_sampler_run, E = 36 geometry): the predictor half-step of PC step i, then -- on the new rank2 block, which
    // the epilogue also wrote back into LDS -- the norms pass of the corrector of step i + 1 (its ScoreNetworkF score to HBM, both
    // norm partials, the hodge projections of the new state into the second buffer set): one block load instead of two.
    int merge;
    unsigned int draw_r2;  // the next step's corrector draw (flat groups)
    float* P0b; float* P1b; float* U1b;
    float* net2; float* part2;
    int ldk, ldh;
    long long* dbg;
    const float* wp;       // packed buffer (Wcat^T of the hodge projections)
    CorrFuse cf;
};

// MT = ceil(E / 16) row tiles (1..4); RS (affine phase 2 only): plain MFMA steps covering E mod 16 behind the MT - 1 full
// 16-wide blocks of the contraction index (0: the last block is taken whole, zero padded -- E mod 16 == 0 or > 12);
// AFFINE: ScoreNetworkF folds to alpha F + beta HF + gamma; GEN1: general (non-affine) mlp_value in the hodge branch.
// Compile-time so that the common variant carries no general-path code.
//
// Row strips (ST, the qm9 geometry E = 36 = 2 x 16 + 4): on gfx950 an fp32 MFMA runs at the fp32 vector rate on the vector
// pipe, so the padding rows of a 16-row tile cost exactly as much as useful ones -- a third row tile holding 4 of 16 rows
// wastes a quarter of all matrix time.  With ST the last E mod 16 <= 4 rows are a 4-ROW STRIP computed by
// v_mfma_f32_4x4x1_16B_f32 (16 independent 4x4x1 blocks per instruction, 8.8 cycles measured: tools/ubench/mfma_4x4x1.hip):
// lane l = 16 kc + 4 cg + j takes strip row (l & 3) as A and column 4 cg + j as B of the block (column group cg, k class kc),
// i.e. 4 rows x 16 columns x 4 k values per instruction -- the SAME operand addresses as the 16x16x4 tile code (B operand of
// column l & 15, k slot group l >> 4), only the A row differs; the four k classes are summed with two lane swaps.
// QM9 (1, 2): the qm9 geometry (E = 36, K = 466, N = 9, LDS strides 488 / 36) as compile-time constants: the index arithmetic on these
// strides folds into immediates (as k_xa<false, XA_PLAIN9>); the host selects the instance only when the plan matches (r2_qm9()).
template <int MT, int RS, bool AFFINE, bool GEN1, int QM9 = 0>
__global__ __launch_bounds__(512, 4) void k_r2(const PlanD* __restrict__ plan, const float* __restrict__ w,
                                            const unsigned char* __restrict__ edges,
                                            const unsigned long long* __restrict__ cells, R2Args ra, RankEpi ep,
                                            NoiseArgs na) {
    CCSD_DYN_SMEM(sm);
    // QM9 with a baked plan (ccsd_baked_qm9.h; r2_qm9() == 2 when the plan's architecture bytes equal the baked ones): every plan
    // field but the weight-derived affine fold -- read through `pw` -- is a compile-time constant
    constexpr bool BAKED = QM9 == 2 && CCSD_BAKED_QM9_SIZE == sizeof(PlanD);
    const PlanD& pw = *plan;
    const PlanD& p = BAKED ? *reinterpret_cast<const PlanD*>(CCSD_BAKED_QM9_PLAN) : *plan;
    const int E = QM9 ? 36 : p.E, K = QM9 ? 466 : p.K, N = QM9 ? 9 : p.N, NN = N * N, ldk = QM9 ? 488 : ra.ldk, ldh = QM9 ? 36 : ra.ldh;
    // (the thread count stays a run-time value even in the QM9 instance: as a constant the block-load and tile loops were unrolled and
    // rescheduled into a slower kernel, 158 -> 178 us)
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
#ifndef CCSD_EMU
    __shared__ int s_hdone;              // H-tile tasks finished (phase 1 -> 2 hand-over)
#endif
    const float* Fg = ra.rank2 + (size_t)b * E * K;
    const FastDiv dK(K);
    constexpr bool ST = AFFINE && !GEN1 && MT == 3 && RS == 1;   // E = 33..36: two full row tiles + one 4-row strip
    constexpr int MTF = ST ? MT - 1 : MT;                 // full 16-row tiles
    constexpr int E0 = 16 * MTF;                          // first strip row (ST)

    // ---- phase 0: masks; rank2 block -> LDS (with the Langevin corrector's work on it, see below); adjacency powers
    stamp(ra.dbg, 0);
#ifndef CCSD_EMU
    if (tid == 0) s_hdone = 0;
#endif
    // switched-off nodes of the complex (k_flagbits): a uniform address, i.e. one scalar load per wave -- no LDS round trip
    const unsigned long long off = ra.offbits[b];
    // Second hodge layer's projections.  Linear mlp_value (every shipped checkpoint): rank2'[e,k] = fl[e] fr[k] (s[e] F[e,k] + b)
    // with s[e] = sum_c w_c a_c[e] from the adjacency powers, so  P_1 = rank2' Wcat_1 = fl (s ((F o fr) Wcat_1) + b (fr Wcat_1)):
    // this kernel delivers the two adjacency-independent factors -- Q_1 = (F o fr) Wcat_1 (in P1) and u_1 = fr Wcat_1 (in U1) --
    // and k_xa, which holds the adjacency, applies s, b and fl.  Only a general (non-linear) mlp_value needs the adjacency
    // powers here (rank2' is then formed element by element in the GEMM's loader).
    const int hodge2 = (p.h_L > 1) && ra.want_p;
    const bool adjpow = hodge2 && p.hl[0].mval.n > 1;
    // mask tables, zero padding of the K tail, adjacency of the general mlp_value path; ends with the barrier that publishes them
    // (`between`: called after the tables' own global loads are requested and before they are used -- the block load puts its first
    // batch there, so that the in-order return of vector loads delivers the few table bytes first)
    auto build_tables = [&](auto&& between) {
        unsigned long long cw[2];
        unsigned int ew = 0;
#pragma unroll
        for (int u = 0; u < 2; ++u) { const int k = tid + u * nth; cw[u] = cells[k < K ? k : K - 1]; }
        if (tid < 64) ew = *reinterpret_cast<const unsigned short*>(edges + 2 * (tid < E ? tid : E - 1));
        between();
#pragma unroll
        for (int u = 0; u < 2; ++u) { const int k = tid + u * nth; if (k < Kp4) sFrb[k] = (k < K && !(cw[u] & off)) ? 1 : 0; }
        for (int k = tid + 2 * nth; k < Kp4; k += nth) sFrb[k] = (k < K && !(cells[k] & off)) ? 1 : 0;
        if (tid < 64) sFl[tid] = (tid < E && !(((off >> (ew & 0xffu)) | (off >> (ew >> 8))) & 1ull)) ? 1.f : 0.f;
        for (int e = tid + nth; e < 64; e += nth) sFl[e] = e < E ? edge_on(off, edges, e) : 0.f;    // (fewer than 64 threads: host emulation)
        for (int t = tid; t < E * (Kp4 - K); t += nth) { const int e = t / (Kp4 - K), k = K + t % (Kp4 - K); sF[e * ldk + k] = 0.f; }
        if (adjpow) {
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
        stamp(ra.dbg, 8);
    };
    // The block load.  The Langevin corrector's rank2 draw is keyed by FLAT groups of four consecutive elements (NoiseArgs::flat_r)
    // = one 16-byte load of this loop, so the corrector's work on rank2 happens here, where the block streams through registers:
    //  * norms launch (MODE_NORMS): the noise norm  sum (z fl fr)^2  of the draw (gen_noise_rank2 + torch.norm, cc_utils.py:613-615,
    //    solver.py:793-797) -- the epilogue then only squares the score;
    //  * predictor launch of ccsd_sampler_run (cf.on): the fused corrector apply  F <- fma(c2, z fl fr, fma(c1, net, F))  with the raw
    //    scores of the norms pass loaded alongside (same expression as k_langevin_apply), one pass, no LDS read-modify-write.
    // The loop is double buffered: the loads of batch i + 1 are in flight while batch i is processed (Philox + Box-Muller are ~100
    // vector instructions per group); the first batch is requested before the mask tables are built.  With K even (and E K a
    // multiple of 4) a group is two aligned pairs (e, k..k+1), (e', k'..k'+1): masks and LDS stores go pair-wise.
    float s_net = 0.f, s_z = 0.f;
    {
        float c1f = 0.f, c2f = 0.f;
        if (ra.cf.on) corr_coef(ra.cf, 2, &c1f, &c2f);
        NoiseArgs nc = na;                                   // the corrector draw of this launch
        if (ra.cf.on) { nc.zr = nullptr; nc.draw_r = ra.cf.draw_r; }
        const bool znorm = ep.mode == MODE_NORMS && na.flat_r, zuse = znorm || ra.cf.on;
        const float* Ng = ra.cf.on ? ra.cf.net_r + (size_t)b * E * K : Fg;
        const int EK = E * K;
        if ((EK & 3) == 0 && (K & 1) == 0) {
            const float4* F4 = reinterpret_cast<const float4*>(Fg);
            const float4* N4 = reinterpret_cast<const float4*>(Ng);
            const float4* Z4 = reinterpret_cast<const float4*>(nc.zr ? nc.zr + (size_t)b * EK : Fg);
            const int n4 = EK >> 2;
            const FastDiv dK2(K >> 1);                        // pair index -> (row, pair within the row)
            auto phase0 = [&](auto LB_, auto CF_, auto ZN_) {
                constexpr int LB = decltype(LB_)::value;      // float4 loads per thread and batch
                constexpr bool CF = decltype(CF_)::value;     // fused corrector apply: the raw scores are loaded alongside
                constexpr bool ZN = decltype(ZN_)::value;     // noise norm of the corrector draw
                constexpr int LN = CF ? LB : 1;
                float4 va[LB], vb[LB], qa[LN], qb[LN];
                auto issue = [&](float4* v, float4* q, int base) {
#pragma unroll
                    for (int u = 0; u < LB; ++u) { const int i4 = base + u * nth; v[u] = F4[i4 < n4 ? i4 : n4 - 1]; }
                    if (CF) {
#pragma unroll
                        for (int u = 0; u < LB; ++u) { const int i4 = base + u * nth; q[CF ? u : 0] = N4[i4 < n4 ? i4 : n4 - 1]; }
                    }
                };
                auto consume = [&](const float4* v, const float4* q, int base) {
#pragma unroll
                    for (int u = 0; u < LB; ++u) {
                        const int i4 = base + u * nth;
                        if (i4 < n4) {
                            int e, kp;
                            dK2.divmod(2 * i4, e, kp);                       // first pair of the group: row e, columns 2 kp, 2 kp + 1
                            const int k = 2 * kp;
                            const bool wrap = k + 2 == K;                    // second pair starts the next row
                            const int e1 = wrap ? e + 1 : e, k1 = wrap ? 0 : k + 2;
                            float vv[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
                            if (CF || ZN) {
                                float z[4];
                                if (nc.zr) { const float4 z4 = Z4[i4]; z[0] = z4.x; z[1] = z4.y; z[2] = z4.z; z[3] = z4.w; }
                                else philox_normal4(nc.seed, nc.draw_r, nc.b_off + b, (unsigned)i4, z);
                                const unsigned f0 = *reinterpret_cast<const unsigned short*>(sFrb + k), f1 = *reinterpret_cast<const unsigned short*>(sFrb + k1);
                                const float fl0 = sFl[e], fl1 = sFl[e1];
                                const float m[4] = {fl0 * (float)(f0 & 0xffu), fl0 * (float)(f0 >> 8), fl1 * (float)(f1 & 0xffu), fl1 * (float)(f1 >> 8)};
                                const float4 n4v = q[CF ? u : 0];
                                const float nn[4] = {n4v.x, n4v.y, n4v.z, n4v.w};
#pragma unroll
                                for (int j = 0; j < 4; ++j) {
                                    const float zz = z[j] * m[j];
                                    if (ZN && !CF) s_z = fmaf(zz, zz, s_z);
                                    if (CF) vv[j] = fmaf(c2f, zz, fmaf(c1f, nn[j], vv[j]));
                                }
                                if (CF && ZN) {
                                    // merged launch: the noise norm of the NEXT step's corrector draw (the norms pass that follows in this
                                    // launch needs it) is taken here too -- same groups, same masks, the same per-thread order as the norms
                                    // launch's own block load (bitwise the same sum) -- while this phase waits for its loads; it used to be a
                                    // loop of its own ahead of the second pass, where nothing hid its ~100 vector instructions per group
                                    float z2[4];
                                    philox_normal4(na.seed, ra.draw_r2, na.b_off + b, (unsigned)i4, z2);
#pragma unroll
                                    for (int j = 0; j < 4; ++j) { const float zz = z2[j] * m[j]; s_z = fmaf(zz, zz, s_z); }
                                }
                            }
                            *reinterpret_cast<float2*>(sF + e * ldk + k) = make_float2(vv[0], vv[1]);
                            *reinterpret_cast<float2*>(sF + e1 * ldk + k1) = make_float2(vv[2], vv[3]);
                        }
                    }
                };
                const int step = LB * nth;
                int base = tid;
                build_tables([&]() { issue(va, qa, base); });
                while (true) {
                    const bool more_b = base - tid + step < n4;              // (uniform)
                    if (more_b) issue(vb, qb, base + step);
                    consume(va, qa, base);
                    if (!more_b) break;
                    base += step;
                    const bool more_a = base - tid + step < n4;
                    if (more_a) issue(va, qa, base + step);
                    consume(vb, qb, base);
                    if (!more_a) break;
                    base += step;
                }
            };
            typedef std::integral_constant<bool, true> T_;
            typedef std::integral_constant<bool, false> F_;
            if (ra.cf.on && ra.merge) phase0(std::integral_constant<int, 3>{}, T_{}, T_{});
            else if (ra.cf.on) phase0(std::integral_constant<int, 3>{}, T_{}, F_{});
            else if (znorm) phase0(std::integral_constant<int, 3>{}, F_{}, T_{});
            else phase0(std::integral_constant<int, 3>{}, F_{}, F_{});
        } else {
            build_tables([]() {});
            for (int t = tid; t < EK; t += nth) {
                int e, k;
                dK.divmod(t, e, k);
                float v = Fg[t];
                if (zuse) {
                    const float z = nc.zr ? nc.zr[(size_t)b * EK + t] : philox_normal1(nc.seed, nc.draw_r, nc.b_off + b, (unsigned)t);
                    const float zz = z * sFl[e] * (float)sFrb[k];
                    if (znorm) s_z = fmaf(zz, zz, s_z);
                    if (ra.cf.on) v = fmaf(c2f, zz, fmaf(c1f, Ng[t], v));
                }
                sF[e * ldk + k] = v;
            }
        }
    }
    stamp(ra.dbg, 9);
    if (adjpow) {
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
    // ---- one pass = phase 1 + phase 2 over the LDS-resident block.  A merged launch (ra.merge) makes two: pass 0 the predictor
    // half-step (its epilogue writes the new rank2 to HBM AND back into the block), pass 1 the next corrector's norms pass on it.
    const int npass = ra.merge ? 2 : 1;
    for (int pass = 0; pass < npass; ++pass) {
    const bool wb = ra.merge && pass == 0;        // write the new state back into sF (pass 0 of a merged launch)
    if (pass == 1) {
        __syncthreads();                          // every wave is done with H and with its columns of the block
        ep.mode = MODE_NORMS; ep.out = ra.net2; ep.mean = nullptr; ep.part = ra.part2;
        na.zr = nullptr; na.draw_r = ra.draw_r2; na.flat_r = 1;
        ra.P0 = ra.P0b; ra.P1 = ra.P1b; ra.U1 = ra.U1b;
#ifndef CCSD_EMU
        if (tid == 0) s_hdone = 0;
#endif
        // (the noise norm of the next corrector draw was taken in phase 0, beside the fused apply)
        __syncthreads();
    }
    stamp(ra.dbg, 1);
#ifdef CCSD_STOP_DIAG
    // diagnostic build (tools/dev/phase_mix.sh): ablation bits written by the host behind the stamp rows -- 1: no projection k loops,
    // 2: no H k loops, 4: no column-tile epilogue, 8: epilogue without noise, 16: no column tiles at all
    const int diag = ra.dbg ? (int)ra.dbg[((size_t)gridDim.x + 254) * 64 + 62] : 0;
#else
    constexpr int diag = 0;
#endif
    const HodgeLayerD& h0 = p.hl[0];
    const HodgeLayerD& h1 = p.hl[1];
    const bool doP0 = ra.want_p && p.h_L > 0, doP1 = hodge2;
    const bool lin1 = doP1 && h0.mval.n == 1;      // rank2' affine in rank2: fold it around the GEMM
    const bool pm = !GEN1 && lin1 && ra.masked;    // masked state: Q_1 as a plain projection tile, U1 untouched
    const int wc0 = doP0 ? h0.wc : 0, wc1 = doP1 ? p.h_pw : 0;   // (layers >= 1: one concatenated projection, PlanD::h_pw)

    // ---- phase 1: H = F F^T (upper-triangle tiles, mirrored), P_0 = F Wcat_0, P_1 = rank2' Wcat_1.
    // One 16x16 output tile over the full K per task; a wave runs two tasks interleaved (independent MFMA
    // chains).  Operands of eight k-steps are fetched at once; the weight fragments, which come from L2,
    // are double-buffered in registers one batch ahead.  No atomics: results are bitwise reproducible.
    stamp(ra.dbg, 2);
    const int ks = Kp4 >> 2;
    int nHtasks = 0;
    // task list: H full tiles (upper triangle), [H strip sets of 16 columns], P_0 full tiles, P_1 full tiles, [P_0 strip sets], [P_1 strip sets]
    const int nHf = p.f_cnum == 2 ? MTF * (MTF + 1) / 2 : 0, nHs = (ST && p.f_cnum == 2) ? MT : 0;
    const int nH = nHf + nHs;
    const int nt0 = doP0 ? (wc0 + 15) >> 4 : 0, nt1 = doP1 ? (wc1 + 15) >> 4 : 0;
    const int nPf = MTF * (nt0 + nt1);
    const int ntask = nH + nPf + (ST ? nt0 + nt1 : 0);
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
            int ll = 1;                                        // layer of concatenated column n (padding columns: zero)
            while (ll + 1 < p.h_L && n >= p.h_poff[ll + 1]) ++ll;
            const HodgeLayerD& hn = ccsd_hl(p, ll);
            const int nl = n - p.h_poff[ll];
            for (int kk = 0; kk < K; ++kk) {
                const float frk = (float)sFrb[kk], wv = nl < hn.wc ? w[hn.wcat + (size_t)kk * hn.wc + nl] : 0.f;
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
            ra.P1[((size_t)b * E + m) * wc1 + n] = acc;
            if (lin1 && m == 0 && !pm) ra.U1[(size_t)b * wc1 + n] = un;
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
    auto run_tile = [&](int t) {
        const int lane = tid & 63, l15 = lane & 15, kq = lane >> 4;
        int type, i, c;                                       // 0: H(i, c >= i); 1: P_0(i, c); 2: P_1(i, c)
        bool strip = false;                                   // ST: rows E0.. as one 4-row strip, c = set of 16 output columns
        if (t < nHf) {
            type = 0; i = 0;
            int rem = t;
            while (rem >= MTF - i) { rem -= MTF - i; ++i; }
            c = i + rem;
        } else if (t < nH) {
            type = 0; i = MTF; c = t - nHf; strip = true;
        } else if (t < nH + MTF * nt0) {
            type = 1; i = (t - nH) / nt0; c = (t - nH) % nt0;
        } else if (t < nH + nPf) {
            type = 2; i = (t - nH - MTF * nt0) / nt1; c = (t - nH - MTF * nt0) % nt1;
        } else if (t < nH + nPf + nt0) {
            type = 1; i = MTF; c = t - nH - nPf; strip = true;
        } else {
            type = 2; i = MTF; c = t - nH - nPf - nt0; strip = true;
        }
        const int ra_ = strip ? E0 + (l15 & 3) : 16 * i + l15;
        const int arow = ra_ < E ? ra_ : E - 1;               // A row of this lane (clamped: never stored beyond E)
        const float* pa = sF + arow * ldk + 4 * kq;
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
        // the k loop, specialised on where the B operand lives (LDS / global), on the masked-A kind and on the MFMA shape
        auto kloop = [&](auto LB, auto K1, auto SR) {
            constexpr bool lb = decltype(LB)::v, k1 = decltype(K1)::v, sr = decltype(SR)::v;
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
                        const int e = arow;
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
                        if (!sr) upart = fmaf(fr0, b4.x, fmaf(fr1, b4.y, fmaf(fr2, b4.z, fmaf(fr3, b4.w, upart))));
                    }
                }
                r2_f32x4& acc = (u & 1) ? acc1 : acc0;            // block parity == slot parity (D even, block counter a multiple of D)
                if (sr) {                                          // 4-row strip: 16 blocks of 4x4x1 (4 column groups x 4 k classes)
                    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a4.x, b4.x, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a4.y, b4.y, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a4.z, b4.z, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a4.w, b4.w, acc, 0, 0, 0);
                } else {
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, b4.x, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, b4.y, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, b4.z, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, b4.w, acc, 0, 0, 0);
                }
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
        if ((diag & 1) && type != 0) { (void)0; }
        else if ((diag & 2) && type == 0) { (void)0; }
        else
        if (ST && strip) {
            if (type == 2 && !pm) kloop(BoolTag<false>{}, BoolTag<true>{}, BoolTag<true>{});
            else if (type == 0) kloop(BoolTag<true>{}, BoolTag<false>{}, BoolTag<true>{});
            else kloop(BoolTag<false>{}, BoolTag<false>{}, BoolTag<true>{});
        } else {
            if (type == 2 && !pm) kloop(BoolTag<false>{}, BoolTag<true>{}, BoolTag<false>{});
            else if (type == 0) kloop(BoolTag<true>{}, BoolTag<false>{}, BoolTag<false>{});
            else kloop(BoolTag<false>{}, BoolTag<false>{}, BoolTag<false>{});
        }
        r2_f32x4 acc = acc0 + acc1;
        const int n = 16 * c + l15;
        if (ST && strip) {
            // the lane holds rows E0 .. E0 + 3 of column n, summed over its k class only: add the four classes (lanes l, l ^ 16,
            // l ^ 32, l ^ 48) -- afterwards every class holds the sums and class kq stores row E0 + kq
            float v = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float tot = rows_allsum(acc[r]);
                v = kq == r ? tot : v;
            }
            const int m = E0 + kq;
            if (type == 0) {
                if (n < E && m < E) {
                    const float hv = (hmask && m == n) ? 0.f : v;   // hodge_mask zeroes the diagonal (cc_utils.py:964-969)
                    sH[m * ldh + n] = hv;
                    sH[n * ldh + m] = hv;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                if (lane == 0) atomicAdd(&s_hdone, 1);
            } else {
                const int wcn = type == 1 ? wc0 : wc1;
                float* const dstp = type == 1 ? ra.P0 : ra.P1;
                if (n < wcn && m < E) dstp[((size_t)b * E + m) * wcn + n] = v;
            }
            return;
        }
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
                float* dst = ra.P1 + ((size_t)b * E + mb) * wc1 + n;      // Q_1 (linear mlp_value) / P_1 (general)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (mb + r < E) dst[(size_t)r * wc1] = acc[r];
                if (!GEN1 && i == 0 && kq == 0 && !pm) ra.U1[(size_t)b * wc1 + n] = un;
            }
        }
    };
    // Phase 1 = the H tiles plus as many projection tiles as it takes to give every wave the same number of tasks.  The other
    // projection tiles (they depend on nothing but F) are run by the waves BETWEEN their column tiles of phase 2 (affine
    // path): phase 2's epilogue is pure VALU work (Philox, Box-Muller, masks, update), the projection tiles pure MFMA work, and
    // the partner waves of a SIMD then feed different pipes instead of queueing for the same one phase after phase.
    nHtasks = nH;
    const int nw1 = nth >> 6, wave1 = wave_index();
    int n1 = ntask;
    if (AFFINE && !wb) {     // (a pass that writes the new state back into the block runs every task first: they read all of it)
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
    if (wb) __syncthreads();
    else if (nHtasks > 0) {
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
    // norms launch: the noise norm of a flat-keyed corrector draw was taken in phase 0; only a draw keyed by 4-row groups (the S4
    // sampler's first draw) is generated -- and squared -- in the epilogue
    const bool eznorm = !na.flat_r;
    const int ntn = (K + 15) >> 4, ksE = Ep4 >> 2;
    auto epi4 = [&](int e0, int k, const float* hf) {
        if (e0 >= E || k >= K) return;
        float z[4] = {0.f, 0.f, 0.f, 0.f};
        if (ep.mode == MODE_PRED || (ep.mode == MODE_NORMS && eznorm)) raw_noise_r4(na, b, e0 >> 2, k, E, K, z);   // one Philox group = 4 edge rows
        const float fr = (float)sFrb[k];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int e = e0 + r;
            if (e >= E) continue;
            const float f = sF[e * ldk + k];
            const float m = sFl[e] * fr;                         // flags_left * flags_right, cc_utils.py:590
            const float hf1[CCSD_MAXCN - 1] = {hf[r], 0.f, 0.f};      // (cnum > 2 takes the tiled kernels: ccsd_plan::fused_r2)
            const float net = fnet_element<AFFINE>(pw, w, f, hf1, m);
            const size_t gi = ((size_t)b * E + e) * K + k;
            if (ep.mode == MODE_SCORE) {
                ep.out[gi] = ep.sscale * net;
            } else {
                const float zz = z[r] * m;                       // gen_noise_rank2, cc_utils.py:613-615
                if (ep.mode == MODE_NORMS) {
                    ep.out[gi] = net;
                    s_net = fmaf(net, net, s_net);
                    if (eznorm) s_z = fmaf(zz, zz, s_z);
                } else {
                    const float mean = fmaf(ep.pa, f, ep.pb * net);
                    if (ep.mean) ep.mean[gi] = mean;
                    const float nv = fmaf(ep.pc, zz, mean);
                    ep.out[gi] = nv;
                    if (wb) sF[e * ldk + k] = nv;
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
    if constexpr (AFFINE && ST) {
        // Affine ScoreNetworkF, E = 16 TF + (1..4) rows (qm9_CC: E = 36).  As the general affine path below, except that the
        // last rows are a 4-row strip:
        //  * per column tile: TF row tiles on v_mfma_f32_16x16x4_f32 (contraction index in the permuted slot order, the lane's B
        //    operands are the F values of its own accumulator rows) + the strip rows on v_mfma_f32_4x4x1_16B_f32 with the SAME B
        //    registers: lane (kq, l15) accumulates rows E0..E0+3 of column l15 over the contraction indices of its k class kq
        //    (16 t + 4 kq + j and E0 + kq -- every index exactly once over the four classes);
        //  * the strip's epilogue runs once per FOUR column tiles of the wave: a reduce-scatter over the k classes (three lane
        //    swaps per register) leaves the finished sums of column tile j in the lanes of class j, so all 64 lanes carry four
        //    real elements (one Philox group each) instead of a quarter-filled third row tile per column tile.
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        const int wave = wave_index(), nw = nth >> 6, lane = tid & 63, l15 = lane & 15, kq = lane >> 4;
        constexpr int TF = MT - 1;                          // full row tiles == full 16-wide blocks of the contraction index
        const int ar0 = l15 * ldh + 4 * kq;                // A, full tiles: row l15 of row tile 0, slot group kq (rows < E0: no clamp)
        const int srow = E0 + (l15 & 3) < E ? E0 + (l15 & 3) : E - 1;
        const int as0 = srow * ldh + 4 * kq;               // A, strip: row E0 + (lane & 3), slot group kq
        const int brow = 4 * kq * ldk;                     // B: row 4 kq of block 0; block t, step j: + (16 t + j) ldk
        const unsigned vo = (unsigned)(4 * kq * K + l15);  // element offset of (row 4 kq, column l15) inside the complex's block
        const bool cn2 = p.f_cnum == 2;
        float* const outp = ep.out + (size_t)b * E * K;
        auto coltile = [&](auto MODE_, auto INJ_, int tn) -> f32x4 {
            constexpr int MODE = decltype(MODE_)::value;   // 0 score, 1 norms, 2 predictor, 3 predictor + mean output
            constexpr bool INJ = decltype(INJ_)::value;    // host-supplied raw draws instead of Philox
            const float s_ = MODE == 0 ? ep.sscale : MODE == 1 ? 1.f : ep.pb;
            const float sa = s_ * pw.f_alpha, sb = s_ * pw.f_beta, sg = s_ * pw.f_gamma;
            const float pa = ep.pa, pc = ep.pc;
            float* const meanp = MODE == 3 ? ep.mean + (size_t)b * E * K : nullptr;
            const float* const zrp = INJ ? na.zr + (size_t)b * E * K : nullptr;
            const int n = 16 * tn + l15;
            const bool nin = n < K;
            const int nc = nin ? n : K - 1;
            // B operands: bv[4 t + j] = F[16 t + 4 kq + j][n] (permuted blocks), bvr[s] = F[E0 + 4 s + kq][n] (remainder)
            float bv[4 * TF], bvr[RS];
            const float* fb = sF + brow + nc;
#pragma unroll
            for (int t = 0; t < TF; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) bv[4 * t + j] = fb[(16 * t + j) * ldk];
#pragma unroll
            for (int s0 = 0; s0 < RS; ++s0) {
                const int c = E0 + 4 * s0 + kq;
                bvr[s0] = sF[(c < E ? c : E - 1) * ldk + nc];
            }
            f32x4 acc[TF], sacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < TF; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (cn2) {
                int ao = ar0, so = as0;
                asm volatile("" : "+v"(ao), "+v"(so));     // opaque per tile: the loop-invariant A loads must not be hoisted into registers
#pragma unroll
                for (int t = 0; t < TF; ++t) {
                    float4 a4[TF];
#pragma unroll
                    for (int i = 0; i < TF; ++i) a4[i] = *reinterpret_cast<const float4*>(sH + ao + 16 * i * ldh + 16 * t);
                    const float4 h4 = *reinterpret_cast<const float4*>(sH + so + 16 * t);
                    // interleaved: the row tiles' chains and the strip's chain are independent of each other
#pragma unroll
                    for (int i = 0; i < TF; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[i].x, bv[4 * t + 0], acc[i], 0, 0, 0);
                    sacc = __builtin_amdgcn_mfma_f32_4x4x1f32(h4.x, bv[4 * t + 0], sacc, 0, 0, 0);
#pragma unroll
                    for (int i = 0; i < TF; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[i].y, bv[4 * t + 1], acc[i], 0, 0, 0);
                    sacc = __builtin_amdgcn_mfma_f32_4x4x1f32(h4.y, bv[4 * t + 1], sacc, 0, 0, 0);
#pragma unroll
                    for (int i = 0; i < TF; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[i].z, bv[4 * t + 2], acc[i], 0, 0, 0);
                    sacc = __builtin_amdgcn_mfma_f32_4x4x1f32(h4.z, bv[4 * t + 2], sacc, 0, 0, 0);
#pragma unroll
                    for (int i = 0; i < TF; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[i].w, bv[4 * t + 3], acc[i], 0, 0, 0);
                    sacc = __builtin_amdgcn_mfma_f32_4x4x1f32(h4.w, bv[4 * t + 3], sacc, 0, 0, 0);
                }
#pragma unroll
                for (int s0 = 0; s0 < RS; ++s0) {
                    const int c = E0 + 4 * s0 + kq;        // contraction index of this lane's slot / k class
                    const int cc = (c < E ? c : E - 1) - 4 * kq;
#pragma unroll
                    for (int i = 0; i < TF; ++i) {
                        const float av = sH[ao + 16 * i * ldh + cc];
                        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(c < E ? av : 0.f, bvr[s0], acc[i], 0, 0, 0);
                    }
                    const float hv = sH[so + cc];
                    sacc = __builtin_amdgcn_mfma_f32_4x4x1f32(c < E ? hv : 0.f, bvr[s0], sacc, 0, 0, 0);
                }
            }
            if (nin && !(diag & 4)) {                      // false only for the padding columns of the last column tile
                const float fr = (float)sFrb[n];
                unsigned gi = vo + 16u * (unsigned)tn;
#pragma unroll
                for (int i = 0; i < TF; ++i) {
                    const int e0 = 16 * i + 4 * kq;
                    float z[4] = {0.f, 0.f, 0.f, 0.f};
                    if ((MODE >= 2 || (MODE == 1 && eznorm)) && !(diag & 8)) {
                        if (INJ) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) z[r] = zrp[gi + (unsigned)(r * K)];
                        } else {
                            philox_normal4(na.seed, na.draw_r, na.b_off + b, (unsigned)((4 * i + kq) * K + n), z);   // one Philox group = 4 edge rows
                        }
                    }
                    const float4 fl4 = *reinterpret_cast<const float4*>(sFl + e0);
                    const float flv[4] = {fl4.x, fl4.y, fl4.z, fl4.w};
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float f = bv[4 * i + r];
                        const float m = flv[r] * fr;                     // flags_left * flags_right, cc_utils.py:590
                        const float net = m * fmaf(sb, acc[i][r], fmaf(sa, f, sg));
                        const unsigned g = gi + (unsigned)(r * K);
                        if (MODE == 0) {
                            outp[g] = net;
                        } else {
                            const float zz = z[r] * m;                   // gen_noise_rank2, cc_utils.py:613-615
                            if (MODE == 1) {
                                outp[g] = net;
                                s_net = fmaf(net, net, s_net);
                                if (eznorm) s_z = fmaf(zz, zz, s_z);
                            } else {
                                const float mean = fmaf(pa, f, net);     // v_mean = pa v + pb net (pb folded into net)
                                if (MODE == 3) meanp[g] = mean;
                                const float nv = fmaf(pc, zz, mean);
                                outp[g] = nv;
                                if (wb) sF[(e0 + r) * ldk + n] = nv;     // merged launch: the next pass works on the new block
                            }
                        }
                    }
                    gi += 16u * (unsigned)K;
                }
            }
            return sacc;
        };
        // strip rows of up to four column tiles tn0, tn0 + nw, ...: p_j = the k-class partials of tile j.  Reduce-scatter: after two
        // 16-lane-row swaps and one half-wave swap the lanes of class j hold the finished sums of tile j (rows E0..E0+3, column l15).
        auto strip_finish = [&](auto MODE_, auto INJ_, const f32x4& p0, const f32x4& p1, const f32x4& p2, const f32x4& p3, int tn0) {
            constexpr int MODE = decltype(MODE_)::value;
            constexpr bool INJ = decltype(INJ_)::value;
            float hf[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float a = p0[r], c = p1[r], d = p2[r], e = p3[r];
                lane_swap16(a, c);
                const float a01 = a + c;          // rows: t0(0+1) t1(0+1) t0(2+3) t1(2+3)
                lane_swap16(d, e);
                const float a23 = d + e;          // rows: t2(0+1) t3(0+1) t2(2+3) t3(2+3)
                float u = a01, v = a23;
                lane_swap32(u, v);
                hf[r] = u + v;                    // rows: t0 t1 t2 t3
            }
            const int tn = tn0 + kq * nw;
            const int n = 16 * tn + l15;
            if (tn < ntn && n < K && !(diag & 4)) {
                const float s_ = MODE == 0 ? ep.sscale : MODE == 1 ? 1.f : ep.pb;
                const float sa = s_ * pw.f_alpha, sb = s_ * pw.f_beta, sg = s_ * pw.f_gamma;
                const float pa = ep.pa, pc = ep.pc;
                float* const meanp = MODE == 3 ? ep.mean + (size_t)b * E * K : nullptr;
                const float fr = (float)sFrb[n];
                const unsigned gi = (unsigned)(E0 * K + n);
                float z[4] = {0.f, 0.f, 0.f, 0.f};
                if ((MODE >= 2 || (MODE == 1 && eznorm)) && !(diag & 8)) {
                    if (INJ) {
                        const float* const zrp = na.zr + (size_t)b * E * K;
#pragma unroll
                        for (int r = 0; r < 4; ++r) z[r] = E0 + r < E ? zrp[gi + (unsigned)(r * K)] : 0.f;
                    } else {
                        philox_normal4(na.seed, na.draw_r, na.b_off + b, (unsigned)((E0 >> 2) * K + n), z);
                    }
                }
                const float4 fl4 = *reinterpret_cast<const float4*>(sFl + E0);   // sFl: 64 entries, zero beyond E
                const float flv[4] = {fl4.x, fl4.y, fl4.z, fl4.w};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int e = E0 + r;
                    if (e < E) {
                        const float f = sF[e * ldk + n];
                        const float m = flv[r] * fr;
                        const float net = m * fmaf(sb, hf[r], fmaf(sa, f, sg));
                        const unsigned g = gi + (unsigned)(r * K);
                        if (MODE == 0) {
                            outp[g] = net;
                        } else {
                            const float zz = z[r] * m;
                            if (MODE == 1) {
                                outp[g] = net;
                                s_net = fmaf(net, net, s_net);
                                if (eznorm) s_z = fmaf(zz, zz, s_z);
                            } else {
                                const float mean = fmaf(pa, f, net);
                                if (MODE == 3) meanp[g] = mean;
                                const float nv = fmaf(pc, zz, mean);
                                outp[g] = nv;
                                if (wb) sF[e * ldk + n] = nv;
                            }
                        }
                    }
                }
            }
        };
        typedef std::integral_constant<bool, false> NoInj;
        typedef std::integral_constant<bool, true> Inj;
        const bool inj = na.zr != nullptr && ep.mode != MODE_SCORE;
        const int cmode = ep.mode == MODE_SCORE ? 0 : ep.mode == MODE_NORMS ? 1 : ep.mean == nullptr ? 2 : 3;
        const int sel = cmode * 2 + (inj ? 1 : 0);
        // Static schedule of a wave: its column tiles tn = wave, wave + nw, ... with its leftover phase-1 tasks in between.  The
        // leftover list is rotated by the number of full H tiles: the waves that had the light (strip) tasks in phase 1 take the
        // full projection tiles.  (Static, hence the per-thread accumulation order of the Langevin norms is fixed and runs are
        // bitwise reproducible.)
        int pt = n1 + (wave + nw - (nHf % nw)) % nw;
        const int pslot = wave < (nw >> 1) ? 0 : 2;
        int cnt = 0;
        const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
        f32x4 sp0 = zero4, sp1 = zero4, sp2 = zero4, sp3 = zero4;
        for (int tn = wave; tn < ntn; tn += nw, ++cnt) {
            if (cnt == pslot && pt < ntask) { run_tile(pt); pt += nw; }
            if (diag & 16) continue;
            f32x4 sacc;
            switch (sel) {
                case 0: case 1: sacc = coltile(std::integral_constant<int, 0>{}, NoInj{}, tn); break;
                case 2: sacc = coltile(std::integral_constant<int, 1>{}, NoInj{}, tn); break;
                case 3: sacc = coltile(std::integral_constant<int, 1>{}, Inj{}, tn); break;
                case 4: sacc = coltile(std::integral_constant<int, 2>{}, NoInj{}, tn); break;
                case 5: sacc = coltile(std::integral_constant<int, 2>{}, Inj{}, tn); break;
                case 6: sacc = coltile(std::integral_constant<int, 3>{}, NoInj{}, tn); break;
                default: sacc = coltile(std::integral_constant<int, 3>{}, Inj{}, tn); break;
            }
            const int c4 = cnt & 3;
            if (c4 == 0) sp0 = sacc; else if (c4 == 1) sp1 = sacc; else if (c4 == 2) sp2 = sacc; else sp3 = sacc;
            if (c4 == 3 || tn + nw >= ntn) {
                const int tn0 = tn - c4 * nw;
                switch (sel) {
                    case 0: case 1: strip_finish(std::integral_constant<int, 0>{}, NoInj{}, sp0, sp1, sp2, sp3, tn0); break;
                    case 2: strip_finish(std::integral_constant<int, 1>{}, NoInj{}, sp0, sp1, sp2, sp3, tn0); break;
                    case 3: strip_finish(std::integral_constant<int, 1>{}, Inj{}, sp0, sp1, sp2, sp3, tn0); break;
                    case 4: strip_finish(std::integral_constant<int, 2>{}, NoInj{}, sp0, sp1, sp2, sp3, tn0); break;
                    case 5: strip_finish(std::integral_constant<int, 2>{}, Inj{}, sp0, sp1, sp2, sp3, tn0); break;
                    case 6: strip_finish(std::integral_constant<int, 3>{}, NoInj{}, sp0, sp1, sp2, sp3, tn0); break;
                    default: strip_finish(std::integral_constant<int, 3>{}, Inj{}, sp0, sp1, sp2, sp3, tn0); break;
                }
                sp0 = sp1 = sp2 = sp3 = zero4;
            }
        }
        for (; pt < ntask; pt += nw) run_tile(pt);
    } else if constexpr (AFFINE) {
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
        const int wave = wave_index(), nw = nth >> 6, lane = tid & 63, l15 = lane & 15, kq = lane >> 4;
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
            const float sa = s_ * pw.f_alpha, sb = s_ * pw.f_beta, sg = s_ * pw.f_gamma;
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
                            if (MODE >= 2 || (MODE == 1 && eznorm)) {
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
                                            if (eznorm) s_z = fmaf(zz, zz, s_z);
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
        const int wave = wave_index(), nw = nth >> 6, lane = tid & 63, l15 = lane & 15, kq = lane >> 4;
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
    }   // pass
    // (no phase 3: the epilogue wrote the results to HBM)
    stamp(ra.dbg, 4);
    if (ep.mode == MODE_NORMS) {
        float t2_[2] = {s_net, s_z};
        block_sums<2>(t2_, sRed);
        const float tn_ = t2_[0], tz_ = t2_[1];
        if (tid == 0) { ep.part[(size_t)b * 2 + 0] = tn_; ep.part[(size_t)b * 2 + 1] = tz_; }
    }
    stamp(ra.dbg, 5);
}
