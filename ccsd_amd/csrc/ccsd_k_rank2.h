// ccsd_k_rank2.h -- k_flagbits and the tiled rank-2 kernels: k_gemm_h (H = F F^T), k_gemm_p / k_gemm_p0 (hodge projections), k_edgecoef, k_hf_score
// Part of the kernel source of libccsd_hip.so (see ccsd_kernels.h for the map).
#pragma once
#include "ccsd_rank2_common.h"

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
// k_masktab: the two flag masks of every complex as BYTE tables in the workspace -- mfr[b][k] = flags_right (cell k switched on),
// mfl[b][e] = flags_left (edge e switched on); get_rank2_flags, cc_utils.py:527-591.  The element-wise kernels of the general path
// (k_ew1, k_langevin_apply, k_noise_norm) then read one 32-bit word of mfr per 16-byte group of rank2 and one byte of mfl instead of
// an 8-byte cell word and two edge-table bytes PER ELEMENT.  Rows are padded to multiples of 4 (Kp, Ep), padding = 0.
__global__ void k_masktab(const unsigned long long* __restrict__ offbits, const unsigned char* __restrict__ edges,
                          const unsigned long long* __restrict__ cells, unsigned char* __restrict__ mfr, unsigned char* __restrict__ mfl,
                          int B, int E, int K, int Kp, int Ep) {
    const long long per = Kp + Ep, n = (long long)B * per;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long long)gridDim.x * blockDim.x) {
        const int b = (int)(t / per), r = (int)(t - (long long)b * per);
        const unsigned long long off = offbits[b];
        if (r < Kp) mfr[(size_t)b * Kp + r] = (r < K && !(cells[r] & off)) ? 1 : 0;
        else {
            const int e = r - Kp;
            mfl[(size_t)b * Ep + e] = (e < E && !(((off >> edges[2 * e]) | (off >> edges[2 * e + 1])) & 1ull)) ? 1 : 0;
        }
    }
}
// Tiles of the tiled rank-2 kernels are numbered so that ALL tiles of one complex run on ONE XCD: workgroups are dealt round-robin
// over the 8 XCDs (linear block id mod 8), each with its own 4 MB L2, and the tiles of a complex share its operands (k_gemm_h:
// every 64-row slab of F feeds 3-4 tiles; k_hf_score: every row tile of a column block reads the same E x 64 slab of F, every
// column block the same rows of H).  id = 8 (T (b / 8) + t) + b % 8  <->  (complex b, tile t): one HBM fetch per operand instead of
// one per sharing tile.  Placement is a speed matter only; grid = xcd_grid(B, T) workgroups, ids with b >= B return at once.
CCSD_DEV bool xcd_sample_tile(int id, int T, int B, int* b, int* t) {
    const int grp = id / (8 * T), rem = id - grp * 8 * T;
    *t = rem >> 3;
    *b = 8 * grp + (rem & 7);
    return *b < B;
}
static inline int xcd_grid(int B, int T) { return ((B + 7) / 8) * 8 * T; }

// ---------------------------------------------------------------------------------------------
// k_gemm_h: H[b] = (F[b] F[b]^T) * hodge_mask           hodge_laplacian + mask, cc_utils.py:929, 964-969
// grid xcd_grid(B, nt (nt + 1) / 2), nt = ceil(E/64): the upper-triangle tiles of a complex (H is symmetric: mirrored on store)
// ---------------------------------------------------------------------------------------------
#define H_BK 32   // k per slab (two 16-wide MFMA k blocks)
#define H_LD 40   // LDS row stride in floats: 16-byte aligned rows, == 8 mod 32 -> conflict-free ds_read_b128 fragments
// EC, KC: E and K as compile-time constants (0: the run-time arguments) -- the instance for the community_small geometry
template <int EC = 0, int KC = 0>
__global__ __launch_bounds__(256) void k_gemm_h(const float* __restrict__ rank2, float* __restrict__ H, int E_, int K_,
                                                int zero_diag, int B) {
    const int E = EC ? EC : E_, K = KC ? KC : K_;
    const int nt = (E + T_BM - 1) / T_BM;
    int b, t;
    if (!xcd_sample_tile((int)blockIdx.x, nt * (nt + 1) / 2, B, &b, &t)) return;
    int ty = 0;                                  // upper-triangle tile t -> (row tile ty, column tile tx >= ty), row-major
    while (t >= nt - ty) { t -= nt - ty; ++ty; }
    const int tx = ty + t;
    const int m0 = ty * T_BM, n0 = tx * T_BN;
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
    const int tid = threadIdx.x, wave = wave_index(), lane = tid & 63, l15 = lane & 15, kq = lane >> 4;
    const bool diag = tx == ty, vec = (K & 3) == 0;
    // Wave -> 16 x 16 sub-tiles.  Off-diagonal tiles: a 32 x 32 block (2 x 2 sub-tiles) per wave.  DIAGONAL tiles (half of a
    // community_small complex's six): H is symmetric, only the ten sub-tiles on or above the diagonal of the 4 x 4 grid are computed and
    // mirrored on store (the same products in the same order: bit-identical to computing both halves) -- waves 0 / 3 take the diagonal
    // blocks without their lower sub-tile (3 each), waves 1 / 2 one sub-tile row of the off-diagonal block each (2 each): 3 sub-tiles per
    // wave at most instead of 4, i.e. 3/4 of the matrix time of such a tile.
    const int wm = diag ? (wave == 0 || wave == 1 ? 0 : wave == 2 ? 16 : 32) : (wave >> 1) * 32;
    const int wn = diag ? (wave == 0 ? 0 : 32) : (wave & 1) * 32;
    const bool row1 = !diag || wave == 0 || wave == 3;     // sub-tile row 1 of the wave's block is live
    const bool r1c0 = !diag;                               // ... its column 0 too (never in a diagonal tile)
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
            const float4 a1 = *reinterpret_cast<const float4*>(As + ((row1 ? wm + 16 : wm) + l15) * H_LD + 16 * t + 4 * kq);
            const float4 b0 = *reinterpret_cast<const float4*>(Bp + (wn + l15) * H_LD + 16 * t + 4 * kq);
            const float4 b1 = *reinterpret_cast<const float4*>(Bp + (wn + 16 + l15) * H_LD + 16 * t + 4 * kq);
            const float av0[4] = {a0.x, a0.y, a0.z, a0.w}, av1[4] = {a1.x, a1.y, a1.z, a1.w};
            const float bv0[4] = {b0.x, b0.y, b0.z, b0.w}, bv1[4] = {b1.x, b1.y, b1.z, b1.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc.a[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av0[j], bv0[j], acc.a[0][0], 0, 0, 0);
                acc.a[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av0[j], bv1[j], acc.a[0][1], 0, 0, 0);
            }
            if (r1c0) {
#pragma unroll
                for (int j = 0; j < 4; ++j) acc.a[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av1[j], bv0[j], acc.a[1][0], 0, 0, 0);
            }
            if (row1) {
#pragma unroll
                for (int j = 0; j < 4; ++j) acc.a[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av1[j], bv1[j], acc.a[1][1], 0, 0, 0);
            }
        }
    }
    {
        const int ldH = h_ld(E);
        float* Hb = H + (size_t)b * E * ldH;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (i == 1 && !(j == 0 ? r1c0 : row1)) continue;          // (dead sub-tiles of a diagonal tile)
                const int n = n0 + wn + 16 * j + l15;
                if (n >= E) continue;
#pragma unroll
                for (int s2 = 0; s2 < 4; ++s2) {
                    const int m = m0 + wm + 16 * i + 4 * kq + s2;
                    if (m < E) {
                        const float hv = (zero_diag && m == n) ? 0.f : acc.a[i][j][s2];
                        Hb[(size_t)m * ldH + n] = hv;
                        Hb[(size_t)n * ldH + m] = hv;                        // mirror (diagonal tiles: their lower sub-tiles)
                    }
                }
            }
    }
    return;
#endif
    const int ldH = h_ld(E);
    float* Hb = H + (size_t)b * E * ldH;
    tile_foreach4(acc, [&](int ml, int nl, const float* v) {
        const int n = n0 + nl;
        if (n >= E) return;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int m = m0 + ml + s;
            if (m < E) {
                const float hv = (zero_diag && m == n) ? 0.f : v[s];
                Hb[(size_t)m * ldH + n] = hv;
                if (tx != ty) Hb[(size_t)n * ldH + m] = hv;
            }
        }
    });
}

#ifndef CCSD_EMU
// ---------------------------------------------------------------------------------------------
// k_gemm_h_full<EC, KC>: H[b] = (F[b] F[b]^T) * hodge_mask with ONE workgroup per complex -- the 12 x 12 grid of 16 x 16 sub-tiles of a
// 192-row block (E <= 192: the community_small geometry, E = 190), upper triangle only (78 sub-tiles), mirrored on store.
// Why: k_gemm_h's six 64 x 64 tiles of a complex each stream their two row slabs of F, and with ~250 workgroups in flight per XCD one
// k step of all of them touches as much as the 4 MB L2 holds -- the sibling tiles miss: 1268 MB of HBM traffic per launch against
// 443 MB of F + 74 MB of H (r04_pmc_community_small_CC.json), 4.2 TB/s, i.e. the kernel is HBM-bound on re-reads.  Here a k slab of F
// ([192][32], 30 KB) is staged ONCE per complex and every sub-tile takes both its operands from it: traffic = F once + H.
// Sub-tile (i, j), i <= j: A rows 16 i .., B rows 16 j .. of the same slab; k order and operand slots as in k_gemm_h (blocks of 16
// ascending, step j of a block = slot 4 kq + j) => bit-identical results.  Waves: 0 / 1 the triangles of rows 0-5 / 6-11 (less one
// sub-tile each), 2 / 3 the rows 0-2 / 3-5 of the off-diagonal 6 x 6 block plus the sub-tile given up by wave 0 / 1, whose fragments
// they hold anyway: 20 / 20 / 19 / 19 sub-tiles.  LDS: two slabs (one barrier per slab); B complexes = B workgroups, two per CU.
// ---------------------------------------------------------------------------------------------
// ---- EXPERIMENT (plan option CCSD_SPLIT_BF16 = 3 | 6, never the default): split-precision contraction on the bf16 matrix pipe.
// An fp32 operand is cut into bf16 pieces: h = its top 16 bits (truncation: exact), then l = bf16(x - h) (round to nearest) -- SPLIT 2,
// |x - h - l| <= 2^-16 |x| -- or m = the top 16 bits of x - h and l = bf16(x - h - m) -- SPLIT 3, <= 2^-24 |x|, i.e. every bit of x.
// A product a b is then the fp32-accumulated sum of h_a h_b + h_a l_b + l_a h_b (three v_mfma_f32_16x16x32_bf16, 16 cycles each for 32 k
// against eight fp32 MFMAs of 32 cycles: "bf16 x 3", dropped terms <= (2^-14 + 2^-15) |a b|) or of the six terms down to m_a m_b
// ("bf16 x 6", dropped terms <= 2^-21 |a b|: fp32-grade).  The lane's eight k values of a 32-wide slab are its two 16-byte fragments.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int SPLIT>
struct SplitFrag { bf16x8 h, m, l; };
template <int SPLIT>
CCSD_DEV SplitFrag<SPLIT> split_frag(const float4& a, const float4& b) {
    const float x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    unsigned hw[4], mw[4];
    float r[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned u0 = __float_as_uint(x[2 * i]), u1 = __float_as_uint(x[2 * i + 1]);
        hw[i] = __builtin_amdgcn_perm(u1, u0, 0x07060302u);
        r[2 * i] = x[2 * i] - __uint_as_float(u0 & 0xffff0000u);
        r[2 * i + 1] = x[2 * i + 1] - __uint_as_float(u1 & 0xffff0000u);
    }
    SplitFrag<SPLIT> f;
    __builtin_memcpy(&f.h, hw, 16);
    if (SPLIT == 3) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned u0 = __float_as_uint(r[2 * i]), u1 = __float_as_uint(r[2 * i + 1]);
            mw[i] = __builtin_amdgcn_perm(u1, u0, 0x07060302u);
            r[2 * i] = r[2 * i] - __uint_as_float(u0 & 0xffff0000u);
            r[2 * i + 1] = r[2 * i + 1] - __uint_as_float(u1 & 0xffff0000u);
        }
        __builtin_memcpy(&f.m, mw, 16);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) f.l[i] = (__bf16)r[i];
    return f;
}
typedef float f32x4_ __attribute__((ext_vector_type(4)));
template <int SPLIT>
CCSD_DEV f32x4_ split_mma(const SplitFrag<SPLIT>& a, const SplitFrag<SPLIT>& b, f32x4_ c) {      // (small terms first)
    if (SPLIT == 3) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.m, b.m, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.l, b.h, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.h, b.l, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.m, b.h, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.h, b.m, c, 0, 0, 0);
    } else {
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.l, b.h, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.h, b.l, c, 0, 0, 0);
    }
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.h, b.h, c, 0, 0, 0);
}

template <int EC, int KC, int SPLIT = 0>        // SPLIT: 0 exact fp32 (the product path); 2 / 3: the split-precision experiment above
__global__ __launch_bounds__(256, 2) void k_gemm_h_full(const float* __restrict__ rank2, float* __restrict__ H, int zero_diag) {
    static_assert(EC > 96 && EC <= 192 && (KC & 3) == 0, "one 192-row block, 16-byte rows");
    constexpr int E = EC, K = KC, NS = (K + H_BK - 1) / H_BK, SLAB = 192 * H_LD;
    __shared__ __align__(16) float Fs[2 * SLAB];
    const int b = blockIdx.x, tid = threadIdx.x, wave = wave_index(), lane = tid & 63, l15 = lane & 15, kq = lane >> 4;
    const float* Fb = rank2 + (size_t)b * E * K;
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    // thread -> (row r0 + 32 u, 4-float column group c4) of the 192 x 32 slab, u < 6
    const int r0 = tid >> 3, c4 = (tid & 7) * 4;
    float4 rg[6];
    auto ldg = [&](int s) {
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            const int row = r0 + 32 * u, k = s * H_BK + c4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < E && k < K) v = *reinterpret_cast<const float4*>(Fb + (size_t)row * K + k);       // (K a multiple of 4: whole groups)
            rg[u] = v;
        }
    };
    auto sts = [&](int buf) {
#pragma unroll
        for (int u = 0; u < 6; ++u) *reinterpret_cast<float4*>(Fs + buf * SLAB + (r0 + 32 * u) * H_LD + c4) = rg[u];
    };
    f32x4 acc[21];
#pragma unroll
    for (int i = 0; i < 21; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool tri = wave < 2;
    const int tb = tri ? 6 * wave : 3 * (wave - 2);        // first row tile of the wave's triangle / rectangle
    // the sub-tile a triangle wave gives up, taken by the rectangle wave that holds its fragments: (0, 1) -> wave 2, (6, 7) -> wave 3
    ldg(0);
    sts(0);
    __syncthreads();
    for (int s = 0; s < NS; ++s) {
        const float* S = Fs + (s & 1) * SLAB;
        if (s + 1 < NS) ldg(s + 1);
        if constexpr (SPLIT != 0) {
            static_assert(H_BK == 32, "one 32-wide bf16 MFMA per slab and term");
            auto sfrag = [&](int rt) {
                const float* q = S + (16 * rt + l15) * H_LD + 4 * kq;
                return split_frag<SPLIT>(*reinterpret_cast<const float4*>(q), *reinterpret_cast<const float4*>(q + 16));
            };
            if (tri) {
                SplitFrag<SPLIT> f[6];
#pragma unroll
                for (int i = 0; i < 6; ++i) f[i] = sfrag(tb + i);
                int a = 0;
#pragma unroll
                for (int i = 0; i < 6; ++i)
#pragma unroll
                    for (int j = i; j < 6; ++j) {
                        if (!(i == 0 && j == 1)) acc[a] = split_mma<SPLIT>(f[i], f[j], acc[a]);
                        ++a;
                    }
            } else {
                SplitFrag<SPLIT> fa[3], fb[6];
#pragma unroll
                for (int i = 0; i < 3; ++i) fa[i] = sfrag(tb + i);
#pragma unroll
                for (int j = 0; j < 6; ++j) fb[j] = sfrag(6 + j);
                int a = 0;
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 6; ++j) { acc[a] = split_mma<SPLIT>(fa[i], fb[j], acc[a]); ++a; }
                if (wave == 2) acc[18] = split_mma<SPLIT>(fa[0], fa[1], acc[18]);
                else acc[18] = split_mma<SPLIT>(fb[0], fb[1], acc[18]);
            }
        } else {
#pragma unroll
        for (int t = 0; t < H_BK / 16; ++t) {
            auto frag = [&](int rt) { return *reinterpret_cast<const float4*>(S + (16 * rt + l15) * H_LD + 16 * t + 4 * kq); };
            if (tri) {
                float4 f[6];
#pragma unroll
                for (int i = 0; i < 6; ++i) f[i] = frag(tb + i);
                int a = 0;
#pragma unroll
                for (int i = 0; i < 6; ++i)
#pragma unroll
                    for (int j = i; j < 6; ++j) {
                        if (!(i == 0 && j == 1)) {                         // (given to the rectangle wave)
                            const float av[4] = {f[i].x, f[i].y, f[i].z, f[i].w}, bv[4] = {f[j].x, f[j].y, f[j].z, f[j].w};
#pragma unroll
                            for (int q = 0; q < 4; ++q) acc[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q], bv[q], acc[a], 0, 0, 0);
                        }
                        ++a;
                    }
            } else {
                float4 fa[3], fb[6];
#pragma unroll
                for (int i = 0; i < 3; ++i) fa[i] = frag(tb + i);
#pragma unroll
                for (int j = 0; j < 6; ++j) fb[j] = frag(6 + j);
                int a = 0;
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 6; ++j) {
                        const float av[4] = {fa[i].x, fa[i].y, fa[i].z, fa[i].w}, bv[4] = {fb[j].x, fb[j].y, fb[j].z, fb[j].w};
#pragma unroll
                        for (int q = 0; q < 4; ++q) acc[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q], bv[q], acc[a], 0, 0, 0);
                        ++a;
                    }
                // the extra sub-tile: wave 2 -> (0, 1) = fa[0] x fa[1]; wave 3 -> (6, 7) = fb[0] x fb[1]
                const float4 xa = wave == 2 ? fa[0] : fb[0], xb = wave == 2 ? fa[1] : fb[1];
                const float av[4] = {xa.x, xa.y, xa.z, xa.w}, bv[4] = {xb.x, xb.y, xb.z, xb.w};
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[18] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q], bv[q], acc[18], 0, 0, 0);
            }
        }
        }
        if (s + 1 < NS) sts((s + 1) & 1);
        __syncthreads();
    }
    constexpr int ldH = (E + 3) & ~3;                    // == h_ld(E)
    float* Hb = H + (size_t)b * E * ldH;
    auto store = [&](int ri, int cj, const f32x4& v) {
        const int n = 16 * cj + l15;
        if (n >= E) return;
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) {
            const int m = 16 * ri + 4 * kq + s2;
            if (m < E) {
                const float hv = (zero_diag && m == n) ? 0.f : v[s2];
                Hb[(size_t)m * ldH + n] = hv;
                Hb[(size_t)n * ldH + m] = hv;
            }
        }
    };
    if (tri) {
        int a = 0;
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = i; j < 6; ++j) {
                if (!(i == 0 && j == 1)) store(tb + i, tb + j, acc[a]);
                ++a;
            }
    } else {
        int a = 0;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 6; ++j) { store(tb + i, 6 + j, acc[a]); ++a; }
        if (wave == 2) store(0, 1, acc[18]); else store(6, 7, acc[18]);
    }
}
#endif

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
                                                const unsigned long long* __restrict__ cells, int kchunk) {
    __shared__ float As[T_BK * T_LD];
    __shared__ float Bs[T_BK * T_LD];
    __shared__ float s_mv[CCSD_MAXLIN * CCSD_HWBLK];   // mlp_value as zero-padded 8x8 blocks (LDS broadcast reads)
    const int m0 = blockIdx.y * T_BM, n0 = blockIdx.x * T_BN;
    // split K (small batches: too few 64-row tiles to fill the chip): slice blockIdx.z sums k in [z kchunk, (z + 1) kchunk) into its own
    // copy of P (P + z rows wc); k_sum_splits adds the slices in a fixed order.  kchunk >= K with gridDim.z == 1: the whole sum, in place.
    const int kbeg = (int)blockIdx.z * kchunk, kend = kbeg + kchunk < K ? kbeg + kchunk : K;
    P += (size_t)blockIdx.z * rows * wc;
    const float* Wc = W + wcat_off;
    TileAcc acc;
    tile_zero(acc);
    if (layer == 1) { stage_mlp_blocks(mval, W, s_mv, (int)threadIdx.x, (int)blockDim.x); __syncthreads(); }
    for (int k0 = kbeg; k0 < kend; k0 += T_BK) {
        for (int idx = threadIdx.x; idx < T_BM * T_BK; idx += blockDim.x) {
            const int r = idx / T_BK, kk = idx % T_BK, k = k0 + kk, row = m0 + r;
            float v = 0.f;
            if (row < rows && k < kend) {
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
            Bs[kk * T_LD + c] = (k < kend && col < wc) ? Wc[(size_t)k * wc + col] : 0.f;
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
// k_sum_splits: P[i] = parts[0][i] + parts[1][i] + ... (the K slices of k_gemm_p, always in this order: reproducible)
__global__ void k_sum_splits(const float* __restrict__ parts, float* __restrict__ P, long long n, int S) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float acc = parts[i];
        for (int z = 1; z < S; ++z) acc += parts[(size_t)z * n + i];
        P[i] = acc;
    }
}

// ---------------------------------------------------------------------------------------------
// k_gemm_p0: layer-0 hodge projections  P_0[r][c] = sum_k rank2[r][k] * Wcat_0[k][c]  (DenseHCNConv out = rank2 @ W,
// hodge_layers.py:185) over the flattened rows r = b*E + e, for narrow outputs (wc <= 64: 16 columns for every shipped
// network).  One workgroup = 64 rows x all pad16(wc) columns: both operands are row copies (rank2 rows, rows of the
// transposed packed weights Wcat^T[col][Kp]) read back as ds_read_b128 permuted-k fragments; wave w owns rows
// 16w..16w+15 and every 16-column tile, so no MFMA is spent on the 64-column padding of the general tile engine.
// ---------------------------------------------------------------------------------------------
// Fused Langevin-corrector work of the tiled path (h_L == 1, K a multiple of 4; ccsd_api.h: tiled_fuse_ok).  The projection kernel is
// the one pass of a half-step that streams rank2 once, row by row, in 16-byte pieces = the flat Philox groups of the corrector's
// draw (NoiseArgs::flat_r), so the corrector's element-wise work rides on it:
//   mode 1 (norms pass):      zrow[row] = sum_k (z fl fr)^2 of the row  -- the noise norm (gen_noise_rank2 + torch.norm,
//                              cc_utils.py:613-615, solver.py:793-797); replaces k_noise_norm;
//   mode 2 (predictor pass):  F1 = fma(c2, z fl fr, fma(c1, net, F)) -- the corrector apply, same expression as k_langevin_apply
//                              (solver.py:797-801) -- is what goes into LDS (P_0 = F1 Wcat_0) AND out to `f1` for k_gemm_h / k_hf_score;
//                              replaces k_langevin_apply's pass over rank2 and the projection's own read of the corrected state.
// `f1` may alias `net` (every element is read, then written, by the same thread).
// Plans whose ScoreNetworkF is affine with cnum = 1 (k_ew1's: net = fl fr (alpha F + gamma), no Hodge Laplacian term -- the N = 38
// substitute for zinc250k_CC) have NO other rank-2 kernel in a half-step, so there the whole rank-2 side rides on the projection pass:
//   mode 3 (norms pass):      per row { sum net^2, sum (z fl fr)^2 } -> zrow[row][2]  (+ the raw score to `net_out` when a separate
//                              ccsd_corrector_apply will want it); the projection is of the state as given;
//   mode 4 (predictor pass):  F1 = corrector apply with the raw score recomputed in place (as k_ew1 does), P_0 = F1 Wcat_0,
//                              new state = fma(pc, z' fl fr, fma(pa, F1, pb net(F1))) -> `out` (+ the mean -> `mean`), z' = the
//                              predictor's flat-keyed draw.  Same expressions, in the same order, as k_ew1 / k_langevin_apply.
// One read of rank2 per norms pass (k_ew1 + k_gemm_p0 read it twice) and one read + one write per predictor pass (k_ew1: one read, two
// writes; k_gemm_p0: one more read): 18 GB instead of 36 GB per PC step at E = 703, K = 8436, B = 256.
struct P0Fuse {
    int mode;
    const float* net; float* f1; float* zrow;
    unsigned long long seed; long long b_off; unsigned int draw;
    MaskTab mt; int E;
    CorrFuse cf;
    // modes 3 / 4
    float alpha, gamma, pa, pb, pc;
    unsigned int draw_pred;
    float* out; float* mean; float* net_out;
};
// element-wise form of the two modes (host emulation, whose projections run through the general k_gemm_p): one thread per flat group
__global__ void k_p0_fuse_ew(const float* __restrict__ rank2, P0Fuse pf, int rows, int K) {
    float c1 = 0.f, c2 = 0.f;
    if (pf.mode == 2 || pf.mode == 4) corr_coef(pf.cf, 2, &c1, &c2);
    const int E = pf.E;
    for (long long row = (long long)blockIdx.x * blockDim.x + threadIdx.x; row < rows; row += (long long)gridDim.x * blockDim.x) {
        const int b = (int)(row / E), e = (int)(row - (long long)b * E);
        float zs = 0.f, ns = 0.f;
        for (int k = 0; k < K; k += 4) {
            float z[4], zp[4] = {0.f, 0.f, 0.f, 0.f}, m[4];
            const unsigned g = (unsigned)(((long long)e * K + k) >> 2);
            philox_normal4(pf.seed, pf.draw, pf.b_off + b, g, z);
            if (pf.mode == 4) philox_normal4(pf.seed, pf.draw_pred, pf.b_off + b, g, zp);
            group_masks(pf.mt, b, E, K, e, k, m);
            for (int j = 0; j < 4; ++j) {
                const float zz = z[j] * m[j];
                const size_t gi = (size_t)row * K + k + j;
                if (pf.mode == 1) zs = fmaf(zz, zz, zs);
                else if (pf.mode == 2) pf.f1[gi] = fmaf(c2, zz, fmaf(c1, pf.net[gi], rank2[gi]));
                else if (pf.mode == 3) {
                    const float net = m[j] * fmaf(pf.alpha, rank2[gi], pf.gamma);
                    ns = fmaf(net, net, ns); zs = fmaf(zz, zz, zs);
                    if (pf.net_out) pf.net_out[gi] = net;
                } else {
                    float f = rank2[gi];
                    float net = m[j] * fmaf(pf.alpha, f, pf.gamma);
                    f = fmaf(c2, zz, fmaf(c1, net, f));
                    pf.f1[gi] = f;
                    net = m[j] * fmaf(pf.alpha, f, pf.gamma);
                    const float mean = fmaf(pf.pa, f, pf.pb * net);
                    if (pf.mean) pf.mean[gi] = mean;
                    pf.out[gi] = fmaf(pf.pc, zp[j] * m[j], mean);
                }
            }
        }
        if (pf.mode == 1) pf.zrow[row] = zs;
        if (pf.mode == 3) { pf.zrow[2 * row] = ns; pf.zrow[2 * row + 1] = zs; }
    }
}
#ifndef CCSD_EMU
#ifndef CCSD_EMU
// ---------------------------------------------------------------------------------------------
// k_hp_full<EC, KC, MODE>: ONE pass over a complex's rank2 block per half-step on the tiled path (community_small geometry, one hodge
// layer, wc <= 16: tiled_fuse_ok + the geometry): what k_gemm_p0<1, KC, MODE> and k_gemm_h_full<EC, KC> do in two --
//   * the Langevin corrector's element-wise work where the block streams through registers (P0Fuse, as in k_gemm_p0):
//     MODE 1 the noise norm of the corrector's draw per row (-> pf.zrow), MODE 2 the corrector apply (corrected rank2 -> pf.f1,
//     in place over the raw scores it consumes); everything below sees the corrected block;
//   * P_0 = F Wcat_0 (12 sub-tiles of 16 rows x 16 columns; WT = Wcat_0^T [16][Kp], a [16][32] slab of it staged beside F's);
//   * H = (F F^T) * hodge_mask (78 upper-triangle sub-tiles, mirrored).
// Per sub-tile the k order and operand slots of the kernels it replaces: P_0 and H are bit-identical to theirs.
// 512 threads = 8 waves, two workgroups per CU (128 VGPRs).  Row tiles in four groups G0..G3 of three: waves 0-5 take one off-diagonal
// 3 x 3 block (Ga, Gb) each (9 sub-tiles, 6 fragments), four of them also the three P_0 sub-tiles of a group they hold; waves 6 / 7 the
// diagonal triangles of (G0, G1) / (G2, G3) (12 sub-tiles): 9 x 2 + 12 x 6 = 90.  LDS (dynamic): two F slabs [192][40] + two W slabs [16][40] = 66.6 KB.
// grid: B workgroups.
// ---------------------------------------------------------------------------------------------
template <int EC, int KC, int MODE>
__global__ __launch_bounds__(512, 2) void k_hp_full(const float* __restrict__ rank2, const float* __restrict__ WT, float* __restrict__ H,
                                                    float* __restrict__ P, int wc, int zero_diag, P0Fuse pf) {
    static_assert(EC > 144 && EC <= 192 && (KC & 3) == 0 && (MODE == 1 || MODE == 2), "one 192-row block, 16-byte rows");
    constexpr int E = EC, K = KC, Kp = (KC + 31) & ~31, NS = Kp / H_BK, SLAB = 192 * H_LD, WSLAB = 16 * H_LD;
    CCSD_DYN_SMEM(sm);
    float* const Fs = sm;                         // [2][SLAB]
    float* const Ws = sm + 2 * SLAB;              // [2][WSLAB]
    const int b = blockIdx.x, tid = threadIdx.x, wave = wave_index(), lane = tid & 63, l15 = lane & 15, kq = lane >> 4;
    const float* Fb = rank2 + (size_t)b * E * K;
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    // thread -> (row r0 + 64 u, 4-float column group c4) of the 192 x 32 slab, u < 3; threads 0..127 also one group of the W slab
    const int r0 = tid >> 3, c4 = (tid & 7) * 4;
    float c1 = 0.f, c2 = 0.f, zacc[3] = {0.f, 0.f, 0.f};
    if (MODE == 2) corr_coef(pf.cf, 2, &c1, &c2);
    float4 rg[3], rn[3], rw;
    auto ldg = [&](int s) {
        const int k = s * H_BK + c4;
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int row = r0 + 64 * u;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f), n = v;
            if (row < E && k < K) {
                v = *reinterpret_cast<const float4*>(Fb + (size_t)row * K + k);       // (K a multiple of 4: whole groups)
                if (MODE == 2) n = *reinterpret_cast<const float4*>(pf.net + ((size_t)b * E + row) * K + k);
            }
            rg[u] = v; rn[u] = n;
        }
        if (tid < 128) rw = *reinterpret_cast<const float4*>(WT + (size_t)r0 * Kp + k);   // (rows 0..15 of Wcat_0^T, zero-padded to Kp)
    };
    // the corrector's work on the slab in registers (same expressions as k_gemm_p0's), then the slab goes to LDS
    auto sts = [&](int s, int buf) {
        const int k = s * H_BK + c4;
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int row = r0 + 64 * u;
            if (row < E && k < K) {
                float z[4], m[4];
                philox_normal4(pf.seed, pf.draw, pf.b_off + b, (unsigned)((row * K + k) >> 2), z);
                group_masks(pf.mt, b, E, K, row, k, m);
                if (MODE == 1) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) { const float zz = z[j] * m[j]; zacc[u] = fmaf(zz, zz, zacc[u]); }
                } else {
                    float4& v = rg[u];
                    const float4 n = rn[u];
                    v.x = fmaf(c2, z[0] * m[0], fmaf(c1, n.x, v.x)); v.y = fmaf(c2, z[1] * m[1], fmaf(c1, n.y, v.y));
                    v.z = fmaf(c2, z[2] * m[2], fmaf(c1, n.z, v.z)); v.w = fmaf(c2, z[3] * m[3], fmaf(c1, n.w, v.w));
                    *reinterpret_cast<float4*>(pf.f1 + ((size_t)b * E + row) * K + k) = v;
                }
            }
            *reinterpret_cast<float4*>(Fs + buf * SLAB + row * H_LD + c4) = rg[u];
        }
        if (tid < 128) *reinterpret_cast<float4*>(Ws + buf * WSLAB + r0 * H_LD + c4) = rw;
    };
    f32x4 acc[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    // waves 0..5: off-diagonal block (Ga, Gb) = (0,1) (0,2) (0,3) (1,2) (1,3) (2,3); their P_0 row tiles: one of Ga's, one of Gb's
    const bool offd = wave < 6;
    const int ga = wave < 3 ? 0 : wave < 5 ? 1 : 2, gb = wave < 3 ? wave + 1 : wave < 5 ? wave - 1 : 3;
    // P_0 (12 sub-tiles): a whole group per wave -- G0 with wave 0, G1 with wave 3, G2 with wave 5 (their Ga), G3 with wave 4 (its Gb)
    const bool pA = wave == 0 || wave == 3 || wave == 5, pB = wave == 4;
    const int d0 = wave == 6 ? 0 : 6;                      // waves 6 / 7: the triangles of row tiles d0 .. d0 + 2 and d0 + 3 .. d0 + 5
    ldg(0);
    sts(0, 0);
    __syncthreads();
    for (int s = 0; s < NS; ++s) {
        const float* S = Fs + (s & 1) * SLAB;
        const float* W = Ws + (s & 1) * WSLAB;
        if (s + 1 < NS) ldg(s + 1);
#pragma unroll
        for (int t = 0; t < H_BK / 16; ++t) {
            auto frag = [&](int rt) { return *reinterpret_cast<const float4*>(S + (16 * rt + l15) * H_LD + 16 * t + 4 * kq); };
            if (offd) {
                float4 fa[3], fb[3];
#pragma unroll
                for (int i = 0; i < 3; ++i) { fa[i] = frag(3 * ga + i); fb[i] = frag(3 * gb + i); }
                const float4 wq = *reinterpret_cast<const float4*>(W + l15 * H_LD + 16 * t + 4 * kq);
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        const float av[4] = {fa[i].x, fa[i].y, fa[i].z, fa[i].w}, bv[4] = {fb[j].x, fb[j].y, fb[j].z, fb[j].w};
#pragma unroll
                        for (int q = 0; q < 4; ++q) acc[3 * i + j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q], bv[q], acc[3 * i + j], 0, 0, 0);
                    }
                if (pA || pB) {
                    const float wv[4] = {wq.x, wq.y, wq.z, wq.w};
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
                        const float4 x = pA ? fa[i] : fb[i];
                        const float av[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
                        for (int q = 0; q < 4; ++q) acc[9 + i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q], wv[q], acc[9 + i], 0, 0, 0);
                    }
                }
            } else {
                float4 f[6];
#pragma unroll
                for (int i = 0; i < 6; ++i) f[i] = frag(d0 + i);
                int a = 0;
#pragma unroll
                for (int g = 0; g < 2; ++g)
#pragma unroll
                    for (int i = 0; i < 3; ++i)
#pragma unroll
                        for (int j = i; j < 3; ++j) {
                            const float4 &fi = f[3 * g + i], &fj = f[3 * g + j];
                            const float av[4] = {fi.x, fi.y, fi.z, fi.w}, bv[4] = {fj.x, fj.y, fj.z, fj.w};
#pragma unroll
                            for (int q = 0; q < 4; ++q) acc[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q], bv[q], acc[a], 0, 0, 0);
                            ++a;
                        }
            }
        }
        if (s + 1 < NS) sts(s + 1, (s + 1) & 1);
        __syncthreads();
    }
    if (MODE == 1) {
        // a row's eight column groups sit in eight consecutive lanes: fixed butterfly, lane 0 of the group stores (as k_gemm_p0)
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            float v = zacc[u];
            v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64);
            const int row = r0 + 64 * u;
            if ((tid & 7) == 0 && row < E) pf.zrow[(size_t)b * E + row] = v;
        }
    }
    constexpr int ldH = (E + 3) & ~3;                    // == h_ld(E)
    float* Hb = H + (size_t)b * E * ldH;
    auto store = [&](int ri, int cj, const f32x4& v) {
        const int n = 16 * cj + l15;
        if (n >= E) return;
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) {
            const int m = 16 * ri + 4 * kq + s2;
            if (m < E) {
                const float hv = (zero_diag && m == n) ? 0.f : v[s2];
                Hb[(size_t)m * ldH + n] = hv;
                Hb[(size_t)n * ldH + m] = hv;
            }
        }
    };
    auto store_p = [&](int ri, const f32x4& v) {
        if (l15 >= wc) return;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = 16 * ri + 4 * kq + r;
            if (m < E) P[((size_t)b * E + m) * wc + l15] = v[r];
        }
    };
    if (offd) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) store(3 * ga + i, 3 * gb + j, acc[3 * i + j]);
        if (pA || pB) {
#pragma unroll
            for (int i = 0; i < 3; ++i) store_p(3 * (pA ? ga : gb) + i, acc[9 + i]);
        }
    } else {
        int a = 0;
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = i; j < 3; ++j) { store(d0 + 3 * g + i, d0 + 3 * g + j, acc[a]); ++a; }
    }
}
#endif

template <int NT, int KC = 0, int MODE = 0>          // KC: K as a compile-time constant (0: the argument); Kp follows; MODE: P0Fuse::mode
__global__ __launch_bounds__(256) void k_gemm_p0(const float* __restrict__ rank2, const float* __restrict__ WT, float* __restrict__ P,
                                                 int rows, int K_, int Kp_, int wc, P0Fuse pf) {
    const int K = KC ? KC : K_, Kp = KC ? ((KC + 31) & ~31) : Kp_;
    __shared__ __align__(16) float As[T_BM * H_LD];
    __shared__ __align__(16) float Bs[16 * NT * H_LD];
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const int tid = threadIdx.x, wave = wave_index(), lane = tid & 63, l15 = lane & 15, kq = lane >> 4;
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
    // fused corrector work (MODE != 0; K % 4 == 0): the thread's two rows of every slab are fixed -> (sample, edge) once
    float c1 = 0.f, c2 = 0.f, zacc[2] = {0.f, 0.f}, nacc[2] = {0.f, 0.f};
    int fb[2] = {0, 0}, fe[2] = {0, 0};
    if (MODE == 2 || MODE == 4) corr_coef(pf.cf, 2, &c1, &c2);
    if (MODE) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int row = m0 + r0 + 32 * u, rc = row < rows ? row : rows - 1;
            fb[u] = rc / pf.E; fe[u] = rc - fb[u] * pf.E;
        }
    }
    auto ldn = [&](int row, int k) -> float4 {         // raw score beside the state (MODE 2)
        const int rc = row < rows ? row : rows - 1, kc = k + 3 < K ? k : K - 4;
        return *reinterpret_cast<const float4*>(pf.net + (size_t)rc * K + kc);
    };
    auto fuse = [&](float4& v, const float4& n, int u, int k) {
        const int row = m0 + r0 + 32 * u;
        if (row >= rows || k >= K) return;               // (clamped rows of the last workgroup, zero padding of the last slab)
        float z[4], m[4];
        philox_normal4(pf.seed, pf.draw, pf.b_off + fb[u], (unsigned)((fe[u] * K + k) >> 2), z);
        group_masks(pf.mt, fb[u], pf.E, K, fe[u], k, m);
        if (MODE == 1) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { const float zz = z[j] * m[j]; zacc[u] = fmaf(zz, zz, zacc[u]); }
        } else if (MODE == 2) {
            v.x = fmaf(c2, z[0] * m[0], fmaf(c1, n.x, v.x)); v.y = fmaf(c2, z[1] * m[1], fmaf(c1, n.y, v.y));
            v.z = fmaf(c2, z[2] * m[2], fmaf(c1, n.z, v.z)); v.w = fmaf(c2, z[3] * m[3], fmaf(c1, n.w, v.w));
            *reinterpret_cast<float4*>(pf.f1 + (size_t)row * K + k) = v;
        } else if (MODE == 3) {
            const float f[4] = {v.x, v.y, v.z, v.w};
            float net[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                net[j] = m[j] * fmaf(pf.alpha, f[j], pf.gamma);                // fnet_element<AFFINE>, cnum = 1 (k_ew1)
                const float zz = z[j] * m[j];
                nacc[u] = fmaf(net[j], net[j], nacc[u]);
                zacc[u] = fmaf(zz, zz, zacc[u]);
            }
            if (pf.net_out) *reinterpret_cast<float4*>(pf.net_out + (size_t)row * K + k) = make_float4(net[0], net[1], net[2], net[3]);
        } else {
            float zp[4];
            philox_normal4(pf.seed, pf.draw_pred, pf.b_off + fb[u], (unsigned)((fe[u] * K + k) >> 2), zp);
            float f[4] = {v.x, v.y, v.z, v.w}, o[4], mu[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float net = m[j] * fmaf(pf.alpha, f[j], pf.gamma);
                f[j] = fmaf(c2, z[j] * m[j], fmaf(c1, net, f[j]));             // k_langevin_apply
                net = m[j] * fmaf(pf.alpha, f[j], pf.gamma);
                mu[j] = fmaf(pf.pa, f[j], pf.pb * net);                        // v_mean = pa*v + pb*net
                o[j] = fmaf(pf.pc, zp[j] * m[j], mu[j]);
            }
            v = make_float4(f[0], f[1], f[2], f[3]);                            // the corrected state: what the projection is taken of
            *reinterpret_cast<float4*>(pf.out + (size_t)row * K + k) = make_float4(o[0], o[1], o[2], o[3]);
            if (pf.mean) *reinterpret_cast<float4*>(pf.mean + (size_t)row * K + k) = make_float4(mu[0], mu[1], mu[2], mu[3]);
        }
    };
    f32x4 acc[NT];
#pragma unroll
    for (int c = 0; c < NT; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // Register double buffering, two slabs ahead in the apply mode (its element-wise work -- Philox, masks, the f1 store -- sits
    // between a slab's arrival and its use: with one slab in flight the memory pipe idles meanwhile), one slab ahead otherwise.
    constexpr int NS = (MODE == 2 || MODE == 4) ? 2 : 1;
    float4 ra[NS][2], rn[NS][2], rb[NS][(NT + 1) / 2];
    auto load_slab = [&](auto S_, int k0) {
        constexpr int S = decltype(S_)::value;
        ra[S][0] = lda(m0 + r0, k0 + c4); ra[S][1] = lda(m0 + r0 + 32, k0 + c4);
        if (MODE == 2) { rn[S][0] = ldn(m0 + r0, k0 + c4); rn[S][1] = ldn(m0 + r0 + 32, k0 + c4); }
#pragma unroll
        for (int u = 0; u < (NT + 1) / 2; ++u) {           // 16*NT weight rows x 8 float4: tid + 256u < 128*NT
            const int idx = tid + 256 * u, wr = idx >> 3;
            rb[S][u] = wr < 16 * NT ? *reinterpret_cast<const float4*>(WT + (size_t)wr * Kp + k0 + (idx & 7) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto slab = [&](auto S_, int k0) {                     // consume stage S (slab k0), refill it with slab k0 + NS * H_BK
        constexpr int S = decltype(S_)::value;
        if (MODE) {                                                                  // (before the barrier: registers only)
            fuse(ra[S][0], MODE == 2 ? rn[S][0] : ra[S][0], 0, k0 + c4);
            fuse(ra[S][1], MODE == 2 ? rn[S][1] : ra[S][1], 1, k0 + c4);
        }
        __syncthreads();
        *reinterpret_cast<float4*>(As + r0 * H_LD + c4) = ra[S][0];
        *reinterpret_cast<float4*>(As + (r0 + 32) * H_LD + c4) = ra[S][1];
#pragma unroll
        for (int u = 0; u < (NT + 1) / 2; ++u) {
            const int idx = tid + 256 * u, wr = idx >> 3;
            if (wr < 16 * NT) *reinterpret_cast<float4*>(Bs + wr * H_LD + (idx & 7) * 4) = rb[S][u];
        }
        __syncthreads();
        if (k0 + NS * H_BK < Kp) load_slab(S_, k0 + NS * H_BK);
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
    };
    typedef std::integral_constant<int, 0> S0;
    typedef std::integral_constant<int, NS - 1> S1;
    load_slab(S0{}, 0);
    if (NS == 2 && H_BK < Kp) load_slab(S1{}, H_BK);
    for (int k0 = 0; k0 < Kp; k0 += NS * H_BK) {
        slab(S0{}, k0);
        if (NS == 2 && k0 + H_BK < Kp) slab(S1{}, k0 + H_BK);
    }
    if (MODE == 1 || MODE == 3) {
        // the row's eight column groups sit in eight consecutive lanes: fixed butterfly, lane 0 of the group stores
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            float v = zacc[u], w2 = nacc[u];
            v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64);
            if (MODE == 3) { w2 += __shfl_xor(w2, 1, 64); w2 += __shfl_xor(w2, 2, 64); w2 += __shfl_xor(w2, 4, 64); }
            const int row = m0 + r0 + 32 * u;
            if ((tid & 7) == 0 && row < rows) {
                if (MODE == 1) pf.zrow[row] = v;
                else { pf.zrow[2 * (size_t)row] = w2; pf.zrow[2 * (size_t)row + 1] = v; }
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
// k_gemm_pow: next Hodge power  Hn = Hp . H1  per complex (pow_tensor_cc with cnum > 2, cc_utils.py:972-977: x_ = bmm(H, x_)
// repeatedly, i.e. channel j = H^j F).  E x E x E per complex: one 16x16 output tile per wave.  grid (ceil(nt^2 / waves), 1, B)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_gemm_pow(const float* __restrict__ Hp, const float* __restrict__ H1, float* __restrict__ Hn, int E) {
    const int b = blockIdx.z, nt = (E + 15) >> 4;
#ifdef CCSD_EMU
    const int wave = 0, nw = 1;
#else
    const int wave = wave_index(), nw = blockDim.x >> 6;
#endif
    const int ldH = h_ld(E);
    const float* A = Hp + (size_t)b * E * ldH;
    const float* Bm = H1 + (size_t)b * E * ldH;
    float* C = Hn + (size_t)b * E * ldH;
    for (int tile = blockIdx.x * nw + wave; tile < nt * nt; tile += gridDim.x * nw) {
        const int ti = tile / nt, tj = tile - ti * nt;
        wave_tile<8>(16 * ti, 16 * tj, (E + 3) >> 2,
                     [&](int r, int k) { const float v = A[(size_t)(r < E ? r : E - 1) * ldH + (k < E ? k : E - 1)]; return (r < E && k < E) ? v : 0.f; },
                     [&](int k, int c) { const float v = Bm[(size_t)(k < E ? k : E - 1) * ldH + (c < E ? c : E - 1)]; return (k < E && c < E) ? v : 0.f; },
                     [&](int r, int c, float acc) { if (r < E && c < E) C[(size_t)r * ldH + c] = acc; });
    }
}

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
// k_hodge_value: the rank-2 features a HodgeAdjAttentionLayer hands to the next one, materialised (general hodge stack: more than
// two layers with a non-affine mlp_value):
//   Rout[b][e][k] = fl[e] * mlp_value(cat_c V_c[e][k]) * fr[k],   V_c = H_c . Rin[b]       (hodge_attention.py:98, 322-323; cc_utils.py:594-615)
// H: the layer's dense hodge adjacency [B][hstride] as [cin][E][E] (k_xa's dump), or nullptr for layer 0, whose hodge adjacency is the
// diagonal acoef[b][c][e] (k_edgecoef): V_c = a_c[e] Rin[e][k] -- the expression of k_gemm_p's layer-1 loader, bit for bit.
// grid (ceil(K / cw), B), cw = 64 columns (32 when E > 128); one workgroup holds the [E][cw] column slab of Rin in LDS (dynamic: E cw floats).
// A wave works on one edge row (two) at a time with its lanes on the columns: H_c[e][.] is wave-uniform, the slab reads conflict-free.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_hodge_value(const float* __restrict__ Rin, const float* __restrict__ H, int hstride,
                                                     const float* __restrict__ acoef, const float* __restrict__ W, MlpD mval, int cin,
                                                     float* __restrict__ Rout, int E, int K, int cw,
                                                     const unsigned long long* __restrict__ offbits,
                                                     const unsigned char* __restrict__ edges,
                                                     const unsigned long long* __restrict__ cells) {
    CCSD_DYN_SMEM(Rs);                                   // [E][cw]
    __shared__ float s_mv[CCSD_MAXLIN * CCSD_HWBLK];
    const int b = blockIdx.y, k0 = blockIdx.x * cw, sh = cw == 64 ? 6 : 5;
    const float* Rb = Rin + (size_t)b * E * K;
    stage_mlp_blocks(mval, W, s_mv, (int)threadIdx.x, (int)blockDim.x);
    for (int idx = threadIdx.x; idx < E * cw; idx += blockDim.x) {
        const int e = idx >> sh, kk = idx & (cw - 1);
        Rs[idx] = k0 + kk < K ? Rb[(size_t)e * K + k0 + kk] : 0.f;
    }
    __syncthreads();
    const unsigned long long off = offbits[b];
    for (int idx = threadIdx.x; idx < E * cw; idx += blockDim.x) {
        const int e = idx >> sh, kk = idx & (cw - 1), k = k0 + kk;
        if (k >= K) continue;
        float in[CCSD_SMALLW], out[CCSD_SMALLW];
#pragma unroll
        for (int c = 0; c < CCSD_SMALLW; ++c) {
            float v = 0.f;
            if (c < cin) {
                if (H) {
                    const float* Hr = H + (size_t)b * hstride + ((size_t)c * E + e) * E;
                    for (int e2 = 0; e2 < E; ++e2) v = fmaf(Hr[e2], Rs[e2 * cw + kk], v);
                } else {
                    v = acoef[((size_t)b * cin + c) * E + e] * Rs[idx];
                }
            }
            in[c] = v;
        }
        small_mlp_lds<CCSD_SMALLW>(s_mv, mval.n, in, out);
        Rout[((size_t)b * E + e) * K + k] = edge_on(off, edges, e) * out[0] * cell_on(off, cells, k);
    }
}

// ---------------------------------------------------------------------------------------------
// k_hf_score: ScoreNetworkF.  Tile (edge rows m0.., cell columns n0..) of  H.F  on MFMA, then per
// element the channel MLP stack of ScoreNetwork_F.py:198-217 and one of three fused epilogues.
// grid xcd_grid(B, ceil(K/64) ceil(E/64))
// ---------------------------------------------------------------------------------------------
// NP: Hodge powers the instantiation can hold (1: cnum <= 2, the common case -- one accumulator, no loop; CCSD_MAXCN - 1 otherwise)
// EC, KC: E and K as compile-time constants (0: from the plan)
template <bool AFFINE, int FW, int NP, int EC = 0, int KC = 0>
__global__ __launch_bounds__(256) void k_hf_score(const PlanD* __restrict__ plan, const float* __restrict__ w,
                                                  const float* __restrict__ rank2, const float* __restrict__ H,
                                                  const unsigned long long* __restrict__ offbits,
                                                  const unsigned char* __restrict__ edges,
                                                  const unsigned long long* __restrict__ cells, RankEpi ep,
                                                  NoiseArgs na, int B, MaskTab mt) {
    __shared__ float red[64];
    const PlanD& p = *plan;
    const int E = EC ? EC : p.E, K = KC ? KC : p.K;
    const int ncb = (K + T_BN - 1) / T_BN, nrt = (E + T_BM - 1) / T_BM;
    int b, tile;
    if (!xcd_sample_tile((int)blockIdx.x, ncb * nrt, B, &b, &tile)) return;       // (see xcd_sample_tile: a complex's tiles share an XCD)
    const int rty = tile / ncb, cbx = tile - rty * ncb;
    const int m0 = rty * T_BM, n0 = cbx * T_BN;
    const float* Fb = rank2 + (size_t)b * E * K;
    // one (H^j F) tile per Hodge power j = 1 .. cnum - 1 (pow_tensor_cc, cc_utils.py:961-979): the powers H^j (B, E, E) lie
    // behind each other in the workspace (k_gemm_h, k_gemm_pow); the F slabs are re-read per power (cnum > 2 only)
    const int npow = p.f_cnum - 1;
    TileAcc accs[NP];
    // the state values the epilogue needs (specialised epilogue, below): requested beside the LAST contraction slab, so that they are
    // in registers when the MFMAs end instead of costing an HBM / L2 round trip after them
    float fpre[2][2][4];
    unsigned char frpre[2];          // flags_right bytes of the wave's two column sub-tiles
    unsigned flpre[2];               // flags_left bytes of its two row groups (four rows each)
    constexpr bool FPRE = AFFINE && NP == 1;
#pragma unroll
    for (int jp = 0; jp < NP; ++jp) tile_zero(accs[jp]);
#pragma unroll
    for (int jp = 0; jp < NP; ++jp) {
    if (jp >= npow) break;
    TileAcc& acc = accs[jp];
    const int ldH = h_ld(E);
    const float* Hb = H + ((size_t)jp * B + b) * E * ldH;
#ifdef CCSD_EMU
    static float As[T_BK * T_LD], Bs[T_BK * T_LD];
    {
        for (int k0 = 0; k0 < E; k0 += T_BK) {
            for (int idx = threadIdx.x; idx < T_BM * T_BK; idx += blockDim.x) {
                const int r = idx / T_BK, kk = idx % T_BK, k = k0 + kk, row = m0 + r;
                As[kk * T_LD + r] = (row < E && k < E) ? Hb[(size_t)row * ldH + k] : 0.f;
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
    // contraction slab: 32 wide.  (64 in the compile-time-E instances was right while the kernel sat at 102 VGPRs / four workgroups per CU --
    // half as many barriers; since the epilogue's inputs are requested beside the last slab it needs 72, and the 19 KB of LDS of the
    // 32-wide slab let more workgroups cover a tile's first-slab latency and epilogue: 541 -> 501 us; 16 wide: 520)
    constexpr int HBK = H_BK, HLD = HBK + 8;       // (HLD == 8 mod 32: conflict-free ds_read_b128 fragments)
    constexpr int NA = T_BM * HBK / 4 / 256, NB = HBK / 16, AG = HBK / 4;   // A slab: NA 16-byte groups per thread, AG groups per row
    __shared__ __align__(16) float As[T_BM * HLD];
    __shared__ __align__(16) float Bs[HBK * BLD];
    {
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        const int tid = threadIdx.x, wave = wave_index(), lane = tid & 63, l15 = lane & 15, kq = lane >> 4;
        const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
        const bool vec = (K & 3) == 0;
        // A slab 64 x HBK: thread -> (row ar + (256 / AG) u, 16-byte column group ak), u < NA -- the rows of H are padded to whole groups
        // in the workspace (h_ld); components beyond E are masked
        const int ar = tid / AG, ak = (tid % AG) * 4;
        // B slab HBK x 64: thread -> (k row bk + 16u, 4-float column group bc4), u < NB
        const int bk = tid >> 4, bc4 = (tid & 15) * 4;
        float4 ra[NA];
        float4 rb[NB];
        auto load_slab = [&](int k0) {
#pragma unroll
            for (int u = 0; u < NA; ++u) {
                const int row = m0 + ar + (256 / AG) * u, k = k0 + ak;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (row < E && k < E) {
                    v = *reinterpret_cast<const float4*>(Hb + (unsigned)(row * ldH + k));       // (uniform base + 32-bit offset: no 64-bit address arithmetic)
                    if (k + 1 >= E) v.y = 0.f;
                    if (k + 2 >= E) v.z = 0.f;
                    if (k + 3 >= E) v.w = 0.f;
                }
                ra[u] = v;
            }
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                const int k = k0 + bk + 16 * u, col = n0 + bc4;
                const float* src = Fb + (unsigned)((k < E ? k : E - 1) * K);
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
        for (int k0 = 0; k0 < E; k0 += HBK) {
            __syncthreads();
#pragma unroll
            for (int u = 0; u < NA; ++u) *reinterpret_cast<float4*>(As + (ar + (256 / AG) * u) * HLD + ak) = ra[u];
#pragma unroll
            for (int u = 0; u < NB; ++u) *reinterpret_cast<float4*>(Bs + (bk + 16 * u) * BLD + bc4) = rb[u];
            __syncthreads();
            if (k0 + HBK < E) load_slab(k0 + HBK);
            else if (FPRE) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int e0 = m0 + wm + 16 * i + 4 * kq, k = n0 + wn + 16 * j + l15;
#pragma unroll
                        for (int s2 = 0; s2 < 4; ++s2)
                            fpre[i][j][s2] = (k < K && e0 + s2 < E) ? Fb[(unsigned)((e0 + s2) * K + k)] : 0.f;
                    }
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int k = n0 + wn + 16 * q + l15, e0 = m0 + wm + 16 * q + 4 * kq;
                    frpre[q] = k < K ? mt.mfr[(size_t)b * mt.Kp + k] : (unsigned char)0;
                    flpre[q] = e0 < E ? *reinterpret_cast<const unsigned*>(mt.mfl + (size_t)b * mt.Ep + e0) : 0u;   // (e0 a multiple of 4, rows padded to Ep)
                }
            }
#pragma unroll
            for (int t = 0; t < HBK / 16; ++t) {
                const float4 a0 = *reinterpret_cast<const float4*>(As + (wm + l15) * HLD + 16 * t + 4 * kq);
                const float4 a1 = *reinterpret_cast<const float4*>(As + (wm + 16 + l15) * HLD + 16 * t + 4 * kq);
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
    }
    const unsigned char* const frb = mt.mfr + (size_t)b * mt.Kp;      // mask byte tables of this complex (k_masktab)
    const unsigned char* const flb = mt.mfl + (size_t)b * mt.Ep;
    float s_net = 0.f, s_z = 0.f;
    const bool znorm = !na.flat_r || na.zr != nullptr;   // host-supplied draws are read per element, whatever the group layout
#ifndef CCSD_EMU
    bool epi_done = false;
    if constexpr (AFFINE && NP == 1) {
        // Affine ScoreNetworkF, one Hodge power (every shipped tiled configuration): the epilogue specialised per mode and noise source
        // OUTSIDE the element loops -- no per-element mode branches, uniform base pointers + one 32-bit element offset per lane,
        // whole 4-row groups without row tests.  Same arithmetic, in the same order, as the general form below (fnet_element<true>):
        // the two agree bit for bit.  (PMC, community_small_CC: the general form issued 67 vector instructions per element.)
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        const int wave = wave_index(), lane = threadIdx.x & 63;
        const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32, l15 = lane & 15, kq = lane >> 4;
        const float fa = p.f_alpha, fgam = p.f_gamma, fbe = p.f_betas[1];
        const bool cn2 = p.f_cnum > 1;
        const float* const Fbb = Fb;
        float* const outb = ep.out + (size_t)b * E * K;
        float* const meanb = ep.mean ? ep.mean + (size_t)b * E * K : nullptr;
        const float* const zrb = na.zr ? na.zr + (size_t)b * E * K : nullptr;
        auto run = [&](auto MODE_, auto ZS_) {
            constexpr int MODE = decltype(MODE_)::value;     // 0 score, 1 norms, 2 predictor, 3 predictor + mean output
            constexpr int ZS = decltype(ZS_)::value;         // 0: no draw, 1: in-kernel Philox (4-row groups), 2: host-supplied draws
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int e0 = m0 + wm + 16 * i + 4 * kq, k = n0 + wn + 16 * j + l15;
                    if (k >= K || e0 >= E) continue;
                    const f32x4 hf = accs[0].a[i][j];
                    const float fr = (float)frpre[j];                                       // (== frb[k], flb + e0: loaded beside the last slab)
                    const unsigned fl4 = flpre[i];
                    const unsigned off = (unsigned)(e0 * K + k);
                    float z[4] = {0.f, 0.f, 0.f, 0.f};
                    if (ZS == 1) philox_normal4(na.seed, na.draw_r, na.b_off + b, (unsigned)((e0 >> 2) * K + k), z);
                    const int nr = E - e0 < 4 ? E - e0 : 4;          // rows of the group inside the block (4 except in the last one)
#pragma unroll
                    for (int s2 = 0; s2 < 4; ++s2) {
                        if (s2 >= nr) break;
                        const unsigned g = off + (unsigned)(s2 * K);
                        const float f = fpre[i][j][s2];                                   // (== Fbb[g], loaded beside the last slab)
                        if (ZS == 2) z[s2] = zrb[g];
                        const float m = (float)((fl4 >> (8 * s2)) & 0xffu) * fr;          // flags_left * flags_right, cc_utils.py:590
                        float t = fmaf(fa, f, fgam);
                        if (cn2) t = fmaf(fbe, hf[s2], t);
                        const float net = m * t;
                        if (MODE == 0) {
                            outb[g] = ep.sscale * net;
                        } else if (MODE == 1) {
                            outb[g] = net;
                            s_net = fmaf(net, net, s_net);
                            const float zz = z[s2] * m;
                            s_z = fmaf(zz, zz, s_z);
                        } else {
                            const float zz = z[s2] * m;                               // gen_noise_rank2, cc_utils.py:613-615
                            const float mean = fmaf(ep.pa, f, ep.pb * net);           // v_mean = pa*v + pb*net
                            if (MODE == 3) meanb[g] = mean;
                            outb[g] = fmaf(ep.pc, zz, mean);
                        }
                    }
                }
        };
#define HF_RUN(M_, Z_) run(std::integral_constant<int, M_>{}, std::integral_constant<int, Z_>{})
        const bool inj = na.zr != nullptr;
        if (ep.mode == MODE_SCORE) HF_RUN(0, 0);
        else if (ep.mode == MODE_NORMS) { if (!znorm) HF_RUN(1, 0); else if (inj) HF_RUN(1, 2); else HF_RUN(1, 1); }
        else if (ep.mean) { if (inj) HF_RUN(3, 2); else HF_RUN(3, 1); }
        else { if (inj) HF_RUN(2, 2); else HF_RUN(2, 1); }
#undef HF_RUN
        epi_done = true;
    }
    if (!epi_done)
#endif
    tile_foreach4n<NP>(accs, [&](int ml, int nl, const float (*hfp)[4]) {
        const int k = n0 + nl, e0 = m0 + ml;
        if (k >= K || e0 >= E) return;
        const float fr = (float)frb[k];
        const unsigned fl4 = *reinterpret_cast<const unsigned*>(flb + e0);     // e0 is a multiple of 4, rows are padded to Ep
        float z[4] = {0.f, 0.f, 0.f, 0.f};
        // (norms launch of a flat-keyed corrector draw with in-kernel Philox: its noise norm comes from k_noise_norm)
        if (ep.mode == MODE_PRED || (ep.mode == MODE_NORMS && znorm)) raw_noise_r4(na, b, e0 >> 2, k, E, K, z);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int e = e0 + s;
            if (e >= E) continue;
            const size_t gi = ((size_t)b * E + e) * K + k;
            const float f = Fb[(size_t)e * K + k];
            const float m = (float)((fl4 >> (8 * s)) & 0xffu) * fr;          // flags_left * flags_right, cc_utils.py:590
            const float hf[CCSD_MAXCN - 1] = {hfp[0][s], NP > 1 ? hfp[NP > 1 ? 1 : 0][s] : 0.f, NP > 2 ? hfp[NP > 2 ? 2 : 0][s] : 0.f};
            const float net = fnet_element<AFFINE, FW>(p, w, f, hf, m);
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
        float t2[2] = {s_net, s_z};
        block_sums<2>(t2, red);                              // (one pair of barriers; each sum in block_sum's order)
        const float tn = t2[0], tz = t2[1];
        if (threadIdx.x == 0) {
            const int nt = ncb * nrt;
            ep.part[((size_t)b * nt + tile) * 2 + 0] = tn;
            ep.part[((size_t)b * nt + tile) * 2 + 1] = tz;
        }
    }
}

