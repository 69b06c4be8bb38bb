/*
 * lfbm5d_group_slab.hip -- the group stage of configurations whose stacks do not fit the LDS (round 5): the Wiener step with 12x12 or
 * 16x16 patches or N = 32, either step on 5x5 ... 17x17 angular windows with a 2-D transform (core:277-481, :1044-1280).
 *
 * A group's stack(s) are nSx x A x k^2 floats (x 2 in the Wiener step): 288 KB for k = 16, N = 16, A = 9.  The three transforms run
 * along three different axes, so no piece of the stack is closed under all of them -- but each stage is closed under a DIFFERENT
 * cut, and only the hand-over between the cuts needs the whole stack somewhere:
 *   stage 1  per PATCH (tau_2D = dct / bior): kThreads / k patches at a time are fetched from the window images row by row into an
 *            LDS work area [patch][k][k+1], transformed there (a thread owns a row, then a column), and their coefficients written to
 *            the workgroup's slice of an HBM scratch buffer -- 16-byte loads and stores throughout;
 *   stage 2  per SLAB of coefficients (pixels when tau_2D = id, then fetched straight from the images): [stack][n][st][SLAB] in LDS --
 *            angular transform, fibres along the matches (threshold / Wiener), inverse angular transform -- and back to the slice
 *            (tau_2D = id: straight to `filt`);
 *   stage 3  per patch again: filtered coefficients from the slice, inverse 2-D transform in the work area, pixels to `filt`.
 * Round 4's general kernel (k_group_big, lfbm5d_group_generic.hip) keeps the whole stack in the slice and runs EVERY stage on it
 * with 4-byte accesses at one wave per SIMD (280-410 registers): 5-45 ms per 304^2 pass where this one takes 2.5-13 (3x3 windows;
 * profiles/r05_e_slab_kernel.txt), 174-800 ms on 9x9 ... 15x15 windows where this one takes 25-110.
 * The same transform routines / the same sums in the same order as the general kernel: the same results.
 * Not here (the general kernel keeps them): stacks of which not even four coefficients of every patch fit 112 KB of LDS (a 17x17 window
 * with N = 16 in the Wiener step), patches above 16x16 or of a side the 2-D routines have no row form for, useSD (its float sums depend
 * on the order of the fibres), the per-SAI BM3D flavour.
 */
#include "lfbm5d_group_device.h"

namespace lfbm5d {

namespace {

constexpr int kSlabFloats = 18432;    /* LDS stack of a slab: 72 KB -- the registers (170-320) allow two workgroups per CU at most */
constexpr int kSlabFloatsMax = 28800; /* ... and 112 KB, one workgroup per CU, for stacks that need it to hold four pixels (13x13 / 15x15 windows, Wiener N = 16) */
constexpr unsigned kSlabBlocks = 1024;

/* ---- 2-D transforms of the patches of a work area [patch][K][K+1], K threads per patch (thread r: row r, then column r) ---- */
template <int K>
__device__ __forceinline__ void dct_tp_rows(float* Tp, int r, bool fwd, const float* nrm, TbPtr tb) {   /* first pass of fwd2d_dct / inv2d_dct */
    constexpr int RS = K + 1;
    float x[K];
    if constexpr (K == 8) {   /* 8x8: the orthonormal butterfly DCT-8 of the dedicated 8x8 kernels (a third of the operations, no table) */
#pragma unroll
        for (int t = 0; t < 8; t++) x[t] = Tp[r * RS + t];
        if (fwd) dct8_fwd(x); else dct8_inv(x);
#pragma unroll
        for (int t = 0; t < 8; t++) Tp[r * RS + t] = x[t];
        return;
    }
    if (fwd) {
#pragma unroll
        for (int t = 0; t < K; t++) x[t] = Tp[r * RS + t];
#pragma unroll
        for (int j = 0; j < K; j++) {
            float a = 0.0f;
#pragma unroll
            for (int t = 0; t < K; t++) a += x[t] * tb->cos2[j * K + t];
            Tp[r * RS + j] = 2.0f * a;
        }
    } else {
#pragma unroll
        for (int v = 0; v < K; v++) x[v] = Tp[r * RS + v] * nrm[v];   /* coef_norm_inv[r][v] */
#pragma unroll
        for (int j = 0; j < K; j++) {
            float a = 0.0f;
#pragma unroll
            for (int v = 1; v < K; v++) a += x[v] * tb->cos2[v * K + j];
            Tp[r * RS + j] = x[0] + 2.0f * a;
        }
    }
}
template <int K>
__device__ __forceinline__ void dct_tp_cols(float* Tp, int r, bool fwd, const float* nrm, TbPtr tb) {   /* second pass: column r, in place */
    constexpr int RS = K + 1;
    float c[K];
#pragma unroll
    for (int t = 0; t < K; t++) c[t] = Tp[t * RS + r];
    if constexpr (K == 8) {
        if (fwd) dct8_fwd(c); else dct8_inv(c);
#pragma unroll
        for (int t = 0; t < 8; t++) Tp[t * RS + r] = c[t];
        return;
    }
    if (fwd) {
#pragma unroll
        for (int i = 0; i < K; i++) {   /* (unrolled: nrm stays in registers; the cosines of a row are one scalar load) */
            float a = 0.0f;
#pragma unroll
            for (int t = 0; t < K; t++) a += c[t] * tb->cos2[i * K + t];
            Tp[i * RS + r] = 2.0f * a * nrm[i];   /* coef_norm[i][r] */
        }
    } else {
        const float c2 = tb->coef2inv;
#pragma unroll 4
        for (int i = 0; i < K; i++) {
            float a = 0.0f;
#pragma unroll
            for (int u = 1; u < K; u++) a += c[u] * tb->cos2[u * K + i];
            Tp[i * RS + r] = c2 * (c[0] + 2.0f * a);
        }
    }
}
template <int K>
__device__ __forceinline__ void bior_tp(float* Tp, int r, bool fwd, TbPtr tb) {   /* all levels; the K threads of a patch share a wavefront */
    if (fwd) bior_fwd_level<K, K>(Tp, r, tb); else bior_inv_level<K, 2>(Tp, r, tb);
}

/* One 2-D stage over `np` patches: `src(p, r, x)` delivers row r of patch p (K floats), `dst(p, r, x)` takes it. */
template <int K, class SRC, class DST>
__device__ __forceinline__ void patches_2d(float* tmp, int np, unsigned tau2, bool fwd, TbPtr tb, SRC src, DST dst, long long* sub = nullptr) {
#ifdef LFBM5D_SLAB_PHASES
    long long tl = (long long)__builtin_readcyclecounter();
#define SUB_MARK(i) do { asm volatile("" ::: "memory"); const long long tn = (long long)__builtin_readcyclecounter(); if (sub) sub[i] += tn - tl; tl = tn; } while (0)
#else
#define SUB_MARK(i) do {} while (0)
#endif
    constexpr int PPI = kThreads / K, RS = K + 1;
    constexpr int PT = K == 8 ? 2 : 1, PR = PPI * PT;   /* 8x8: two patches per thread and round -- twice the rows in flight, half the rounds */
    const int tid = threadIdx.x, slot = tid / K, r = tid % K;
    float* Tp = tmp + slot * K * RS;   /* two work areas of PR patches each; a thread's patches PPI apart */
    /* K = 8, 16: the K threads of a patch share a wavefront and no other thread touches their part of the work area -- no workgroup
     * barrier anywhere in the stage (a wave's DS operations execute in order), the waves drift apart and hide each other's loads */
    constexpr bool wave_local = (64 % K) == 0;
    /* the per-thread norms, once per stage (indexed by the thread's row / column: vector loads, which inside the transform
     * cost a memory round trip per output) */
    float nrm[K];
    if (tau2 == 5 && K != 8) {
#pragma unroll
        for (int t = 0; t < K; t++) nrm[t] = fwd ? tb->cn2[t * K + r] : tb->cni2[r * K + t];
    }
#define SLAB_SYNC() do { if (wave_local) __builtin_amdgcn_wave_barrier(); else __syncthreads(); } while (0)
    /* Software pipeline over the rounds: the rows of the NEXT round are requested before this round is transformed, and the results
     * of the PREVIOUS round leave while this one is transformed (two work areas): the wait for the next rows -- vmcnt counts loads
     * and stores in order -- then finds stores that have had a whole round to be acknowledged */
    float xn[PT][K];
#pragma unroll
    for (int h = 0; h < PT; h++) if (slot < PPI && slot + h * PPI < np) src(std::integral_constant<int, K>{}, slot + h * PPI, r, xn[h]);
    float* TpPrev = Tp + PR * K * RS;
    for (int p0 = 0; p0 < np; p0 += PR) {
        bool on[PT];
#pragma unroll
        for (int h = 0; h < PT; h++) {
            on[h] = slot < PPI && p0 + slot + h * PPI < np;
            if (on[h]) {
#pragma unroll
                for (int t = 0; t < K; t++) Tp[h * PPI * K * RS + r * RS + t] = xn[h][t];
            }
        }
        if (p0 > 0 && slot < PPI) {   /* (the previous round was full) */
#pragma unroll
            for (int h = 0; h < PT; h++) {
                float x[K];
#pragma unroll
                for (int t = 0; t < K; t++) x[t] = TpPrev[h * PPI * K * RS + r * RS + t];
                dst(std::integral_constant<int, K>{}, p0 - PR + slot + h * PPI, r, x);
            }
        }
#pragma unroll
        for (int h = 0; h < PT; h++) if (slot < PPI && p0 + PR + slot + h * PPI < np) src(std::integral_constant<int, K>{}, p0 + PR + slot + h * PPI, r, xn[h]);
        SLAB_SYNC();
        SUB_MARK(0);
        if (tau2 == 5) {
#pragma unroll
            for (int h = 0; h < PT; h++) if (on[h]) dct_tp_rows<K>(Tp + h * PPI * K * RS, r, fwd, nrm, tb);
            SLAB_SYNC();
            SUB_MARK(1);
#pragma unroll
            for (int h = 0; h < PT; h++) if (on[h]) dct_tp_cols<K>(Tp + h * PPI * K * RS, r, fwd, nrm, tb);
        } else if constexpr (K != 12) {
#pragma unroll
            for (int h = 0; h < PT; h++) if (on[h]) bior_tp<K>(Tp + h * PPI * K * RS, r, fwd, tb);
        }
        SLAB_SYNC();
        SUB_MARK(2);
        { float* const sw = Tp; Tp = TpPrev; TpPrev = sw; }
    }
    if (np > 0 && slot < PPI) {   /* the last round's results */
        const int base = ((np - 1) / PR) * PR;
#pragma unroll
        for (int h = 0; h < PT; h++) {
            const int patch = base + slot + h * PPI;
            if (patch < np) {
                float x[K];
#pragma unroll
                for (int t = 0; t < K; t++) x[t] = TpPrev[h * PPI * K * RS + r * RS + t];
                dst(std::integral_constant<int, K>{}, patch, r, x);
            }
        }
    }
    SUB_MARK(3);
#undef SLAB_SYNC
#undef SUB_MARK
    __syncthreads();   /* (what the stage wrote is read by other waves next) */
}

/* ---- the kernel ---- */
template <int STEP, int WA, int MAXN>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(2, 8))) void k_group_slab(   /* at most 256 registers: two workgroups per CU (caps
                                                                                                                         * for three / four waves per SIMD: +-10 %) */
    GroupArgs a, float* scratch, unsigned long long slice_floats, int ls) {
    constexpr int A = WA * WA, NST = STEP == 2 ? 2 : 1;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ unsigned pos[MAXN * A < kSlabFloatsMax / 4 ? MAXN * A : kSlabFloatsMax / 4];   /* (group_uses_slab: N * A <= kSlabFloatsMax / 4) */
    __shared__ float red[kThreads / 64];
    __shared__ float cn4s[A], cni4s[A];
    __shared__ int sa_tab[WA > 3 ? SaLayout<WA>::words : 1];   /* shape-adaptive groups of the larger windows (sa_fill) */
    const int tid = threadIdx.x;
    const int k = a.k, k2 = k * k, N = a.N;
    const int SLAB = 1 << ls, P2 = SLAB >> 1;
    const size_t plane = (size_t)a.Wb * a.Hb;
    const TbPtr tb = (TbPtr)a.tb;
    const bool id2 = a.tau2 == 4;
    float* const slice = id2 ? nullptr : scratch + (size_t)blockIdx.x * slice_floats;   /* [stack][n * A + st][k2] coefficients (tau_2D = id: none) */
    for (int i = tid; i < A; i += kThreads) { cn4s[i] = tb->cn4[i]; cni4s[i] = tb->cni4[i]; }   /* (A = 289 > kThreads) */
    const unsigned items = a.n_groups * a.C;
    for (unsigned it = blockIdx.x; it < items; it += gridDim.x) {
        const unsigned g = a.ref_begin + it / a.C;
        const int c = (int)(it % a.C);
        const int nSx = (int)a.self_cnt[g], NSA = nSx * A;
        for (int i = tid; i < NSA; i += kThreads) pos[i] = a.gpos[(size_t)g * N * A + i];
        typedef typename std::conditional<(WA > 7), ShRefBig, ShRef>::type SH;   /* (windows beyond 7x7: the large shape record) */
        SH sh = [&]() -> SH { if constexpr (WA > 7) return group_shape_big(a, g); else return group_shape(a, g); }();
        const bool use_sadct = a.tau4 == 6 && sh.use_sadct;
        const bool do_dct4 = a.tau4 == 5 || (a.tau4 == 6 && !use_sadct);
        const bool do_sa4 = !do_dct4 && a.tau4 == 6;
        if constexpr (WA > 3) { if (do_sa4) sa_fill<WA, SH>(sa_tab, sh, tb, tid, kThreads); }   /* (read behind the barrier below) */
        const float sig = a.sigma[c];
        const float T = a.lambda * sig * 1.41421356237309505f;   /* core:2431 */
        const float sig2 = sig * sig;
        float* const out = a.filt;   /* + filt_patch(a, g, n, st, k2) */
        __syncthreads();
#ifdef LFBM5D_SLAB_PHASES   /* development builds: cycles per stage, thread 0 of every 64th workgroup (lfbm5d_api.hip prints counters 4..15) */
        long long tq[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tlast = (long long)__builtin_readcyclecounter();
#define SLAB_MARK(i) do { asm volatile("" ::: "memory"); const long long tn = (long long)__builtin_readcyclecounter(); tq[i] += tn - tlast; tlast = tn; } while (0)
#else
#define SLAB_MARK(i) do {} while (0)
#endif

        /* ---- stage 1: 2-D forward, images -> slice ---- */
        if (!id2) {
            auto src = [&](auto kt, int p, int r, float* x) {
                constexpr int K = decltype(kt)::value;
                const int stack = p / NSA, ns = p - stack * NSA;
                const unsigned pp = pos[ns];
                const float* row = (STEP == 2 && stack ? a.basic : a.noisy) + ((size_t)(ns % A) * a.C + c) * plane + (pp == 0xffffffffu ? 0u : pp) + (size_t)r * a.Wb;
#pragma unroll
                for (int t = 0; t < K; t += 4) {
                    f4u q = f4u{{0.0f, 0.0f, 0.0f, 0.0f}};
                    if (pp != 0xffffffffu) q = *reinterpret_cast<const f4u*>(row + t);
                    x[t] = q.v[0]; x[t + 1] = q.v[1]; x[t + 2] = q.v[2]; x[t + 3] = q.v[3];
                }
            };
            auto dst = [&](auto kt, int p, int r, const float* x) {
                constexpr int K = decltype(kt)::value;
                float* o = slice + (size_t)p * K * K + r * K;
#pragma unroll
                for (int t = 0; t < K; t += 4) filt_put4(reinterpret_cast<v4f*>(o + t), v4f{x[t], x[t + 1], x[t + 2], x[t + 3]});
            };
#ifdef LFBM5D_SLAB_PHASES
            long long* const subp = tq + 8;
#else
            long long* const subp = nullptr;
#endif
            if (k == 16) patches_2d<16>(lds, NST * NSA, a.tau2, true, tb, src, dst, subp);
            else if (k == 12) patches_2d<12>(lds, NST * NSA, a.tau2, true, tb, src, dst, subp);
            else patches_2d<8>(lds, NST * NSA, a.tau2, true, tb, src, dst, subp);
            /* (the slice is read back by other threads of this workgroup: its writes go through the CU's write-through L1,
             * the barrier that ended patches_2d orders them) */
        }

        SLAB_MARK(0);
        /* ---- stage 2: slabs of SLAB coefficients / pixels ---- */
        float wacc = 0.0f;
        float* const S0 = lds;
        float* const S1 = lds + ((size_t)NSA << ls);
        float* const F = STEP == 2 ? S1 : S0;
        float* const fslice = id2 ? nullptr : slice + (STEP == 2 ? (size_t)NSA * k2 : 0);   /* the filtered stack returns to its own place */
        for (int s0 = 0; s0 < k2; s0 += SLAB) {
            const int npx = min(SLAB, k2 - s0);
            /* load: [stack][ns][q], q fastest */
            if ((k & 3) == 0 && ls >= 2) {
                constexpr int G = 12;
                const int QS = SLAB >> 2, total = (NST * NSA) << (ls - 2);
                for (int e0 = tid; e0 < total; e0 += kThreads * G) {
                    v4f v[G];
#pragma unroll
                    for (int u = 0; u < G; u++) {
                        const int e = e0 + u * kThreads;
                        v[u] = v4f{0.0f, 0.0f, 0.0f, 0.0f};
                        if (e < total) {
                            const int q = 4 * (e & (QS - 1)), pidx = e >> (ls - 2);
                            if (q < npx) {
                                if (id2) {
                                    const int stack = pidx / NSA, ns = pidx - stack * NSA, pq = s0 + q;
                                    const unsigned pp = pos[ns];
                                    if (pp != 0xffffffffu) {
                                        const float* img = (STEP == 2 && stack ? a.basic : a.noisy) + ((size_t)(ns % A) * a.C + c) * plane;
                                        const f4u w = *reinterpret_cast<const f4u*>(img + pp + (size_t)(pq / k) * a.Wb + pq % k);
                                        v[u] = v4f{w.v[0], w.v[1], w.v[2], w.v[3]};
                                    }
                                } else
                                    v[u] = *reinterpret_cast<const v4f*>(slice + (size_t)pidx * k2 + s0 + q);
                            }
                        }
                    }
#pragma unroll
                    for (int u = 0; u < G; u++) { const int e = e0 + u * kThreads; if (e < total) *reinterpret_cast<v4f*>(lds + 4 * (size_t)e) = v[u]; }
                }
            } else {   /* one value per load: tau_2D = id with a patch side that is no multiple of four; slabs of two (17x17 windows, Wiener N = 16) */
                const int total = (NST * NSA) << ls;
                for (int e = tid; e < total; e += kThreads) {
                    const int q = e & (SLAB - 1), pidx = e >> ls, stack = pidx / NSA, ns = pidx - stack * NSA, pq = s0 + q;
                    float v = 0.0f;
                    if (q < npx) {
                        if (id2) {
                            const unsigned pp = pos[ns];
                            if (pp != 0xffffffffu)
                                v = ((STEP == 2 && stack ? a.basic : a.noisy) + ((size_t)(ns % A) * a.C + c) * plane)[pp + (size_t)(pq / k) * a.Wb + pq % k];
                        } else
                            v = slice[(size_t)pidx * k2 + pq];
                    }
                    lds[e] = v;
                }
            }
            __syncthreads();
            SLAB_MARK(1);
            /* angular transform forward (core:353-360) on both stacks */
            if (do_dct4 || do_sa4) {
                if constexpr (WA == 3) {
                    if (do_dct4) {   /* two pixels per lane: the scalar routine's operation sequence on packed values */
                        v2f* const S2 = reinterpret_cast<v2f*>(lds);
                        for (int f = tid; f < (NST * nSx) << (ls - 1); f += kThreads) {
                            const int pp = f & (P2 - 1), sn = f >> (ls - 1);   /* sn = stack * nSx + n */
                            v2f* B = S2 + ((size_t)sn * 9 << (ls - 1)) + pp;
                            v2f x[9];
#pragma unroll
                            for (int st = 0; st < 9; st++) x[st] = B[st << (ls - 1)];
                            dct9_fwd2(x, tb);
#pragma unroll
                            for (int st = 0; st < 9; st++) B[st << (ls - 1)] = x[st];
                        }
                    } else
                    for (int f = tid; f < (NST * nSx) << ls; f += kThreads) {
                        const int q = f & (SLAB - 1), sn = f >> ls;
                        float* B = lds + ((size_t)sn * 9 << ls) + q;
                        float x[9];
#pragma unroll
                        for (int st = 0; st < 9; st++) x[st] = B[st << ls];
                        sadct9_fwd(x, sh, tb);
#pragma unroll
                        for (int st = 0; st < 9; st++) B[st << ls] = x[st];
                    }
                } else if (do_dct4) {
                    /* separable, pixel pairs (lfbm5d_group_wide.hip): rows of the aw x aw block, then columns times coef_norm_4d */
                    v2f* const S2 = reinterpret_cast<v2f*>(lds);
                    for (int e = tid; e < (NST * nSx * WA) << (ls - 1); e += kThreads) {
                        const int pp = e & (P2 - 1), r = e >> (ls - 1);
                        v2f* row = S2 + ((size_t)r * WA << (ls - 1)) + pp;
                        v2f x[WA], t[WA];
#pragma unroll
                        for (int j = 0; j < WA; j++) x[j] = row[j << (ls - 1)];
#pragma unroll
                        for (int u = 0; u < WA; u++) {
                            v2f acc = {0.0f, 0.0f};
#pragma unroll
                            for (int j = 0; j < WA; j++) acc += x[j] * tb->cosw[u * WA + j];
                            t[u] = 2.0f * acc;
                        }
#pragma unroll
                        for (int u = 0; u < WA; u++) row[u << (ls - 1)] = t[u];
                    }
                    __syncthreads();
                    for (int e = tid; e < (NST * nSx * WA) << (ls - 1); e += kThreads) {
                        const int pp = e & (P2 - 1), r = e >> (ls - 1), u = r % WA, sn = r / WA;
                        v2f* col = S2 + ((size_t)(sn * A + u) << (ls - 1)) + pp;
                        v2f t[WA], x[WA];
#pragma unroll
                        for (int j = 0; j < WA; j++) t[j] = col[(j * WA) << (ls - 1)];
#pragma unroll
                        for (int v = 0; v < WA; v++) {
                            v2f acc = {0.0f, 0.0f};
#pragma unroll
                            for (int j = 0; j < WA; j++) acc += t[j] * tb->cosw[v * WA + j];
                            x[v] = 2.0f * acc * cn4s[v * WA + u];
                        }
#pragma unroll
                        for (int v = 0; v < WA; v++) col[(v * WA) << (ls - 1)] = x[v];
                    }
                } else {   /* shape-adaptive groups of the larger windows: the same separable form with the shape record's lengths */
                    v2f* const S2 = reinterpret_cast<v2f*>(lds);
                    for (int e = tid; e < (NST * nSx * WA) << (ls - 1); e += kThreads) {
                        const int pp = e & (P2 - 1), r = e >> (ls - 1);
                        sadctw_rows_fwd2<WA>((LdsV2)(S2 + ((size_t)r * WA << (ls - 1)) + pp), P2, r % WA, (SaTab)sa_tab);
                    }
                    __syncthreads();
                    for (int e = tid; e < (NST * nSx * WA) << (ls - 1); e += kThreads) {
                        const int pp = e & (P2 - 1), r = e >> (ls - 1), u = r % WA, sn = r / WA;
                        sadctw_cols_fwd2<WA>((LdsV2)(S2 + ((size_t)(sn * A + u) << (ls - 1)) + pp), P2, u, (SaTab)sa_tab);
                    }
                }
                __syncthreads();
            }
            SLAB_MARK(2);
            /* the fibres along the matches (core:371-410 / :1118-1160) */
            {
                float s1 = 0.0f, s2 = 0.0f;   /* (useSD is not served here) */
                for (int f = tid; f < A << ls; f += kThreads) {
                    const int q = f & (SLAB - 1), st = f >> ls;
                    if (q >= npx) continue;
                    const bool in_shape = !use_sadct || sh.mask_dct[st];
                    const int base = (st << ls) + q, stride = A << ls;
                    switch (nSx) {
                        case 1:  filter5<1, STEP>(S0, S1, base, stride, a.tau5, T, sig2, in_shape, wacc, s1, s2, tb); break;
                        case 2:  filter5<2, STEP>(S0, S1, base, stride, a.tau5, T, sig2, in_shape, wacc, s1, s2, tb); break;
                        case 4:  filter5<4, STEP>(S0, S1, base, stride, a.tau5, T, sig2, in_shape, wacc, s1, s2, tb); break;
                        case 8:  filter5<8, STEP>(S0, S1, base, stride, a.tau5, T, sig2, in_shape, wacc, s1, s2, tb); break;
                        case 16: filter5<16, STEP>(S0, S1, base, stride, a.tau5, T, sig2, in_shape, wacc, s1, s2, tb); break;
                        default: if constexpr (MAXN >= 32) filter5<32, STEP>(S0, S1, base, stride, a.tau5, T, sig2, in_shape, wacc, s1, s2, tb); break;
                    }
                }
            }
            __syncthreads();
            SLAB_MARK(3);
            /* angular transform inverse (core:431-451) on the filtered stack */
            if (do_dct4 || do_sa4) {
                if constexpr (WA == 3) {
                    if (do_dct4) {
                        v2f* const F2 = reinterpret_cast<v2f*>(F);
                        for (int f = tid; f < nSx << (ls - 1); f += kThreads) {
                            const int pp = f & (P2 - 1), n = f >> (ls - 1);
                            v2f* B = F2 + ((size_t)n * 9 << (ls - 1)) + pp;
                            v2f x[9];
#pragma unroll
                            for (int st = 0; st < 9; st++) x[st] = B[st << (ls - 1)];
                            dct9_inv2(x, tb);
#pragma unroll
                            for (int st = 0; st < 9; st++) B[st << (ls - 1)] = x[st];
                        }
                    } else
                    for (int f = tid; f < nSx << ls; f += kThreads) {
                        const int q = f & (SLAB - 1), n = f >> ls;
                        float* B = F + ((size_t)n * 9 << ls) + q;
                        float x[9];
#pragma unroll
                        for (int st = 0; st < 9; st++) x[st] = B[st << ls];
                        sadct9_inv(x, sh, tb);
#pragma unroll
                        for (int st = 0; st < 9; st++) B[st << ls] = x[st];
                    }
                } else if (do_dct4) {
                    v2f* const F2 = reinterpret_cast<v2f*>(F);
                    for (int e = tid; e < (nSx * WA) << (ls - 1); e += kThreads) {
                        const int pp = e & (P2 - 1), r = e >> (ls - 1), s = r % WA;
                        v2f* row = F2 + ((size_t)r * WA << (ls - 1)) + pp;
                        v2f x[WA], t[WA];
#pragma unroll
                        for (int u = 0; u < WA; u++) x[u] = row[u << (ls - 1)] * cni4s[s * WA + u];
#pragma unroll
                        for (int j = 0; j < WA; j++) {
                            v2f acc = {0.0f, 0.0f};
#pragma unroll
                            for (int u = 1; u < WA; u++) acc += x[u] * tb->cosw[u * WA + j];
                            t[j] = x[0] + 2.0f * acc;
                        }
#pragma unroll
                        for (int j = 0; j < WA; j++) row[j << (ls - 1)] = t[j];
                    }
                    __syncthreads();
                    for (int e = tid; e < (nSx * WA) << (ls - 1); e += kThreads) {
                        const int pp = e & (P2 - 1), r = e >> (ls - 1), j = r % WA, n = r / WA;
                        v2f* col = F2 + ((size_t)(n * A + j) << (ls - 1)) + pp;
                        v2f t[WA], y[WA];
#pragma unroll
                        for (int v = 0; v < WA; v++) t[v] = col[(v * WA) << (ls - 1)];
#pragma unroll
                        for (int i = 0; i < WA; i++) {
                            v2f acc = {0.0f, 0.0f};
#pragma unroll
                            for (int v = 1; v < WA; v++) acc += t[v] * tb->cosw[v * WA + i];
                            y[i] = (t[0] + 2.0f * acc) * tb->coef4inv;
                        }
#pragma unroll
                        for (int i = 0; i < WA; i++) col[(i * WA) << (ls - 1)] = y[i];
                    }
                } else {
                    v2f* const F2 = reinterpret_cast<v2f*>(F);
                    for (int e = tid; e < (nSx * WA) << (ls - 1); e += kThreads) {
                        const int pp = e & (P2 - 1), r = e >> (ls - 1), u = r % WA, n = r / WA;
                        sadctw_cols_inv2<WA>((LdsV2)(F2 + ((size_t)(n * A + u) << (ls - 1)) + pp), P2, u, (SaTab)sa_tab);
                    }
                    __syncthreads();
                    for (int e = tid; e < (nSx * WA) << (ls - 1); e += kThreads) {
                        const int pp = e & (P2 - 1), r = e >> (ls - 1);
                        sadctw_rows_inv2<WA>((LdsV2)(F2 + ((size_t)r * WA << (ls - 1)) + pp), P2, r % WA, (SaTab)sa_tab);
                    }
                }
                __syncthreads();
            }
            SLAB_MARK(4);
            /* the filtered slab: back to the slice, or (tau_2D = id) out */
            if ((k & 3) == 0 && ls >= 2) {
                const int QS = SLAB >> 2, total = NSA << (ls - 2);
                for (int e = tid; e < total; e += kThreads) {
                    const int q = 4 * (e & (QS - 1)), ns = e >> (ls - 2);
                    if (q < npx) {
                        const v4f v = *reinterpret_cast<const v4f*>(F + 4 * (size_t)e);
                        float* o = id2 ? out + filt_patch(a, g, ns / A, ns % A, k2) + (size_t)c * k2 + s0 + q : fslice + (size_t)ns * k2 + s0 + q;
                        if (id2) filt_put4(reinterpret_cast<v4f*>(o), v); else *reinterpret_cast<v4f*>(o) = v;   /* (the slice is read back by this workgroup) */
                    }
                }
            } else {
                for (int e = tid; e < NSA << ls; e += kThreads) {
                    const int q = e & (SLAB - 1), ns = e >> ls;
                    if (q < npx) {
                        if (id2) filt_put(&out[filt_patch(a, g, ns / A, ns % A, k2) + (size_t)c * k2 + s0 + q], F[e]);
                        else fslice[(size_t)ns * k2 + s0 + q] = F[e];
                    }
                }
            }
            __syncthreads();
            SLAB_MARK(5);
        }

        /* group weight (core:412-421) */
        for (int o = 32; o > 0; o >>= 1) wacc += __shfl_xor(wacc, o);
        if ((tid & 63) == 0) red[tid >> 6] = wacc;
        __syncthreads();
        if (tid == 0) {
            float w = 0.0f;
            for (int i = 0; i < kThreads / 64; i++) w += red[i];
            a.wgt[(size_t)g * a.C + c] = w > 0.0f ? (sig > 0.0f ? 1.0f / (sig * sig * w) : 1.0f / w) : 1.0f;
            if (c == 0) {
                atomicAdd(&a.counters[0], (unsigned long long)nSx);
                if (use_sadct) atomicAdd(&a.counters[1], 1ull);
            }
        }

        /* ---- stage 3: 2-D inverse, slice -> filt ---- */
        if (!id2) {
            auto src = [&](auto kt, int p, int r, float* x) {
                constexpr int K = decltype(kt)::value;
                const float* row = fslice + (size_t)p * K * K + r * K;
#pragma unroll
                for (int t = 0; t < K; t += 4) { const v4f q = *reinterpret_cast<const v4f*>(row + t); x[t] = q[0]; x[t + 1] = q[1]; x[t + 2] = q[2]; x[t + 3] = q[3]; }
            };
            auto dst = [&](auto kt, int p, int r, const float* x) {
                constexpr int K = decltype(kt)::value;
                float* o = out + filt_patch(a, g, p / A, p % A, K * K) + (size_t)c * K * K + r * K;
#pragma unroll
                for (int t = 0; t < K; t += 4) filt_put4(reinterpret_cast<v4f*>(o + t), v4f{x[t], x[t + 1], x[t + 2], x[t + 3]});
            };
            if (k == 16) patches_2d<16>(lds, NSA, a.tau2, false, tb, src, dst);
            else if (k == 12) patches_2d<12>(lds, NSA, a.tau2, false, tb, src, dst);
            else patches_2d<8>(lds, NSA, a.tau2, false, tb, src, dst);
        }
        __syncthreads();   /* pos / red / the LDS are reused by the next item */
        SLAB_MARK(6);
#ifdef LFBM5D_SLAB_PHASES
        if (tid == 0 && blockIdx.x % 64 == 5) { for (int i = 0; i < 7; i++) atomicAdd(&a.counters[4 + i], (unsigned long long)tq[i]); atomicAdd(&a.counters[11], 1ull); for (int i = 0; i < 4; i++) atomicAdd(&a.counters[12 + i], (unsigned long long)tq[8 + i]); }
#endif
    }
}

int slab_log2(const GroupArgs& a) {
    const int per_px = (a.step == 2 ? 2 : 1) * (int)a.N * (int)a.A;
    int ls = 6;
    while (ls > 2 && (per_px << ls) > kSlabFloats) ls--;
    if ((per_px << ls) > kSlabFloatsMax) ls = 1;   /* slabs of two values, one packed pair per item (17x17 windows, Wiener N = 16: 74 KB) */
    return ls;
}
size_t slab_lds_bytes(const GroupArgs& a) {
    const size_t stack = (size_t)((a.step == 2 ? 2 : 1) * a.N * a.A) << slab_log2(a);
    const size_t work = a.tau2 == 4 ? 0 : (size_t)2 * (a.k == 8 ? 2 : 1) * (kThreads / a.k) * a.k * (a.k + 1);   /* two work areas of a round's patches (patches_2d) */
    return std::max(stack, work) * sizeof(float);
}

} /* namespace */

/* Which configurations the slab kernel takes (the dedicated kernels have been asked first): those the general kernel would run
 * from HBM slices or with one workgroup per CU. */
static int slab_window_side(unsigned A) {   /* 3, 5, ... 17, or 0 */
    for (int w = 3; w <= 17; w += 2) if (A == (unsigned)(w * w)) return w;
    return 0;
}
bool group_uses_slab(const GroupArgs& a) {
    if (a.opt & kOptNoSlabKernel) return false;
    const int wa = slab_window_side(a.A);
    if (a.bm3d || a.useSD || !wa || a.N > (wa > 9 ? 16u : 32u)) return false;   /* (windows beyond 9x9: the N <= 16 instances only) */
    if (a.tau2 == 5 && !(a.k == 8 || a.k == 12 || a.k == 16)) return false;
    if (a.tau2 == 7 && !(a.k == 8 || a.k == 16)) return false;
    if (a.tau2 == 4 && a.k > 16) return false;
    if (((size_t)(a.step == 2 ? 2 : 1) * a.N * a.A << 1) > (size_t)kSlabFloatsMax) return false;   /* (not even a pair of pixels per slab) */
    /* stacks the general kernel would keep in HBM slices (beyond ~150 KB); with a 2-D transform also those it would hold in LDS at
     * one workgroup per CU (measured: N = 32, k = 8, Wiener: dct 3.9 ms here / 4.8 there, id 3.1 / 2.7) */
    const size_t stacks = (size_t)(a.step == 2 ? 2 : 1) * a.N * a.A * a.k * a.k * sizeof(float);
    return stacks > (a.tau2 == 4 ? (size_t)150 : (size_t)64) * 1024;
}
size_t group_slab_scratch_bytes(const GroupArgs& a) {
    if (!group_uses_slab(a) || a.tau2 == 4) return 0;
    return (size_t)kSlabBlocks * (a.step == 2 ? 2 : 1) * a.N * a.A * a.k * a.k * sizeof(float);
}

template <int WA, int MAXN>
static hipError_t prepare_slab() {
    const void* fns[] = {reinterpret_cast<const void*>(&k_group_slab<1, WA, MAXN>), reinterpret_cast<const void*>(&k_group_slab<2, WA, MAXN>)};
    for (const void* f : fns) {
        /* the cap follows the SAME bound as eligibility (group_uses_slab admits any stack whose four-pixel slab is within
         * kSlabFloatsMax floats, e.g. 9x9 windows, Wiener step, N = 32: 81 KB): a cap per window side once refused such a launch */
        const hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 116 * 1024);   /* slab stack <= 112 KB, two work areas <= 35 KB */
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}
hipError_t prepare_group_slab() {
    hipError_t e = prepare_slab<3, 16>();
#define LFBM5D_PS(WA, MAXN) if (e == hipSuccess) e = prepare_slab<WA, MAXN>()
    LFBM5D_PS(3, 32); LFBM5D_PS(5, 16); LFBM5D_PS(5, 32); LFBM5D_PS(7, 16); LFBM5D_PS(7, 32); LFBM5D_PS(9, 16); LFBM5D_PS(9, 32);
    LFBM5D_PS(11, 16); LFBM5D_PS(13, 16); LFBM5D_PS(15, 16); LFBM5D_PS(17, 16);
#undef LFBM5D_PS
    return e;
}

template <int WA, int MAXN>
static void launch_slab(hipStream_t s, const GroupArgs& a, unsigned blocks, size_t lds, unsigned long long slice, int ls) {
    if (a.step == 2) hipLaunchKernelGGL((k_group_slab<2, WA, MAXN>), dim3(blocks), dim3(kThreads), lds, s, a, a.scratch, slice, ls);
    else             hipLaunchKernelGGL((k_group_slab<1, WA, MAXN>), dim3(blocks), dim3(kThreads), lds, s, a, a.scratch, slice, ls);
}

hipError_t launch_group_slab(hipStream_t s, const GroupArgs& a, bool* launched) {
    *launched = false;
    if (!group_uses_slab(a)) return hipSuccess;
    const unsigned long long slice = (unsigned long long)(a.step == 2 ? 2 : 1) * a.N * a.A * a.k * a.k;
    if (a.tau2 != 4 && (!a.scratch || a.scratch_floats < slice * kSlabBlocks)) return hipErrorInvalidValue;
    *launched = true;
    const unsigned blocks = std::min<unsigned>(kSlabBlocks, a.n_groups * a.C);
    const int ls = slab_log2(a);
    const size_t lds = slab_lds_bytes(a);
    const bool n16 = a.N <= 16;
#define LFBM5D_LS(WA) do { if (n16) launch_slab<WA, 16>(s, a, blocks, lds, slice, ls); else launch_slab<WA, 32>(s, a, blocks, lds, slice, ls); } while (0)
    switch (slab_window_side(a.A)) {
        case 3: LFBM5D_LS(3); break;
        case 5: LFBM5D_LS(5); break;
        case 7: LFBM5D_LS(7); break;
        case 9: LFBM5D_LS(9); break;
        case 11: launch_slab<11, 16>(s, a, blocks, lds, slice, ls); break;
        case 13: launch_slab<13, 16>(s, a, blocks, lds, slice, ls); break;
        case 15: launch_slab<15, 16>(s, a, blocks, lds, slice, ls); break;
        default: launch_slab<17, 16>(s, a, blocks, lds, slice, ls); break;
    }
#undef LFBM5D_LS
    return hipGetLastError();
}

} /* namespace lfbm5d */
