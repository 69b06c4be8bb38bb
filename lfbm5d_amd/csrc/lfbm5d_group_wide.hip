/*
 * lfbm5d_group_wide.hip -- dedicated group kernel of the hard-thresholding step with tau_2D = id on 5x5, 7x7 and 9x9 angular windows
 * (aswSize 2 / 3 / 4; core:277-481 with aheight x awidth from bm5d.cpp:215-218), round 5.
 *
 * With no 2-D transform a pixel of the patches never mixes with other pixels, so a (group, channel) is processed in SLABS of
 * pixels, a workgroup per slab: the slab's nSx x A x SLAB stack lives in LDS (51 KB for a 5x5 window and 64 pixels, 50 KB for 7x7
 * and 32: three workgroups per CU) --
 *   gather the slab's pixels of every patch: 16-byte loads (4-byte aligned) of four pixels of a patch row, all 13 of a thread in
 *   flight at once, the patch positions read straight from the table (no LDS round trip, no barrier before the first image load);
 *   the aw x aw angular DCT as two separable passes over the stack -- item = (match, row of the block, pixel PAIR), then (match,
 *   column, pair): aw packed values per thread (a thread holding all 49 values of a 7x7 block needed 325 registers; the fused form
 *   of the 5x5 block spills at 168);
 *   the fibres along the matches, two per lane: Haar / Hadamard / DCT, hard threshold, inverse;
 *   the inverse angular passes, the second writing four filtered pixels per 16-byte store straight to `filt`.
 * The general kernel keeps the whole 200 KB stack of such a group in an HBM scratch slice: 5.8 / 12.8 ms per 304^2 pass (5x5 / 7x7,
 * k = 16, N = 8; 9x9: 200 ms) where this one takes 1.01 / 2.26 ms (9x9: 5.8); same operations in the same order per value, hence the same
 * results.  Shape-adaptive groups (2 / 9 / 223 of 3721 there) run the same separable passes with the row / column lengths of their shape
 * record (sadctw_*2, out of line, on an LDS copy of the record): as a per-pixel block in private memory nine such groups cost the
 * 7x7 pass 0.4 ms.
 * Where the time goes now (in-kernel clocks of a sample of workgroups, profiles/r05_d_wide_window.txt): a workgroup lives 14 us of
 * which the gather is 4, the four angular passes 6, the fibres 2.7; VALU issue is ~55 % busy at three waves per SIMD (1900 VALU
 * instructions per wave, a fifth of them the transforms' FMAs), and `filt` itself is 2.3 / 4.5 GB per pass.
 */
#include "lfbm5d_group_device.h"

namespace lfbm5d {

namespace {

/* 5th-dimension filter of the hard-thresholding step (core:2408-2505 / :2281-2391) on TWO fibres per lane -- neighbouring pixels of
 * the LDS stack: 8-byte LDS accesses and packed arithmetic, the same operations in the same order per fibre as filter5 /
 * shrink_fibre.  The survivor count is a whole number: exact in any order. */
template <int NS>
__device__ __forceinline__ void filter5_pair(v2f* S2, int base, int stride, unsigned tau5, float T, bool in_shape, float& wacc, TbPtr tb) {
    v2f o[NS];
#pragma unroll
    for (int n = 0; n < NS; n++) o[n] = S2[base + n * stride];
    const bool haar = tau5 == 9, dct = tau5 == 5;
    if (dct) dct5_fwd<NS>(o, tb);
    else if (NS > 1) { if (haar) haar_fwd<NS>(o); else hadamard<NS>(o); }
    if (in_shape) {
        const float Th = haar ? T : (dct ? T * 2.0f : T * sqrtf((float)NS));   /* shrink_fibre */
#pragma unroll
        for (int n = 0; n < NS; n++) {
            if (fabsf(o[n].x) > Th) wacc += 1.0f; else o[n].x = 0.0f;
            if (fabsf(o[n].y) > Th) wacc += 1.0f; else o[n].y = 0.0f;
        }
    }
    if (dct) dct5_inv<NS>(o, tb);
    else if (NS > 1) {
        if (haar) haar_inv<NS>(o);
        else {
            hadamard<NS>(o);
            const float hc = 1.0f / (float)NS;
#pragma unroll
            for (int n = 0; n < NS; n++) o[n] *= hc;
        }
    }
#pragma unroll
    for (int n = 0; n < NS; n++) S2[base + n * stride] = o[n];
}

/* SPLIT: a workgroup per (group, channel, SLAB of pixels) -- blockIdx.z = slab -- instead of one per (group, channel) that walks the
 * slabs: four to eight times the workgroups, three of them per CU in different phases, so that gathers, transforms and stores of
 * different slabs overlap.  The survivor count of a (group, channel) is then summed with atomics in wgt (whole numbers: exact in any
 * order; the buffer is zeroed before the launch) and turned into the weight by k_group_idw_weight.  useSD (float sums whose value
 * depends on the order) keeps the walking form. */
/* slabs a workgroup of the SPLIT form takes one after the other (one prologue for them): two up to 7x7, four beyond (measured:
 * 5x5 1.09 / 1.06 / 1.13 ms with 1 / 2 / 4, 7x7 2.45 / 2.40 / 2.48, 9x9 7.8 / 7.5 / 7.2, 13x13 34.2 (2) / 33.7 (4)) */
template <int AW> constexpr int wide_spw() { return AW <= 7 ? 2 : 4; }
template <int AW, int SLAB, bool SPLIT>
__global__ __launch_bounds__(256) void k_group_idw(GroupArgs a) {   /* (512 threads at six waves per SIMD: slower, 80 registers spill) */
    constexpr int A = AW * AW, NT = 256;
    extern __shared__ float S[];                      /* [n][st][SLAB] */
    __shared__ float red[3][NT / 64];
    __shared__ float cn4s[A], cni4s[A];               /* coef_norm_4d / coef_norm_inv_4d: indexed per thread in the column / row passes */
    __shared__ int sa_tab[SaLayout<AW>::words];       /* shape-adaptive groups: the shape record and the tables of every row length (sa_fill) */
    constexpr int P2 = SLAB / 2;                      /* the angular passes work on pixel PAIRS: v_pk_fma_f32, 8-byte LDS accesses */
    v2f* const S2 = reinterpret_cast<v2f*>(S);
    const int tid = threadIdx.x;
    const unsigned gi = xcd_group_index(a);
    if (gi >= a.n_groups) return;
    const unsigned g = a.ref_begin + gi, slab = blockIdx.z;
    const int c = blockIdx.y;
    const int k = a.k, k2 = k * k, N = a.N;
    const size_t plane = (size_t)a.Wb * a.Hb;
    const TbPtr tb = (TbPtr)a.tb;
    for (int i = tid; i < A; i += NT) { cn4s[i] = tb->cn4[i]; cni4s[i] = tb->cni4[i]; }   /* (first read behind the barrier that follows the gather; A = 289 > NT) */
    const int nSx = (int)a.self_cnt[g];
    /* prologue: everything the workgroup needs from memory is requested at once -- the group's size and shape, sigma, the norm
     * tables, and the positions of the patches each thread will fetch (16-byte gather: straight into registers, no LDS round trip
     * and no barrier before the first image load) */
    constexpr int QS = SLAB / 4, GQ = (8 * A * QS + NT - 1) / NT;   /* 16-byte loads per thread: 13 */
    unsigned pv[GQ];
    const bool quads = (k & 3) == 0;
    if (quads) {
#pragma unroll
        for (int u = 0; u < GQ; u++) pv[u] = a.gpos[(size_t)g * N * A + min((tid + u * NT) / QS, N * A - 1)];
    }   /* (patch sides that are no multiple of four: one pixel per load, the positions from the table load by load -- no LDS copy) */
    /* (windows beyond 7x7: the large shape record of the pre-pass) */
    typedef typename std::conditional<(AW > 7), ShRefBig, ShRef>::type SH;
    SH sh = [&]() -> SH { if constexpr (AW > 7) return group_shape_big(a, g); else return group_shape(a, g); }();
    const bool use_sadct = a.tau4 == 6 && sh.use_sadct;
    const bool do_dct4 = a.tau4 == 5 || (a.tau4 == 6 && !use_sadct);
    const bool do_sa4 = !do_dct4 && a.tau4 == 6;
    if (do_sa4) sa_fill<AW, SH>(sa_tab, sh, tb, tid, NT);   /* (read behind the barrier that follows the first gather) */
    const float sig = a.sigma[c];
    const float T = a.lambda * sig * 1.41421356237309505f;   /* core:2431 */
    const float sig2 = sig * sig;
    float wacc = 0.0f, s1 = 0.0f, s2 = 0.0f;
    float* const out = a.filt;   /* + filt_patch(a, g, n, st, k2): group-major, or SAI-major on windows of 11 x 11 SAIs and more */
    const float* const img = a.noisy + (size_t)c * plane;
#ifdef LFBM5D_WIDE_PHASES   /* development builds (with -DLFBM5D_WIDE_PHASES, which makes lfbm5d_api.hip print them): cycles per phase of the slab loop, summed over the workgroups */
    long long tq[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tlast = (long long)__builtin_readcyclecounter();
    const long long treal0 = (long long)__builtin_amdgcn_s_memrealtime();
#define WIDE_MARK(i) do { asm volatile("" ::: "memory"); const long long tn = (long long)__builtin_readcyclecounter(); tq[i] += tn - tlast; tlast = tn; } while (0)
#else
#define WIDE_MARK(i) do {} while (0)
#endif
    for (int p0 = SPLIT ? (int)slab * SLAB * wide_spw<AW>() : 0; p0 < (SPLIT ? min(k2, ((int)slab + 1) * SLAB * wide_spw<AW>()) : k2); p0 += SLAB) {
        const int npx = min(SLAB, k2 - p0);
        /* gather (core:286-299): item = (patch, pixel), pixel fastest */
        if (quads) {   /* four pixels of a patch row per 16-byte load (4-byte aligned), every load of the slab in flight at once */
            constexpr int G = GQ;
            const int total = nSx * A * QS;           /* <= 8 * 25 * 16 = 3200, 8 * 49 * 8 = 3136: at most 13 per thread */
            const unsigned cplane = (unsigned)(a.C * plane);   /* (window images of up to 2^31 floats: launch_group_wide) */
            const int px = 4 * (tid % QS), pq = p0 + px;   /* NT is a multiple of QS: the same pixels for every load of the thread */
            const unsigned poff = (unsigned)(pq / k) * a.Wb + (unsigned)(pq % k);
            f4u v[G];
#pragma unroll
            for (int u = 0; u < G; u++) {
                const int e = tid + u * NT, ns = e / QS;
                v[u] = f4u{{0.0f, 0.0f, 0.0f, 0.0f}};
                if (e < total && pv[u] != 0xffffffffu && px < npx) v[u] = *reinterpret_cast<const f4u*>(img + ((unsigned)(ns % A) * cplane + pv[u] + poff));
            }
#ifdef LFBM5D_WIDE_PHASES
            WIDE_MARK(9);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            WIDE_MARK(6);
#endif
#pragma unroll
            for (int u = 0; u < G; u++) {
                const int e = tid + u * NT;
                if (e < total) *reinterpret_cast<v4f*>(S + 4 * e) = v4f{v[u].v[0], v[u].v[1], v[u].v[2], v[u].v[3]};
            }
        } else {
            constexpr int G = 10;   /* loads in flight per thread */
            const int total = nSx * A * SLAB;
            const unsigned cplane = (unsigned)(a.C * plane);
            for (int e0 = tid; e0 < total; e0 += NT * G) {
                float v[G];
#pragma unroll
                for (int u = 0; u < G; u++) {
                    const int e = e0 + u * NT;
                    v[u] = 0.0f;
                    if (e < total) {
                        const int px = e % SLAB, ns = e / SLAB, pq = p0 + px;
                        const unsigned p = a.gpos[(size_t)g * N * A + ns];
                        if (p != 0xffffffffu && px < npx) v[u] = img[(unsigned)(ns % A) * cplane + p + (unsigned)(pq / k) * a.Wb + (unsigned)(pq % k)];
                    }
                }
#pragma unroll
                for (int u = 0; u < G; u++) { const int e = e0 + u * NT; if (e < total) S[e] = v[u]; }
            }
        }
        __syncthreads();
        WIDE_MARK(0);
        if (do_dct4) {
            /* forward angular DCT (dct_4d_process, core:1862-1901), separable: rows of the aw x aw block ... */
            for (int e = tid; e < nSx * AW * P2; e += NT) {
                const int pp = e % P2, r = e / P2;                       /* r = match * AW + row of the block */
                v2f* row = S2 + (size_t)r * AW * P2 + pp;
                v2f x[AW], t[AW];
#pragma unroll
                for (int j = 0; j < AW; j++) x[j] = row[j * P2];
#pragma unroll
                for (int u = 0; u < AW; u++) {
                    v2f acc = {0.0f, 0.0f};
#pragma unroll
                    for (int j = 0; j < AW; j++) acc += x[j] * tb->cosw[u * AW + j];
                    t[u] = 2.0f * acc;
                }
#pragma unroll
                for (int u = 0; u < AW; u++) row[u * P2] = t[u];
            }
            __syncthreads();
            WIDE_MARK(1);
            /* ... then columns, times coef_norm_4d */
            for (int e = tid; e < nSx * AW * P2; e += NT) {
                const int pp = e % P2, r = e / P2, u = r % AW, n = r / AW;
                v2f* col = S2 + (size_t)(n * A + u) * P2 + pp;
                v2f t[AW], x[AW];
#pragma unroll
                for (int j = 0; j < AW; j++) t[j] = col[j * AW * P2];
#pragma unroll
                for (int v = 0; v < AW; v++) {
                    v2f acc = {0.0f, 0.0f};
#pragma unroll
                    for (int j = 0; j < AW; j++) acc += t[j] * tb->cosw[v * AW + j];
                    x[v] = 2.0f * acc * cn4s[v * AW + u];
                }
#pragma unroll
                for (int v = 0; v < AW; v++) col[v * AW * P2] = x[v];
            }
            __syncthreads();
        } else if (do_sa4) {   /* shape-adaptive groups: the same separable form, row / column lengths from the group's shape record */
            for (int e = tid; e < nSx * AW * P2; e += NT) {
                const int pp = e % P2, r = e / P2;
                sadctw_rows_fwd2<AW>((LdsV2)(S2 + (size_t)r * AW * P2 + pp), P2, r % AW, (SaTab)sa_tab);
            }
            __syncthreads();
            for (int e = tid; e < nSx * AW * P2; e += NT) {
                const int pp = e % P2, r = e / P2, u = r % AW, n = r / AW;
                sadctw_cols_fwd2<AW>((LdsV2)(S2 + (size_t)(n * A + u) * P2 + pp), P2, u, (SaTab)sa_tab);
            }
            __syncthreads();
        }
        WIDE_MARK(2);
        /* the fibres along the matches (core:371-410): pairs of pixels; the standard-deviation weights (useSD: float sums whose
         * value depends on the order) keep the one-fibre form and its order */
        if (!SPLIT && a.useSD) {
            for (int f = tid; f < A * SLAB; f += NT) {
                const int st = f / SLAB, q = f % SLAB;
                if (q >= npx) continue;
                const bool in_shape = !use_sadct || sh.mask_dct[st];
                const int base = st * SLAB + q, stride = A * SLAB;
                switch (nSx) {
                    case 1:  filter5<1, 1>(S, nullptr, base, stride, a.tau5, T, sig2, in_shape, wacc, s1, s2, tb); break;
                    case 2:  filter5<2, 1>(S, nullptr, base, stride, a.tau5, T, sig2, in_shape, wacc, s1, s2, tb); break;
                    case 4:  filter5<4, 1>(S, nullptr, base, stride, a.tau5, T, sig2, in_shape, wacc, s1, s2, tb); break;
                    default: filter5<8, 1>(S, nullptr, base, stride, a.tau5, T, sig2, in_shape, wacc, s1, s2, tb); break;
                }
            }
        } else {
            for (int f = tid; f < A * P2; f += NT) {
                const int st = f / P2, q2 = f % P2;
                if (2 * q2 >= npx) continue;   /* (an odd tail pixel's partner holds zeros: they stay zeros and count nothing) */
                const bool in_shape = !use_sadct || sh.mask_dct[st];
                const int base = st * P2 + q2, stride = A * P2;
                switch (nSx) {
                    case 1:  filter5_pair<1>(S2, base, stride, a.tau5, T, in_shape, wacc, tb); break;
                    case 2:  filter5_pair<2>(S2, base, stride, a.tau5, T, in_shape, wacc, tb); break;
                    case 4:  filter5_pair<4>(S2, base, stride, a.tau5, T, in_shape, wacc, tb); break;
                    default: filter5_pair<8>(S2, base, stride, a.tau5, T, in_shape, wacc, tb); break;
                }
            }
        }
        __syncthreads();
        WIDE_MARK(3);
        if (do_dct4) {
            /* inverse angular DCT (dct_4d_inverse, core:1913-1954): times coef_norm_inv, rows ... */
            for (int e = tid; e < nSx * AW * P2; e += NT) {
                const int pp = e % P2, r = e / P2, s = r % AW;
                v2f* row = S2 + (size_t)r * AW * P2 + pp;
                v2f x[AW], t[AW];
#pragma unroll
                for (int u = 0; u < AW; u++) x[u] = row[u * P2] * cni4s[s * AW + u];
#pragma unroll
                for (int j = 0; j < AW; j++) {
                    v2f acc = {0.0f, 0.0f};
#pragma unroll
                    for (int u = 1; u < AW; u++) acc += x[u] * tb->cosw[u * AW + j];
                    t[j] = x[0] + 2.0f * acc;
                }
#pragma unroll
                for (int j = 0; j < AW; j++) row[j * P2] = t[j];
            }
            __syncthreads();
            WIDE_MARK(4);
            /* ... then columns, and the filtered pixels straight out: filt[g][n][st][c][pq] -- four pixels per thread and 16-byte
             * stores where the patch area allows (the CU's store path is issue-bound: half the instructions of 8-byte stores) */
            if ((k2 & 3) == 0) {
                constexpr int P4 = SLAB / 4;
                const v4f* const S4 = reinterpret_cast<const v4f*>(S);
                for (int e = tid; e < nSx * AW * P4; e += NT) {
                    const int pp = e % P4, r = e / P4, j = r % AW, n = r / AW, px = 4 * pp;
                    const v4f* col = S4 + (size_t)(n * A + j) * P4 + pp;
                    v4f t[AW];
#pragma unroll
                    for (int v = 0; v < AW; v++) t[v] = col[v * AW * P4];
                    if (px < npx) {   /* (npx is a multiple of four here) */
#pragma unroll
                        for (int i = 0; i < AW; i++) {
                            v4f acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                            for (int v = 1; v < AW; v++) acc += t[v] * tb->cosw[v * AW + i];
                            filt_put4(reinterpret_cast<v4f*>(out + filt_patch(a, g, n, i * AW + j, k2) + (size_t)c * k2 + p0 + px), (t[0] + 2.0f * acc) * tb->coef4inv);
                        }
                    }
                }
            } else
            for (int e = tid; e < nSx * AW * P2; e += NT) {
                const int pp = e % P2, r = e / P2, j = r % AW, n = r / AW, px = 2 * pp;
                const v2f* col = S2 + (size_t)(n * A + j) * P2 + pp;
                v2f t[AW];
#pragma unroll
                for (int v = 0; v < AW; v++) t[v] = col[v * AW * P2];
                if (px < npx) {
                    const bool both = px + 1 < npx, even = (k2 & 1) == 0;   /* even patch area: every pair of `filt` is 8-byte aligned */
#pragma unroll
                    for (int i = 0; i < AW; i++) {
                        v2f acc = {0.0f, 0.0f};
#pragma unroll
                        for (int v = 1; v < AW; v++) acc += t[v] * tb->cosw[v * AW + i];
                        const v2f y = (t[0] + 2.0f * acc) * tb->coef4inv;
                        float* o = out + filt_patch(a, g, n, i * AW + j, k2) + (size_t)c * k2 + p0 + px;
                        if (both && even) filt_put2(reinterpret_cast<v2f*>(o), y);
                        else { filt_put(o, y.x); if (both) filt_put(o + 1, y.y); }
                    }
                }
            }
        } else {
            if (do_sa4) {
                for (int e = tid; e < nSx * AW * P2; e += NT) {
                    const int pp = e % P2, r = e / P2, u = r % AW, n = r / AW;
                    sadctw_cols_inv2<AW>((LdsV2)(S2 + (size_t)(n * A + u) * P2 + pp), P2, u, (SaTab)sa_tab);
                }
                __syncthreads();
                for (int e = tid; e < nSx * AW * P2; e += NT) {
                    const int pp = e % P2, r = e / P2;
                    sadctw_rows_inv2<AW>((LdsV2)(S2 + (size_t)r * AW * P2 + pp), P2, r % AW, (SaTab)sa_tab);
                }
                __syncthreads();
            }
            for (int e = tid; e < nSx * A * SLAB; e += NT) {
                const int px = e % SLAB, ns = e / SLAB;
                if (px < npx) filt_put(&out[filt_patch(a, g, ns / A, ns % A, k2) + (size_t)c * k2 + p0 + px], S[e]);
            }
        }
        WIDE_MARK(7);
        __syncthreads();
        WIDE_MARK(5);
    }
#ifdef LFBM5D_WIDE_PHASES
    if (tid == 0 && blockIdx.x % 64 == 5) {   /* a sample: same-address atomics of every workgroup would queue in L2 and slow every load */
        for (int i = 0; i < 6; i++) atomicAdd(&a.counters[4 + i], (unsigned long long)tq[i]); atomicAdd(&a.counters[12], (unsigned long long)tq[6]); atomicAdd(&a.counters[13], (unsigned long long)tq[7]); atomicAdd(&a.counters[14], (unsigned long long)tq[8]); atomicAdd(&a.counters[15], (unsigned long long)tq[9]); atomicAdd(&a.counters[10], 1ull); atomicAdd(&a.counters[11], (unsigned long long)((long long)__builtin_amdgcn_s_memrealtime() - treal0)); }
#endif
    /* group weight (core:412-421, sd_weighting_5d core:3140-3173) */
    for (int o = 32; o > 0; o >>= 1) { wacc += __shfl_xor(wacc, o); s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
    if ((tid & 63) == 0) { red[0][tid >> 6] = wacc; red[1][tid >> 6] = s1; red[2][tid >> 6] = s2; }
    __syncthreads();
    if (tid == 0) {
        float w = 0.0f, m = 0.0f, q = 0.0f;
        for (int i = 0; i < NT / 64; i++) { w += red[0][i]; m += red[1][i]; q += red[2][i]; }
        float wx;
        if (a.useSD) {
            const float Nn = (float)(nSx * A);
            const float res = (q - m * m / Nn) / (Nn - 1.0f);
            wx = res > 0.0f ? 1.0f / sqrtf(res) : 0.0f;
        } else
            wx = w > 0.0f ? (sig > 0.0f ? 1.0f / (sig * sig * w) : 1.0f / w) : 1.0f;
        if (SPLIT) { if (w != 0.0f) atomicAdd(&a.wgt[(size_t)g * a.C + c], w); }
        else a.wgt[(size_t)g * a.C + c] = wx;
        if (c == 0 && (!SPLIT || slab == 0)) {
            atomicAdd(&a.counters[0], (unsigned long long)nSx);
            if (use_sadct) atomicAdd(&a.counters[1], 1ull);
        }
    }
}

/* SPLIT form: survivor counts -> weights (core:413-421) */
__global__ void k_group_idw_weight(GroupArgs a) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n_groups * a.C) return;
    const size_t o = (size_t)a.ref_begin * a.C + i;
    const float w = a.wgt[o], sig = a.sigma[i % a.C];
    a.wgt[o] = w > 0.0f ? (sig > 0.0f ? 1.0f / (sig * sig * w) : 1.0f / w) : 1.0f;
}

} /* namespace */

/* window side -> pixels per slab (the slab's stack of up to 8 matches within 51 KB; a power of two, so that a thread's 16-byte loads
 * all fetch the same pixels) */
template <int AW> struct WideSlab { static constexpr int value = AW == 5 ? 64 : AW == 7 ? 32 : AW == 9 ? 16 : AW <= 15 ? 8 : 4; };   /* (15x15: 58 KB -- its 190
 * registers allow two workgroups per CU anyway: 23 ms per launch against 46 with four-pixel slabs; 17x17 with eight: 74 + 17 KB, one workgroup per CU, 51 against 40) */

template <int AW>
static hipError_t prepare_idw() {
    constexpr int SLAB = WideSlab<AW>::value;
    const void* fns[] = {reinterpret_cast<const void*>(&k_group_idw<AW, SLAB, false>), reinterpret_cast<const void*>(&k_group_idw<AW, SLAB, true>)};
    for (const void* f : fns) {
        const hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * AW * AW * SLAB * (int)sizeof(float));   /* N <= 8 */
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}
hipError_t prepare_group_wide() {
    hipError_t e = prepare_idw<5>();
    if (e == hipSuccess) e = prepare_idw<7>();
    if (e == hipSuccess) e = prepare_idw<9>();
    if (e == hipSuccess) e = prepare_idw<11>();
    if (e == hipSuccess) e = prepare_idw<13>();
    if (e == hipSuccess) e = prepare_idw<15>();
    if (e == hipSuccess) e = prepare_idw<17>();
    return e;
}

template <int AW>
static void launch_idw(hipStream_t s, const GroupArgs& a, bool split) {
    constexpr int SLAB = WideSlab<AW>::value;
    const unsigned gx = ((a.n_groups + 7) / 8) * 8;   /* xcd_group_index */
    const size_t lds = (size_t)a.N * AW * AW * SLAB * sizeof(float);
    if (split) hipLaunchKernelGGL((k_group_idw<AW, SLAB, true>), dim3(gx, a.C, (a.k * a.k + SLAB * wide_spw<AW>() - 1) / (SLAB * wide_spw<AW>())), dim3(256), lds, s, a);
    else       hipLaunchKernelGGL((k_group_idw<AW, SLAB, false>), dim3(gx, a.C), dim3(256), lds, s, a);
}

hipError_t launch_group_wide(hipStream_t s, const GroupArgs& a, bool* launched) {
    *launched = false;
    int aw = 0;
    for (int w = 5; w <= 17; w += 2) if (a.A == (unsigned)(w * w)) aw = w;
    if (!(aw && a.tau2 == 4 && a.step == 1 && !a.bm3d && a.N <= 8 && a.k * a.k <= 256 && (size_t)a.A * a.C * a.Wb * a.Hb * 4 < 0x7fffffffull)) return hipSuccess;
    *launched = true;
    const bool split = !(a.useSD || (a.opt & kOptWideNoSplit));
    if (split) {
        const hipError_t e = hipMemsetAsync(a.wgt + (size_t)a.ref_begin * a.C, 0, (size_t)a.n_groups * a.C * sizeof(float), s);
        if (e != hipSuccess) return e;
    }
    switch (aw) {
        case 5: launch_idw<5>(s, a, split); break;
        case 7: launch_idw<7>(s, a, split); break;
        case 9: launch_idw<9>(s, a, split); break;
        case 11: launch_idw<11>(s, a, split); break;
        case 13: launch_idw<13>(s, a, split); break;
        case 15: launch_idw<15>(s, a, split); break;
        default: launch_idw<17>(s, a, split); break;
    }
    if (split) hipLaunchKernelGGL(k_group_idw_weight, grid1d((size_t)a.n_groups * a.C), dim3(256), 0, s, a);
    return hipGetLastError();
}

} /* namespace lfbm5d */
