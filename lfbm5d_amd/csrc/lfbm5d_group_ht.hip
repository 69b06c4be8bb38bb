/*
 * lfbm5d_group_ht.hip -- dedicated group kernels of the HARD-THRESHOLDING step on 3x3 windows (core:277-481) for gfx950:
 * k_group_id_* (tau_2D = id: one thread per pixel holds the pixel of all nSx x 9 patches in registers, packed fp32) and
 * k_group_bior16_* / k_group_dct16_* (16x16 patches with a bior1.5 / DCT 2-D stage in front of and behind the same register
 * stage).  Split from lfbm5d_kernels.hip in round 5.
 */
#include "lfbm5d_group_device.h"

namespace lfbm5d {

namespace {

/* ------------------------------------------------------------------------------------------
 * Register-resident variant for tau_2D = id (the README hard-thresholding configuration): with no
 * 2-D transform, pixel pq of the group never mixes with other pixels, so one thread owns pixel pq
 * of all nSx * A patches (72 registers for N = 8), loads them straight from the window images and
 * runs the 3x3 angular DCT / SADCT, the Haar/Hadamard fibre transforms, the shrinkage and the
 * inverses without touching LDS (only the group weight is reduced through it).  No LDS stack means
 * occupancy is set by registers, not by the 72 KiB stack of k_group.
 * ------------------------------------------------------------------------------------------ */
/* One pixel of all NS * 9 patches of the group (hard-thresholding step).  Loads and stores go
 * through buffer resources with the per-patch part of the address in a scalar register (the patch
 * positions are uniform), so none of the 2 * NS * 9 memory operations needs address VGPRs; the
 * 3x3 angular DCTs run on pairs of patches (n, n + 1) with packed fp32 arithmetic. */
/* the angular transform, the 5th-dimension transform with the hard threshold and their inverses on one pixel's NS * 9 values
 * V[h][st] = {patch h, patch h + NS/2} (the register stage shared by the tau_2D = id kernel and the 16x16 kernels) */
/* SA_MODE: how the (rare) shape-adaptive transform is reached -- 0: calls (scratch vector; keeps its code out of the caller's
 * register allocation), 1: inline on sa_lds, nine floats of LDS of this thread's, 2: inline in registers (the *_sa kernels, which
 * the host launches for windows with an empty SAI, where EVERY group is shape-adaptive, and over the list of the few such groups
 * of an ordinary window), 3: none -- the kernel skips shape-adaptive groups (round 6: k_group_id_haar / _any) */
/* Round 6: the same stage for groups that take the full 3x3 DCT and Haar fibres (the README configuration; every group of a
 * window without empty SAIs except the one in a thousand whose shape is not the whole window) with EVERY normalisation constant
 * moved out of the transforms, and the reference-order form as its referee.  Measured (profiles/r06_c_*): k_group_id_haar
 * alone, without its loads and stores, took 0.55 of its 0.86 ms -- the kernel is bound by VALU cycles, and packed fp32
 * instructions take 4 cycles, not 2 (profiles/r06_a_valu_rate.txt).  The reference-order form spends 153 packed instructions
 * per pair of 3x3 blocks and 24 + 8 per pair of Haar fibres; here
 *   forward 3x3    unnormalised rows, then columns: (x0 + x2) + x1, x0 - x2, (x0 + x2) - 2 x1                  24 packed
 *   Haar           sums and differences only; a coefficient of level l carries 2^(l/2)                           6 packed + 2
 *   threshold      |c| > T / (F[st] 2^(-l/2)) =: Tq, F = alpha_v alpha_u coef_norm_4d, alpha = (2, sqrt 3, 1): the reference's
 *                  comparison (core:2431-2437), scaled; Tq comes from the host (GroupArgs::ht3_T); survivors counted
 *                  from the comparison masks (s_bcnt1) instead of a select and an add per value
 *   inverse Haar   fma by 2 and by 4 (exact); the 1 / nSx goes into the next constant                            6 packed + 2
 *   inverse 3x3    one multiplication per value by ht3_g[st] F[st] / nSx, then additions and fma by -2            33 packed
 * -- about 0.4 of the VALU cycles.  Mathematically the same numbers; in floating point a coefficient within round-off of
 * the threshold could decide differently than the reference-order form (tools/flip_count.py, profiles/r06_c_ht_fast_flips.txt:
 * 1, 1, 1 decisions of 864 M per 560 x 560 pass against the CPU oracle, where the reference-order form has 0, 0, 1).  So the
 * chain also watches its distance to the threshold: a wave in which ANY coefficient comes within a guard band of its threshold
 * (2^-16 ... 2^-18 of it, ht_guard) reports it, and its (group, channel) is redone in the reference-order form by the list launch behind the kernel
 * (one to two in a hundred on natural data, whose coefficients are far denser around the threshold than noise alone).  The two forms differ by float round-off of sums of at most 72 pixel values -- below 1e-5 of
 * a threshold for 8-bit-range data, a tenth of the guard band -- so every decision is the reference-order form's decision:
 * the survivor counts, hence the weights and `den`, are bit-identical to rounds 1-5; the filtered values agree to an ulp or two.
 * Returns true when the caller has to fall back.  LFBM5D_HT_REFERENCE_ORDER (build flag): no fast chain at all. */
/* guard band, relative to the threshold, by angular frequency: emulated over 3 M near-threshold coefficients of 8-bit-range data
 * (brightness 0..255, contrast up to +-100, sigma 25) the two forms differ by at most 7.9e-6 of a threshold at st = 0 (sums of
 * 72 bright pixels), 2.3e-6 at st = 3, 6 and 1.2e-6 elsewhere */
__device__ __forceinline__ constexpr float ht_guard(int st) { return st == 0 ? 1.0f / 65536.0f : (st == 3 || st == 6) ? 1.0f / 131072.0f : 1.0f / 262144.0f; }
template <int NS> __device__ __forceinline__ void haar_fwd_pairs_u(v2f* P) {
    if (NS == 8) {
        const v2f S01 = P[0] + P[1], D01 = P[0] - P[1], S23 = P[2] + P[3], D23 = P[2] - P[3];
        const v2f SS = S01 + S23, DD = S01 - S23;
        P[0] = v2f{SS.x + SS.y, SS.x - SS.y}; P[1] = DD; P[2] = D01; P[3] = D23;
    } else if (NS == 4) {
        const v2f S = P[0] + P[1], D = P[0] - P[1];
        P[0] = v2f{S.x + S.y, S.x - S.y}; P[1] = D;
    } else if (NS == 2) {
        P[0] = v2f{P[0].x + P[0].y, P[0].x - P[0].y};
    }
}
template <int NS> __device__ __forceinline__ void haar_inv_pairs_u(v2f* P) {   /* nSx times the orthonormal inverse of the orthonormal coefficients */
    if (NS == 8) {
        const v2f X = v2f{P[0].x + P[0].y, P[0].x - P[0].y};
        const v2f U = P[1] * 2.0f + X, V = X - P[1] * 2.0f, D01 = P[2], D23 = P[3];
        P[0] = D01 * 4.0f + U; P[1] = U - D01 * 4.0f; P[2] = D23 * 4.0f + V; P[3] = V - D23 * 4.0f;
    } else if (NS == 4) {
        const v2f X = v2f{P[0].x + P[0].y, P[0].x - P[0].y}, D = P[1];
        P[0] = D * 2.0f + X; P[1] = X - D * 2.0f;
    } else if (NS == 2) {
        P[0] = v2f{P[0].x + P[0].y, P[0].x - P[0].y};
    }
}
template <int NS>
__device__ __forceinline__ bool group_id_compute_fast(const GroupArgs& a, int c, v2f (&V)[NS > 1 ? NS / 2 : 1][9], float& wacc, float& s1, float& s2) {
    constexpr int NH = NS > 1 ? NS / 2 : 1;
    constexpr int LV = NS == 8 ? 3 : NS == 4 ? 2 : NS == 2 ? 1 : 0;       /* Haar levels; the pair P[h] holds coefficients of level lvl(h) */
    const TbPtr tb = (TbPtr)a.tb;
    /* forward 3x3, unnormalised: along u (x[s*3 + u]), then along s */
#pragma unroll
    for (int h = 0; h < NH; h++) {
        v2f t[9];
#pragma unroll
        for (int s = 0; s < 3; s++) {
            const v2f p = V[h][s * 3] + V[h][s * 3 + 2];
            t[s * 3] = p + V[h][s * 3 + 1]; t[s * 3 + 1] = V[h][s * 3] - V[h][s * 3 + 2]; t[s * 3 + 2] = p - 2.0f * V[h][s * 3 + 1];
        }
#pragma unroll
        for (int u = 0; u < 3; u++) {
            const v2f p = t[u] + t[6 + u];
            V[h][u] = p + t[3 + u]; V[h][3 + u] = t[u] - t[6 + u]; V[h][6 + u] = p - 2.0f * t[3 + u];
        }
    }
    const float inv_n = 1.0f / (float)NS;
    unsigned kept = 0;                 /* survivors of this WAVE (uniform) */
    unsigned long long near = 0ull;    /* lanes with a coefficient inside the guard band (uniform) */
#pragma unroll
    for (int st = 0; st < 9; st++) {
        v2f P[NH];
#pragma unroll
        for (int h = 0; h < NH; h++) P[h] = V[h][st];
        haar_fwd_pairs_u<NS>(P);
#pragma unroll
        for (int h = 0; h < NH; h++) {
            const int lvl = h == 0 ? LV : (NS == 8 && h == 1) ? 2 : 1;
            const float Tq = a.ht3_T[c][st][lvl], Gq = Tq * ht_guard(st);
            const float ax = fabsf(P[h].x), ay = fabsf(P[h].y);
            const bool kx = ax > Tq, ky = (NS > 1) && ay > Tq;
            near |= __ballot(fabsf(ax - Tq) < Gq);
            if (NS > 1) near |= __ballot(fabsf(ay - Tq) < Gq);
            kept += (unsigned)__popcll(__ballot(kx)) + (NS > 1 ? (unsigned)__popcll(__ballot(ky)) : 0u);
            P[h].x = kx ? P[h].x : 0.0f;
            P[h].y = ky ? P[h].y : 0.0f;
        }
        haar_inv_pairs_u<NS>(P);
#pragma unroll
        for (int h = 0; h < NH; h++) V[h][st] = P[h];
    }
    if (near) return true;
    wacc += (__lane_id() == 0) ? (float)kept : 0.0f;   /* (the caller sums wacc over the lanes) */
    if (a.useSD) {   /* sd_weighting_5d on the filtered 4-D coefficients (core:3140-3173): back to their normalised scale */
#pragma unroll
        for (int h = 0; h < NH; h++)
#pragma unroll
            for (int st = 0; st < 9; st++) {
                const v2f y = V[h][st] * (tb->ht3_f[st] * inv_n);
                s1 += y.x; s2 += y.x * y.x;
                if (NS > 1) { s1 += y.y; s2 += y.y * y.y; }
            }
    }
    /* inverse 3x3: Z = g f / nSx * value, then along u, then along s */
#pragma unroll
    for (int h = 0; h < NH; h++) {
        v2f t[9];
#pragma unroll
        for (int s = 0; s < 3; s++) {
            const v2f Z0 = V[h][s * 3] * (tb->ht3_gf[s * 3] * inv_n), Z1 = V[h][s * 3 + 1] * (tb->ht3_gf[s * 3 + 1] * inv_n), Z2 = V[h][s * 3 + 2] * (tb->ht3_gf[s * 3 + 2] * inv_n);
            const v2f p = Z0 + Z2;
            t[s * 3] = p + Z1; t[s * 3 + 1] = Z0 - 2.0f * Z2; t[s * 3 + 2] = p - Z1;
        }
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const v2f p = t[j] + t[6 + j];
            V[h][j] = p + t[3 + j]; V[h][3 + j] = t[j] - 2.0f * t[6 + j]; V[h][6 + j] = p - t[3 + j];
        }
    }
    return false;
}

template <int NS, bool HAAR, int SA_MODE = 0>
__device__ __forceinline__ void group_id_compute(const GroupArgs& a, int c, ShRef sh, bool use_sadct, v2f (&V)[NS > 1 ? NS / 2 : 1][9],
                                                 float& wacc, float& s1, float& s2, float* sa_lds = nullptr) {
    constexpr int NH = NS > 1 ? NS / 2 : 1;
    const TbPtr tb = (TbPtr)a.tb;
    const bool do_dct4 = a.tau4 == 5 || (a.tau4 == 6 && !use_sadct);
    const bool do_sa4 = SA_MODE != 3 && !do_dct4 && a.tau4 == 6;   /* SA_MODE 3: the caller never hands over a shape-adaptive group */
    auto sadct_pairs = [&](bool fwd) {   /* rare: shape-adaptive transform on the scalar path, staged through t9 */
#pragma unroll
        for (int h = 0; h < NH; h++)
#pragma unroll
            for (int half = 0; half < (NS > 1 ? 2 : 1); half++) {
                if (SA_MODE == 1) {
#pragma unroll
                    for (int i = 0; i < 9; i++) sa_lds[i] = half ? V[h][i].y : V[h][i].x;
                    if (fwd) sadct9_fwd_lds(sa_lds, sh, tb); else sadct9_inv_lds(sa_lds, sh, tb);
#pragma unroll
                    for (int i = 0; i < 9; i++) { if (half) V[h][i].y = sa_lds[i]; else V[h][i].x = sa_lds[i]; }
                } else {
                    float t9[9];
#pragma unroll
                    for (int i = 0; i < 9; i++) t9[i] = half ? V[h][i].y : V[h][i].x;
                    if (SA_MODE == 2) { if (fwd) sadct9_fwd_sel(t9, sh, tb); else sadct9_inv_sel(t9, sh, tb); }
                    else if (fwd) sadct9_fwd(t9, sh, tb); else sadct9_inv(t9, sh, tb);
#pragma unroll
                    for (int i = 0; i < 9; i++) { if (half) V[h][i].y = t9[i]; else V[h][i].x = t9[i]; }
                }
            }
    };
    if (do_dct4) {
#pragma unroll
        for (int h = 0; h < NH; h++) dct9_fwd2(V[h], tb);
    } else if (do_sa4) sadct_pairs(true);
    const float sig = a.sigma[c];
    const float T = a.lambda * sig * 1.41421356237309505f;
#pragma unroll
    for (int st = 0; st < 9; st++) {
        const bool in_shape = !use_sadct || sh.mask_dct[st];
        if (HAAR) {
            v2f P[NH];
#pragma unroll
            for (int h = 0; h < NH; h++) P[h] = V[h][st];
            haar_fwd_pairs<NS>(P);
            if (in_shape) {
#pragma unroll
                for (int h = 0; h < NH; h++) {
                    const bool kx = fabsf(P[h].x) > T, ky = (NS > 1) && fabsf(P[h].y) > T;
                    wacc += (kx ? 1.0f : 0.0f) + (ky ? 1.0f : 0.0f);
                    P[h].x = kx ? P[h].x : 0.0f;
                    P[h].y = ky ? P[h].y : 0.0f;
                }
            }
            haar_inv_pairs<NS>(P);
#pragma unroll
            for (int h = 0; h < NH; h++) V[h][st] = P[h];
        } else {
            float o[NS], e[1] = {0.0f};
#pragma unroll
            for (int n = 0; n < NS; n++) o[n] = n < NH ? V[n][st].x : V[n - NH][st].y;
            shrink_fibre<NS, 1>(o, e, a.tau5, T, sig * sig, in_shape, wacc, tb);
#pragma unroll
            for (int n = 0; n < NS; n++) { if (n < NH) V[n][st].x = o[n]; else V[n - NH][st].y = o[n]; }
        }
    }
    if (a.useSD) {
#pragma unroll
        for (int h = 0; h < NH; h++)
#pragma unroll
            for (int st = 0; st < 9; st++) {
                s1 += V[h][st].x; s2 += V[h][st].x * V[h][st].x;
                if (NS > 1) { s1 += V[h][st].y; s2 += V[h][st].y * V[h][st].y; }
            }
    }
    if (do_dct4) {
#pragma unroll
        for (int h = 0; h < NH; h++) dct9_inv2(V[h], tb);
    } else if (do_sa4) sadct_pairs(false);
}

/* FAST: the unnormalised chain (group_id_compute_fast) instead of the reference-order form; returns true when the wave came within
 * the guard band of a threshold, i.e. when its results must not be used (the caller lists the group for the reference-order launch) */
template <int NS, bool HAAR, bool LDSW = false, int SA_MODE = 0, bool FAST = false>   /* LDSW: values come from / go back to an LDS work area [patch][k][k+1] (2-D transformed patches) */
__device__ __forceinline__ bool group_id_body(const GroupArgs& a, unsigned g, int c, int pq, const __attribute__((address_space(4))) unsigned* pos,
                                              ShRef sh, bool use_sadct, float& wacc, float& s1, float& s2, float* work = nullptr) {
    const int k = a.k, k2 = k * k, A = 9;
    const unsigned plane = a.Wb * a.Hb;
    const unsigned kRsrcFlags = 0x00020000u;
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)a.noisy, 0, (int)((size_t)a.A * a.C * plane * 4), kRsrcFlags);
    float* const out = a.filt + (size_t)g * a.N * A * a.C * k2;
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc((void*)out, 0, (int)((size_t)a.N * A * a.C * k2 * 4), kRsrcFlags);
    /* the pixel's NS * 9 values as pairs of patches: V[h][st] = {patch h, patch h + NS/2} (NS = 1: .y unused) */
    constexpr int NH = NS > 1 ? NS / 2 : 1;
    v2f V[NH][9];
    const int voff = (int)(((unsigned)(pq / k) * a.Wb + pq % k) * 4u);
    const int woff = (pq / k) * (k + 1) + pq % k;      /* this pixel inside a work-area patch */
    unsigned okbits[NS];
    typedef const __attribute__((address_space(4))) unsigned* cuptr_;
    const cuptr_ ofs = (cuptr_)(a.gofs + (size_t)g * a.N * A), ok = (cuptr_)(a.gok + (size_t)g * a.N);
    const unsigned cbase = (unsigned)c * plane * 4u;
    auto load_all = [&]() {
        if (LDSW) {
#pragma unroll
            for (int n = 0; n < NS; n++)
#pragma unroll
                for (int st = 0; st < 9; st++) {
                    const float x = work[(n * A + st) * kT16Patch + woff];
                    if (n < NH) V[n][st].x = x; else V[n - NH][st].y = x;
                    okbits[n] = 0x1ffu;
                }
        } else
#pragma unroll
        for (int n = 0; n < NS; n++) {
            okbits[n] = ok[n];                            /* uniform: scalar loads (pre-pass: k_group_pos) */
#pragma unroll
            for (int st = 0; st < 9; st++) {
                const unsigned so = ofs[n * A + st] + cbase;   /* absent patches read offset 0 and are zeroed below */
#if defined(LFBM5D_HT_EXP) && (LFBM5D_HT_EXP & 1)   /* timing experiment: one gather per thread instead of NS * 9 */
                const float x = (n | st) ? V[0][0].x * 1.0001f + (float)(n + st) : __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_in, voff, (int)so, 0));
#else
                const float x = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_in, voff, (int)so, 0));
#endif
                if (n < NH) V[n][st].x = x; else V[n - NH][st].y = x;
            }
        }
        if (NS == 1) {
#pragma unroll
            for (int st = 0; st < 9; st++) V[0][st].y = 0.0f;
        }
#pragma unroll
        for (int n = 0; n < NS; n++)
            if (okbits[n] != 0x1ffu) {   /* uniform, rare: patches of empty SAIs / never-filled table column read as zeros */
#pragma unroll
                for (int st = 0; st < 9; st++) {
                    if (n < NH) V[n][st].x = ((okbits[n] >> st) & 1) ? V[n][st].x : 0.0f;
                    else V[n - NH][st].y = ((okbits[n] >> st) & 1) ? V[n - NH][st].y : 0.0f;
                }
            }
    };
    load_all();
    bool near = false;
    if (FAST) near = group_id_compute_fast<NS>(a, c, V, wacc, s1, s2);   /* true: the wave's results are not to be used (the caller lists the group) */
    else
    group_id_compute<NS, HAAR, SA_MODE>(a, c, sh, use_sadct, V, wacc, s1, s2);
    const int vout = pq * 4;
#pragma unroll
    for (int n = 0; n < NS; n++)
#pragma unroll
        for (int st = 0; st < 9; st++) {
            const float r = n < NH ? V[n][st].x : V[n - NH][st].y;
            if (LDSW) work[(n * A + st) * kT16Patch + woff] = r;
#if defined(LFBM5D_HT_EXP) && (LFBM5D_HT_EXP & 2)   /* timing experiment: the stores are issued at an out-of-range offset (dropped by the buffer's bounds check) */
            else __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, r), rs_out, (n | st) ? 0x7ffffff0 : vout, (int)((((unsigned)(n * A + st) * a.C + c) * k2) * 4u), LFBM5D_FILT_STORE_AUX);
#else
            else __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, r), rs_out, vout, (int)((((unsigned)(n * A + st) * a.C + c) * k2) * 4u), LFBM5D_FILT_STORE_AUX);
#endif
        }
    return near;
}

/* SA_MODE 3 (the kernels of ordinary windows): groups whose shape is not the whole window are left to the list launch;
 * SA_MODE 2: every group handed over goes through the inline shape-adaptive form where its shape asks for it */
template <bool HAAR, int SA_MODE, bool FAST = false>
__device__ __forceinline__ void group_id_one(const GroupArgs& a, const unsigned g, const int c, float (*red)[4]) {
    const int tid = threadIdx.x;
    const int A = 9, N = a.N;
    const int nSx = (int)a.self_cnt[g];
    /* positions are uniform per workgroup and constant during this kernel: constant address space -> scalar loads */
    typedef const __attribute__((address_space(4))) unsigned* cuptr;
    const cuptr pos = (cuptr)(a.gpos + (size_t)g * N * A);
    ShRef sh = group_shape(a, g);
    const bool use_sadct = a.tau4 == 6 && sh.use_sadct;
    if (SA_MODE == 3 && use_sadct) return;   /* (uniform) */
    float wacc = 0.0f, s1 = 0.0f, s2 = 0.0f;
    bool near = false;   /* FAST: wave-uniform */
    if (tid < (int)(a.k * a.k)) {
        switch (nSx) {
            case 1:  near = group_id_body<1, HAAR, false, SA_MODE, FAST>(a, g, c, tid, pos, sh, use_sadct, wacc, s1, s2); break;
            case 2:  near = group_id_body<2, HAAR, false, SA_MODE, FAST>(a, g, c, tid, pos, sh, use_sadct, wacc, s1, s2); break;
            case 4:  near = group_id_body<4, HAAR, false, SA_MODE, FAST>(a, g, c, tid, pos, sh, use_sadct, wacc, s1, s2); break;
            default: near = group_id_body<8, HAAR, false, SA_MODE, FAST>(a, g, c, tid, pos, sh, use_sadct, wacc, s1, s2); break;
        }
    }
    for (int o = 32; o > 0; o >>= 1) { wacc += __shfl_xor(wacc, o); s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
    if ((tid & 63) == 0) { red[0][tid >> 6] = wacc; red[1][tid >> 6] = s1; red[2][tid >> 6] = s2; if (FAST) red[3][tid >> 6] = near ? 1.0f : 0.0f; }
    __syncthreads();
    if (tid == 0) {
        if (FAST) {   /* a wave came within the guard band of a threshold: the (group, channel) goes on the list; the reference-order launch overwrites what was stored here */
            float nr = 0.0f;
            for (unsigned i = 0; i < (blockDim.x + 63) / 64; i++) nr += red[3][i];
            if (nr != 0.0f) {
#ifdef LFBM5D_HT_COUNT_NEAR   /* development: the guard-band cases show up as "shape-adaptive groups" in the pass statistics */
                atomicAdd(&a.counters[1], 1ull);
#endif
                a.sa_list[1u + atomicAdd(&a.sa_list[0], 1u)] = g | (1u << (29 + c)); return;
            }
        }
        float w = 0.0f, m = 0.0f, q = 0.0f;
        for (unsigned i = 0; i < (blockDim.x + 63) / 64; i++) { w += red[0][i]; m += red[1][i]; q += red[2][i]; }
        float wx;
        if (a.useSD) {
            const float Nn = (float)(nSx * A);
            const float res = (q - m * m / Nn) / (Nn - 1.0f);
            wx = res > 0.0f ? 1.0f / sqrtf(res) : 0.0f;
        } else {
            const float sig = a.sigma[c];
            wx = w > 0.0f ? (sig > 0.0f ? 1.0f / (sig * sig * w) : 1.0f / w) : 1.0f;
        }
        a.wgt[(size_t)g * a.C + c] = wx;
        if (c == 0) {
            atomicAdd(&a.counters[0], (unsigned long long)nSx);
            if (use_sadct) atomicAdd(&a.counters[1], 1ull);
        }
    }
}

/* Ordinary windows (round 6).  The kernels carry no shape-adaptive code -- as a call it cost every group 74 VGPRs (values live
 * across a call sit in the sparse callee-saved registers), inline a second copy of the register stage -- and k_group_id_haar_fast
 * no reference-order code either (inline it doubled the registers; as a call the argument block went through scratch memory: 3 ms):
 * what they cannot finish goes on a LIST (GroupArgs::sa_list: group number | channel mask << 29) that k_group_id_*_list
 * (reference-order arithmetic, inline shape-adaptive form; a workgroup per entry) works off behind them -- the one group in a
 * thousand whose angular shape is not the whole window (all channels; listed by k_group_shape), and the (group, channel) pairs
 * in which a wave of the fast chain came within the guard band of a threshold (one to two in a hundred). */
#ifndef LFBM5D_HT_WAVES
#define LFBM5D_HT_WAVES 0
#endif
template <bool HAAR, int SA_MODE, bool FAST>
__device__ __forceinline__ void group_id_kernel(const GroupArgs& a) {
    __shared__ float red[4][4];
    const unsigned gi = xcd_group_index(a);
    if (gi >= a.n_groups) return;
    group_id_one<HAAR, SA_MODE, FAST>(a, a.ref_begin + gi, (int)blockIdx.y, red);
}
template <bool HAAR>
__device__ __forceinline__ void group_id_list_kernel(const GroupArgs& a) {
    __shared__ float red[4][4];
    const unsigned n = a.sa_list[0];
    for (unsigned i = blockIdx.x; i < n; i += gridDim.x) {
        const unsigned e = a.sa_list[1u + i];
        for (unsigned c = 0; c < a.C; c++)
            if ((e >> (29 + c)) & 1u) {
                group_id_one<HAAR, 2>(a, e & 0x1fffffffu, (int)c, red);
                __syncthreads();   /* red is reused */
            }
    }
}
#if LFBM5D_HT_WAVES > 0
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(LFBM5D_HT_WAVES, LFBM5D_HT_WAVES))) void k_group_id_haar_fast(GroupArgs a) { group_id_kernel<true, 3, true>(a); }
#else
__global__ __launch_bounds__(256) void k_group_id_haar_fast(GroupArgs a) { group_id_kernel<true, 3, true>(a); }
#endif
__global__ __launch_bounds__(256) void k_group_id_haar(GroupArgs a) { group_id_kernel<true, 3, false>(a); }   /* tau_4D = id (no angular DCT to speed up) and -DLFBM5D_HT_REFERENCE_ORDER builds */
__global__ __launch_bounds__(256) void k_group_id_any(GroupArgs a) { group_id_kernel<false, 3, false>(a); }
/* windows with an empty SAI (every group shape-adaptive): the transform inline, in registers; 168 VGPRs: three waves per SIMD */
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_group_id_haar_sa(GroupArgs a) { group_id_kernel<true, 2, false>(a); }
__global__ __launch_bounds__(256) void k_group_id_any_sa(GroupArgs a) { group_id_kernel<false, 2, false>(a); }
/* ... and the listed groups of an ordinary window */
constexpr unsigned kSaListBlocks = 2048;   /* an entry per workgroup while the list is shorter: the launch lasts one group's latency */
__global__ __launch_bounds__(256) void k_group_id_haar_list(GroupArgs a) { group_id_list_kernel<true>(a); }
__global__ __launch_bounds__(256) void k_group_id_any_list(GroupArgs a) { group_id_list_kernel<false>(a); }

/* ------------------------------------------------------------------------------------------
 * Hard-thresholding step with tau_2D = bior1.5 or dct and 16x16 patches (BASELINE configurations 2, 4 and 5):
 * the register-resident kernel above with a 2-D stage in front and behind it.  All nSx * A patches of the group
 * go through an LDS work area [patch][16][17]: the first pass of the forward transform reads whole patch rows from
 * the window images, the per-pixel threads run the angular and 5th-dimension stages on the work area
 * (group_id_body), the last pass of the inverse transform stores the filtered rows.  bior1.5: two rows / columns
 * per thread as packed pairs (bior_taps2), levels 8, 4, 2 with the patches re-dealt to fewer threads each; DCT: 16
 * threads per patch (dct16_fwd / dct16_inv).
 * ------------------------------------------------------------------------------------------ */
/* MULTI (N = 1, BASELINE configuration 5): a group is nine patches, so a workgroup takes kT16Groups consecutive groups
 * through the 2-D stages together (their patches are contiguous in gpos and filt) and runs the per-pixel stage once
 * per group.  Measured at 560^2: 0.90 ms with one group per workgroup, 0.34 / 0.33 / 0.36 / 0.38 / 0.46 / 0.47 / 0.71 ms
 * with 2 / 3 / 4 / 5 / 7 / 8 / 14 -- three groups fill one round of the 16x16 level (216 of 256 threads) and leave
 * room for five workgroups per CU. */
#ifndef LFBM5D_T16_GROUPS
#define LFBM5D_T16_GROUPS 3
#endif
constexpr int kT16Groups = LFBM5D_T16_GROUPS;
#ifndef LFBM5D_T16_ROUND
#define LFBM5D_T16_ROUND 40
#endif
#ifndef LFBM5D_T16_WAVES
#define LFBM5D_T16_WAVES 3
#endif
constexpr int kT16Half = LFBM5D_T16_ROUND;   /* patches per round of 2-D transforms: the work area of the Haar kernels (40: two rounds for a full group of 72, three workgroups per CU) */
template <bool HAAR, bool BIOR, bool MULTI, bool SPLIT = false, bool SA = false>   /* SA: windows with an empty SAI -- the shape-adaptive transform inline, in registers (N = 1 form) */
__device__ __forceinline__ void group_t16_kernel(const GroupArgs& a) {
    extern __shared__ float lds[];
    __shared__ float red[MULTI ? kT16Groups : 1][3][4];
    constexpr int K = 16, RS = K + 1, PSZ = kT16Patch, A = 9;
    const int tid = threadIdx.x;
    const unsigned g = a.ref_begin + blockIdx.x * (MULTI ? kT16Groups : 1);     /* first group of the workgroup */
    const int ngr = MULTI ? (int)min((unsigned)kT16Groups, a.ref_begin + a.n_groups - g) : 1;
    const int c = blockIdx.y;
    const int N = a.N;
    const int nSx = MULTI ? 1 : (int)a.self_cnt[g], NP = MULTI ? ngr * A : nSx * A;
    typedef const __attribute__((address_space(4))) unsigned* cuptr;
    const cuptr pos = (cuptr)(a.gpos + (size_t)g * N * A);
    ShRef sh = group_shape(a, g);
    const bool use_sadct = a.tau4 == 6 && sh.use_sadct;
    const TbPtr tb = (TbPtr)a.tb;
    float* work = lds;
#ifdef LFBM5D_PHASE_TIMING
    long long tq[6]; int tqi = 0;
#define T16_MARK() do { if (tid == 0) tq[tqi] = (long long)__builtin_readcyclecounter(); tqi++; } while (0)
    T16_MARK();
#else
#define T16_MARK() do {} while (0)
#endif
    /* forward 2-D transform; its first pass (the rows of the 16x16 level) takes the patches straight from the window
     * images: a thread loads whole 64-byte patch rows (four 16-byte loads at 4-byte alignment), transforms them and
     * parks the result in the work area -- no separate gather, and 8 (bior) / 4 (DCT) loads per thread and round where
     * a thread-per-pixel gather issues one 4-byte load per patch */
    const size_t plane = (size_t)a.Wb * a.Hb;
    auto patch_src = [&](int patch, bool& ok) -> const float* {
        const unsigned p = a.gpos[(size_t)g * N * A + patch];
        ok = p != 0xffffffffu;            /* empty SAI / never-filled table column: zeros */
#if defined(LFBM5D_T16_EXP) && LFBM5D_T16_EXP == 1   /* timing experiment (results garbage): every patch row from one cached place */
        return a.noisy + (size_t)c * plane + (patch & 7) * 16;
#endif
        return a.noisy + ((size_t)(patch % A) * a.C + c) * plane + (ok ? p : 0u);
    };
    /* patches base .. base + np - 1 of the group -> work area slots 0 .. np - 1 */
    auto fwd2d = [&](const int base, const int np) {
        if (BIOR) {
            constexpr int TPP = 8, PPI = kThreads / TPP;   /* rows r and r + 8 per thread */
            const int slot = tid / TPP, r = tid % TPP;
            for (int p0 = 0; p0 < np; p0 += PPI) {
                const int patch = p0 + slot;
                if (patch < np) {
                    bool ok;
                    const float* src = patch_src(base + patch, ok);
                    v2f v[K], o[K];
    #pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const f4u lo = *reinterpret_cast<const f4u*>(src + (size_t)r * a.Wb + 4 * q);
                        const f4u hi = *reinterpret_cast<const f4u*>(src + (size_t)(r + 8) * a.Wb + 4 * q);
    #pragma unroll
                        for (int e = 0; e < 4; e++) v[4 * q + e] = ok ? v2f{lo.v[e], hi.v[e]} : v2f{0.0f, 0.0f};
                    }
                    bior_taps2<K, true>(v, o, tb);
                    float* Tp = work + patch * PSZ;
    #pragma unroll
                    for (int cc = 0; cc < K; cc++) { Tp[r * RS + cc] = o[cc].x; Tp[(r + 8) * RS + cc] = o[cc].y; }
                    __builtin_amdgcn_wave_barrier();
                    bior16_pass2<K, true, false>(Tp, r, tb);
                }
            }
            __syncthreads();
            bior16_level_all<8, true>(work, np, tid, tb);
            bior16_level_all<4, true>(work, np, tid, tb);
            bior16_level_all<2, true>(work, np, tid, tb);
        } else {
            const int slot = tid / K, r = tid % K;         /* DCT: 16 threads per patch, thread = row, then column */
            for (int p0 = 0; p0 < np; p0 += kThreads / K) {
                const int patch = p0 + slot;
                if (patch < np) {
                    bool ok;
                    const float* src = patch_src(base + patch, ok);
                    float x[K];
    #pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const f4u t4 = *reinterpret_cast<const f4u*>(src + (size_t)r * a.Wb + 4 * q);
    #pragma unroll
                        for (int e = 0; e < 4; e++) x[4 * q + e] = ok ? t4.v[e] : 0.0f;
                    }
                    dct16_fwd(x);
                    float* Tp = work + patch * PSZ;
    #pragma unroll
                    for (int cc = 0; cc < K; cc++) Tp[r * RS + cc] = x[cc];
                    __builtin_amdgcn_wave_barrier();
    #pragma unroll
                    for (int i = 0; i < K; i++) x[i] = Tp[i * RS + r];
                    dct16_fwd(x);
    #pragma unroll
                    for (int i = 0; i < K; i++) Tp[i * RS + r] = x[i];
                }
            }
            __syncthreads();
        }
    };
    /* inverse 2-D transform; its last pass (the rows of the 16x16 level) stores the filtered patches: filt[g][n][st][c][256] */
    float* const out = a.filt + (size_t)g * N * A * a.C * K * K;
    auto inv2d = [&](const int base, const int np) {
        if (BIOR) {
            bior16_level_all<2, false>(work, np, tid, tb);
            bior16_level_all<4, false>(work, np, tid, tb);
            bior16_level_all<8, false>(work, np, tid, tb);
            constexpr int TPP = 8, PPI = kThreads / TPP;
            const int slot = tid / TPP, r = tid % TPP;
            for (int p0 = 0; p0 < np; p0 += PPI) {
                const int patch = p0 + slot;
                if (patch < np) {
                    float* Tp = work + patch * PSZ;
                    bior16_pass2<K, false, false>(Tp, r, tb);
                    __builtin_amdgcn_wave_barrier();
                    v2f v[K], o[K];
    #pragma unroll
                    for (int cc = 0; cc < K; cc++) v[cc] = v2f{Tp[r * RS + cc], Tp[(r + 8) * RS + cc]};
                    bior_taps2<K, false>(v, o, tb);
                    float4* dst = reinterpret_cast<float4*>(out + ((size_t)(base + patch) * a.C + c) * K * K);
    #pragma unroll
                    for (int q = 0; q < 4; q++) {
                        filt_put4(&dst[r * 4 + q], make_float4(o[4 * q].x, o[4 * q + 1].x, o[4 * q + 2].x, o[4 * q + 3].x));
                        filt_put4(&dst[(r + 8) * 4 + q], make_float4(o[4 * q].y, o[4 * q + 1].y, o[4 * q + 2].y, o[4 * q + 3].y));
                    }
                }
            }
        } else {
            const int slot = tid / K, r = tid % K;
            for (int p0 = 0; p0 < np; p0 += kThreads / K) {
                const int patch = p0 + slot;
                if (patch < np) {
                    float* Tp = work + patch * PSZ;
                    float x[K];
    #pragma unroll
                    for (int i = 0; i < K; i++) x[i] = Tp[i * RS + r];
                    dct16_inv(x);
    #pragma unroll
                    for (int i = 0; i < K; i++) Tp[i * RS + r] = x[i];
                    __builtin_amdgcn_wave_barrier();
    #pragma unroll
                    for (int cc = 0; cc < K; cc++) x[cc] = Tp[r * RS + cc];
                    dct16_inv(x);
                    float4* dst = reinterpret_cast<float4*>(out + ((size_t)(base + patch) * a.C + c) * K * K);
    #pragma unroll
                    for (int q = 0; q < 4; q++) filt_put4(&dst[r * 4 + q], make_float4(x[4 * q], x[4 * q + 1], x[4 * q + 2], x[4 * q + 3]));
                }
            }
        }
    };
    float wacc[MULTI ? kT16Groups : 1], s1[MULTI ? kT16Groups : 1], s2[MULTI ? kT16Groups : 1];
    const bool split = SPLIT && !MULTI;
    if (!split) { fwd2d(0, NP); T16_MARK(); }
    if (MULTI) {
#pragma unroll
        for (int gi = 0; gi < kT16Groups; gi++) {
            wacc[gi] = 0.0f; s1[gi] = 0.0f; s2[gi] = 0.0f;
            if (gi < ngr) {
                ShRef shg = group_shape(a, g + gi);
                group_id_body<1, HAAR, true, SA ? 2 : 0>(a, g + gi, c, tid, pos + gi * A, shg, a.tau4 == 6 && shg.use_sadct, wacc[gi], s1[gi], s2[gi],
                                                         work + gi * A * PSZ);
            }
        }
    } else if (SPLIT) {
        /* Round 4: a work area of kT16Half patches instead of the whole group's (three workgroups per CU instead of two).  A thread
         * collects its pixel's coefficients in registers from the forward transforms -- one round of them for groups of up to four
         * matches, two for the full group --, runs the register stage of the tau_2D = id kernel on them, and the results go back
         * through the same area for the inverse transforms.  Between the two the area is free: the shape-adaptive transform of the
         * rare groups that need it runs inline on nine floats of it per thread (as a call it costs every group 44 VGPRs). */
        wacc[0] = 0.0f; s1[0] = 0.0f; s2[0] = 0.0f;
        const int woff = (tid / K) * RS + tid % K;
        auto rounds = [&](auto ns_tag) {
            constexpr int NS = decltype(ns_tag)::value, NH = NS > 1 ? NS / 2 : 1, NPc = NS * A, PR = kT16Half, NR = (NPc + PR - 1) / PR, F0 = NPc - (NR - 1) * PR;
            v2f V[NH][9];
            auto put = [&](const int pch, const float x) { if (pch / 9 < NH) V[pch / 9][pch % 9].x = x; else V[pch / 9 - NH][pch % 9].y = x; };
            auto get = [&](const int pch) { return pch / 9 < NH ? V[pch / 9][pch % 9].x : V[pch / 9 - NH][pch % 9].y; };
            if (NS == 1) {
#pragma unroll
                for (int st = 0; st < 9; st++) V[0][st].y = 0.0f;
            }
#pragma unroll
            for (int r = 0; r < NR; r++) {   /* the first round is the short one: fewest coefficients in registers while the transforms of the others run */
                const int b0 = r == 0 ? 0 : F0 + (r - 1) * PR, cnt = r == 0 ? F0 : PR;
                fwd2d(b0, cnt);
#pragma unroll
                for (int q = 0; q < PR; q++) if (q < cnt) put(b0 + q, work[q * PSZ + woff]);
                __syncthreads();
            }
            T16_MARK();
            /* the shape-adaptive transform inline: in registers in the wavelet kernel (4.2 -> 3.0 ms per pass of shape-adaptive groups), on LDS
             * scratch in the DCT kernel, whose register allocation the register form upsets (69 spills, +20 % on ordinary groups) */
            group_id_compute<NS, HAAR, BIOR ? 2 : 1>(a, c, sh, use_sadct, V, wacc[0], s1[0], s2[0], work + tid * 9);
            __syncthreads();
            T16_MARK();
#pragma unroll
            for (int r = NR - 1; r >= 0; r--) {
                const int b0 = r == 0 ? 0 : F0 + (r - 1) * PR, cnt = r == 0 ? F0 : PR;
#pragma unroll
                for (int q = 0; q < PR; q++) if (q < cnt) work[q * PSZ + woff] = get(b0 + q);
                __syncthreads();
                inv2d(b0, cnt);
                if (r > 0) __syncthreads();
            }
        };
        switch (nSx) {
            case 1:  rounds(std::integral_constant<int, 1>{}); break;
            case 2:  rounds(std::integral_constant<int, 2>{}); break;
            case 4:  rounds(std::integral_constant<int, 4>{}); break;
            default: rounds(std::integral_constant<int, 8>{}); break;
        }
    } else {
        wacc[0] = 0.0f; s1[0] = 0.0f; s2[0] = 0.0f;
        switch (nSx) {
            case 1:  group_id_body<1, HAAR, true>(a, g, c, tid, pos, sh, use_sadct, wacc[0], s1[0], s2[0], work); break;
            case 2:  group_id_body<2, HAAR, true>(a, g, c, tid, pos, sh, use_sadct, wacc[0], s1[0], s2[0], work); break;
            case 4:  group_id_body<4, HAAR, true>(a, g, c, tid, pos, sh, use_sadct, wacc[0], s1[0], s2[0], work); break;
            default: group_id_body<8, HAAR, true>(a, g, c, tid, pos, sh, use_sadct, wacc[0], s1[0], s2[0], work); break;
        }
    }
    if (!split) { __syncthreads(); T16_MARK(); inv2d(0, NP); }
    T16_MARK();
#ifdef LFBM5D_PHASE_TIMING
    T16_MARK();
    if (tid == 0) { for (int i = 0; i < 5; i++) atomicAdd(&a.counters[4 + i], (unsigned long long)(tq[i + 1] - tq[i])); atomicAdd(&a.counters[9], 1ull); }
#endif
#pragma unroll
    for (int gi = 0; gi < (MULTI ? kT16Groups : 1); gi++) {
        for (int o = 32; o > 0; o >>= 1) { wacc[gi] += __shfl_xor(wacc[gi], o); s1[gi] += __shfl_xor(s1[gi], o); s2[gi] += __shfl_xor(s2[gi], o); }
        if ((tid & 63) == 0) { red[gi][0][tid >> 6] = wacc[gi]; red[gi][1][tid >> 6] = s1[gi]; red[gi][2][tid >> 6] = s2[gi]; }
    }
    __syncthreads();
    if (tid < ngr) {   /* group weights (core:412-421, sd_weighting_5d core:3140-3173) */
        const int gi = tid;
        float w = 0.0f, m = 0.0f, q = 0.0f;
        for (int i = 0; i < 4; i++) { w += red[gi][0][i]; m += red[gi][1][i]; q += red[gi][2][i]; }
        float wx;
        if (a.useSD) {
            const float Nn = (float)(nSx * A);
            const float res = (q - m * m / Nn) / (Nn - 1.0f);
            wx = res > 0.0f ? 1.0f / sqrtf(res) : 0.0f;
        } else {
            const float sig = a.sigma[c];
            wx = w > 0.0f ? (sig > 0.0f ? 1.0f / (sig * sig * w) : 1.0f / w) : 1.0f;
        }
        a.wgt[(size_t)(g + gi) * a.C + c] = wx;
        if (c == 0) {
            atomicAdd(&a.counters[0], (unsigned long long)nSx);
            if (a.tau4 == 6 && group_shape(a, g + gi).use_sadct) atomicAdd(&a.counters[1], 1ull);
        }
    }
}
#ifndef LFBM5D_T16_NOSPLIT
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(LFBM5D_T16_WAVES, LFBM5D_T16_WAVES))) void k_group_bior16_haar(GroupArgs a) { group_t16_kernel<true, true, false, true>(a); }
#else
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_group_bior16_haar(GroupArgs a) { group_t16_kernel<true, true, false>(a); }
#endif
__global__ __launch_bounds__(256) void k_group_bior16_any(GroupArgs a) { group_t16_kernel<false, true, false>(a); }
#ifndef LFBM5D_T16_NOSPLIT
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(LFBM5D_T16_WAVES, LFBM5D_T16_WAVES))) void k_group_dct16_haar(GroupArgs a) { group_t16_kernel<true, false, false, true>(a); }
#else
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_group_dct16_haar(GroupArgs a) { group_t16_kernel<true, false, false>(a); }
#endif
__global__ __launch_bounds__(256) void k_group_dct16_any(GroupArgs a) { group_t16_kernel<false, false, false>(a); }
/* N = 1: kT16Groups groups per workgroup (the 5th-dimension transform is the identity, HAAR or not) */
__global__ __launch_bounds__(256) void k_group_bior16_n1(GroupArgs a) { group_t16_kernel<true, true, true>(a); }
__global__ __launch_bounds__(256) void k_group_dct16_n1(GroupArgs a) { group_t16_kernel<true, false, true>(a); }
__global__ __launch_bounds__(256) void k_group_bior16_n1_sa(GroupArgs a) { group_t16_kernel<true, true, true, false, true>(a); }
__global__ __launch_bounds__(256) void k_group_dct16_n1_sa(GroupArgs a) { group_t16_kernel<true, false, true, false, true>(a); }

} /* namespace */

hipError_t prepare_group_ht() {
    const void* fns[] = {
        reinterpret_cast<const void*>(&k_group_bior16_haar), reinterpret_cast<const void*>(&k_group_bior16_any),
        reinterpret_cast<const void*>(&k_group_dct16_haar), reinterpret_cast<const void*>(&k_group_dct16_any),
        reinterpret_cast<const void*>(&k_group_bior16_n1), reinterpret_cast<const void*>(&k_group_dct16_n1),
        reinterpret_cast<const void*>(&k_group_bior16_n1_sa), reinterpret_cast<const void*>(&k_group_dct16_n1_sa)};
    for (const void* f : fns) {
        const hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, kDedicatedLdsLimit);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

hipError_t launch_group_ht(hipStream_t s, const GroupArgs& a, bool all_sa, bool* launched) {
    *launched = true;
    /* no 2-D transform and a stack small enough for registers: register-resident kernel */
    if (a.tau2 == 4 && a.N <= 8 && a.k * a.k <= 256 && a.step == 1 && a.A == 9 && (size_t)a.A * a.C * a.Wb * a.Hb * 4 < 0x7fffffffull) {   /* 32-bit byte offsets into the window */
        const unsigned threads = ((a.k * a.k + 63) / 64) * 64;
        const unsigned gx = ((a.n_groups + 7) / 8) * 8;   /* xcd_group_index */
        if (all_sa) {
            if (a.tau5 == 9) hipLaunchKernelGGL(k_group_id_haar_sa, dim3(gx, a.C), dim3(threads), 0, s, a);
            else             hipLaunchKernelGGL(k_group_id_any_sa, dim3(gx, a.C), dim3(threads), 0, s, a);
        }
        else {
            if (!a.sa_list) return hipErrorInvalidValue;
#ifndef LFBM5D_HT_REFERENCE_ORDER
            const bool fast = a.tau5 == 9 && (a.tau4 == 5 || a.tau4 == 6) && a.C <= 3;
#else
            const bool fast = false;
#endif
            if (fast)             hipLaunchKernelGGL(k_group_id_haar_fast, dim3(gx, a.C), dim3(threads), 0, s, a);
            else if (a.tau5 == 9) hipLaunchKernelGGL(k_group_id_haar, dim3(gx, a.C), dim3(threads), 0, s, a);
            else                  hipLaunchKernelGGL(k_group_id_any, dim3(gx, a.C), dim3(threads), 0, s, a);
            if (a.tau4 == 6 || fast) {   /* what the kernel above has left: shape-adaptive groups (k_group_shape's entries), guard-band cases */
                if (a.tau5 == 9) hipLaunchKernelGGL(k_group_id_haar_list, dim3(kSaListBlocks), dim3(threads), 0, s, a);
                else             hipLaunchKernelGGL(k_group_id_any_list, dim3(kSaListBlocks), dim3(threads), 0, s, a);
            }
        }
        return hipGetLastError();
    }
    if ((a.tau2 == 7 || a.tau2 == 5) && a.k == 16 && a.N <= 8 && a.step == 1 && a.A == 9) {   /* bior1.5 / DCT on 16x16 patches, HT step */
        const size_t lb = (size_t)a.N * 9 * kT16Patch * sizeof(float);
        const dim3 grid(a.n_groups, a.C), block(256);
        if (a.N == 1 && a.tau5 != 5) {   /* nine patches per group: a few groups share a workgroup (Haar / Hadamard of one patch: identity) */
            const dim3 grid8((a.n_groups + kT16Groups - 1) / kT16Groups, a.C);
            const size_t l1 = (size_t)kT16Groups * 9 * kT16Patch * sizeof(float);
            if (all_sa) {
                if (a.tau2 == 7) hipLaunchKernelGGL(k_group_bior16_n1_sa, grid8, block, l1, s, a);
                else             hipLaunchKernelGGL(k_group_dct16_n1_sa, grid8, block, l1, s, a);
            }
            else if (a.tau2 == 7) hipLaunchKernelGGL(k_group_bior16_n1, grid8, block, l1, s, a);
            else                  hipLaunchKernelGGL(k_group_dct16_n1, grid8, block, l1, s, a);
            return hipGetLastError();
        }
        if (a.tau2 == 7) {
#ifndef LFBM5D_T16_NOSPLIT   /* two rounds through a work area of 40 patches: groups of fewer than eight matches fit it whole (N <= 4: 36 patches) */
            if (a.tau5 == 9) hipLaunchKernelGGL(k_group_bior16_haar, grid, block, std::min(lb, (size_t)kT16Half * kT16Patch * sizeof(float)), s, a);
#else
            if (a.tau5 == 9) hipLaunchKernelGGL(k_group_bior16_haar, grid, block, lb, s, a);
#endif
            else             hipLaunchKernelGGL(k_group_bior16_any, grid, block, lb, s, a);
        } else {
#ifndef LFBM5D_T16_NOSPLIT
            if (a.tau5 == 9) hipLaunchKernelGGL(k_group_dct16_haar, grid, block, std::min(lb, (size_t)kT16Half * kT16Patch * sizeof(float)), s, a);
#else
            if (a.tau5 == 9) hipLaunchKernelGGL(k_group_dct16_haar, grid, block, lb, s, a);
#endif
            else             hipLaunchKernelGGL(k_group_dct16_any, grid, block, lb, s, a);
        }
        return hipGetLastError();
    }
    *launched = false;
    return hipSuccess;
}
} /* namespace lfbm5d */
