/*
 * lfbm5d_group_generic.hip -- the group stage of a core pass (core:277-481 / :1054-1282) for gfx950: the geometry pre-pass
 * (k_group_pos, k_group_shape), the GENERAL group kernel -- any patch size, any transform combination, any angular window,
 * stacks in LDS (k_group) or in HBM scratch slices (k_group_big) -- and launch_group, which hands a configuration to its dedicated
 * kernel (lfbm5d_group_ht.hip, lfbm5d_group_wiener.hip) when there is one.  Split from lfbm5d_kernels.hip in round 5.
 */
#include "lfbm5d_group_device.h"

namespace lfbm5d {

namespace {

/* ------------------------------------------------------------------------------------------
 * Group geometry pre-pass (core:286-323, :503): for every group of the launch, the window position
 * of each of its N x A patches (0xffffffff: no patch), the positions the aggregation kernel adds
 * them at, and the 9-bit angular shape.  Doing this once, fully parallel, takes the
 * self_idx -> best -> patch dependent-load chain out of every group workgroup.
 * ------------------------------------------------------------------------------------------ */
__global__ __launch_bounds__(256) void k_group_pos(GroupArgs a) {
    /* one thread per (group, match): the A disparity look-ups of a match are independent loads */
    const int A = a.A, N = a.N, NA = N * A;
    const size_t plane = (size_t)a.Wb * a.Hb;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx == 0 && a.sa_list) a.sa_list[0] = 0u;   /* (k_group_shape, the next launch on the stream, appends) */
    if (idx >= (size_t)a.n_groups * N) return;
    const unsigned g = a.ref_begin + (unsigned)(idx / N);
    const int n = (int)(idx % N);
    const int nSx = (int)a.self_cnt[g];
    const unsigned k_r = a.refs[g];
    const unsigned ind_pst = n < nSx ? a.self_idx[(size_t)g * N + n] : 0u;
    /* byte offsets and presence bits for the scalar loads of the register-resident HT kernel (launch_group) */
    const bool want_ofs = A == 9 && a.tau2 == 4 && a.step == 1 && (size_t)A * a.C * plane * 4 < 0x7fffffffull;
    unsigned bits = 0;
    for (int st = 0; st < A; st++) {
        const bool masked = a.mask_bits.test((unsigned)st);
        unsigned p = 0xffffffffu;
        if (n < nSx && masked) p = (st == (int)a.pst) ? ind_pst : a.best[(size_t)st * plane + ind_pst];
        /* gather position: patches whose column equals Wb-k read the reference's never-filled table
         * column, i.e. zeros (core:1697, bm3d.cpp:737) on the centre path -- they are still aggregated */
        const bool zero_patch = a.fill_quirk && p != 0xffffffffu && (p % a.Wb) >= a.Wb - a.k;
        const int i = n * A + st;
        a.gpos[(size_t)g * NA + i] = zero_patch ? 0xffffffffu : p;
        if (want_ofs) {
            const bool there = !zero_patch && p != 0xffffffffu;
            a.gofs[(size_t)g * NA + i] = there ? (unsigned)(((size_t)st * a.C * plane + p) * 4) : 0u;
            bits |= there ? 1u << st : 0u;
        }
        const bool in_shape = st == (int)a.pst || (masked && a.shape[(size_t)st * plane + k_r]);
        /* positions the aggregation kernel will add this group's patches at; 0xffffffff = none
         * (match slot unused, empty SAI, or SAI outside the SADCT shape, core:503) */
        /* stored as (row << 16) | column: the aggregation kernel tests each position against many tiles */
        a.aggpos[((size_t)st * a.n_refs_total + g) * N + n] = ((a.tau4 != 6 || in_shape) && p != 0xffffffffu) ? ((p / a.Wb) << 16) | (p % a.Wb) : 0xffffffffu;
    }
    if (want_ofs) a.gok[(size_t)g * N + n] = bits;
}
template <bool BIG>   /* BIG: windows of more than 7x7 SAIs (records of kShapeInfoBigBytes, built in place in global memory) */
__global__ __launch_bounds__(256) void k_group_shape(GroupArgs a) {
    typedef typename std::conditional<BIG, ShapeInfoBig, ShapeInfo>::type SH;
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n_groups) return;
    const unsigned g = a.ref_begin + i;
    const size_t plane = (size_t)a.Wb * a.Hb;
    const unsigned k_r = a.refs[g];
    /* only what the group kernels read is written: the first A entries of each array (the record is sized for a 7x7
     * window, a 3x3 one uses a fifth of it), and nothing but the flag when the angular transform is not shape-adaptive */
    SH* out = reinterpret_cast<SH*>(a.gshape) + g;
    if (a.tau4 != 6) { out->use_sadct = 0; return; }
    const int A = (int)a.A, aw = window_side(A);
    if (BIG) {
        int* m = out->mask_dct;   /* (scratch until build_shape overwrites it: the shape goes in through a second array) */
        int full = 0;
        for (int st = 0; st < A; st++) {
            const bool masked = a.mask_bits.test((unsigned)st);
            m[st] = (st == (int)a.pst || (masked && a.shape[(size_t)st * plane + k_r])) ? 1 : 0;
            full += m[st];
        }
        if (full == A) { out->use_sadct = 0; return; }
        for (int st = 0; st < A; st++) out->mask[st] = m[st];
        build_shape(*out, out->mask, aw);
        return;
    }
    SH sh;
    int m[BIG ? 1 : kMaxA], full = 0;
    for (int st = 0; st < A; st++) {
        const bool masked = a.mask_bits.test((unsigned)st);
        m[st] = (st == (int)a.pst || (masked && a.shape[(size_t)st * plane + k_r])) ? 1 : 0;
        full += m[st];
    }
    /* the usual case, every SAI in the shape: the plain angular DCT, and the group kernels read nothing but the flag */
    if (full == A) { out->use_sadct = 0; return; }
    build_shape(sh, m, aw);
    for (int q = 0; q < A; q++) {
        out->mask[q] = sh.mask[q]; out->idx[q] = sh.idx[q]; out->mask_col[q] = sh.mask_col[q];
        out->idx_col[q] = sh.idx_col[q]; out->mask_dct[q] = sh.mask_dct[q];
    }
    for (int q = 0; q < aw; q++) { out->row_n[q] = sh.row_n[q]; out->col_n[q] = sh.col_n[q]; }
    out->use_sadct = sh.use_sadct;
    if (sh.use_sadct && a.sa_list) a.sa_list[1u + atomicAdd(&a.sa_list[0], 1u)] = g | (7u << 29);   /* every channel */
}
/* One (group, channel) of the generic path.  S0 / S1: the group's stack(s) [n][st][pq] -- in LDS (k_group) or, when the
 * stacks do not fit the 160 KiB, in a per-workgroup slice of an HBM scratch buffer (k_group_big); tmp: the 2-D stage's
 * LDS work area.  Any patch size, any transform combination, 3x3 and 5x5 angular windows. */
template <int STEP, bool BIG = false>
__device__ __forceinline__ void group_generic(const GroupArgs& a, const unsigned g, const int c, float* S0, float* S1, float* tmp,
                                              unsigned* pos, float (*red)[kThreads / 64]) {
    const int tid = threadIdx.x;
    const int k = a.k, k2 = k * k, A = a.A, N = a.N;
    const int aw = window_side(A);
    const int nSx = (int)a.self_cnt[g];
    const size_t plane = (size_t)a.Wb * a.Hb;
    const int stack = nSx * A * k2;
    const TbPtr tb = (TbPtr)a.tb;

    /* patch positions (core:286-299) and the SADCT shape of this group (core:302-323): from the pre-pass */
    for (int i = tid; i < nSx * A; i += kThreads) pos[i] = a.gpos[(size_t)g * N * A + i];   /* up to 32 x 9 > 256 */
    typename std::conditional<BIG, ShRefBig, ShRef>::type sh = [&]() -> typename std::conditional<BIG, ShRefBig, ShRef>::type {
        if constexpr (BIG) return group_shape_big(a, g); else return group_shape(a, g); }();
    __syncthreads();
    const bool use_sadct = a.tau4 == 6 && sh.use_sadct;

    /* gather (core:286-299).  Patches whose column equals Wb-k read the reference's never-filled
     * table column, i.e. zeros (core:1697, bm3d.cpp:737) -- reproduce. */
    {
        constexpr int G = 12; /* loads in flight per thread and stack */
        for (int e0 = tid; e0 < stack; e0 += kThreads * G) {
            float v0[G], v1[G];
#pragma unroll
            for (int u = 0; u < G; u++) {
                const int e = e0 + u * kThreads;
                v0[u] = 0.0f; v1[u] = 0.0f;
                if (e < stack) {
                    const int pq = e % k2, ns = e / k2;
                    const int st = ns % A;
                    const unsigned p = pos[ns];
                    if (p != 0xffffffffu) {   /* the pre-pass folds the never-filled table column in */
                        const size_t off = ((size_t)st * a.C + c) * plane + p + (size_t)(pq / k) * a.Wb + pq % k;
                        v0[u] = a.noisy[off];
                        if (STEP == 2) v1[u] = a.basic[off];
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < G; u++) {
                const int e = e0 + u * kThreads;
                if (e < stack) { S0[e] = v0[u]; if (STEP == 2) S1[e] = v1[u]; }
            }
        }
    }
    __syncthreads();

    if (a.tau2 != 4) {
        fwd2d(S0, tmp, nSx * A, k, a.tau2, tb);
        if (STEP == 2) fwd2d(S1, tmp, nSx * A, k, a.tau2, tb);
    }

    /* 4-D forward (core:353-360): one (n, pq) fibre of A values per thread */
    const bool do_dct4 = a.tau4 == 5 || (a.tau4 == 6 && !use_sadct);
    const bool do_sa4 = !do_dct4 && a.tau4 == 6;
    if (do_dct4 || do_sa4) {
        for (int f = tid; f < nSx * k2; f += kThreads) {
            const int n = f / k2, pq = f % k2;
            for (int s = 0; s < (STEP == 2 ? 2 : 1); s++) {
                float* S = s ? S1 : S0;
                if constexpr (BIG) {   /* more than 7x7 SAIs: run-time sizes, the vector in scratch memory */
                    float x[kBigA], t[kBigA];
                    for (int st = 0; st < A; st++) x[st] = S[(size_t)(n * A + st) * k2 + pq];
                    if (do_dct4) dctw_fwd_rt(x, t, aw, tb); else sadctw_fwd<ShRefBig>(x, aw, sh, tb);
                    for (int st = 0; st < A; st++) S[(size_t)(n * A + st) * k2 + pq] = x[st];
                } else
                if (A == 9) {
                    float x[9];
#pragma unroll
                    for (int st = 0; st < 9; st++) x[st] = S[(n * A + st) * k2 + pq];
                    if (do_dct4) dct9_fwd(x, tb); else sadct9_fwd(x, sh, tb);
#pragma unroll
                    for (int st = 0; st < 9; st++) S[(n * A + st) * k2 + pq] = x[st];
                } else if (do_dct4 && A == 25) {   /* 5x5 / 7x7 window, plain DCT: in registers */
                    float x[25];
#pragma unroll
                    for (int st = 0; st < 25; st++) x[st] = S[(n * 25 + st) * k2 + pq];
                    dctw_fwd_t<5>(x, tb);
#pragma unroll
                    for (int st = 0; st < 25; st++) S[(n * 25 + st) * k2 + pq] = x[st];
                } else if (do_dct4) {
                    float x[49];
#pragma unroll
                    for (int st = 0; st < 49; st++) x[st] = S[(n * 49 + st) * k2 + pq];
                    dctw_fwd_t<7>(x, tb);
#pragma unroll
                    for (int st = 0; st < 49; st++) S[(n * 49 + st) * k2 + pq] = x[st];
                } else {   /* shape-adaptive: the call form */
                    float x[kMaxA];
                    for (int st = 0; st < A; st++) x[st] = S[(n * A + st) * k2 + pq];
                    if constexpr (!BIG) sadctw_fwd<ShRef>(x, aw, sh, tb);
                    for (int st = 0; st < A; st++) S[(n * A + st) * k2 + pq] = x[st];
                }
            }
        }
        __syncthreads();
    }

    /* 5th dimension + shrinkage (core:371-410): one (st, pq) fibre of nSx values per thread */
    float wacc = 0.0f, s1 = 0.0f, s2 = 0.0f;
    {
        const float sig = a.sigma[c];
        const float T = a.bm3d ? a.lambda * sig : a.lambda * sig * 1.41421356237309505f; /* core:2431; bm3d.cpp:941 */
        const float sig2 = sig * sig;
        for (int f = tid; f < A * k2; f += kThreads) {
            const int st = f / k2, pq = f % k2;
            const bool in_shape = !use_sadct || sh.mask_dct[st];
            const int base = st * k2 + pq, stride = A * k2;
            switch (nSx) {
                case 1:  filter5<1, STEP>(S0, S1, base, stride, a.tau5, T, sig2, in_shape, wacc, s1, s2, tb); break;
                case 2:  filter5<2, STEP>(S0, S1, base, stride, a.tau5, T, sig2, in_shape, wacc, s1, s2, tb); break;
                case 4:  filter5<4, STEP>(S0, S1, base, stride, a.tau5, T, sig2, in_shape, wacc, s1, s2, tb); break;
                case 8:  filter5<8, STEP>(S0, S1, base, stride, a.tau5, T, sig2, in_shape, wacc, s1, s2, tb); break;
                case 16: filter5<16, STEP>(S0, S1, base, stride, a.tau5, T, sig2, in_shape, wacc, s1, s2, tb); break;
                default: filter5<32, STEP>(S0, S1, base, stride, a.tau5, T, sig2, in_shape, wacc, s1, s2, tb); break;   
            }
        }
    }
    /* group weight (core:412-421, sd_weighting_5d core:3140-3173) */
    for (int o = 32; o > 0; o >>= 1) { wacc += __shfl_xor(wacc, o); s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
    if ((tid & 63) == 0) { red[0][tid >> 6] = wacc; red[1][tid >> 6] = s1; red[2][tid >> 6] = s2; }
    __syncthreads();
    if (tid == 0) {
        float w = 0.0f, m = 0.0f, q = 0.0f;
        for (int i = 0; i < kThreads / 64; i++) { w += red[0][i]; m += red[1][i]; q += red[2][i]; }
        float wx;
        if (a.useSD) {
            const float Nn = a.bm3d ? (float)(nSx * k2) : (float)(nSx * A);
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
    float* F = STEP == 2 ? S1 : S0;

    /* 4-D inverse (core:431-451) */
    if (do_dct4 || do_sa4) {
        for (int f = tid; f < nSx * k2; f += kThreads) {
            const int n = f / k2, pq = f % k2;
            if constexpr (BIG) {
                float x[kBigA], t[kBigA];
                for (int st = 0; st < A; st++) x[st] = F[(size_t)(n * A + st) * k2 + pq];
                if (do_dct4) dctw_inv_rt(x, t, aw, tb); else sadctw_inv<ShRefBig>(x, aw, sh, tb);
                for (int st = 0; st < A; st++) F[(size_t)(n * A + st) * k2 + pq] = x[st];
            } else
            if (A == 9) {
                float x[9];
#pragma unroll
                for (int st = 0; st < 9; st++) x[st] = F[(n * A + st) * k2 + pq];
                if (do_dct4) dct9_inv(x, tb); else sadct9_inv(x, sh, tb);
#pragma unroll
                for (int st = 0; st < 9; st++) F[(n * A + st) * k2 + pq] = x[st];
            } else if (do_dct4 && A == 25) {
                float x[25];
#pragma unroll
                for (int st = 0; st < 25; st++) x[st] = F[(n * 25 + st) * k2 + pq];
                dctw_inv_t<5>(x, tb);
#pragma unroll
                for (int st = 0; st < 25; st++) F[(n * 25 + st) * k2 + pq] = x[st];
            } else if (do_dct4) {
                float x[49];
#pragma unroll
                for (int st = 0; st < 49; st++) x[st] = F[(n * 49 + st) * k2 + pq];
                dctw_inv_t<7>(x, tb);
#pragma unroll
                for (int st = 0; st < 49; st++) F[(n * 49 + st) * k2 + pq] = x[st];
            } else {
                float x[kMaxA];
                for (int st = 0; st < A; st++) x[st] = F[(n * A + st) * k2 + pq];
                if constexpr (!BIG) sadctw_inv<ShRef>(x, aw, sh, tb);
                for (int st = 0; st < A; st++) F[(n * A + st) * k2 + pq] = x[st];
            }
        }
    }
    __syncthreads();
    if (a.tau2 != 4) inv2d(F, tmp, nSx * A, k, a.tau2, tb);

    /* filtered patches out: [g][n][st][c][k2] */
    for (int e = tid; e < stack; e += kThreads) {
        const int pq = e % k2, ns = e / k2;
        filt_put(&a.filt[filt_patch(a, g, ns / A, ns % A, k2) + (size_t)c * k2 + pq], F[e]);
    }
}

template <int STEP>
__global__ __launch_bounds__(kThreads) void k_group(GroupArgs a) {
    extern __shared__ float lds[];
    __shared__ unsigned pos[kMaxN3 * kMaxA];
    __shared__ float red[3][kThreads / 64];
    const unsigned g = a.ref_begin + blockIdx.x;
    const int stack = (int)a.self_cnt[g] * (int)a.A * (int)(a.k * a.k);
    group_generic<STEP>(a, g, (int)blockIdx.y, lds, STEP == 2 ? lds + stack : nullptr, lds + (STEP == 2 ? 2 : 1) * stack, pos, red);
}

/* Stacks beyond the LDS (a 5x5 window with the README's 16x16 patches: 8 x 25 x 256 floats = 200 KiB per stack; N = 32
 * with 16x16 patches): a persistent launch, every workgroup owns a slice of an HBM scratch buffer for its stack(s) and
 * walks over (group, channel) items.  Global memory written by a workgroup is visible to it after a barrier (one CU, one
 * vector L1), so the phases are the LDS kernel's, only slower; the 2-D stage's work area stays in LDS. */
template <int STEP, bool BIG>
__global__ __launch_bounds__(kThreads) void k_group_big(GroupArgs a, float* scratch, unsigned long long slice_floats, unsigned tmp_floats) {
    extern __shared__ float lds[];
    __shared__ unsigned pos_small[BIG ? 1 : kMaxN3 * kMaxA];
    __shared__ float red[3][kThreads / 64];
    unsigned* pos = BIG ? reinterpret_cast<unsigned*>(lds + tmp_floats) : pos_small;   /* BIG: N x A positions behind the 2-D work area */
    float* S0 = scratch + (size_t)blockIdx.x * slice_floats;
    const unsigned items = a.n_groups * a.C;
    for (unsigned it = blockIdx.x; it < items; it += gridDim.x) {
        const unsigned g = a.ref_begin + it / a.C;
        const int stack = (int)a.self_cnt[g] * (int)a.A * (int)(a.k * a.k);
        group_generic<STEP, BIG>(a, g, (int)(it % a.C), S0, STEP == 2 ? S0 + stack : nullptr, lds, pos, red);
        __syncthreads();   /* pos / red / the scratch slice are reused by the next item */
    }
}

} /* namespace */

/* Kernels whose LDS stack can exceed the 64 KiB a launch gets by default: raise the limit once per device
 * (called from lfbm5d_create after hipSetDevice; the attribute belongs to the device's code object). */
constexpr int kGenericLdsLimit = 160 * 1024 - 8192;   /* dynamic LDS of k_group: its static part (positions of up to 32 x 49 patches) is 6.3 KB */
hipError_t prepare_group_kernels() {
    const void* fns[] = {reinterpret_cast<const void*>(&k_group<1>), reinterpret_cast<const void*>(&k_group<2>)};
    for (const void* f : fns) {
        const hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, kGenericLdsLimit);
        if (e != hipSuccess) return e;
    }
    hipError_t e = prepare_group_ht();
    if (e == hipSuccess) e = prepare_group_wiener();
    if (e == hipSuccess) e = prepare_group_wide();
    if (e == hipSuccess) e = prepare_group_slab();
    return e;
}

constexpr unsigned kBigBlocks = 1024;   /* persistent workgroups of k_group_big (four per CU) */
static size_t group_tmp_floats(const GroupArgs& a) {
    /* the 2-D stage's work area: one patch per wave-quarter for the generic path, [patch][k][k+1] for bior1.5 */
    const bool rows_form = (a.tau2 == 7 && (a.k == 8 || a.k == 16)) || (a.tau2 == 5 && (a.k == 8 || a.k == 12 || a.k == 16));   /* bior2d_fast, fwd2d_dct */
    return rows_form ? (size_t)(kThreads / a.k) * a.k * (a.k + 1) : std::max<size_t>(256, (size_t)a.k * a.k);
}
size_t group_lds_bytes(const GroupArgs& a) {
    const size_t stack = (size_t)a.N * a.A * a.k * a.k;
    return ((a.step == 2 ? 2 : 1) * stack + group_tmp_floats(a)) * sizeof(float);
}
static bool group_uses_generic(const GroupArgs& a) {   /* mirrors the dispatch of launch_group_ht / launch_group_wiener */
    if (a.opt & kOptGroupGeneric) return true;
    bool wide_window = false;
    for (unsigned w = 5; w <= 17; w += 2) wide_window |= a.A == w * w;
    if (wide_window && a.tau2 == 4 && a.step == 1 && !a.bm3d && a.N <= 8 && a.k * a.k <= 256 && (size_t)a.A * a.C * a.Wb * a.Hb * 4 < 0x7fffffffull) return false;   /* lfbm5d_group_wide.hip */
    if (a.A != 9 && !(a.bm3d && a.A == 1)) return true;
    if (a.A == 9 && a.tau2 == 4 && a.N <= 8 && a.k * a.k <= 256 && a.step == 1) return false;
    if (a.A == 9 && (a.tau2 == 7 || a.tau2 == 5) && a.k == 16 && a.N <= 8 && a.step == 1) return false;
    if (a.A == 9 && a.tau2 == 7 && a.k == 8 && a.step == 2 && a.N <= (unsigned)kMaxN) return false;
    if (a.A == 9 && a.tau2 == 5 && a.k == 8 && a.N <= (unsigned)kMaxN) return false;
    if (a.bm3d && a.A == 1 && a.k == 8 && a.tau5 == 8 && (a.tau2 == 5 || a.tau2 == 7)) return false;
    return true;
}
size_t group_scratch_bytes(const GroupArgs& a) {
    if (group_uses_generic(a) && !(a.opt & kOptGroupGeneric) && group_uses_slab(a)) return group_slab_scratch_bytes(a);
    if (!group_uses_generic(a) || (group_lds_bytes(a) <= (size_t)kGenericLdsLimit && a.A <= (unsigned)kMaxA)) return 0;
    return (size_t)kBigBlocks * (a.step == 2 ? 2 : 1) * a.N * a.A * a.k * a.k * sizeof(float);
}
hipError_t launch_group(hipStream_t s, const GroupArgs& a) {
    /* geometry pre-pass: patch positions, aggregation positions, angular shapes */
    hipLaunchKernelGGL(k_group_pos, grid1d((size_t)a.n_groups * a.N), dim3(256), 0, s, a);
    const bool bigA = a.A > (unsigned)kMaxA;
    if (bigA) hipLaunchKernelGGL(k_group_shape<true>, grid1d(a.n_groups), dim3(256), 0, s, a);
    else      hipLaunchKernelGGL(k_group_shape<false>, grid1d(a.n_groups), dim3(256), 0, s, a);
    /* a window with an empty SAI: tau_4D is the shape-adaptive transform (bm5d.cpp:276-280) and every group uses it */
    const bool all_sa = a.tau4 == 6 && !a.mask_bits.holds_all(a.A) && !(a.opt & kOptNoSaKernels);
    /* option group_generic: test hook, every configuration through the generic LDS kernel (the dedicated kernels' cross-check) */
    if (!(a.opt & kOptGroupGeneric)) {
        bool launched = false;
        hipError_t e = launch_group_ht(s, a, all_sa, &launched);
        if (launched) return e;
        e = launch_group_wiener(s, a, all_sa, &launched);
        if (launched) return e;
        e = launch_group_wide(s, a, &launched);
        if (launched) return e;
        e = launch_group_slab(s, a, &launched);
        if (launched) return e;
    }
    const size_t lds = group_lds_bytes(a);
    if (lds > (size_t)kGenericLdsLimit || bigA) {   /* stacks in HBM scratch slices, persistent workgroups */
        const unsigned long long slice = (unsigned long long)(a.step == 2 ? 2 : 1) * a.N * a.A * a.k * a.k;
        if (!a.scratch || a.scratch_floats < slice * kBigBlocks) return hipErrorInvalidValue;
        const unsigned blocks = std::min<unsigned>(kBigBlocks, a.n_groups * a.C);
        const unsigned tf = (unsigned)group_tmp_floats(a);
        const size_t ltmp = (size_t)tf * sizeof(float) + (bigA ? (size_t)a.N * a.A * sizeof(unsigned) : 0);
        if (bigA) {
            if (a.step == 2) hipLaunchKernelGGL((k_group_big<2, true>), dim3(blocks), dim3(kThreads), ltmp, s, a, a.scratch, slice, tf);
            else             hipLaunchKernelGGL((k_group_big<1, true>), dim3(blocks), dim3(kThreads), ltmp, s, a, a.scratch, slice, tf);
        }
        else if (a.step == 2) hipLaunchKernelGGL((k_group_big<2, false>), dim3(blocks), dim3(kThreads), ltmp, s, a, a.scratch, slice, tf);
        else                  hipLaunchKernelGGL((k_group_big<1, false>), dim3(blocks), dim3(kThreads), ltmp, s, a, a.scratch, slice, tf);
        return hipGetLastError();
    }
    if (a.step == 2) hipLaunchKernelGGL(k_group<2>, dim3(a.n_groups, a.C), dim3(kThreads), lds, s, a);
    else             hipLaunchKernelGGL(k_group<1>, dim3(a.n_groups, a.C), dim3(kThreads), lds, s, a);
    return hipGetLastError();
}
} /* namespace lfbm5d */
