/*
 * lfbm5d_group_wide.hip -- dedicated group kernel of the hard-thresholding step with tau_2D = id on 5x5 and 7x7 angular windows
 * (aswSize 2 / 3; core:277-481 with aheight x awidth from bm5d.cpp:215-218), round 5.
 *
 * With no 2-D transform a pixel of the patches never mixes with other pixels, so a (group, channel) is processed in SLABS of
 * pixels: the slab's nSx x A x SLAB stack lives in LDS (51 KB for a 5x5 window and 64 pixels, 50 KB for 7x7 and 32) --
 *   gather the slab's pixels of every patch (pixel fastest: 256 / 128 contiguous bytes per patch row piece);
 *   the aw x aw angular DCT as two separable passes over the stack -- item = (match, row of the block, pixel), then (match, column,
 *   pixel): aw values per thread, so registers never limit the occupancy (a thread holding all 49 values of a 7x7 block needed 325);
 *   the fibres along the matches: Haar / Hadamard / DCT, hard threshold, inverse (filter5, shared with the general kernel);
 *   the inverse angular passes, the second writing the filtered pixels straight to `filt`
 * -- slab after slab, the survivor count carried over for the group weight.  The general kernel keeps the whole 200 KB stack of
 * such a group in an HBM scratch slice and took 5.8 / 12.8 ms per 304^2 pass (5x5 / 7x7) where this one takes well under 1 / 2 ms;
 * same transforms in the same order, hence the same results.
 */
#include "lfbm5d_group_device.h"

namespace lfbm5d {

namespace {

/* SPLIT: a workgroup per (group, channel, SLAB of pixels) -- blockIdx.z = slab -- instead of one per (group, channel) that walks the
 * slabs: four to eight times the workgroups, three of them per CU in different phases, so that gathers, transforms and stores of
 * different slabs overlap.  The survivor count of a (group, channel) is then summed with atomics in wgt (whole numbers: exact in any
 * order; the buffer is zeroed before the launch) and turned into the weight by k_group_idw_weight.  useSD (float sums whose value
 * depends on the order) keeps the walking form. */
template <int AW, int SLAB, bool SPLIT>
__global__ __launch_bounds__(256) void k_group_idw(GroupArgs a) {
    constexpr int A = AW * AW, NT = 256;
    extern __shared__ float S[];                      /* [n][st][SLAB] */
    __shared__ unsigned pos[8 * A];
    __shared__ float red[3][NT / 64];
    const int tid = threadIdx.x;
    const unsigned gi = xcd_group_index(a);
    if (gi >= a.n_groups) return;
    const unsigned g = a.ref_begin + gi;
    const int c = blockIdx.y;
    const int k = a.k, k2 = k * k, N = a.N;
    const int nSx = (int)a.self_cnt[g];
    const size_t plane = (size_t)a.Wb * a.Hb;
    const TbPtr tb = (TbPtr)a.tb;
    for (int i = tid; i < nSx * A; i += NT) pos[i] = a.gpos[(size_t)g * N * A + i];
    ShRef sh = group_shape(a, g);
    __syncthreads();
    const bool use_sadct = a.tau4 == 6 && sh.use_sadct;
    const bool do_dct4 = a.tau4 == 5 || (a.tau4 == 6 && !use_sadct);
    const bool do_sa4 = !do_dct4 && a.tau4 == 6;
    const float sig = a.sigma[c];
    const float T = a.lambda * sig * 1.41421356237309505f;   /* core:2431 */
    const float sig2 = sig * sig;
    float wacc = 0.0f, s1 = 0.0f, s2 = 0.0f;
    float* const out = a.filt + (size_t)g * N * A * a.C * k2;
    const float* const img = a.noisy + (size_t)c * plane;
    for (int p0 = SPLIT ? (int)blockIdx.z * SLAB : 0; p0 < (SPLIT ? min(k2, ((int)blockIdx.z + 1) * SLAB) : k2); p0 += SLAB) {
        const int npx = min(SLAB, k2 - p0);
        /* gather (core:286-299): item = (patch, pixel), pixel fastest */
        {
            constexpr int G = 10;   /* loads in flight per thread */
            const int total = nSx * A * SLAB;
            const unsigned cplane = (unsigned)(a.C * plane);   /* (window images of up to 2^31 floats: lfbm5d_api.hip keeps A * C * plane * 4 below 2 GiB for this kernel) */
            for (int e0 = tid; e0 < total; e0 += NT * G) {
                float v[G];
#pragma unroll
                for (int u = 0; u < G; u++) {
                    const int e = e0 + u * NT;
                    v[u] = 0.0f;
                    if (e < total) {
                        const int px = e % SLAB, ns = e / SLAB, pq = p0 + px;
                        const unsigned p = pos[ns];
                        if (p != 0xffffffffu && px < npx) v[u] = img[(unsigned)(ns % A) * cplane + p + (unsigned)(pq / k) * a.Wb + (unsigned)(pq % k)];
                    }
                }
#pragma unroll
                for (int u = 0; u < G; u++) { const int e = e0 + u * NT; if (e < total) S[e] = v[u]; }
            }
        }
        __syncthreads();
        if (do_dct4) {
            /* forward angular DCT (dct_4d_process, core:1862-1901), separable: rows of the aw x aw block ... */
#pragma unroll 2
            for (int e = tid; e < nSx * AW * SLAB; e += NT) {
                const int px = e % SLAB, r = e / SLAB, s = r % AW, n = r / AW;
                float* row = S + (size_t)(n * A + s * AW) * SLAB + px;
                float x[AW], t[AW];
#pragma unroll
                for (int j = 0; j < AW; j++) x[j] = row[j * SLAB];
#pragma unroll
                for (int u = 0; u < AW; u++) {
                    float acc = 0.0f;
#pragma unroll
                    for (int j = 0; j < AW; j++) acc += x[j] * tb->cosw[u * AW + j];
                    t[u] = 2.0f * acc;
                }
#pragma unroll
                for (int u = 0; u < AW; u++) row[u * SLAB] = t[u];
            }
            __syncthreads();
            /* ... then columns, times coef_norm_4d */
#pragma unroll 2
            for (int e = tid; e < nSx * AW * SLAB; e += NT) {
                const int px = e % SLAB, r = e / SLAB, u = r % AW, n = r / AW;
                float* col = S + (size_t)(n * A + u) * SLAB + px;
                float t[AW], x[AW];
#pragma unroll
                for (int j = 0; j < AW; j++) t[j] = col[j * AW * SLAB];
#pragma unroll
                for (int v = 0; v < AW; v++) {
                    float acc = 0.0f;
#pragma unroll
                    for (int j = 0; j < AW; j++) acc += t[j] * tb->cosw[v * AW + j];
                    x[v] = 2.0f * acc * tb->cn4[v * AW + u];
                }
#pragma unroll
                for (int v = 0; v < AW; v++) col[v * AW * SLAB] = x[v];
            }
            __syncthreads();
        } else if (do_sa4) {   /* the rare shape-adaptive groups: the call form, one (match, pixel) vector per thread */
            for (int e = tid; e < nSx * SLAB; e += NT) {
                const int px = e % SLAB, n = e / SLAB;
                float y[kMaxA];
                for (int st = 0; st < A; st++) y[st] = S[(size_t)(n * A + st) * SLAB + px];
                sadctw_fwd<ShRef>(y, AW, sh, tb);
                for (int st = 0; st < A; st++) S[(size_t)(n * A + st) * SLAB + px] = y[st];
            }
            __syncthreads();
        }
        /* the fibres along the matches (core:371-410) */
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
        __syncthreads();
        if (do_dct4) {
            /* inverse angular DCT (dct_4d_inverse, core:1913-1954): times coef_norm_inv, rows ... */
#pragma unroll 2
            for (int e = tid; e < nSx * AW * SLAB; e += NT) {
                const int px = e % SLAB, r = e / SLAB, s = r % AW, n = r / AW;
                float* row = S + (size_t)(n * A + s * AW) * SLAB + px;
                float x[AW], t[AW];
#pragma unroll
                for (int u = 0; u < AW; u++) x[u] = row[u * SLAB] * tb->cni4[s * AW + u];
#pragma unroll
                for (int j = 0; j < AW; j++) {
                    float acc = 0.0f;
#pragma unroll
                    for (int u = 1; u < AW; u++) acc += x[u] * tb->cosw[u * AW + j];
                    t[j] = x[0] + 2.0f * acc;
                }
#pragma unroll
                for (int j = 0; j < AW; j++) row[j * SLAB] = t[j];
            }
            __syncthreads();
            /* ... then columns, and the filtered pixels straight out: filt[g][n][st][c][pq] */
#pragma unroll 2
            for (int e = tid; e < nSx * AW * SLAB; e += NT) {
                const int px = e % SLAB, r = e / SLAB, j = r % AW, n = r / AW;
                const float* col = S + (size_t)(n * A + j) * SLAB + px;
                float t[AW];
#pragma unroll
                for (int v = 0; v < AW; v++) t[v] = col[v * AW * SLAB];
                if (px < npx) {
#pragma unroll
                    for (int i = 0; i < AW; i++) {
                        float acc = 0.0f;
#pragma unroll
                        for (int v = 1; v < AW; v++) acc += t[v] * tb->cosw[v * AW + i];
                        out[((size_t)(n * A + i * AW + j) * a.C + c) * k2 + p0 + px] = (t[0] + 2.0f * acc) * tb->coef4inv;
                    }
                }
            }
        } else {
            if (do_sa4) {
                for (int e = tid; e < nSx * SLAB; e += NT) {
                    const int px = e % SLAB, n = e / SLAB;
                    float y[kMaxA];
                    for (int st = 0; st < A; st++) y[st] = S[(size_t)(n * A + st) * SLAB + px];
                    sadctw_inv<ShRef>(y, AW, sh, tb);
                    for (int st = 0; st < A; st++) S[(size_t)(n * A + st) * SLAB + px] = y[st];
                }
                __syncthreads();
            }
            for (int e = tid; e < nSx * A * SLAB; e += NT) {
                const int px = e % SLAB, ns = e / SLAB;
                if (px < npx) out[((size_t)ns * a.C + c) * k2 + p0 + px] = S[e];
            }
        }
        __syncthreads();
    }
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
        if (c == 0 && (!SPLIT || blockIdx.z == 0)) {
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

hipError_t prepare_group_wide() {
    const void* fns[] = {reinterpret_cast<const void*>(&k_group_idw<5, 64, false>), reinterpret_cast<const void*>(&k_group_idw<7, 32, false>),
                         reinterpret_cast<const void*>(&k_group_idw<5, 64, true>), reinterpret_cast<const void*>(&k_group_idw<7, 32, true>)};
    for (const void* f : fns) {
        const hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, kDedicatedLdsLimit);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

hipError_t launch_group_wide(hipStream_t s, const GroupArgs& a, bool* launched) {
    *launched = false;
    if (!(a.tau2 == 4 && a.step == 1 && !a.bm3d && a.N <= 8 && a.k * a.k <= 256 && (a.A == 25 || a.A == 49) && (size_t)a.A * a.C * a.Wb * a.Hb * 4 < 0x7fffffffull)) return hipSuccess;
    *launched = true;
    const unsigned gx = ((a.n_groups + 7) / 8) * 8;   /* xcd_group_index */
    const unsigned k2 = a.k * a.k;
    if (a.useSD || getenv("LFBM5D_WIDE_NOSPLIT")) {
        if (a.A == 25) hipLaunchKernelGGL((k_group_idw<5, 64, false>), dim3(gx, a.C), dim3(256), (size_t)a.N * 25 * 64 * sizeof(float), s, a);
        else           hipLaunchKernelGGL((k_group_idw<7, 32, false>), dim3(gx, a.C), dim3(256), (size_t)a.N * 49 * 32 * sizeof(float), s, a);
        return hipGetLastError();
    }
    hipError_t e = hipMemsetAsync(a.wgt + (size_t)a.ref_begin * a.C, 0, (size_t)a.n_groups * a.C * sizeof(float), s);
    if (e != hipSuccess) return e;
    if (a.A == 25) hipLaunchKernelGGL((k_group_idw<5, 64, true>), dim3(gx, a.C, (k2 + 63) / 64), dim3(256), (size_t)a.N * 25 * 64 * sizeof(float), s, a);
    else           hipLaunchKernelGGL((k_group_idw<7, 32, true>), dim3(gx, a.C, (k2 + 31) / 32), dim3(256), (size_t)a.N * 49 * 32 * sizeof(float), s, a);
    hipLaunchKernelGGL(k_group_idw_weight, grid1d((size_t)a.n_groups * a.C), dim3(256), 0, s, a);
    return hipGetLastError();
}

} /* namespace lfbm5d */
