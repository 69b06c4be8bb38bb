/*
 * lfbm5d_group_wiener.hip -- dedicated group kernels for 8x8 patches on 3x3 windows, for gfx950: k_group_dct8w3 (the README's Wiener
 * step: one image in LDS at a time, shrinkage coefficients in registers), k_group_dct8w2 (its two-image predecessor: Hadamard /
 * DCT fibres, windows of 1.9 GB and more), k_group_dct8w (bior1.5 in the 2-D stage), k_group_dct8 (hard-thresholding step with an 8x8
 * DCT) and k_group_bm3d8 (per-SAI BM3D: a group per wavefront).  core:1054-1282, bm3d.cpp:315-690.  Split from lfbm5d_kernels.hip
 * in round 5.
 */
#include "lfbm5d_group_device.h"

namespace lfbm5d {

namespace {

/* ------------------------------------------------------------------------------------------
 * 8x8 2-D DCT variant (the README Wiener configuration: k = 8, tau_2D = dct).  The 2-D transform
 * is done with ONE THREAD PER PATCH entirely in registers: the thread loads its 8x8 patch from the
 * window image, runs 8 row + 8 column 8-point DCTs (even/odd factorisation, orthonormal scaling
 * = REDFT10 x REDFT10 x coef_norm of bm3d.cpp:745-757,1148-1168) and scatters the 64 coefficients
 * into an LDS stack laid out [coefficient pq][patch] (+1 padding, conflict-free for every phase).
 * That replaces the gather + per-coefficient LDS matrix products of k_group (about 20 LDS
 * operations per stacked pixel) by one LDS write per pixel.  4-D / 5th-dimension phases work on
 * fibres of that stack as in k_group; the inverse 2-D DCT is again one thread per patch, reading
 * its 64 coefficients and storing the 64 pixels of the filtered patch as four-float vectors.
 * ------------------------------------------------------------------------------------------ */
constexpr int kDct8Threads = 320;

template <int STEP>
__global__ __launch_bounds__(kDct8Threads) void k_group_dct8(GroupArgs a) {
    extern __shared__ float lds[];
    __shared__ unsigned pos[kMaxN * kA3];
    __shared__ float red[3][kDct8Threads / 64];
    const int tid = threadIdx.x;
    const unsigned g = a.ref_begin + blockIdx.x;
    const int c = blockIdx.y;
    constexpr int A = 9, K2 = 64;
    const int N = a.N;
    const int nSx = (int)a.self_cnt[g];
    const size_t plane = (size_t)a.Wb * a.Hb;
    const int NP = nSx * A;               /* patches per stack */
    const int NPp = (N * A) | 1;          /* row stride of the [pq][patch] stack, odd */
    float* S0 = lds;
    float* S1 = lds + K2 * NPp;
    const TbPtr tb = (TbPtr)a.tb;
    constexpr int S = STEP == 2 ? 2 : 1;

    for (int i = tid; i < NP; i += kDct8Threads) pos[i] = a.gpos[(size_t)g * N * A + i];
    ShRef sh = group_shape(a, g);
    __syncthreads();
    const bool use_sadct = a.tau4 == 6 && sh.use_sadct;

    /* gather + forward 2-D DCT, one thread per patch */
    for (int task = tid; task < S * NP; task += kDct8Threads) {
        const int s = task / NP, patch = task % NP, st = patch % A;
        const unsigned p = pos[patch];
        const bool ok = p != 0xffffffffu;
        const float* img = (s ? a.basic : a.noisy) + ((size_t)st * a.C + c) * plane + (ok ? p : 0u);
        float x[8][8];
#pragma unroll
        for (int i = 0; i < 8; i++)
#pragma unroll
            for (int j = 0; j < 8; j++) { const float v = img[(size_t)i * a.Wb + j]; x[i][j] = ok ? v : 0.0f; }
#pragma unroll
        for (int i = 0; i < 8; i++) dct8_fwd(x[i]);
#pragma unroll
        for (int j = 0; j < 8; j++) {
            float col[8];
#pragma unroll
            for (int i = 0; i < 8; i++) col[i] = x[i][j];
            dct8_fwd(col);
#pragma unroll
            for (int i = 0; i < 8; i++) x[i][j] = col[i];
        }
        float* dst = (s ? S1 : S0) + patch;
#pragma unroll
        for (int i = 0; i < 8; i++)
#pragma unroll
            for (int j = 0; j < 8; j++) dst[(i * 8 + j) * NPp] = x[i][j];
    }
    __syncthreads();

    /* 4-D forward: one (n, pq) fibre of 9 values per thread */
    const bool do_dct4 = a.tau4 == 5 || (a.tau4 == 6 && !use_sadct);
    const bool do_sa4 = !do_dct4 && a.tau4 == 6;
    if (do_dct4 || do_sa4) {
        for (int f = tid; f < S * nSx * K2; f += kDct8Threads) {
            const int s = f / (nSx * K2), r = f % (nSx * K2), n = r / K2, pq = r % K2;
            float* base = (s ? S1 : S0) + pq * NPp + n * A;
            float x[9];
#pragma unroll
            for (int st = 0; st < 9; st++) x[st] = base[st];
            if (do_dct4) dct9_fwd(x, tb); else sadct9_fwd(x, sh, tb);
#pragma unroll
            for (int st = 0; st < 9; st++) base[st] = x[st];
        }
        __syncthreads();
    }

    /* 5th dimension + shrinkage: one (st, pq) fibre of nSx values per thread */
    float wacc = 0.0f, s1 = 0.0f, s2 = 0.0f;
    {
        const float sig = a.sigma[c];
        const float T = a.lambda * sig * 1.41421356237309505f;
        const float sig2 = sig * sig;
        for (int f = tid; f < A * K2; f += kDct8Threads) {
            const int st = f / K2, pq = f % K2;
            const bool in_shape = !use_sadct || sh.mask_dct[st];
            const int base = pq * NPp + st;
            switch (nSx) {
                case 1:  filter5<1, STEP>(S0, S1, base, A, a.tau5, T, sig2, in_shape, wacc, s1, s2, tb); break;
                case 2:  filter5<2, STEP>(S0, S1, base, A, a.tau5, T, sig2, in_shape, wacc, s1, s2, tb); break;
                case 4:  filter5<4, STEP>(S0, S1, base, A, a.tau5, T, sig2, in_shape, wacc, s1, s2, tb); break;
                case 8:  filter5<8, STEP>(S0, S1, base, A, a.tau5, T, sig2, in_shape, wacc, s1, s2, tb); break;
                default: filter5<16, STEP>(S0, S1, base, A, a.tau5, T, sig2, in_shape, wacc, s1, s2, tb); break;
            }
        }
    }
    for (int o = 32; o > 0; o >>= 1) { wacc += __shfl_xor(wacc, o); s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
    if ((tid & 63) == 0) { red[0][tid >> 6] = wacc; red[1][tid >> 6] = s1; red[2][tid >> 6] = s2; }
    __syncthreads();
    if (tid == 0) {
        float w = 0.0f, m = 0.0f, q = 0.0f;
        for (int i = 0; i < kDct8Threads / 64; i++) { w += red[0][i]; m += red[1][i]; q += red[2][i]; }
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
    float* F = STEP == 2 ? S1 : S0;

    /* 4-D inverse */
    if (do_dct4 || do_sa4) {
        for (int f = tid; f < nSx * K2; f += kDct8Threads) {
            const int n = f / K2, pq = f % K2;
            float* base = F + pq * NPp + n * A;
            float x[9];
#pragma unroll
            for (int st = 0; st < 9; st++) x[st] = base[st];
            if (do_dct4) dct9_inv(x, tb); else sadct9_inv(x, sh, tb);
#pragma unroll
            for (int st = 0; st < 9; st++) base[st] = x[st];
        }
    }
    __syncthreads();

    /* inverse 2-D DCT + store, one thread per patch: filt[g][n][st][c][64] */
    for (int patch = tid; patch < NP; patch += kDct8Threads) {
        float x[8][8];
        const float* src = F + patch;
#pragma unroll
        for (int i = 0; i < 8; i++)
#pragma unroll
            for (int j = 0; j < 8; j++) x[i][j] = src[(i * 8 + j) * NPp];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            float col[8];
#pragma unroll
            for (int i = 0; i < 8; i++) col[i] = x[i][j];
            dct8_inv(col);
#pragma unroll
            for (int i = 0; i < 8; i++) x[i][j] = col[i];
        }
#pragma unroll
        for (int i = 0; i < 8; i++) dct8_inv(x[i]);
        float4* out = reinterpret_cast<float4*>(a.filt + ((size_t)g * N * A + patch) * a.C * K2 + (size_t)c * K2);
#pragma unroll
        for (int i = 0; i < 8; i++) {
            filt_put4(&out[2 * i], make_float4(x[i][0], x[i][1], x[i][2], x[i][3]));
            filt_put4(&out[2 * i + 1], make_float4(x[i][4], x[i][5], x[i][6], x[i][7]));
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * Wiener step of the 8x8 DCT configuration (step 2, k = 8, tau_2D = dct): packed-fp32 variant.
 * The noisy and the pilot (basic) stacks go through identical forward transforms, so they are kept
 * as the two halves of a float2 everywhere: one LDS stack of float2 laid out [coefficient pq][patch],
 * 64-bit LDS accesses, and v_pk_{add,mul,fma}_f32 arithmetic that transforms both stacks at once.
 * Where only the filtered stack remains (inverse transforms) two fibres / two patches are paired
 * instead.  The arithmetic per element is the same sequence as in k_group_dct8<2>.
 * Phases (256 threads, barriers between them):
 *   1  cooperative 16-byte gather of both images through LDS, then thread = patch: 16 packed 8-point DCTs, 64 LDS writes
 *   2  thread = (n, pq) fibre over the 9 SAIs: packed 3x3 DCT (shape-adaptive variant on the scalar path)
 *   3  thread = (st, pq) fibre over the nSx patches: packed Haar, Wiener shrinkage, inverse Haar
 *   4  thread = two (n, pq) fibres of the filtered stack: packed inverse 3x3 DCT
 *   5  thread = two patches: packed inverse 8x8 DCT, 16-byte stores of the filtered patches
 * ------------------------------------------------------------------------------------------ */

/* 8-point orthonormal DCT-II / DCT-III of the Wiener kernels (T = float or a packed pair).  The 1/2 of the orthonormal
 * scaling is folded into the cosines (round 1 multiplied by it separately: a seventh more instructions, no effect then
 * because the kernel was latency-bound; at 60 % VALU utilisation it counts).  No threshold follows these transforms, so
 * the last-bit differences against the unfolded form stay far inside the Wiener step's tolerance. */
template <class T> __device__ __forceinline__ void dct8_fwd_t(T* x) {
    const float a0 = 0.35355339059327376f;
    const float c1 = 0.5f * 0.98078528040323044f, c2 = 0.5f * 0.92387953251128674f, c3 = 0.5f * 0.83146961230254524f,
                c4 = 0.5f * 0.70710678118654752f, c5 = 0.5f * 0.55557023301960222f, c6 = 0.5f * 0.38268343236508977f,
                c7 = 0.5f * 0.19509032201612827f;
    const T s0 = x[0] + x[7], s1 = x[1] + x[6], s2 = x[2] + x[5], s3 = x[3] + x[4];
    const T d0 = x[0] - x[7], d1 = x[1] - x[6], d2 = x[2] - x[5], d3 = x[3] - x[4];
    const T p0 = s0 + s3, p1 = s1 + s2, m0 = s0 - s3, m1 = s1 - s2;
    x[0] = a0 * (p0 + p1);
    x[4] = c4 * (p0 - p1);
    x[2] = c2 * m0 + c6 * m1;
    x[6] = c6 * m0 - c2 * m1;
    x[1] = c1 * d0 + c3 * d1 + c5 * d2 + c7 * d3;
    x[3] = c3 * d0 - c7 * d1 - c1 * d2 - c5 * d3;
    x[5] = c5 * d0 - c1 * d1 + c7 * d2 + c3 * d3;
    x[7] = c7 * d0 - c5 * d1 + c3 * d2 - c1 * d3;
}
template <class T> __device__ __forceinline__ void dct8_inv_t(T* X) {
    const float a0 = 0.35355339059327376f;
    const float c1 = 0.5f * 0.98078528040323044f, c2 = 0.5f * 0.92387953251128674f, c3 = 0.5f * 0.83146961230254524f,
                c4 = 0.5f * 0.70710678118654752f, c5 = 0.5f * 0.55557023301960222f, c6 = 0.5f * 0.38268343236508977f,
                c7 = 0.5f * 0.19509032201612827f;
    const T e0 = a0 * X[0] + c4 * X[4], e1 = a0 * X[0] - c4 * X[4];
    const T f0 = c2 * X[2] + c6 * X[6], f1 = c6 * X[2] - c2 * X[6];
    const T E0 = e0 + f0, E1 = e1 + f1, E2 = e1 - f1, E3 = e0 - f0;
    const T O0 = c1 * X[1] + c3 * X[3] + c5 * X[5] + c7 * X[7];
    const T O1 = c3 * X[1] - c7 * X[3] - c1 * X[5] - c5 * X[7];
    const T O2 = c5 * X[1] - c1 * X[3] + c7 * X[5] + c3 * X[7];
    const T O3 = c7 * X[1] - c5 * X[3] + c3 * X[5] - c1 * X[7];
    X[0] = E0 + O0; X[7] = E0 - O0;
    X[1] = E1 + O1; X[6] = E1 - O1;
    X[2] = E2 + O2; X[5] = E2 - O2;
    X[3] = E3 + O3; X[4] = E3 - O3;
}
template <int NS> __device__ __forceinline__ void haar_fwd2(v2f* v) {
    const float s = 0.70710678118654752f;
#pragma unroll
    for (int n = NS; n > 1; n /= 2) {
        v2f t[NS > 1 ? NS : 1];
#pragma unroll
        for (int i = 0; i < n / 2; i++) { t[i] = (v[2 * i] + v[2 * i + 1]) * s; t[n / 2 + i] = (v[2 * i] - v[2 * i + 1]) * s; }
#pragma unroll
        for (int i = 0; i < n; i++) v[i] = t[i];
    }
}

template <int NS> __device__ __forceinline__ void haar_inv2(v2f* v) {
    const float s = 0.70710678118654752f;
#pragma unroll
    for (int h = 1; h < NS; h *= 2) {
        v2f t[NS > 1 ? NS : 1];
#pragma unroll
        for (int i = 0; i < h; i++) { t[2 * i] = (v[i] + v[h + i]) * s; t[2 * i + 1] = (v[i] - v[h + i]) * s; }
#pragma unroll
        for (int i = 0; i < 2 * h; i++) v[i] = t[i];
    }
}

/* Round 6: the Haar pair without its 1/sqrt 2 factors.  Forward: sums and differences only, the coefficient at index n (the
 * reference's order: approximation, then details from the coarsest level L = log2 NS down to level 1) carries 2^(l/2), l = its
 * level.  Inverse: of coefficients in that scale, NS times the orthonormal inverse -- additions and fma by the exact weights
 * 1, 2, 4, 8.  The Wiener step has no threshold an ulp could flip, its shrinkage e^2 / (e^2 + sigma^2) is evaluated with
 * sigma^2 scaled to the coefficient's own scale, and every dropped factor is a power of two or ends in one constant of the
 * inverse angular transform (w3_body). */
template <int NS> __device__ __forceinline__ constexpr int haar_level(int n) {   /* level of coefficient n of an NS-point transform (NS = 1: 0) */
    int l = 0, m = NS;
    while (m > 1) { m /= 2; l++; }          /* l = log2 NS */
    int lo = 1;                              /* n = 0, 1: level L; [2, 4): L - 1; [4, 8): L - 2; ... */
    int lev = l;
    while (lo * 2 <= n) { lo *= 2; lev--; }
    return NS == 1 ? 0 : lev;
}
template <int NS, class T> __device__ __forceinline__ void haar_fwd_u(T* v) {
#pragma unroll
    for (int n = NS; n > 1; n /= 2) {
        T t[NS > 1 ? NS : 1];
#pragma unroll
        for (int i = 0; i < n / 2; i++) { t[i] = v[2 * i] + v[2 * i + 1]; t[n / 2 + i] = v[2 * i] - v[2 * i + 1]; }
#pragma unroll
        for (int i = 0; i < n; i++) v[i] = t[i];
    }
}
template <int NS, class T> __device__ __forceinline__ void haar_inv_u(T* v) {
#pragma unroll
    for (int h = 1; h < NS; h *= 2) {
        T t[NS > 1 ? NS : 1];
        const float w = (float)h;
#pragma unroll
        for (int i = 0; i < h; i++) { t[2 * i] = v[i] + w * v[h + i]; t[2 * i + 1] = v[i] - w * v[h + i]; }
#pragma unroll
        for (int i = 0; i < 2 * h; i++) v[i] = t[i];
    }
}
/* the 3x3 angular DCT without its constants: forward = the reference's value / (alpha_v alpha_u coef_norm_4d), alpha = (2, sqrt 3, 1)
 * (GroupTables::ht3_f); the inverse takes values pre-multiplied by ht3_gf[st] / NS */
__device__ __forceinline__ void dct9_fwd2_u(v2f* x) {
    v2f t[9];
#pragma unroll
    for (int s = 0; s < 3; s++) {
        const v2f p = x[s * 3] + x[s * 3 + 2];
        t[s * 3] = p + x[s * 3 + 1]; t[s * 3 + 1] = x[s * 3] - x[s * 3 + 2]; t[s * 3 + 2] = p - 2.0f * x[s * 3 + 1];
    }
#pragma unroll
    for (int u = 0; u < 3; u++) {
        const v2f p = t[u] + t[6 + u];
        x[u] = p + t[3 + u]; x[3 + u] = t[u] - t[6 + u]; x[6 + u] = p - 2.0f * t[3 + u];
    }
}
__device__ __forceinline__ void dct9_inv2_u(v2f* x, TbPtr tb, const float scale) {
    const auto& g = tb->ht3_gf;
    v2f t[9];
#pragma unroll
    for (int s = 0; s < 3; s++) {
        const v2f Z0 = x[s * 3] * (g[s * 3] * scale), Z1 = x[s * 3 + 1] * (g[s * 3 + 1] * scale), Z2 = x[s * 3 + 2] * (g[s * 3 + 2] * scale);
        const v2f p = Z0 + Z2;
        t[s * 3] = p + Z1; t[s * 3 + 1] = Z0 - 2.0f * Z2; t[s * 3 + 2] = p - Z1;
    }
#pragma unroll
    for (int j = 0; j < 3; j++) {
        const v2f p = t[j] + t[6 + j];
        x[j] = p + t[3 + j]; x[3 + j] = t[j] - 2.0f * t[6 + j]; x[6 + j] = p - t[3 + j];
    }
}

/* a / b for the Wiener coefficient e^2 / (e^2 + sigma^2) (0 <= a < b, both normal or a = 0): reciprocal estimate and
 * one correction step instead of the IEEE division sequence -- within one ulp of the quotient, which is well inside
 * the float tolerance of this stage (the division was a tenth of the kernel's instructions) */
__device__ __forceinline__ float wiener_div(float a, float b) {
#ifdef LFBM5D_WIENER_DIV_REFINED
    const float r = __builtin_amdgcn_rcpf(b);
    const float q = a * r;
    return __builtin_fmaf(__builtin_fmaf(-b, q, a), r, q);
#else
    return a * __builtin_amdgcn_rcpf(b);   /* v_rcp_f32 is good to 1 ulp: the quotient to 2 ulp, a Wiener coefficient needs no more */
#endif
}

/* phase 3 of k_group_dct8w on one (st, pq) fibre of nSx = NS float2 entries (x: noisy, y: pilot) */
template <int NS, bool HAAR>
__device__ __forceinline__ void wiener_fibre2(v2f* stack, int base, int stride, unsigned tau5, float sig2, bool in_shape,
                                              bool useSD, float& wacc, float& s1, float& s2, TbPtr tb) {
    v2f f[NS];
#pragma unroll
    for (int n = 0; n < NS; n++) f[n] = stack[base + n * stride];
    float o[NS], e[NS];
    if (HAAR) {
        if (NS > 1) haar_fwd2<NS>(f);
#pragma unroll
        for (int n = 0; n < NS; n++) { o[n] = f[n].x; e[n] = f[n].y; }
        if (in_shape) {
#pragma unroll
            for (int n = 0; n < NS; n++) {
                float value = e[n] * e[n];
                value = wiener_div(value, value + sig2);
                e[n] = o[n] * value;
                wacc += value;
            }
        }
        if (NS > 1) haar_inv<NS>(e);
    } else {
#pragma unroll
        for (int n = 0; n < NS; n++) { o[n] = f[n].x; e[n] = f[n].y; }
        shrink_fibre<NS, 2>(o, e, tau5, 0.0f, sig2, in_shape, wacc, tb);
    }
    float* dst = reinterpret_cast<float*>(stack);
#pragma unroll
    for (int n = 0; n < NS; n++) dst[2 * (base + n * stride) + 1] = e[n];
    if (useSD) {
#pragma unroll
        for (int n = 0; n < NS; n++) { s1 += e[n]; s2 += e[n] * e[n]; }
    }
}

#ifndef LFBM5D_DCT8W_THREADS
#define LFBM5D_DCT8W_THREADS 256
#endif
constexpr int kDct8wThreads = LFBM5D_DCT8W_THREADS;

/* bior1.5 on an 8x8 patch held by ONE thread (rows x[i][0..8)), all three levels in registers; T = float or a packed pair.
 * Same taps, order and unfused arithmetic as bior_fwd_level / bior_inv_level (lib_transforms.cpp:46-204). */
template <int N1, class T> __device__ __forceinline__ void bior_fwd_vec(T* v, TbPtr tb) {
#pragma clang fp contract(off)
    constexpr int N2 = N1 / 2;
    T o[N1];
#pragma unroll
    for (int j = 0; j < N2; j++) {
        T acc = v[bior_ext(2 * j, N1)] * tb->lpd[0];
#pragma unroll
        for (int t = 1; t < 10; t++) acc += v[bior_ext(t + 2 * j, N1)] * tb->lpd[t];
        o[j] = acc;
        T hi = v[bior_ext(4 + 2 * j, N1)] * tb->hpd[4];
        hi += v[bior_ext(5 + 2 * j, N1)] * tb->hpd[5];
        o[N2 + j] = hi;
    }
#pragma unroll
    for (int j = 0; j < N1; j++) v[j] = o[j];
}
template <int N1, class T> __device__ __forceinline__ void bior_inv_vec(T* v, TbPtr tb) {
#pragma clang fp contract(off)
    constexpr int N2 = N1 / 2;
    T o[N1];
#pragma unroll
    for (int m = 0; m < N2; m++) {
        T acc = v[m % N1] * tb->hpr[0];
#pragma unroll
        for (int t = 1; t < 10; t++) acc += v[(t * N2 + m) % N1] * tb->hpr[t];
        o[2 * m] = acc;
        T lo = v[(4 * N2 + m) % N1] * tb->lpr[4];
        lo += v[(5 * N2 + m) % N1] * tb->lpr[5];
        o[2 * m + 1] = lo;
    }
#pragma unroll
    for (int j = 0; j < N1; j++) v[j] = o[j];
}
template <int N1, bool FWD, bool ROWS, class T> __device__ __forceinline__ void bior8_pass(T (*x)[8], TbPtr tb) {
#pragma unroll
    for (int a = 0; a < N1; a++) {
        T v[N1];
#pragma unroll
        for (int b = 0; b < N1; b++) v[b] = ROWS ? x[a][b] : x[b][a];
        if (FWD) bior_fwd_vec<N1>(v, tb); else bior_inv_vec<N1>(v, tb);
#pragma unroll
        for (int b = 0; b < N1; b++) { if (ROWS) x[a][b] = v[b]; else x[b][a] = v[b]; }
    }
}
template <class T> __device__ __forceinline__ void bior8_fwd_2d(T (*x)[8], TbPtr tb) {
    bior8_pass<8, true, true>(x, tb); bior8_pass<8, true, false>(x, tb);
    bior8_pass<4, true, true>(x, tb); bior8_pass<4, true, false>(x, tb);
    bior8_pass<2, true, true>(x, tb); bior8_pass<2, true, false>(x, tb);
}
template <class T> __device__ __forceinline__ void bior8_inv_2d(T (*x)[8], TbPtr tb) {
    bior8_pass<2, false, false>(x, tb); bior8_pass<2, false, true>(x, tb);
    bior8_pass<4, false, false>(x, tb); bior8_pass<4, false, true>(x, tb);
    bior8_pass<8, false, false>(x, tb); bior8_pass<8, false, true>(x, tb);
}

template <bool HAAR, bool BIOR = false>   /* BIOR: tau_2D = bior1.5 instead of the DCT */
__global__ __launch_bounds__(kDct8wThreads) void k_group_dct8w(GroupArgs a) {
    extern __shared__ float lds[];
    __shared__ float red[3][kDct8wThreads / 64];
    const int tid = threadIdx.x;
    const unsigned g = a.ref_begin + blockIdx.x;
    const int c = blockIdx.y;
    constexpr int A = 9, K2 = 64;
    const int N = a.N;
    const int nSx = (int)a.self_cnt[g];
    const size_t plane = (size_t)a.Wb * a.Hb;
    const int NP = nSx * A;               /* patches per stack */
    const int NPp = (N * A) | 1;          /* row stride of the [pq][patch] stack (float2 units), odd */
    v2f* stack = reinterpret_cast<v2f*>(lds);
    float* stackf = lds;
    const TbPtr tb = (TbPtr)a.tb;

    ShRef sh = group_shape(a, g);

#ifdef LFBM5D_PHASE_TIMING
    long long tc[6]; int tci = 0;
#define PHASE_MARK() do { if (tid == 0) tc[tci] = (long long)__builtin_readcyclecounter(); tci++; } while (0)
    PHASE_MARK();
#else
#define PHASE_MARK() do {} while (0)
#endif
    /* 1a: cooperative gather.  The unit is one 16-byte half of a patch row: 16 adjacent lanes fetch the 8 rows of
     * one patch (two lanes per 32-byte row, one cache-line request), instead of every lane walking its own
     * patch.  The pieces are parked in LDS, [image][patch][16 pieces], piece index XOR-ed with the patch
     * index so that phase 1b reads them without bank conflicts; the area is reused by the stack afterwards. */
    {
        __shared__ unsigned pos[kMaxN * kA3];
        for (int i = tid; i < NP; i += kDct8wThreads) pos[i] = a.gpos[(size_t)g * N * A + i];
        __syncthreads();
        constexpr int kItems = (kMaxN * kA3 * 16 + kDct8wThreads - 1) / kDct8wThreads;   /* 9 */
        f4u v0[kItems], v1[kItems];
#pragma unroll
        for (int j = 0; j < kItems; j++) {
            const int it = tid + j * kDct8wThreads, patch = it >> 4, piece = it & 15;
            if (patch < NP) {
                const unsigned p = pos[patch];
                const size_t off = ((size_t)(patch % A) * a.C + c) * plane + (p != 0xffffffffu ? p : 0u)
                                   + (size_t)(piece >> 1) * a.Wb + 4 * (piece & 1);
                v0[j] = *reinterpret_cast<const f4u*>(a.noisy + off);
                v1[j] = *reinterpret_cast<const f4u*>(a.basic + off);
            }
        }
        v4f* stage = reinterpret_cast<v4f*>(lds);
#pragma unroll
        for (int j = 0; j < kItems; j++) {
            const int it = tid + j * kDct8wThreads, patch = it >> 4, piece = it & 15;
            if (patch < NP) {
                const bool ok = pos[patch] != 0xffffffffu;     /* empty SAI / never-filled table column: zeros */
                const int slot = patch * 16 + (piece ^ (patch & 15));
                stage[slot] = ok ? v4f{v0[j].v[0], v0[j].v[1], v0[j].v[2], v0[j].v[3]} : v4f{0.f, 0.f, 0.f, 0.f};
                stage[NP * 16 + slot] = ok ? v4f{v1[j].v[0], v1[j].v[1], v1[j].v[2], v1[j].v[3]} : v4f{0.f, 0.f, 0.f, 0.f};
            }
        }
        __syncthreads();
    }
    /* 1b: forward 2-D DCT of both images, one thread per patch */
    {
        const int patch = tid;       /* NP <= 144 < 256 threads */
        v2f x[8][8];
        if (patch < NP) {
            const v4f* stage = reinterpret_cast<const v4f*>(lds);
#pragma unroll
            for (int piece = 0; piece < 16; piece++) {
                const v4f l = stage[patch * 16 + (piece ^ (patch & 15))], r = stage[NP * 16 + patch * 16 + (piece ^ (patch & 15))];
#pragma unroll
                for (int j = 0; j < 4; j++) x[piece >> 1][4 * (piece & 1) + j] = v2f{l[j], r[j]};
            }
        }
        __syncthreads();             /* every patch is in registers: the area becomes the stack */
        if (patch < NP) {
            if (BIOR) bior8_fwd_2d(x, tb);
            else {
#pragma unroll
                for (int i = 0; i < 8; i++) dct8_fwd_t(x[i]);
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    v2f col[8];
#pragma unroll
                    for (int i = 0; i < 8; i++) col[i] = x[i][j];
                    dct8_fwd_t(col);
#pragma unroll
                    for (int i = 0; i < 8; i++) x[i][j] = col[i];
                }
            }
            v2f* dst = stack + patch;
#pragma unroll
            for (int i = 0; i < 8; i++)
#pragma unroll
                for (int j = 0; j < 8; j++) dst[(i * 8 + j) * NPp] = x[i][j];
        }
    }
    __syncthreads();
    PHASE_MARK();

    /* 2: 4-D forward, one (n, pq) fibre of 9 float2 per thread */
    const bool use_sadct = a.tau4 == 6 && sh.use_sadct;
    const bool do_dct4 = a.tau4 == 5 || (a.tau4 == 6 && !use_sadct);
    const bool do_sa4 = !do_dct4 && a.tau4 == 6;
    if (do_dct4 || do_sa4) {
        for (int f = tid; f < nSx * K2; f += kDct8wThreads) {
            const int n = f / K2, pq = f % K2;
            v2f* base = stack + pq * NPp + n * A;
            v2f x[9];
#pragma unroll
            for (int st = 0; st < 9; st++) x[st] = base[st];
            if (do_dct4) dct9_fwd2(x, tb);
            else {   /* rare: shape-adaptive transform on the scalar path */
                float t9[9];
#pragma unroll
                for (int i = 0; i < 9; i++) t9[i] = x[i].x;
                sadct9_fwd(t9, sh, tb);
#pragma unroll
                for (int i = 0; i < 9; i++) { x[i].x = t9[i]; t9[i] = x[i].y; }
                sadct9_fwd(t9, sh, tb);
#pragma unroll
                for (int i = 0; i < 9; i++) x[i].y = t9[i];
            }
#pragma unroll
            for (int st = 0; st < 9; st++) base[st] = x[st];
        }
        __syncthreads();
    }

    PHASE_MARK();
    /* 3: 5th dimension + Wiener shrinkage, one (st, pq) fibre of nSx float2 per thread; result -> .y */
    float wacc = 0.0f, s1 = 0.0f, s2 = 0.0f;
    {
        const float sig = a.sigma[c];
        const float sig2 = sig * sig;
        const bool useSD = a.useSD != 0;
        for (int f = tid; f < A * K2; f += kDct8wThreads) {
            const int st = f / K2, pq = f % K2;
            const bool in_shape = !use_sadct || sh.mask_dct[st];
            const int base = pq * NPp + st;
            switch (nSx) {
                case 1:  wiener_fibre2<1, HAAR>(stack, base, A, a.tau5, sig2, in_shape, useSD, wacc, s1, s2, tb); break;
                case 2:  wiener_fibre2<2, HAAR>(stack, base, A, a.tau5, sig2, in_shape, useSD, wacc, s1, s2, tb); break;
                case 4:  wiener_fibre2<4, HAAR>(stack, base, A, a.tau5, sig2, in_shape, useSD, wacc, s1, s2, tb); break;
                case 8:  wiener_fibre2<8, HAAR>(stack, base, A, a.tau5, sig2, in_shape, useSD, wacc, s1, s2, tb); break;
                default: wiener_fibre2<16, HAAR>(stack, base, A, a.tau5, sig2, in_shape, useSD, wacc, s1, s2, tb); break;
            }
        }
    }
    for (int o = 32; o > 0; o >>= 1) { wacc += __shfl_xor(wacc, o); s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
    if ((tid & 63) == 0) { red[0][tid >> 6] = wacc; red[1][tid >> 6] = s1; red[2][tid >> 6] = s2; }
    __syncthreads();
    PHASE_MARK();
    /* 4: 4-D inverse of the filtered stack (.y), two (n, pq) fibres per thread: pq and pq + 32 */
    if (do_dct4 || do_sa4) {
        for (int f = tid; f < nSx * (K2 / 2); f += kDct8wThreads) {
            const int n = f / (K2 / 2), pq = f % (K2 / 2);
            float* b0 = stackf + 2 * (pq * NPp + n * A) + 1;
            float* b1 = stackf + 2 * ((pq + K2 / 2) * NPp + n * A) + 1;
            v2f x[9];
#pragma unroll
            for (int st = 0; st < 9; st++) x[st] = v2f{b0[2 * st], b1[2 * st]};
            if (do_dct4) dct9_inv2(x, tb);
            else {
                float t9[9];
#pragma unroll
                for (int i = 0; i < 9; i++) t9[i] = x[i].x;
                sadct9_inv(t9, sh, tb);
#pragma unroll
                for (int i = 0; i < 9; i++) { x[i].x = t9[i]; t9[i] = x[i].y; }
                sadct9_inv(t9, sh, tb);
#pragma unroll
                for (int i = 0; i < 9; i++) x[i].y = t9[i];
            }
#pragma unroll
            for (int st = 0; st < 9; st++) { b0[2 * st] = x[st].x; b1[2 * st] = x[st].y; }
        }
    }
    __syncthreads();

    if (tid == kDct8wThreads - 64) {   /* the last wave takes no part in phase 5 (NP / 2 <= 72 patch pairs): the weight costs nothing there */
        float w = 0.0f, m = 0.0f, q = 0.0f;
        for (int i = 0; i < kDct8wThreads / 64; i++) { w += red[0][i]; m += red[1][i]; q += red[2][i]; }
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

    PHASE_MARK();
    /* 5: inverse 2-D DCT + store, two patches per thread (patch, patch + NPh): filt[g][n][st][c][64] */
    const int NPh = (NP + 1) / 2;
    for (int pa = tid; pa < NPh; pa += kDct8wThreads) {
        const int pb = pa + NPh;
        const bool has_b = pb < NP;
        const float* sa = stackf + 2 * pa + 1;
        const float* sb = stackf + 2 * (has_b ? pb : pa) + 1;
        v2f x[8][8];
#pragma unroll
        for (int i = 0; i < 8; i++)
#pragma unroll
            for (int j = 0; j < 8; j++) x[i][j] = v2f{sa[2 * (i * 8 + j) * NPp], sb[2 * (i * 8 + j) * NPp]};
        if (BIOR) bior8_inv_2d(x, tb);
        else {
#pragma unroll
            for (int j = 0; j < 8; j++) {
                v2f col[8];
#pragma unroll
                for (int i = 0; i < 8; i++) col[i] = x[i][j];
                dct8_inv_t(col);
#pragma unroll
                for (int i = 0; i < 8; i++) x[i][j] = col[i];
            }
#pragma unroll
            for (int i = 0; i < 8; i++) dct8_inv_t(x[i]);
        }
        float4* oa = reinterpret_cast<float4*>(a.filt + ((size_t)g * N * A + pa) * a.C * K2 + (size_t)c * K2);
#pragma unroll
        for (int i = 0; i < 8; i++) {
            filt_put4(&oa[2 * i], make_float4(x[i][0].x, x[i][1].x, x[i][2].x, x[i][3].x));
            filt_put4(&oa[2 * i + 1], make_float4(x[i][4].x, x[i][5].x, x[i][6].x, x[i][7].x));
        }
        if (has_b) {
            float4* ob = reinterpret_cast<float4*>(a.filt + ((size_t)g * N * A + pb) * a.C * K2 + (size_t)c * K2);
#pragma unroll
            for (int i = 0; i < 8; i++) {
                filt_put4(&ob[2 * i], make_float4(x[i][0].y, x[i][1].y, x[i][2].y, x[i][3].y));
                filt_put4(&ob[2 * i + 1], make_float4(x[i][4].y, x[i][5].y, x[i][6].y, x[i][7].y));
            }
        }
    }
#ifdef LFBM5D_PHASE_TIMING
    PHASE_MARK();
    if (tid == 0) {
        for (int i = 0; i < 5; i++) atomicAdd(&a.counters[4 + i], (unsigned long long)(tc[i + 1] - tc[i]));
        atomicAdd(&a.counters[9], 1ull);
    }
#endif
}

/* ------------------------------------------------------------------------------------------
 * Wiener step, 8x8 DCT, second generation: the same arithmetic as k_group_dct8w with the 2-D stages dealt to ALL
 * threads.  k_group_dct8w runs the forward 2-D DCT with one thread per patch (144 of 256 threads busy, fed through an
 * LDS staging area) and the inverse with one thread per patch PAIR (72 of 256): those two phases were 63 % of its time
 * at a quarter to a half of the lanes.  Here a 2-D DCT is two passes over the LDS stack with an item = one 8-point
 * packed transform:
 *   1a  item = (row i, patch): the thread loads its 32-byte row of both images straight from the window (no staging),
 *       transforms the noisy / pilot pair, writes 8 float2 into the stack [coefficient][patch]
 *   1b  item = (column j, patch): 8-point transform down the column, in place
 *   2-4 as before (3x3 angular DCT per (n, pq) fibre; Haar + Wiener + inverse Haar per (st, pq) fibre; inverse 3x3)
 *   5a  item = (column j, patch pair): inverse transform of the filtered stack, two patches packed
 *   5b  item = (row i, patch pair): inverse transform along the row, two 32-byte stores
 * Workgroups are 512 threads (eight wavefronts, two workgroups per CU at the stack's 74 KiB: 16 waves per CU; 192 threads --
 * every phase divides evenly -- measured three times slower, odd wave counts leave SIMDs unevenly loaded).
 * Items are numbered patch-fastest so that the stack accesses of a wavefront are consecutive float2 (no bank conflicts);
 * the global accesses are 32-byte row segments either way.
 * ------------------------------------------------------------------------------------------ */
#ifndef LFBM5D_DCT8W2_THREADS
#define LFBM5D_DCT8W2_THREADS 512
#endif
constexpr int kDct8w2Threads = LFBM5D_DCT8W2_THREADS;

template <bool HAAR>
__global__ __launch_bounds__(kDct8w2Threads) void k_group_dct8w2(GroupArgs a) {
    extern __shared__ float lds[];
    __shared__ float red[3][kDct8w2Threads / 64];
    __shared__ unsigned pos[kMaxN * kA3];
    constexpr int TH = kDct8w2Threads;
    const int tid = threadIdx.x;
    const unsigned gi = xcd_group_index(a);
    if (gi >= a.n_groups) return;
    const unsigned g = a.ref_begin + gi;
    const int c = blockIdx.y;
    constexpr int A = 9, K2 = 64;
    const int N = a.N;
    const int nSx = (int)a.self_cnt[g];
    const size_t plane = (size_t)a.Wb * a.Hb;
    const int NP = nSx * A;               /* patches per stack */
    const int NPp = (N * A) | 1;          /* row stride of the [pq][patch] stack (float2 units), odd */
    v2f* stack = reinterpret_cast<v2f*>(lds);
    float* stackf = lds;
    const TbPtr tb = (TbPtr)a.tb;
    /* item index -> (row or column, patch): division by NP / NPh through a 32-bit reciprocal (exact for items < 2^16) */
    const unsigned rcpNP = 0xffffffffu / (unsigned)NP + 1u, rcpNPh = 0xffffffffu / (unsigned)((NP + 1) / 2) + 1u;

    ShRef sh = group_shape(a, g);
    for (int i = tid; i < NP; i += TH) pos[i] = a.gpos[(size_t)g * N * A + i];
    __syncthreads();
#ifdef LFBM5D_PHASE_TIMING
    long long tc[6]; int tci = 0;
    PHASE_MARK();
#endif

    /* 1a: rows.  All loads of a thread's items are issued before the first transform */
    {
        constexpr int kIt = (kMaxN * kA3 * 8 + TH - 1) / TH;   /* 6 */
        f4u n0[kIt], n1[kIt], b0[kIt], b1[kIt];
#pragma unroll
        for (int q = 0; q < kIt; q++) {
            const int it = tid + q * TH;
#ifndef LFBM5D_W2_LOAD_ROWFAST
            const int i = (int)__umulhi((unsigned)it, rcpNP), patch = it - i * NP;
#else
            const int patch = it >> 3, i = it & 7;     /* row-fastest (measured: the stack writes then conflict, 1.46 vs 1.44 ms) */
#endif
            if (it < NP * 8) {
                const unsigned p = pos[patch];
                const size_t off = ((size_t)(patch % A) * a.C + c) * plane + (p != 0xffffffffu ? p : 0u) + (size_t)i * a.Wb;
                n0[q] = *reinterpret_cast<const f4u*>(a.noisy + off); n1[q] = *reinterpret_cast<const f4u*>(a.noisy + off + 4);
                b0[q] = *reinterpret_cast<const f4u*>(a.basic + off); b1[q] = *reinterpret_cast<const f4u*>(a.basic + off + 4);
            }
        }
#pragma unroll
        for (int q = 0; q < kIt; q++) {
            const int it = tid + q * TH;
#ifndef LFBM5D_W2_LOAD_ROWFAST
            const int i = (int)__umulhi((unsigned)it, rcpNP), patch = it - i * NP;
#else
            const int patch = it >> 3, i = it & 7;
#endif
            if (it < NP * 8) {
                const bool ok = pos[patch] != 0xffffffffu;     /* empty SAI / never-filled table column: zeros */
                v2f x[8];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    x[j] = ok ? v2f{n0[q].v[j], b0[q].v[j]} : v2f{0.f, 0.f};
                    x[4 + j] = ok ? v2f{n1[q].v[j], b1[q].v[j]} : v2f{0.f, 0.f};
                }
                dct8_fwd_t(x);
                v2f* dst = stack + (i * 8) * NPp + patch;
#pragma unroll
                for (int j = 0; j < 8; j++) dst[j * NPp] = x[j];
            }
        }
    }
    __syncthreads();
    PHASE_MARK();
    /* 1b: columns, in place */
    for (int it = tid; it < NP * 8; it += TH) {
        const int j = (int)__umulhi((unsigned)it, rcpNP), patch = it - j * NP;
        v2f* col = stack + j * NPp + patch;
        v2f x[8];
#pragma unroll
        for (int i = 0; i < 8; i++) x[i] = col[(i * 8) * NPp];
        dct8_fwd_t(x);
#pragma unroll
        for (int i = 0; i < 8; i++) col[(i * 8) * NPp] = x[i];
    }
    __syncthreads();
    PHASE_MARK();

    /* 2: 4-D forward, one (n, pq) fibre of 9 float2 per thread */
    const bool use_sadct = a.tau4 == 6 && sh.use_sadct;
    const bool do_dct4 = a.tau4 == 5 || (a.tau4 == 6 && !use_sadct);
    const bool do_sa4 = !do_dct4 && a.tau4 == 6;
    if (do_dct4 || do_sa4) {
        for (int f = tid; f < nSx * K2; f += TH) {
            const int n = f / K2, pq = f % K2;
            v2f* base = stack + pq * NPp + n * A;
            v2f x[9];
#pragma unroll
            for (int st = 0; st < 9; st++) x[st] = base[st];
            if (do_dct4) dct9_fwd2(x, tb);
            else {   /* rare: shape-adaptive transform on the scalar path */
                float t9[9];
#pragma unroll
                for (int i = 0; i < 9; i++) t9[i] = x[i].x;
                sadct9_fwd(t9, sh, tb);
#pragma unroll
                for (int i = 0; i < 9; i++) { x[i].x = t9[i]; t9[i] = x[i].y; }
                sadct9_fwd(t9, sh, tb);
#pragma unroll
                for (int i = 0; i < 9; i++) x[i].y = t9[i];
            }
#pragma unroll
            for (int st = 0; st < 9; st++) base[st] = x[st];
        }
        __syncthreads();
    }

    /* 3: 5th dimension + Wiener shrinkage, one (st, pq) fibre of nSx float2 per thread; result -> .y */
    float wacc = 0.0f, s1 = 0.0f, s2 = 0.0f;
    {
        const float sig = a.sigma[c];
        const float sig2 = sig * sig;
        const bool useSD = a.useSD != 0;
        for (int f = tid; f < A * K2; f += TH) {
            const int st = f / K2, pq = f % K2;
            const bool in_shape = !use_sadct || sh.mask_dct[st];
            const int base = pq * NPp + st;
            switch (nSx) {
                case 1:  wiener_fibre2<1, HAAR>(stack, base, A, a.tau5, sig2, in_shape, useSD, wacc, s1, s2, tb); break;
                case 2:  wiener_fibre2<2, HAAR>(stack, base, A, a.tau5, sig2, in_shape, useSD, wacc, s1, s2, tb); break;
                case 4:  wiener_fibre2<4, HAAR>(stack, base, A, a.tau5, sig2, in_shape, useSD, wacc, s1, s2, tb); break;
                case 8:  wiener_fibre2<8, HAAR>(stack, base, A, a.tau5, sig2, in_shape, useSD, wacc, s1, s2, tb); break;
                default: wiener_fibre2<16, HAAR>(stack, base, A, a.tau5, sig2, in_shape, useSD, wacc, s1, s2, tb); break;
            }
        }
    }
    for (int o = 32; o > 0; o >>= 1) { wacc += __shfl_xor(wacc, o); s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
    if ((tid & 63) == 0) { red[0][tid >> 6] = wacc; red[1][tid >> 6] = s1; red[2][tid >> 6] = s2; }
    __syncthreads();
    /* 4: 4-D inverse of the filtered stack (.y), two (n, pq) fibres per thread: pq and pq + 32 */
    if (do_dct4 || do_sa4) {
        for (int f = tid; f < nSx * (K2 / 2); f += TH) {
            const int n = f / (K2 / 2), pq = f % (K2 / 2);
            float* b0 = stackf + 2 * (pq * NPp + n * A) + 1;
            float* b1 = stackf + 2 * ((pq + K2 / 2) * NPp + n * A) + 1;
            v2f x[9];
#pragma unroll
            for (int st = 0; st < 9; st++) x[st] = v2f{b0[2 * st], b1[2 * st]};
            if (do_dct4) dct9_inv2(x, tb);
            else {
                float t9[9];
#pragma unroll
                for (int i = 0; i < 9; i++) t9[i] = x[i].x;
                sadct9_inv(t9, sh, tb);
#pragma unroll
                for (int i = 0; i < 9; i++) { x[i].x = t9[i]; t9[i] = x[i].y; }
                sadct9_inv(t9, sh, tb);
#pragma unroll
                for (int i = 0; i < 9; i++) x[i].y = t9[i];
            }
#pragma unroll
            for (int st = 0; st < 9; st++) { b0[2 * st] = x[st].x; b1[2 * st] = x[st].y; }
        }
    }
    __syncthreads();

    if (tid == 0) {
        float w = 0.0f, m = 0.0f, q = 0.0f;
        for (int i = 0; i < TH / 64; i++) { w += red[0][i]; m += red[1][i]; q += red[2][i]; }
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

    PHASE_MARK();
    /* 5a: inverse transform down the columns of the filtered stack (.y), two patches (pa, pa + NPh) packed, in place */
    const int NPh = (NP + 1) / 2;
    for (int it = tid; it < NPh * 8; it += TH) {
        const int j = (int)__umulhi((unsigned)it, rcpNPh), pa = it - j * NPh;
        const int pb = pa + NPh < NP ? pa + NPh : pa;
        float* ca = stackf + 2 * (j * NPp + pa) + 1;
        float* cb = stackf + 2 * (j * NPp + pb) + 1;
        v2f x[8];
#pragma unroll
        for (int i = 0; i < 8; i++) x[i] = v2f{ca[2 * (i * 8) * NPp], cb[2 * (i * 8) * NPp]};
        dct8_inv_t(x);
#pragma unroll
        for (int i = 0; i < 8; i++) { ca[2 * (i * 8) * NPp] = x[i].x; if (pb != pa) cb[2 * (i * 8) * NPp] = x[i].y; }
    }
    __syncthreads();
    PHASE_MARK();
    /* 5b: inverse transform along the rows + store: filt[g][n][st][c][64].  Items are numbered row-fastest: the eight
     * lanes that hold the rows of one patch write its 256 contiguous bytes */
#ifdef LFBM5D_W2_ROWS_PAIRED
    for (int it = tid; it < NPh * 8; it += TH) {
        const int i = it / NPh, pa = it - i * NPh;
        const int pb = pa + NPh;
        const bool has_b = pb < NP;
        const float* ra = stackf + 2 * ((i * 8) * NPp + pa) + 1;
        const float* rb = stackf + 2 * ((i * 8) * NPp + (has_b ? pb : pa)) + 1;
        v2f x[8];
#pragma unroll
        for (int j = 0; j < 8; j++) x[j] = v2f{ra[2 * j * NPp], rb[2 * j * NPp]};
        dct8_inv_t(x);
        float4* oa = reinterpret_cast<float4*>(a.filt + ((size_t)g * N * A + pa) * a.C * K2 + (size_t)c * K2 + i * 8);
        filt_put4(&oa[0], make_float4(x[0].x, x[1].x, x[2].x, x[3].x));
        filt_put4(&oa[1], make_float4(x[4].x, x[5].x, x[6].x, x[7].x));
        if (has_b) {
            float4* ob = reinterpret_cast<float4*>(a.filt + ((size_t)g * N * A + pb) * a.C * K2 + (size_t)c * K2 + i * 8);
            filt_put4(&ob[0], make_float4(x[0].y, x[1].y, x[2].y, x[3].y));
            filt_put4(&ob[1], make_float4(x[4].y, x[5].y, x[6].y, x[7].y));
        }
    }
#else
    for (int it = tid; it < NP * 8; it += TH) {
        const int patch = it >> 3, i = it & 7;
        const float* ra = stackf + 2 * ((i * 8) * NPp + patch) + 1;
        float x[8];
#pragma unroll
        for (int j = 0; j < 8; j++) x[j] = ra[2 * j * NPp];
        dct8_inv_t(x);
        float4* oa = reinterpret_cast<float4*>(a.filt + ((size_t)g * N * A + patch) * a.C * K2 + (size_t)c * K2 + i * 8);
        filt_put4(&oa[0], make_float4(x[0], x[1], x[2], x[3]));
        filt_put4(&oa[1], make_float4(x[4], x[5], x[6], x[7]));
    }
#endif
#ifdef LFBM5D_PHASE_TIMING
    PHASE_MARK();
    if (tid == 0) {
        for (int i = 0; i < 5; i++) atomicAdd(&a.counters[4 + i], (unsigned long long)(tc[i + 1] - tc[i]));
        atomicAdd(&a.counters[9], 1ull);
    }
#endif
}

/* ------------------------------------------------------------------------------------------
 * Wiener step, 8x8 DCT, Haar along the matches, third generation: ONE image in LDS at a time.  k_group_dct8w2 holds the
 * noisy and the pilot stack side by side (float2, 74 KiB: two workgroups per CU) and the time of a pass follows the
 * number of INDEPENDENT workgroups on a CU, not the number of waves (one workgroup of 8 or 16 waves per CU: 2.2 / 2.1 ms;
 * two of 8 waves: 1.4 ms -- a workgroup spends its phases waiting on one kind of unit, a second one in another phase
 * fills it).  Here the pilot goes through the forward transforms first, its Wiener coefficients e^2 / (e^2 + sigma^2)
 * stay in REGISTERS of the thread that owns the (st, pq) fibre, the noisy image then takes the same LDS: a float stack
 * of 36.5 KiB, four workgroups per CU.  The arithmetic per value is that of k_group_dct8w2 (same transforms on packed
 * pairs -- here two PATCHES per lane where that kernel packs the two images -- same shrinkage).
 *   P1-P3  pilot:  rows from the window (item = row i of patches 2pp, 2pp + 1), columns in place, 3x3 angular
 *   P4     Haar of the pilot fibre -> coefficient (or, outside the SADCT shape, the pilot value: quirk 10) in registers
 *   P5-P7  noisy: as P1-P3
 *   P8     Haar of the noisy fibre, times the coefficient, inverse Haar, in place
 *   P9-P11 inverse 3x3 (fibres pq, pq + 32 packed), inverse columns (patch pairs), inverse rows + store
 * Stack [pq][patch] floats, row stride 146: pairs of patches are 8-byte aligned, and 146 = 18 mod 32 with the lane
 * numbering of the fibre phases (16 values of pq x 2 neighbours in n or st) touches 32 distinct banks.
 * ------------------------------------------------------------------------------------------ */
#ifndef LFBM5D_DCT8W3_THREADS
#define LFBM5D_DCT8W3_THREADS 256
#endif
constexpr int kDct8w3Threads = LFBM5D_DCT8W3_THREADS;
constexpr int kW3Stride = 146;
constexpr unsigned kW3Lds = 64 * kW3Stride * sizeof(float);
constexpr unsigned kW3Empty = 0xf0000000u;   /* byte offset of an absent patch: beyond any window this kernel is launched on */

/* the gathered rows of one image: item q of a thread = row i of the patch pair (2 pp, 2 pp + 1), 32 bytes of each */
template <int NS, int TH> struct W3Rows {
    static constexpr int kIt = (((NS * 9 + 1) / 2) * 8 + TH - 1) / TH;
    v4f ra0[kIt], ra1[kIt], rb0[kIt], rb1[kIt];
};
template <int NS, int TH>
__device__ __forceinline__ void w3_load_rows(__amdgpu_buffer_rsrc_t img, unsigned row_bytes, const unsigned* pos, int tid, W3Rows<NS, TH>& r) {
    constexpr int A = 9, NP = NS * A, NPh = (NP + 1) / 2;
    /* pos[] holds byte offsets into the window (out of range for an empty SAI / never-filled column: the buffer load returns zeros) */
#pragma unroll
    for (int q = 0; q < W3Rows<NS, TH>::kIt; q++) {
        const int it = tid + q * TH;
        if (it < NPh * 8) {
            const int i = it / NPh, pp = it - i * NPh;
            const int pA = 2 * pp, pB = pA + 1 < NP ? pA + 1 : pA;
#if defined(LFBM5D_W3_EXP) && (LFBM5D_W3_EXP & 4)   /* timing experiment: the same loads at consecutive addresses (eight lines per instruction instead of 64) */
            const int oa = (int)((pos[0] & ~15u) + (unsigned)((tid & 63) * 64 + q * 4096)), ob = oa + 32;
            (void)pA; (void)pB; (void)i;
#else
            const int oa = (int)(pos[pA] + (unsigned)i * row_bytes), ob = (int)(pos[pB] + (unsigned)i * row_bytes);
#endif
            r.ra0[q] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(img, oa, 0, 0));
            r.ra1[q] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(img, oa + 16, 0, 0));
            r.rb0[q] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(img, ob, 0, 0));
            r.rb1[q] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(img, ob + 16, 0, 0));
        }
    }
}
template <int NS, int TH, bool SA, bool U = false>   /* U: the unnormalised chain (full-shape groups only) */
__device__ __forceinline__ void w3_forward(__amdgpu_buffer_rsrc_t img, unsigned row_bytes, float* S, const unsigned* pos,
                                           int tid, ShRef sh, bool do_dct4, bool do_sa4, TbPtr tb, const W3Rows<NS, TH>* pre = nullptr) {
    constexpr int A = 9, NP = NS * A, NPh = (NP + 1) / 2, NPf = kW3Stride;
    constexpr int kIt = (NPh * 8 + TH - 1) / TH;
    {
        W3Rows<NS, TH> own;
        if (!pre) w3_load_rows<NS, TH>(img, row_bytes, pos, tid, own);
        const W3Rows<NS, TH>& R = pre ? *pre : own;
        const v4f (&ra0)[kIt] = R.ra0; const v4f (&ra1)[kIt] = R.ra1; const v4f (&rb0)[kIt] = R.rb0; const v4f (&rb1)[kIt] = R.rb1;
#pragma unroll
        for (int q = 0; q < kIt; q++) {
            const int it = tid + q * TH;
            if (it < NPh * 8) {
                const int i = it / NPh, pp = it - i * NPh;
                v2f x[8];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    x[j] = v2f{ra0[q][j], rb0[q][j]};
                    x[4 + j] = v2f{ra1[q][j], rb1[q][j]};
                }
                dct8_fwd_t(x);
                float* dst = S + (i * 8) * NPf + 2 * pp;
#pragma unroll
                for (int j = 0; j < 8; j++) *reinterpret_cast<v2f*>(dst + j * NPf) = x[j];
            }
        }
    }
    __syncthreads();
    for (int it = tid; it < NPh * 8; it += TH) {
        const int j = it / NPh, pp = it - j * NPh;
        float* col = S + j * NPf + 2 * pp;
        v2f x[8];
#pragma unroll
        for (int i = 0; i < 8; i++) x[i] = *reinterpret_cast<const v2f*>(col + (i * 8) * NPf);
        dct8_fwd_t(x);
#pragma unroll
        for (int i = 0; i < 8; i++) *reinterpret_cast<v2f*>(col + (i * 8) * NPf) = x[i];
    }
    __syncthreads();
    if (do_dct4 || do_sa4) {
        for (int f = tid; f < NS * 32; f += TH) {
            int n, pq;
            if (NS > 1) { pq = (f & 15) | ((f >> 1) & 16); n = ((f >> 4) & 1) | ((f >> 5) & ~1); }
            else { n = 0; pq = f; }
            float* b0 = S + pq * NPf + n * A;
            float* b1 = b0 + 32 * NPf;
            v2f x[9];
#pragma unroll
            for (int st = 0; st < 9; st++) x[st] = v2f{b0[st], b1[st]};
            if (U) dct9_fwd2_u(x);
            else if (do_dct4) dct9_fwd2_fast(x, tb);
            else {   /* rare: shape-adaptive transform on the scalar path */
                float t9[9];
#pragma unroll
                for (int i = 0; i < 9; i++) t9[i] = x[i].x;
                if (SA) sadct9_fwd_sel(t9, sh, tb); else sadct9_fwd(t9, sh, tb);
#pragma unroll
                for (int i = 0; i < 9; i++) { x[i].x = t9[i]; t9[i] = x[i].y; }
                if (SA) sadct9_fwd_sel(t9, sh, tb); else sadct9_fwd(t9, sh, tb);
#pragma unroll
                for (int i = 0; i < 9; i++) x[i].y = t9[i];
            }
#pragma unroll
            for (int st = 0; st < 9; st++) { b0[st] = x[st].x; b1[st] = x[st].y; }
        }
        __syncthreads();
    }
}

template <int NS, int TH, bool SA>
__device__ __forceinline__ void w3_body(const GroupArgs& a, float* S, const unsigned* pos, float (*red)[TH / 64], int tid,
                                        unsigned g, int c) {
    constexpr int A = 9, K2 = 64, NP = NS * A, NPh = (NP + 1) / 2, NPf = kW3Stride;
    static_assert(TH == 256, "the fibre phases deal st 0..3 / 4..7 to 256 threads");
    const int N = a.N;
    const unsigned win_bytes = (unsigned)((size_t)A * a.C * a.Wb * a.Hb * 4);
    const TbPtr tb = (TbPtr)a.tb;
    ShRef sh = group_shape(a, g);
    const bool use_sadct = a.tau4 == 6 && sh.use_sadct;
    const bool do_dct4 = a.tau4 == 5 || (a.tau4 == 6 && !use_sadct);
    const bool do_sa4 = !do_dct4 && a.tau4 == 6;
    const float sig = a.sigma[c];
    const float sig2 = sig * sig;
    const bool useSD = a.useSD != 0;

    w3_forward<NS, TH, SA>(__builtin_amdgcn_make_buffer_rsrc((void*)a.basic, 0, (int)win_bytes, 0x00020000u), a.Wb * 4u, S, pos, tid, sh, do_dct4, do_sa4, tb);
    /* fibres (st, pq): every thread owns st and st + 4 (st < 4) as a packed pair, the first wave also st = 8 */
    const int fpq = (tid & 15) | ((tid >> 1) & 48), fst = ((tid >> 4) & 1) | ((tid >> 6) & 2);
    float* const fbase = S + fpq * NPf + fst;
    float* const f8base = S + (tid & 63) * NPf + 8;
    const bool own8 = tid < 64;
    v2f vv[NS];
    float v8[NS];
    {
        v2f e[NS];
#pragma unroll
        for (int n = 0; n < NS; n++) e[n] = v2f{fbase[n * A], fbase[n * A + 4]};
        if (NS > 1) haar_fwd2<NS>(e);
#pragma unroll
        for (int n = 0; n < NS; n++) {
            const v2f value = e[n] * e[n], den = value + sig2;
            vv[n] = v2f{wiener_div(value.x, den.x), wiener_div(value.y, den.y)};
        }
        if (use_sadct) {   /* outside the shape the pilot's own coefficient passes through (quirk 10) */
#pragma unroll
            for (int n = 0; n < NS; n++) {
                if (!sh.mask_dct[fst]) vv[n].x = e[n].x;
                if (!sh.mask_dct[fst + 4]) vv[n].y = e[n].y;
            }
        }
    }
    if (own8) {
        float e[NS];
#pragma unroll
        for (int n = 0; n < NS; n++) e[n] = f8base[n * A];
        if (NS > 1) haar_fwd<NS>(e);
        const bool in8 = !use_sadct || sh.mask_dct[8];
#pragma unroll
        for (int n = 0; n < NS; n++) {
            const float value = e[n] * e[n];
            v8[n] = in8 ? wiener_div(value, value + sig2) : e[n];
        }
    }
    __syncthreads();
    w3_forward<NS, TH, SA>(__builtin_amdgcn_make_buffer_rsrc((void*)a.noisy, 0, (int)win_bytes, 0x00020000u), a.Wb * 4u, S, pos, tid, sh, do_dct4, do_sa4, tb);
    float wacc = 0.0f, s1 = 0.0f, s2 = 0.0f;
    {
        v2f o[NS];
#pragma unroll
        for (int n = 0; n < NS; n++) o[n] = v2f{fbase[n * A], fbase[n * A + 4]};
        if (NS > 1) haar_fwd2<NS>(o);
        if (!use_sadct) {
            v2f w2 = v2f{0.f, 0.f};
#pragma unroll
            for (int n = 0; n < NS; n++) { o[n] = o[n] * vv[n]; w2 += vv[n]; }
            wacc = w2.x + w2.y;
        } else {
            const bool ina = sh.mask_dct[fst], inb = sh.mask_dct[fst + 4];
#pragma unroll
            for (int n = 0; n < NS; n++) {
                o[n].x = ina ? o[n].x * vv[n].x : vv[n].x;
                o[n].y = inb ? o[n].y * vv[n].y : vv[n].y;
                wacc += (ina ? vv[n].x : 0.f) + (inb ? vv[n].y : 0.f);
            }
        }
        if (NS > 1) haar_inv2<NS>(o);
#pragma unroll
        for (int n = 0; n < NS; n++) { fbase[n * A] = o[n].x; fbase[n * A + 4] = o[n].y; }
        if (useSD) {
#pragma unroll
            for (int n = 0; n < NS; n++) { s1 += o[n].x + o[n].y; s2 += o[n].x * o[n].x + o[n].y * o[n].y; }
        }
    }
    if (own8) {
        float o[NS];
#pragma unroll
        for (int n = 0; n < NS; n++) o[n] = f8base[n * A];
        if (NS > 1) haar_fwd<NS>(o);
        if (!use_sadct || sh.mask_dct[8]) {
#pragma unroll
            for (int n = 0; n < NS; n++) { o[n] = o[n] * v8[n]; wacc += v8[n]; }
        } else {
#pragma unroll
            for (int n = 0; n < NS; n++) o[n] = v8[n];
        }
        if (NS > 1) haar_inv<NS>(o);
#pragma unroll
        for (int n = 0; n < NS; n++) f8base[n * A] = o[n];
        if (useSD) {
#pragma unroll
            for (int n = 0; n < NS; n++) { s1 += o[n]; s2 += o[n] * o[n]; }
        }
    }
    for (int o = 32; o > 0; o >>= 1) { wacc += __shfl_xor(wacc, o); s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
    if ((tid & 63) == 0) { red[0][tid >> 6] = wacc; red[1][tid >> 6] = s1; red[2][tid >> 6] = s2; }
    __syncthreads();
    if (do_dct4 || do_sa4) {
        for (int f = tid; f < NS * 32; f += TH) {
            int n, pq;
            if (NS > 1) { pq = (f & 15) | ((f >> 1) & 16); n = ((f >> 4) & 1) | ((f >> 5) & ~1); }
            else { n = 0; pq = f; }
            float* b0 = S + pq * NPf + n * A;
            float* b1 = b0 + 32 * NPf;
            v2f x[9];
#pragma unroll
            for (int st = 0; st < 9; st++) x[st] = v2f{b0[st], b1[st]};
            if (do_dct4) dct9_inv2_fast(x, tb);
            else {
                float t9[9];
#pragma unroll
                for (int i = 0; i < 9; i++) t9[i] = x[i].x;
                if (SA) sadct9_inv_sel(t9, sh, tb); else sadct9_inv(t9, sh, tb);
#pragma unroll
                for (int i = 0; i < 9; i++) { x[i].x = t9[i]; t9[i] = x[i].y; }
                if (SA) sadct9_inv_sel(t9, sh, tb); else sadct9_inv(t9, sh, tb);
#pragma unroll
                for (int i = 0; i < 9; i++) x[i].y = t9[i];
            }
#pragma unroll
            for (int st = 0; st < 9; st++) { b0[st] = x[st].x; b1[st] = x[st].y; }
        }
    }
    __syncthreads();
    if (tid == 0) {
        float w = 0.0f, m = 0.0f, q = 0.0f;
        for (int i = 0; i < TH / 64; i++) { w += red[0][i]; m += red[1][i]; q += red[2][i]; }
        float wx;
        if (a.useSD) {
            const float Nn = (float)(NS * A);
            const float res = (q - m * m / Nn) / (Nn - 1.0f);
            wx = res > 0.0f ? 1.0f / sqrtf(res) : 0.0f;
        } else {
            wx = w > 0.0f ? (sig > 0.0f ? 1.0f / (sig * sig * w) : 1.0f / w) : 1.0f;
        }
        a.wgt[(size_t)g * a.C + c] = wx;
        if (c == 0) {
            atomicAdd(&a.counters[0], (unsigned long long)NS);
            if (use_sadct) atomicAdd(&a.counters[1], 1ull);
        }
    }
    for (int it = tid; it < NPh * 8; it += TH) {
        const int j = it / NPh, pp = it - j * NPh;
        float* col = S + j * NPf + 2 * pp;
        v2f x[8];
#pragma unroll
        for (int i = 0; i < 8; i++) x[i] = *reinterpret_cast<const v2f*>(col + (i * 8) * NPf);
        dct8_inv_t(x);
#pragma unroll
        for (int i = 0; i < 8; i++) *reinterpret_cast<v2f*>(col + (i * 8) * NPf) = x[i];
    }
    __syncthreads();
    /* rows + store: filt[g][n][st][c][64].  Patches (pa, pa + NPh) packed, row-fastest: the eight lanes with the rows of one
     * patch write its 256 bytes */
    for (int it = tid; it < NPh * 8; it += TH) {
        const int pa = it >> 3, i = it & 7;
        const bool has_b = pa + NPh < NP;
        const float* ra = S + (i * 8) * NPf + pa;
        v2f x[8];
#pragma unroll
        for (int j = 0; j < 8; j++) x[j] = v2f{ra[j * NPf], ra[j * NPf + (NP > NPh ? NPh : 0)]};
        dct8_inv_t(x);
        v4f* oa = reinterpret_cast<v4f*>(a.filt + ((size_t)g * N * A + pa) * a.C * K2 + (size_t)c * K2 + i * 8);
        filt_put4_nt(oa, v4f{x[0].x, x[1].x, x[2].x, x[3].x});
        filt_put4_nt(oa + 1, v4f{x[4].x, x[5].x, x[6].x, x[7].x});
        if (has_b) {
            v4f* ob = oa + (size_t)NPh * a.C * (K2 / 4);
            filt_put4_nt(ob, v4f{x[0].y, x[1].y, x[2].y, x[3].y});
            filt_put4_nt(ob + 1, v4f{x[4].y, x[5].y, x[6].y, x[7].y});
        }
    }
}

/* Round 6: the same body for groups whose angular shape is the whole window (all but one in a thousand), on the UNNORMALISED
 * chain: angular 3x3 by additions (24 packed instructions instead of 39 forward, 33 instead of 42 backward), Haar by additions
 * forward and fma by 1, 2, 4, 8 backward (30 instead of 60 each).  A coefficient (st, n) in this scale is the reference's divided
 * by ht3_f[st] 2^(-l/2), l = haar_level(n): the shrinkage e^2 / (e^2 + sigma^2) compares in the coefficient's own scale with
 * sigma^2 2^l / ht3_f[st]^2 (four constants per fibre, formed once per thread), is itself scale-free, and the filtered values go
 * back through ht3_gf[st] / nSx.  No shape-adaptive code: k_group_dct8w3_u skips the groups k_group_shape has listed
 * (GroupArgs::sa_list) and k_group_dct8w3_list takes them through the normalised body behind it. */
template <int NS, int TH>
__device__ __forceinline__ void w3_body_u(const GroupArgs& a, float* S, const unsigned* pos, float (*red)[TH / 64], int tid, unsigned g, int c) {
    constexpr int A = 9, K2 = 64, NP = NS * A, NPh = (NP + 1) / 2, NPf = kW3Stride;
    constexpr int L = NS == 16 ? 4 : NS == 8 ? 3 : NS == 4 ? 2 : NS == 2 ? 1 : 0;
    static_assert(TH == 256, "the fibre phases deal st 0..3 / 4..7 to 256 threads");
    const int N = a.N;
    const unsigned win_bytes = (unsigned)((size_t)A * a.C * a.Wb * a.Hb * 4);
    const TbPtr tb = (TbPtr)a.tb;
    ShRef sh = group_shape(a, g);   /* (unused by the U forms; the signature of w3_forward) */
    const float sig = a.sigma[c];
    const float sig2 = sig * sig;
    const bool useSD = a.useSD != 0;

    w3_forward<NS, TH, false, true>(__builtin_amdgcn_make_buffer_rsrc((void*)a.basic, 0, (int)win_bytes, 0x00020000u), a.Wb * 4u, S, pos, tid, sh, true, false, tb);
    /* fibres (st, pq): every thread owns st and st + 4 (st < 4) as a packed pair, the first wave also st = 8 */
    const int fpq = (tid & 15) | ((tid >> 1) & 48), fst = ((tid >> 4) & 1) | ((tid >> 6) & 2);
    float* const fbase = S + fpq * NPf + fst;
    float* const f8base = S + (tid & 63) * NPf + 8;
    const bool own8 = tid < 64;
    /* sigma^2 in the scale of level l of this thread's fibres: sigma^2 2^l / F^2 */
    v2f kq[L + 1];
    float k8[L + 1];
    {
        const float fa = tb->ht3_f[fst], fb = tb->ht3_f[fst + 4], f8 = tb->ht3_f[8];
        const v2f q = v2f{sig2 / (fa * fa), sig2 / (fb * fb)};
        const float q8 = sig2 / (f8 * f8);
#pragma unroll
        for (int l = 0; l <= L; l++) { kq[l] = q * (float)(1 << l); k8[l] = q8 * (float)(1 << l); }
    }
    v2f vv[NS];
    float v8[NS];
    {
        v2f e[NS];
#pragma unroll
        for (int n = 0; n < NS; n++) e[n] = v2f{fbase[n * A], fbase[n * A + 4]};
        haar_fwd_u<NS>(e);
#pragma unroll
        for (int n = 0; n < NS; n++) {
            const v2f value = e[n] * e[n], den = value + kq[haar_level<NS>(n)];
            vv[n] = v2f{wiener_div(value.x, den.x), wiener_div(value.y, den.y)};
        }
    }
    if (own8) {
        float e[NS];
#pragma unroll
        for (int n = 0; n < NS; n++) e[n] = f8base[n * A];
        haar_fwd_u<NS>(e);
#pragma unroll
        for (int n = 0; n < NS; n++) {
            const float value = e[n] * e[n];
            v8[n] = wiener_div(value, value + k8[haar_level<NS>(n)]);
        }
    }
    __syncthreads();
    /* (asking for these rows before the pilot's fibre phase -- 128 VGPRs, still four waves per SIMD -- changes nothing: 1.00 ms either way;
     * timing builds, profiles/r06_f_w3_experiments.txt: without the gathers 0.80, with them at consecutive addresses 0.92, without the stores 0.92) */
    w3_forward<NS, TH, false, true>(__builtin_amdgcn_make_buffer_rsrc((void*)a.noisy, 0, (int)win_bytes, 0x00020000u), a.Wb * 4u, S, pos, tid, sh, true, false, tb);
    float wacc = 0.0f, s1 = 0.0f, s2 = 0.0f;
    const float inv_n = 1.0f / (float)NS;
    {
        v2f o[NS];
#pragma unroll
        for (int n = 0; n < NS; n++) o[n] = v2f{fbase[n * A], fbase[n * A + 4]};
        haar_fwd_u<NS>(o);
        v2f w2 = v2f{0.f, 0.f};
#pragma unroll
        for (int n = 0; n < NS; n++) { o[n] = o[n] * vv[n]; w2 += vv[n]; }
        wacc = w2.x + w2.y;
        haar_inv_u<NS>(o);
#pragma unroll
        for (int n = 0; n < NS; n++) { fbase[n * A] = o[n].x; fbase[n * A + 4] = o[n].y; }
        if (useSD) {   /* sd_weighting_5d sums the filtered 4-D coefficients in the reference's scale */
            const v2f f2 = v2f{tb->ht3_f[fst], tb->ht3_f[fst + 4]} * inv_n;
#pragma unroll
            for (int n = 0; n < NS; n++) { const v2f y = o[n] * f2; s1 += y.x + y.y; s2 += y.x * y.x + y.y * y.y; }
        }
    }
    if (own8) {
        float o[NS];
#pragma unroll
        for (int n = 0; n < NS; n++) o[n] = f8base[n * A];
        haar_fwd_u<NS>(o);
#pragma unroll
        for (int n = 0; n < NS; n++) { o[n] = o[n] * v8[n]; wacc += v8[n]; }
        haar_inv_u<NS>(o);
#pragma unroll
        for (int n = 0; n < NS; n++) f8base[n * A] = o[n];
        if (useSD) {
            const float f2 = tb->ht3_f[8] * inv_n;
#pragma unroll
            for (int n = 0; n < NS; n++) { const float y = o[n] * f2; s1 += y; s2 += y * y; }
        }
    }
    for (int o = 32; o > 0; o >>= 1) { wacc += __shfl_xor(wacc, o); s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
    if ((tid & 63) == 0) { red[0][tid >> 6] = wacc; red[1][tid >> 6] = s1; red[2][tid >> 6] = s2; }
    __syncthreads();
    for (int f = tid; f < NS * 32; f += TH) {
        int n, pq;
        if (NS > 1) { pq = (f & 15) | ((f >> 1) & 16); n = ((f >> 4) & 1) | ((f >> 5) & ~1); }
        else { n = 0; pq = f; }
        float* b0 = S + pq * NPf + n * A;
        float* b1 = b0 + 32 * NPf;
        v2f x[9];
#pragma unroll
        for (int st = 0; st < 9; st++) x[st] = v2f{b0[st], b1[st]};
        dct9_inv2_u(x, tb, inv_n);
#pragma unroll
        for (int st = 0; st < 9; st++) { b0[st] = x[st].x; b1[st] = x[st].y; }
    }
    __syncthreads();
    if (tid == 0) {
        float w = 0.0f, m = 0.0f, q = 0.0f;
        for (int i = 0; i < TH / 64; i++) { w += red[0][i]; m += red[1][i]; q += red[2][i]; }
        float wx;
        if (a.useSD) {
            const float Nn = (float)(NS * A);
            const float res = (q - m * m / Nn) / (Nn - 1.0f);
            wx = res > 0.0f ? 1.0f / sqrtf(res) : 0.0f;
        } else {
            wx = w > 0.0f ? (sig > 0.0f ? 1.0f / (sig * sig * w) : 1.0f / w) : 1.0f;
        }
        a.wgt[(size_t)g * a.C + c] = wx;
        if (c == 0) atomicAdd(&a.counters[0], (unsigned long long)NS);
    }
    for (int it = tid; it < NPh * 8; it += TH) {
        const int j = it / NPh, pp = it - j * NPh;
        float* col = S + j * NPf + 2 * pp;
        v2f x[8];
#pragma unroll
        for (int i = 0; i < 8; i++) x[i] = *reinterpret_cast<const v2f*>(col + (i * 8) * NPf);
        dct8_inv_t(x);
#pragma unroll
        for (int i = 0; i < 8; i++) *reinterpret_cast<v2f*>(col + (i * 8) * NPf) = x[i];
    }
    __syncthreads();
    for (int it = tid; it < NPh * 8; it += TH) {
        const int pa = it >> 3, i = it & 7;
        const bool has_b = pa + NPh < NP;
        const float* ra = S + (i * 8) * NPf + pa;
        v2f x[8];
#pragma unroll
        for (int j = 0; j < 8; j++) x[j] = v2f{ra[j * NPf], ra[j * NPf + (NP > NPh ? NPh : 0)]};
        dct8_inv_t(x);
#if defined(LFBM5D_W3_EXP) && (LFBM5D_W3_EXP & 2)   /* timing experiment: the whole workgroup stores into one patch's 256 bytes (no HBM write stream) */
        v4f* oa = reinterpret_cast<v4f*>(a.filt + (size_t)g * N * A * a.C * K2 + i * 8);
#else
        v4f* oa = reinterpret_cast<v4f*>(a.filt + ((size_t)g * N * A + pa) * a.C * K2 + (size_t)c * K2 + i * 8);
#endif
        filt_put4_nt(oa, v4f{x[0].x, x[1].x, x[2].x, x[3].x});
        filt_put4_nt(oa + 1, v4f{x[4].x, x[5].x, x[6].x, x[7].x});
        if (has_b) {
#if defined(LFBM5D_W3_EXP) && (LFBM5D_W3_EXP & 2)
            v4f* ob = oa;
#else
            v4f* ob = oa + (size_t)NPh * a.C * (K2 / 4);
#endif
            filt_put4_nt(ob, v4f{x[0].y, x[1].y, x[2].y, x[3].y});
            filt_put4_nt(ob + 1, v4f{x[4].y, x[5].y, x[6].y, x[7].y});
        }
    }
}

#ifndef LFBM5D_W3SA_WAVES
#define LFBM5D_W3SA_WAVES 4   /* 128 VGPRs (a few spills): four workgroups per CU beat 129 without */
#endif
template <bool SA>   /* SA: for windows with an empty SAI (every group shape-adaptive): the transform inline, in registers */
__global__ __launch_bounds__(kDct8w3Threads) __attribute__((amdgpu_waves_per_eu(SA ? LFBM5D_W3SA_WAVES : 1))) void k_group_dct8w3(GroupArgs a) {
    extern __shared__ float lds[];
    __shared__ float red[3][kDct8w3Threads / 64];
    __shared__ unsigned pos[kMaxN * kA3];
    constexpr int TH = kDct8w3Threads;
    const int tid = threadIdx.x;
    const unsigned gi = xcd_group_index(a);
    if (gi >= a.n_groups) return;
    const unsigned g = a.ref_begin + gi;
    const int c = blockIdx.y;
    const int nSx = (int)a.self_cnt[g];
    const unsigned plane = a.Wb * a.Hb;
    for (int i = tid; i < nSx * 9; i += TH) {
        const unsigned p = a.gpos[(size_t)g * a.N * 9 + i];
        pos[i] = p != 0xffffffffu ? (((unsigned)(i % 9) * a.C + c) * plane + p) * 4u : kW3Empty;
    }
    __syncthreads();
#ifdef LFBM5D_W3_ONLY16
    w3_body<16, TH, SA>(a, lds, pos, red, tid, g, c); return;
#endif
    switch (nSx) {
        case 1:  w3_body<1, TH, SA>(a, lds, pos, red, tid, g, c); break;
        case 2:  w3_body<2, TH, SA>(a, lds, pos, red, tid, g, c); break;
        case 4:  w3_body<4, TH, SA>(a, lds, pos, red, tid, g, c); break;
        case 8:  w3_body<8, TH, SA>(a, lds, pos, red, tid, g, c); break;
        default: w3_body<16, TH, SA>(a, lds, pos, red, tid, g, c); break;
    }
}

/* ordinary windows (round 6): full-shape groups on the unnormalised chain, the listed shape-adaptive ones left to k_group_dct8w3_list */
#ifndef LFBM5D_W3U_WAVES
#define LFBM5D_W3U_WAVES 0
#endif
#if LFBM5D_W3U_WAVES > 0
__global__ __launch_bounds__(kDct8w3Threads) __attribute__((amdgpu_waves_per_eu(LFBM5D_W3U_WAVES, LFBM5D_W3U_WAVES))) void k_group_dct8w3_u(GroupArgs a) {
#else
__global__ __launch_bounds__(kDct8w3Threads) void k_group_dct8w3_u(GroupArgs a) {
#endif
    extern __shared__ float lds[];
    __shared__ float red[3][kDct8w3Threads / 64];
    __shared__ unsigned pos[kMaxN * kA3];
    constexpr int TH = kDct8w3Threads;
    const int tid = threadIdx.x;
    const unsigned gi = xcd_group_index(a);
    if (gi >= a.n_groups) return;
    const unsigned g = a.ref_begin + gi;
    if (a.tau4 == 6 && group_shape(a, g).use_sadct) return;   /* (uniform) */
    const int c = blockIdx.y;
    const int nSx = (int)a.self_cnt[g];
    const unsigned plane = a.Wb * a.Hb;
    for (int i = tid; i < nSx * 9; i += TH) {
        const unsigned p = a.gpos[(size_t)g * a.N * 9 + i];
        pos[i] = p != 0xffffffffu ? (((unsigned)(i % 9) * a.C + c) * plane + p) * 4u : kW3Empty;
#if defined(LFBM5D_W3_EXP) && (LFBM5D_W3_EXP & 1)   /* timing experiment: every patch absent -- the gathers go out of range and move nothing */
        pos[i] = kW3Empty;
#endif
    }
    __syncthreads();
    switch (nSx) {
        case 1:  w3_body_u<1, TH>(a, lds, pos, red, tid, g, c); break;
        case 2:  w3_body_u<2, TH>(a, lds, pos, red, tid, g, c); break;
        case 4:  w3_body_u<4, TH>(a, lds, pos, red, tid, g, c); break;
        case 8:  w3_body_u<8, TH>(a, lds, pos, red, tid, g, c); break;
        default: w3_body_u<16, TH>(a, lds, pos, red, tid, g, c); break;
    }
}
constexpr unsigned kW3ListBlocks = 256;
__global__ __launch_bounds__(kDct8w3Threads) void k_group_dct8w3_list(GroupArgs a) {
    extern __shared__ float lds[];
    __shared__ float red[3][kDct8w3Threads / 64];
    __shared__ unsigned pos[kMaxN * kA3];
    constexpr int TH = kDct8w3Threads;
    const int tid = threadIdx.x;
    const unsigned n_list = a.sa_list[0];
    const int c = blockIdx.y;
    for (unsigned li = blockIdx.x; li < n_list; li += gridDim.x) {
        const unsigned g = a.sa_list[1u + li] & 0x1fffffffu;
        const int nSx = (int)a.self_cnt[g];
        const unsigned plane = a.Wb * a.Hb;
        for (int i = tid; i < nSx * 9; i += TH) {
            const unsigned p = a.gpos[(size_t)g * a.N * 9 + i];
            pos[i] = p != 0xffffffffu ? (((unsigned)(i % 9) * a.C + c) * plane + p) * 4u : kW3Empty;
        }
        __syncthreads();
        switch (nSx) {
            case 1:  w3_body<1, TH, true>(a, lds, pos, red, tid, g, c); break;
            case 2:  w3_body<2, TH, true>(a, lds, pos, red, tid, g, c); break;
            case 4:  w3_body<4, TH, true>(a, lds, pos, red, tid, g, c); break;
            case 8:  w3_body<8, TH, true>(a, lds, pos, red, tid, g, c); break;
            default: w3_body<16, TH, true>(a, lds, pos, red, tid, g, c); break;
        }
        __syncthreads();   /* pos / red / the stack are reused */
    }
}

/* ------------------------------------------------------------------------------------------
 * Per-SAI BM3D flavour, 8x8 patches (LFBM3Ddenoising's parameters: bm3d.cpp:315-690 with kHard = kWien = 8).
 * A group is nSx <= 32 patches of ONE image, so a whole group fits a WAVEFRONT: four groups per workgroup, no
 * workgroup barrier anywhere.  Lane = patch for the 2-D stages (the 8x8 patch and its transform in registers, the
 * noisy / pilot pair packed in the Wiener step), lane = coefficient for the Hadamard + shrinkage along the stack
 * (ht_filtering_hadamard :914-966, wiener_filtering_hadamard :980-1027), through an LDS stack [coefficient][patch]
 * of the wave's own.  The generic group kernel spends a 256-thread workgroup on such a group.
 * ------------------------------------------------------------------------------------------ */
constexpr int kBm3dWaves = 4;
template <int STEP> struct Bm3dT { typedef float type; };
template <> struct Bm3dT<2> { typedef v2f type; };
__device__ __forceinline__ float bm3d_first(float x) { return x; }
__device__ __forceinline__ float bm3d_first(v2f x) { return x.x; }

template <int STEP, bool BIOR>
__global__ __launch_bounds__(64 * kBm3dWaves) void k_group_bm3d8(GroupArgs a) {
    extern __shared__ float lds[];
    typedef typename Bm3dT<STEP>::type T;
    constexpr int K2 = 64, ST = kMaxN3 + 1;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const unsigned gi = blockIdx.x * kBm3dWaves + wave;
    if (gi >= a.n_groups) return;                      /* the whole wavefront leaves: nothing below synchronises waves */
    const unsigned g = a.ref_begin + gi;
    const int c = blockIdx.y;
    const int N = a.N, nSx = (int)a.self_cnt[g];
    const size_t plane = (size_t)a.Wb * a.Hb;
    const TbPtr tb = (TbPtr)a.tb;
    T* S = reinterpret_cast<T*>(lds) + (size_t)wave * K2 * ST;

    /* A: lane = patch: load, forward 2-D transform, scatter to the stack */
    if (lane < nSx) {
        const unsigned p = a.gpos[(size_t)g * N + lane];
        const bool ok = p != 0xffffffffu;              /* never-filled table column: zeros (bm3d.cpp:737, :857) */
        const size_t off = (size_t)c * plane + (ok ? p : 0u);
        T x[8][8];
#pragma unroll
        for (int i = 0; i < 8; i++)
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const f4u q0 = *reinterpret_cast<const f4u*>(a.noisy + off + (size_t)i * a.Wb + 4 * h);
                if (STEP == 2) {
                    const f4u q1 = *reinterpret_cast<const f4u*>(a.basic + off + (size_t)i * a.Wb + 4 * h);
#pragma unroll
                    for (int e = 0; e < 4; e++) reinterpret_cast<v2f&>(x[i][4 * h + e]) = ok ? v2f{q0.v[e], q1.v[e]} : v2f{0.0f, 0.0f};
                } else {
#pragma unroll
                    for (int e = 0; e < 4; e++) reinterpret_cast<float&>(x[i][4 * h + e]) = ok ? q0.v[e] : 0.0f;
                }
            }
        if (BIOR) bior8_fwd_2d(x, tb);
        else {
#pragma unroll
            for (int i = 0; i < 8; i++) dct8_fwd_t(x[i]);
#pragma unroll
            for (int j = 0; j < 8; j++) {
                T col[8];
#pragma unroll
                for (int i = 0; i < 8; i++) col[i] = x[i][j];
                dct8_fwd_t(col);
#pragma unroll
                for (int i = 0; i < 8; i++) x[i][j] = col[i];
            }
        }
#pragma unroll
        for (int i = 0; i < 8; i++)
#pragma unroll
            for (int j = 0; j < 8; j++) S[(i * 8 + j) * ST + lane] = x[i][j];
    }
    __builtin_amdgcn_wave_barrier();

    /* B: lane = coefficient: Hadamard along the stack, shrinkage, inverse; filtered value back in place (Wiener: into .y) */
    float wacc = 0.0f, s1 = 0.0f, s2 = 0.0f;
    {
        const float sig = a.sigma[c];
        const float Tthr = a.lambda * sig;             /* * sqrt(nSx) inside shrink_fibre (bm3d.cpp:941) */
        T* F = S + lane * ST;
        auto fibre = [&](auto ns_tag) {
            constexpr int NS = decltype(ns_tag)::value;
            float o[NS], e[NS];
#pragma unroll
            for (int n = 0; n < NS; n++) {
                if (STEP == 2) { const v2f t = reinterpret_cast<const v2f&>(F[n]); o[n] = t.x; e[n] = t.y; }
                else { o[n] = reinterpret_cast<const float&>(F[n]); e[n] = 0.0f; }
            }
            shrink_fibre<NS, STEP>(o, e, 8u, Tthr, sig * sig, true, wacc, tb);
#pragma unroll
            for (int n = 0; n < NS; n++) {
                const float r = STEP == 1 ? o[n] : e[n];
                s1 += r; s2 += r * r;
                if (STEP == 2) reinterpret_cast<v2f&>(F[n]).y = r; else reinterpret_cast<float&>(F[n]) = r;
            }
        };
        switch (nSx) {
            case 2:  fibre(std::integral_constant<int, 2>{}); break;
            case 4:  fibre(std::integral_constant<int, 4>{}); break;
            case 8:  fibre(std::integral_constant<int, 8>{}); break;
            case 16: fibre(std::integral_constant<int, 16>{}); break;
            default: fibre(std::integral_constant<int, 32>{}); break;
        }
    }
    for (int o = 32; o > 0; o >>= 1) { wacc += __shfl_xor(wacc, o); s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
    if (lane == 0) {
        float wx;
        if (a.useSD) {                                 /* sd_weighting, bm3d.cpp:1345-1373 */
            const float Nn = (float)(nSx * K2);
            const float res = (s2 - s1 * s1 / Nn) / (Nn - 1.0f);
            wx = res > 0.0f ? 1.0f / sqrtf(res) : 0.0f;
        } else {
            const float sig = a.sigma[c];
            wx = wacc > 0.0f ? (sig > 0.0f ? 1.0f / (sig * sig * wacc) : 1.0f / wacc) : 1.0f;
        }
        a.wgt[(size_t)g * a.C + c] = wx;
        if (c == 0) atomicAdd(&a.counters[0], (unsigned long long)nSx);
    }
    __builtin_amdgcn_wave_barrier();

    /* C: lane = patch: inverse 2-D transform of the filtered coefficients, 16-byte stores: filt[g][n][c][64] */
    if (lane < nSx) {
        float x[8][8];
#pragma unroll
        for (int i = 0; i < 8; i++)
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const T t = S[(i * 8 + j) * ST + lane];
                if (STEP == 2) x[i][j] = reinterpret_cast<const v2f&>(t).y; else x[i][j] = bm3d_first(t);
            }
        if (BIOR) bior8_inv_2d(x, tb);
        else {
#pragma unroll
            for (int j = 0; j < 8; j++) {
                float col[8];
#pragma unroll
                for (int i = 0; i < 8; i++) col[i] = x[i][j];
                dct8_inv_t(col);
#pragma unroll
                for (int i = 0; i < 8; i++) x[i][j] = col[i];
            }
#pragma unroll
            for (int i = 0; i < 8; i++) dct8_inv_t(x[i]);
        }
        float4* dst = reinterpret_cast<float4*>(a.filt + (((size_t)g * N + lane) * a.C + c) * K2);
#pragma unroll
        for (int i = 0; i < 8; i++) {
            filt_put4(&dst[2 * i], make_float4(x[i][0], x[i][1], x[i][2], x[i][3]));
            filt_put4(&dst[2 * i + 1], make_float4(x[i][4], x[i][5], x[i][6], x[i][7]));
        }
    }
}

} /* namespace */

hipError_t prepare_group_wiener() {
    const void* fns[] = {
        reinterpret_cast<const void*>(&k_group_dct8<1>),
        reinterpret_cast<const void*>(&k_group_dct8w2<true>), reinterpret_cast<const void*>(&k_group_dct8w2<false>),
        reinterpret_cast<const void*>(&k_group_dct8w<true, true>), reinterpret_cast<const void*>(&k_group_dct8w<false, true>),
        reinterpret_cast<const void*>(&k_group_bm3d8<2, true>), reinterpret_cast<const void*>(&k_group_bm3d8<2, false>)};
    for (const void* f : fns) {
        const hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, kDedicatedLdsLimit);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

hipError_t launch_group_wiener(hipStream_t s, const GroupArgs& a, bool all_sa, bool* launched) {
    *launched = true;
    if (a.tau2 == 7 && a.k == 8 && a.A == 9 && a.step == 2 && a.N <= (unsigned)kMaxN) {   /* 8x8 bior1.5, Wiener step: round 1's kernel with the wavelet in its 2-D stage */
        const size_t l8 = (size_t)2 * 64 * ((a.N * 9) | 1) * sizeof(float);
        if (a.tau5 == 9) hipLaunchKernelGGL((k_group_dct8w<true, true>), dim3(a.n_groups, a.C), dim3(kDct8wThreads), l8, s, a);
        else             hipLaunchKernelGGL((k_group_dct8w<false, true>), dim3(a.n_groups, a.C), dim3(kDct8wThreads), l8, s, a);
        return hipGetLastError();
    }
    if (a.tau2 == 5 && a.k == 8 && a.A == 9 && a.N <= (unsigned)kMaxN) {   /* 8x8 DCT */
        const size_t l8 = (size_t)(a.step == 2 ? 2 : 1) * 64 * ((a.N * 9) | 1) * sizeof(float);
        if (a.step == 2) {
            const unsigned gx = ((a.n_groups + 7) / 8) * 8;   /* xcd_group_index */
            /* the README's Wiener step: k_group_dct8w3.  Its two-image predecessor k_group_dct8w2 stays for Hadamard / DCT fibres and for
             * windows of 1.9 GB and more (32-bit offsets); option dct8w_v2: test hook, that kernel for every configuration.  (Round 1's
             * k_group_dct8w and the unpacked k_group_dct8<2> were retired as Wiener DCT kernels in round 5: nothing selected them.) */
            if (a.tau5 == 9 && (size_t)9 * a.C * a.Wb * a.Hb * 4 < 0x70000000ull && !(a.opt & kOptDct8wV2)) {
                if (all_sa) hipLaunchKernelGGL(k_group_dct8w3<true>, dim3(gx, a.C), dim3(kDct8w3Threads), kW3Lds, s, a);
#ifdef LFBM5D_W3_NORMALISED   /* build flag: round 3-5's normalised body for every group (A/B runs) */
                else        hipLaunchKernelGGL(k_group_dct8w3<false>, dim3(gx, a.C), dim3(kDct8w3Threads), kW3Lds, s, a);
#else
                else if (a.tau4 == 5 || a.tau4 == 6) {
                    if (!a.sa_list) return hipErrorInvalidValue;
                    hipLaunchKernelGGL(k_group_dct8w3_u, dim3(gx, a.C), dim3(kDct8w3Threads), kW3Lds, s, a);
                    if (a.tau4 == 6) hipLaunchKernelGGL(k_group_dct8w3_list, dim3(kW3ListBlocks, a.C), dim3(kDct8w3Threads), kW3Lds, s, a);   /* the groups skipped above (usually none) */
                }
                else        hipLaunchKernelGGL(k_group_dct8w3<false>, dim3(gx, a.C), dim3(kDct8w3Threads), kW3Lds, s, a);   /* tau_4D = id */
#endif
            }
            else if (a.tau5 == 9) hipLaunchKernelGGL((k_group_dct8w2<true>), dim3(gx, a.C), dim3(kDct8w2Threads), l8, s, a);
            else                  hipLaunchKernelGGL((k_group_dct8w2<false>), dim3(gx, a.C), dim3(kDct8w2Threads), l8, s, a);
            return hipGetLastError();
        }
        hipLaunchKernelGGL(k_group_dct8<1>, dim3(a.n_groups, a.C), dim3(kDct8Threads), l8, s, a);   /* hard-thresholding step: one thread per patch for the 2-D stage */
        return hipGetLastError();
    }
    if (a.bm3d && a.A == 1 && a.k == 8 && a.tau5 == 8 && (a.tau2 == 5 || a.tau2 == 7)) {   /* per-SAI BM3D, 8x8 patches: a group per wavefront */
        const dim3 grid((a.n_groups + kBm3dWaves - 1) / kBm3dWaves, a.C), block(64 * kBm3dWaves);
        const size_t lb = (size_t)kBm3dWaves * 64 * (kMaxN3 + 1) * (a.step == 2 ? sizeof(v2f) : sizeof(float));
        if (a.step == 2) { if (a.tau2 == 7) hipLaunchKernelGGL((k_group_bm3d8<2, true>), grid, block, lb, s, a); else hipLaunchKernelGGL((k_group_bm3d8<2, false>), grid, block, lb, s, a); }
        else             { if (a.tau2 == 7) hipLaunchKernelGGL((k_group_bm3d8<1, true>), grid, block, lb, s, a); else hipLaunchKernelGGL((k_group_bm3d8<1, false>), grid, block, lb, s, a); }
        return hipGetLastError();
    }
    *launched = false;
    return hipSuccess;
}
} /* namespace lfbm5d */
