/*
 * lfbm5d_group_device.h -- device-side building blocks of the group kernels (lfbm5d_group_*.hip): the per-group SADCT record,
 * the 3x3 / general angular transforms (core:1862-2264), Haar / Hadamard / DCT along the stack with hard thresholding and
 * Wiener shrinkage (core:2281-3123), bior1.5 and the 2-D DCT (lib_transforms.cpp:46-277, bm3d.cpp:705-895, :1039-1086).
 * Internal to liblfbm5d_hip.so; every translation unit that includes it gets its own copies (anonymous namespace).
 */
#ifndef LFBM5D_GROUP_DEVICE_H
#define LFBM5D_GROUP_DEVICE_H

#include "lfbm5d_kernels.h"

#include <algorithm>
#include <cstdlib>
#include <type_traits>

namespace lfbm5d {

/* dispatch pieces of launch_group (lfbm5d_group_generic.hip): the dedicated kernels of the hard-thresholding step
 * (lfbm5d_group_ht.hip) and of the Wiener step / 8x8 patches / per-SAI BM3D (lfbm5d_group_wiener.hip).  *launched = false: the
 * configuration is not theirs */
hipError_t launch_group_ht(hipStream_t s, const GroupArgs& a, bool all_sa, bool* launched);
hipError_t launch_group_wiener(hipStream_t s, const GroupArgs& a, bool all_sa, bool* launched);
hipError_t launch_group_wide(hipStream_t s, const GroupArgs& a, bool* launched);   /* lfbm5d_group_wide.hip: HT, tau_2D = id, 5x5 / 7x7 windows */
hipError_t prepare_group_wide();
/* lfbm5d_group_slab.hip: stacks beyond the LDS (Wiener k = 12 / 16, N = 32, 5x5 / 7x7 windows with a 2-D transform) */
hipError_t launch_group_slab(hipStream_t s, const GroupArgs& a, bool* launched);
hipError_t prepare_group_slab();
bool group_uses_slab(const GroupArgs& a);
size_t group_slab_scratch_bytes(const GroupArgs& a);
hipError_t prepare_group_ht();
hipError_t prepare_group_wiener();
constexpr int kDedicatedLdsLimit = 160 * 1024 - 4096;

namespace {

/* transform tables are constant during a kernel: constant address space, so uniform reads become scalar loads */
typedef const __attribute__((address_space(4))) GroupTables* TbPtr;
typedef const __attribute__((address_space(4))) float* TbFloats;
static inline dim3 grid1d(size_t n, unsigned b = 256) { return dim3((unsigned)((n + b - 1) / b)); }

/* ================================== group kernel ========================================== */

constexpr int kThreads = 256;

template <int MA, int MW>
struct ShapeInfoT {          /* SADCT bookkeeping of one group (core:302-323, :2036-2049, :2102-2104) */
    int mask[MA], idx[MA], mask_col[MA], idx_col[MA], mask_dct[MA];   /* [s * aw + t], aw = window side */
    int row_n[MW], col_n[MW];
    int use_sadct;
};
typedef ShapeInfoT<kMaxA, kMaxAw> ShapeInfo;        /* windows of up to 7x7 SAIs */
typedef ShapeInfoT<kBigA, kBigAw> ShapeInfoBig;     /* larger windows (general forms only) */
static_assert(sizeof(ShapeInfo) == kShapeInfoBytes && sizeof(ShapeInfoBig) == kShapeInfoBigBytes, "GroupArgs::gshape stride");
/* the per-group ShapeInfo written by the pre-pass is constant during the group kernels: scalar loads */
typedef const __attribute__((address_space(4))) ShapeInfo& ShRef;
typedef const __attribute__((address_space(4))) ShapeInfoBig& ShRefBig;
/* side of a square window of A SAIs */
__device__ __forceinline__ int window_side(int A) { int w = 1; while (w * w < A) w++; return w; }

template <class SH>
__device__ void build_shape(SH& sh, const int* m, int aw) {
    const int A = aw * aw;
    int size = 0;
    for (int i = 0; i < A; i++) { sh.mask[i] = m[i]; sh.idx[i] = 0; sh.mask_col[i] = 0; sh.idx_col[i] = 0; sh.mask_dct[i] = 0; size += m[i]; }
    for (int s = 0; s < aw; s++) {
        int r = 0;
        for (int t = 0; t < aw; t++) if (m[s * aw + t]) sh.idx[s * aw + r++] = t;
        sh.row_n[s] = r;
        for (int t = 0; t < r; t++) sh.mask_col[s * aw + t] = 1;
    }
    for (int t = 0; t < aw; t++) {
        int r = 0;
        for (int s = 0; s < aw; s++) if (sh.mask_col[s * aw + t]) sh.idx_col[(r++) * aw + t] = s;
        sh.col_n[t] = r;
        for (int s = 0; s < r; s++) sh.mask_dct[s * aw + t] = 1;
    }
    sh.use_sadct = size != A;
}

/* orthonormalised 3x3 angular DCT as the reference applies it (core:1862-1954) */
__device__ __forceinline__ void dct9_fwd(float* x, TbPtr tb) {
    float t[9];
#pragma unroll
    for (int s = 0; s < 3; s++)
#pragma unroll
        for (int u = 0; u < 3; u++)
            t[s * 3 + u] = 2.0f * (x[s * 3] * tb->cos3[u * 3] + x[s * 3 + 1] * tb->cos3[u * 3 + 1] + x[s * 3 + 2] * tb->cos3[u * 3 + 2]);
#pragma unroll
    for (int v = 0; v < 3; v++)
#pragma unroll
        for (int u = 0; u < 3; u++)
            x[v * 3 + u] = 2.0f * (t[u] * tb->cos3[v * 3] + t[3 + u] * tb->cos3[v * 3 + 1] + t[6 + u] * tb->cos3[v * 3 + 2]) * tb->cn4[v * 3 + u];
}
__device__ __forceinline__ void dct9_inv(float* x, TbPtr tb) {
    float t[9];
#pragma unroll
    for (int i = 0; i < 9; i++) x[i] *= tb->cni4[i];
#pragma unroll
    for (int s = 0; s < 3; s++)
#pragma unroll
        for (int j = 0; j < 3; j++)
            t[s * 3 + j] = x[s * 3] + 2.0f * (x[s * 3 + 1] * tb->cos3[3 + j] + x[s * 3 + 2] * tb->cos3[6 + j]);
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++)
            x[i * 3 + j] = (t[j] + 2.0f * (t[3 + j] * tb->cos3[3 + i] + t[6 + j] * tb->cos3[6 + i])) * tb->coef4inv;
}

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));   /* two values per lane: v_pk_{add,mul,fma}_f32 */
struct __attribute__((packed, aligned(4))) f4u { float v[4]; };   /* 16-byte load at 4-byte alignment */
/* dct9_fwd / dct9_inv on a pair of fibres: the same operation sequence as the scalar versions (folding the
 * constant factors into the stage matrices saves a third of the multiplies but moves results by an ulp, enough
 * to flip the odd hard-threshold decision against the reference) */
__device__ __forceinline__ void dct9_fwd2(v2f* x, TbPtr tb) {
    v2f t[9];
#pragma unroll
    for (int s = 0; s < 3; s++)
#pragma unroll
        for (int u = 0; u < 3; u++)
            t[s * 3 + u] = 2.0f * (x[s * 3] * tb->cos3[u * 3] + x[s * 3 + 1] * tb->cos3[u * 3 + 1] + x[s * 3 + 2] * tb->cos3[u * 3 + 2]);
#pragma unroll
    for (int v = 0; v < 3; v++)
#pragma unroll
        for (int u = 0; u < 3; u++)
            x[v * 3 + u] = 2.0f * (t[u] * tb->cos3[v * 3] + t[3 + u] * tb->cos3[v * 3 + 1] + t[6 + u] * tb->cos3[v * 3 + 2]) * tb->cn4[v * 3 + u];
}
__device__ __forceinline__ void dct9_inv2(v2f* x, TbPtr tb) {
    v2f t[9];
#pragma unroll
    for (int i = 0; i < 9; i++) x[i] *= tb->cni4[i];
#pragma unroll
    for (int s = 0; s < 3; s++)
#pragma unroll
        for (int j = 0; j < 3; j++)
            t[s * 3 + j] = x[s * 3] + 2.0f * (x[s * 3 + 1] * tb->cos3[3 + j] + x[s * 3 + 2] * tb->cos3[6 + j]);
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++)
            x[i * 3 + j] = (t[j] + 2.0f * (t[3 + j] * tb->cos3[3 + i] + t[6 + j] * tb->cos3[6 + i])) * tb->coef4inv;
}
/* The same 3x3 transforms with the table's symmetries used (cos3 = {1, 1, 1; c, 0, -c; 1/2, -1, 1/2}: the table's
 * middle entry is cos(pi/2) in double, 6e-17, taken as 0): less than half the operations, results within an ulp or two of
 * dct9_fwd2 / dct9_inv2.  For the Wiener kernels only -- a hard-threshold decision can flip on an ulp. */
__device__ __forceinline__ void dct9_fwd2_fast(v2f* x, TbPtr tb) {
    const float c2 = 2.0f * tb->cos3[3];
    v2f t[9];
#pragma unroll
    for (int s = 0; s < 3; s++) {
        const v2f a = x[s * 3] + x[s * 3 + 2];
        t[s * 3] = 2.0f * (a + x[s * 3 + 1]);
        t[s * 3 + 1] = c2 * (x[s * 3] - x[s * 3 + 2]);
        t[s * 3 + 2] = a - 2.0f * x[s * 3 + 1];
    }
#pragma unroll
    for (int u = 0; u < 3; u++) {
        const v2f a = t[u] + t[6 + u];
        x[u] = (a + t[3 + u]) * (2.0f * tb->cn4[u]);
        x[3 + u] = (t[u] - t[6 + u]) * (c2 * tb->cn4[3 + u]);
        x[6 + u] = (a - 2.0f * t[3 + u]) * tb->cn4[6 + u];
    }
}
__device__ __forceinline__ void dct9_inv2_fast(v2f* x, TbPtr tb) {
    const float c2 = 2.0f * tb->cos3[3];
    v2f t[9];
#pragma unroll
    for (int s = 0; s < 3; s++) {
        const v2f X0 = x[s * 3] * tb->cni4[s * 3], X1 = x[s * 3 + 1] * (c2 * tb->cni4[s * 3 + 1]), X2 = x[s * 3 + 2] * tb->cni4[s * 3 + 2];
        const v2f p = X0 + X2;
        t[s * 3] = p + X1; t[s * 3 + 1] = X0 - 2.0f * X2; t[s * 3 + 2] = p - X1;
    }
#pragma unroll
    for (int j = 0; j < 3; j++) {
        const v2f T0 = t[j] * tb->coef4inv, T1 = t[3 + j] * (c2 * tb->coef4inv), T2 = t[6 + j] * tb->coef4inv;
        const v2f p = T0 + T2;
        x[j] = p + T1; x[3 + j] = T0 - 2.0f * T2; x[6 + j] = p - T1;
    }
}
/* Haar over the NS patches of a group held as pairs P[h] = {patch h, patch h + NS/2} (lib_transforms.cpp:403-471
 * / :290-321, same butterflies and scaling, evaluated two at a time).  Forward leaves the coefficients in
 * C[0..NS/2) (a permutation of the reference order -- the shrinkage treats all coefficients alike); the inverse
 * takes that layout back to P. */
template <int NS> __device__ __forceinline__ void haar_fwd_pairs(v2f* P) {
    const float s = 0.70710678118654752f;
    if (NS == 8) {
        const v2f S01 = (P[0] + P[1]) * s, D01 = (P[0] - P[1]) * s, S23 = (P[2] + P[3]) * s, D23 = (P[2] - P[3]) * s;
        const v2f SS = (S01 + S23) * s, DD = (S01 - S23) * s;
        P[0] = v2f{(SS.x + SS.y) * s, (SS.x - SS.y) * s}; P[1] = DD; P[2] = D01; P[3] = D23;
    } else if (NS == 4) {
        const v2f S = (P[0] + P[1]) * s, D = (P[0] - P[1]) * s;
        P[0] = v2f{(S.x + S.y) * s, (S.x - S.y) * s}; P[1] = D;
    } else if (NS == 2) {
        P[0] = v2f{(P[0].x + P[0].y) * s, (P[0].x - P[0].y) * s};
    }
}
template <int NS> __device__ __forceinline__ void haar_inv_pairs(v2f* P) {
    const float s = 0.70710678118654752f;
    if (NS == 8) {
        const v2f X = v2f{(P[0].x + P[0].y) * s, (P[0].x - P[0].y) * s};
        const v2f U = (X + P[1]) * s, V = (X - P[1]) * s, D01 = P[2], D23 = P[3];
        P[0] = (U + D01) * s; P[1] = (U - D01) * s; P[2] = (V + D23) * s; P[3] = (V - D23) * s;
    } else if (NS == 4) {
        const v2f X = v2f{(P[0].x + P[0].y) * s, (P[0].x - P[0].y) * s}, D = P[1];
        P[0] = (X + D) * s; P[1] = (X - D) * s;
    } else if (NS == 2) {
        P[0] = v2f{(P[0].x + P[0].y) * s, (P[0].x - P[0].y) * s};
    }
}
/* 1-D REDFT10 / REDFT01 of runtime length n <= 3 (SADCT rows / columns) */
__device__ void r10_small(const float* x, float* y, int n, TbPtr tb) {
    for (int u = 0; u < n; u++) {
        float a = 0.0f;
        for (int j = 0; j < n; j++) a += x[j] * tb->cos1[n][u * n + j];
        y[u] = 2.0f * a;
    }
}
__device__ void r01_small(const float* x, float* y, int n, TbPtr tb) {
    for (int j = 0; j < n; j++) {
        float a = 0.0f;
        for (int u = 1; u < n; u++) a += x[u] * tb->cos1[n][u * n + j];
        y[j] = x[0] + 2.0f * a;
    }
}
/* core:1969-2116 on one 3x3 vector */
__device__ __noinline__ void sadct9_fwd(float* v, ShRef sh, TbPtr tb) {
    float x[3], y[3];
    for (int s = 0; s < 3; s++) {
        const int n = sh.row_n[s];
        if (n == 1) v[s * 3] = v[s * 3 + sh.idx[s * 3]];
        else if (n > 1) {
            for (int t = 0; t < n; t++) x[t] = v[s * 3 + sh.idx[s * 3 + t]];
            r10_small(x, y, n, tb);
            for (int t = 0; t < n; t++) v[s * 3 + t] = y[t] * tb->cn1[n][t];
        }
    }
    for (int t = 0; t < 3; t++) {
        const int n = sh.col_n[t];
        if (n == 1) v[t] = v[sh.idx_col[t] * 3 + t];
        else if (n > 1) {
            for (int s = 0; s < n; s++) x[s] = v[sh.idx_col[s * 3 + t] * 3 + t];
            r10_small(x, y, n, tb);
            for (int s = 0; s < n; s++) v[s * 3 + t] = y[s] * tb->cn1[n][s];
        }
    }
    const float coef = 0.5f * 0.70710678118654752f;
    for (int i = 0; i < 9; i++) v[i] *= (float)sh.mask_dct[i] * coef;
}
/* core:2131-2264 */
__device__ __noinline__ void sadct9_inv(float* v, ShRef sh, TbPtr tb) {
    float x[3], y[3];
    const float coef = 2.0f * 1.41421356237309505f;
    for (int t = 0; t < 3; t++) {
        const int n = sh.col_n[t];
        if (n == 1) v[sh.idx_col[t] * 3 + t] = v[t] * coef;
        else if (n > 1) {
            for (int s = 0; s < n; s++) x[s] = v[s * 3 + t] * tb->cni1[n][s] * coef;
            r01_small(x, y, n, tb);
            for (int s = 0; s < n; s++) v[sh.idx_col[s * 3 + t] * 3 + t] = y[s] * tb->c1inv[n];
        }
    }
    for (int s = 0; s < 3; s++) {
        const int n = sh.row_n[s];
        if (n == 1) v[s * 3 + sh.idx[s * 3]] = v[s * 3];
        else if (n > 1) {
            for (int t = 0; t < n; t++) x[t] = v[s * 3 + t] * tb->cni1[n][t];
            r01_small(x, y, n, tb);
            for (int t = 0; t < n; t++) v[s * 3 + sh.idx[s * 3 + t]] = y[t] * tb->c1inv[n];
        }
    }
    for (int i = 0; i < 9; i++) v[i] *= (float)sh.mask[i];
}
/* The same two transforms for callers that cannot afford a call: inline, the vector in LDS (dynamic indices), the length-n
 * transforms unrolled to three with the terms past n left out -- the same products added in the same order.  (Around a call the
 * 72 live values of the 16x16 kernels' register stage have to sit in the sparse callee-saved registers: 124 VGPRs become 168 and
 * 80 spills for every group, shape-adaptive or not.) */
__device__ __forceinline__ void r10_small3(const float (&x)[3], float (&y)[3], int n, TbPtr tb) {
#pragma unroll
    for (int u = 0; u < 3; u++) {
        float a = 0.0f;
#pragma unroll
        for (int j = 0; j < 3; j++) if (j < n && u < n) a += x[j] * tb->cos1[n][u * n + j];
        y[u] = 2.0f * a;
    }
}
__device__ __forceinline__ void r01_small3(const float (&x)[3], float (&y)[3], int n, TbPtr tb) {
#pragma unroll
    for (int j = 0; j < 3; j++) {
        float a = 0.0f;
#pragma unroll
        for (int u = 1; u < 3; u++) if (u < n && j < n) a += x[u] * tb->cos1[n][u * n + j];
        y[j] = x[0] + 2.0f * a;
    }
}
/* ... and in REGISTERS: a group's shape is uniform, so every index of the transform is a scalar -- the gathers become two selects on
 * scalar conditions, the scatters three conditional moves per destination, the loops over the run lengths predicated code; no
 * dynamic indexing, no call, no scratch.  Same products in the same order as sadct9_fwd / sadct9_inv.  (A window with an empty SAI
 * makes every group shape-adaptive, bm5d.cpp:276-280: with the call form such a 560^2 pass took 13.6 instead of 0.94 ms in the
 * HT group kernel and 9.3 instead of 1.1 ms in the Wiener one.) */
__device__ __forceinline__ float pick3(float a, float b, float c, int i) { return i == 0 ? a : (i == 1 ? b : c); }
/* the length-N transforms with N a compile-time constant (the run length is a scalar: one uniform branch per row / column selects
 * the instance; a full row -- two of three in a window with one empty SAI -- needs no gather at all) */
template <int N> __device__ __forceinline__ void r10_n(const float (&x)[3], float (&y)[3], TbPtr tb) {
#pragma unroll
    for (int u = 0; u < N; u++) {
        float a = 0.0f;
#pragma unroll
        for (int j = 0; j < N; j++) a += x[j] * tb->cos1[N][u * N + j];
        y[u] = 2.0f * a;
    }
}
template <int N> __device__ __forceinline__ void r01_n(const float (&x)[3], float (&y)[3], TbPtr tb) {
#pragma unroll
    for (int j = 0; j < N; j++) {
        float a = 0.0f;
#pragma unroll
        for (int u = 1; u < N; u++) a += x[u] * tb->cos1[N][u * N + j];
        y[j] = x[0] + 2.0f * a;
    }
}
__device__ __forceinline__ void sadct9_fwd_sel(float (&v)[9], ShRef sh, TbPtr tb) {
    float x[3], y[3];
#pragma unroll
    for (int s = 0; s < 3; s++) {
        const int n = sh.row_n[s];
        if (n == 3) {   /* idx = 0, 1, 2 */
            x[0] = v[s * 3]; x[1] = v[s * 3 + 1]; x[2] = v[s * 3 + 2];
            r10_n<3>(x, y, tb);
#pragma unroll
            for (int t = 0; t < 3; t++) v[s * 3 + t] = y[t] * tb->cn1[3][t];
        } else if (n == 2) {
            x[0] = pick3(v[s * 3], v[s * 3 + 1], v[s * 3 + 2], sh.idx[s * 3]);
            x[1] = pick3(v[s * 3], v[s * 3 + 1], v[s * 3 + 2], sh.idx[s * 3 + 1]);
            r10_n<2>(x, y, tb);
            v[s * 3] = y[0] * tb->cn1[2][0]; v[s * 3 + 1] = y[1] * tb->cn1[2][1];
        } else if (n == 1) v[s * 3] = pick3(v[s * 3], v[s * 3 + 1], v[s * 3 + 2], sh.idx[s * 3]);
    }
#pragma unroll
    for (int t = 0; t < 3; t++) {
        const int n = sh.col_n[t];
        if (n == 3) {
            x[0] = v[t]; x[1] = v[3 + t]; x[2] = v[6 + t];
            r10_n<3>(x, y, tb);
#pragma unroll
            for (int s2 = 0; s2 < 3; s2++) v[s2 * 3 + t] = y[s2] * tb->cn1[3][s2];
        } else if (n == 2) {
            x[0] = pick3(v[t], v[3 + t], v[6 + t], sh.idx_col[t]);
            x[1] = pick3(v[t], v[3 + t], v[6 + t], sh.idx_col[3 + t]);
            r10_n<2>(x, y, tb);
            v[t] = y[0] * tb->cn1[2][0]; v[3 + t] = y[1] * tb->cn1[2][1];
        } else if (n == 1) v[t] = pick3(v[t], v[3 + t], v[6 + t], sh.idx_col[t]);
    }
    const float coef = 0.5f * 0.70710678118654752f;
#pragma unroll
    for (int i = 0; i < 9; i++) v[i] *= (float)sh.mask_dct[i] * coef;
}
__device__ __forceinline__ void sadct9_inv_sel(float (&v)[9], ShRef sh, TbPtr tb) {
    float x[3], y[3];
    const float coef = 2.0f * 1.41421356237309505f;
#pragma unroll
    for (int t = 0; t < 3; t++) {
        const int n = sh.col_n[t];
        if (n == 3) {   /* rows 0, 1, 2: in place */
#pragma unroll
            for (int s2 = 0; s2 < 3; s2++) x[s2] = v[s2 * 3 + t] * tb->cni1[3][s2] * coef;
            r01_n<3>(x, y, tb);
#pragma unroll
            for (int s2 = 0; s2 < 3; s2++) v[s2 * 3 + t] = y[s2] * tb->c1inv[3];
        } else if (n >= 1) {
            if (n == 1) { y[0] = v[t] * coef; y[1] = 0.0f; }
            else {
                x[0] = v[t] * tb->cni1[2][0] * coef; x[1] = v[3 + t] * tb->cni1[2][1] * coef;
                r01_n<2>(x, y, tb);
                y[0] *= tb->c1inv[2]; y[1] *= tb->c1inv[2];
            }
            /* v[idx_col[s2][t]][t] = y[s2] for s2 < n: the destinations are distinct rows */
#pragma unroll
            for (int r = 0; r < 3; r++) {
                float w = v[r * 3 + t];
                w = sh.idx_col[t] == r ? y[0] : w;
                w = (n == 2 && sh.idx_col[3 + t] == r) ? y[1] : w;
                v[r * 3 + t] = w;
            }
        }
    }
#pragma unroll
    for (int s = 0; s < 3; s++) {
        const int n = sh.row_n[s];
        if (n == 3) {
#pragma unroll
            for (int t = 0; t < 3; t++) x[t] = v[s * 3 + t] * tb->cni1[3][t];
            r01_n<3>(x, y, tb);
#pragma unroll
            for (int t = 0; t < 3; t++) v[s * 3 + t] = y[t] * tb->c1inv[3];
        } else if (n >= 1) {
            if (n == 1) { y[0] = v[s * 3]; y[1] = 0.0f; }
            else {
                x[0] = v[s * 3] * tb->cni1[2][0]; x[1] = v[s * 3 + 1] * tb->cni1[2][1];
                r01_n<2>(x, y, tb);
                y[0] *= tb->c1inv[2]; y[1] *= tb->c1inv[2];
            }
#pragma unroll
            for (int q = 0; q < 3; q++) {
                float w = v[s * 3 + q];
                w = sh.idx[s * 3] == q ? y[0] : w;
                w = (n == 2 && sh.idx[s * 3 + 1] == q) ? y[1] : w;
                v[s * 3 + q] = w;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 9; i++) v[i] *= (float)sh.mask[i];
}
__device__ __forceinline__ void sadct9_fwd_lds(float* v, ShRef sh, TbPtr tb) {
    float x[3], y[3];
#pragma unroll
    for (int s = 0; s < 3; s++) {
        const int n = sh.row_n[s];
        if (n == 1) v[s * 3] = v[s * 3 + sh.idx[s * 3]];
        else if (n > 1) {
#pragma unroll
            for (int t = 0; t < 3; t++) x[t] = t < n ? v[s * 3 + sh.idx[s * 3 + t]] : 0.0f;
            r10_small3(x, y, n, tb);
#pragma unroll
            for (int t = 0; t < 3; t++) if (t < n) v[s * 3 + t] = y[t] * tb->cn1[n][t];
        }
    }
#pragma unroll
    for (int t = 0; t < 3; t++) {
        const int n = sh.col_n[t];
        if (n == 1) v[t] = v[sh.idx_col[t] * 3 + t];
        else if (n > 1) {
#pragma unroll
            for (int s2 = 0; s2 < 3; s2++) x[s2] = s2 < n ? v[sh.idx_col[s2 * 3 + t] * 3 + t] : 0.0f;
            r10_small3(x, y, n, tb);
#pragma unroll
            for (int s2 = 0; s2 < 3; s2++) if (s2 < n) v[s2 * 3 + t] = y[s2] * tb->cn1[n][s2];
        }
    }
    const float coef = 0.5f * 0.70710678118654752f;
#pragma unroll
    for (int i = 0; i < 9; i++) v[i] *= (float)sh.mask_dct[i] * coef;
}
__device__ __forceinline__ void sadct9_inv_lds(float* v, ShRef sh, TbPtr tb) {
    float x[3], y[3];
    const float coef = 2.0f * 1.41421356237309505f;
#pragma unroll
    for (int t = 0; t < 3; t++) {
        const int n = sh.col_n[t];
        if (n == 1) v[sh.idx_col[t] * 3 + t] = v[t] * coef;
        else if (n > 1) {
#pragma unroll
            for (int s2 = 0; s2 < 3; s2++) x[s2] = s2 < n ? v[s2 * 3 + t] * tb->cni1[n][s2] * coef : 0.0f;
            r01_small3(x, y, n, tb);
#pragma unroll
            for (int s2 = 0; s2 < 3; s2++) if (s2 < n) v[sh.idx_col[s2 * 3 + t] * 3 + t] = y[s2] * tb->c1inv[n];
        }
    }
#pragma unroll
    for (int s = 0; s < 3; s++) {
        const int n = sh.row_n[s];
        if (n == 1) v[s * 3 + sh.idx[s * 3]] = v[s * 3];
        else if (n > 1) {
#pragma unroll
            for (int t = 0; t < 3; t++) x[t] = t < n ? v[s * 3 + t] * tb->cni1[n][t] : 0.0f;
            r01_small3(x, y, n, tb);
#pragma unroll
            for (int t = 0; t < 3; t++) if (t < n) v[s * 3 + sh.idx[s * 3 + t]] = y[t] * tb->c1inv[n];
        }
    }
#pragma unroll
    for (int i = 0; i < 9; i++) v[i] *= (float)sh.mask[i];
}

/* General aw x aw angular window (aswSize 2: 5x5): the same transforms with run-time sizes, generic kernel only.
 * dct_4d_process / dct_4d_inverse (core:1862-1954) and sadct_4d_process / _inverse (core:1969-2264) on one vector. */
/* The angular DCT of a 5x5 / 7x7 window (dct_4d_process / dct_4d_inverse, core:1862-1954) with the window side a compile-time
 * constant: the vector and the intermediate in registers, the loops unrolled.  (Rounds 2-3 had call forms that walked a scratch
 * vector with run-time indices: a 5x5 window's group kernel took 24-30 ms per 304^2 pass, a 7x7 window's 73-78 ms.) */
template <int AW>
__device__ __forceinline__ void dctw_fwd_t(float (&x)[AW * AW], TbPtr tb) {
    float t[AW * AW];
#pragma unroll
    for (int s = 0; s < AW; s++)
#pragma unroll
        for (int u = 0; u < AW; u++) {
            float acc = 0.0f;
#pragma unroll
            for (int j = 0; j < AW; j++) acc += x[s * AW + j] * tb->cosw[u * AW + j];
            t[s * AW + u] = 2.0f * acc;
        }
#pragma unroll
    for (int v = 0; v < AW; v++)
#pragma unroll
        for (int u = 0; u < AW; u++) {
            float acc = 0.0f;
#pragma unroll
            for (int j = 0; j < AW; j++) acc += t[j * AW + u] * tb->cosw[v * AW + j];
            x[v * AW + u] = 2.0f * acc * tb->cn4[v * AW + u];
        }
}
template <int AW>
__device__ __forceinline__ void dctw_inv_t(float (&x)[AW * AW], TbPtr tb) {
    float t[AW * AW];
#pragma unroll
    for (int i = 0; i < AW * AW; i++) x[i] *= tb->cni4[i];
#pragma unroll
    for (int s = 0; s < AW; s++)
#pragma unroll
        for (int j = 0; j < AW; j++) {
            float acc = 0.0f;
#pragma unroll
            for (int u = 1; u < AW; u++) acc += x[s * AW + u] * tb->cosw[u * AW + j];
            t[s * AW + j] = x[s * AW] + 2.0f * acc;
        }
#pragma unroll
    for (int i = 0; i < AW; i++)
#pragma unroll
        for (int j = 0; j < AW; j++) {
            float acc = 0.0f;
#pragma unroll
            for (int v = 1; v < AW; v++) acc += t[v * AW + j] * tb->cosw[v * AW + i];
            x[i * AW + j] = (t[j] + 2.0f * acc) * tb->coef4inv;
        }
}
/* ... and with a run-time window side, on a vector in scratch memory (windows larger than 7x7) */
__device__ __noinline__ void dctw_fwd_rt(float* x, float* t, int aw, TbPtr tb) {
    for (int s = 0; s < aw; s++)
        for (int u = 0; u < aw; u++) {
            float acc = 0.0f;
            for (int j = 0; j < aw; j++) acc += x[s * aw + j] * tb->cosw[u * aw + j];
            t[s * aw + u] = 2.0f * acc;
        }
    for (int v = 0; v < aw; v++)
        for (int u = 0; u < aw; u++) {
            float acc = 0.0f;
            for (int j = 0; j < aw; j++) acc += t[j * aw + u] * tb->cosw[v * aw + j];
            x[v * aw + u] = 2.0f * acc * tb->cn4[v * aw + u];
        }
}
__device__ __noinline__ void dctw_inv_rt(float* x, float* t, int aw, TbPtr tb) {
    for (int i = 0; i < aw * aw; i++) x[i] *= tb->cni4[i];
    for (int s = 0; s < aw; s++)
        for (int j = 0; j < aw; j++) {
            float acc = 0.0f;
            for (int u = 1; u < aw; u++) acc += x[s * aw + u] * tb->cosw[u * aw + j];
            t[s * aw + j] = x[s * aw] + 2.0f * acc;
        }
    for (int i = 0; i < aw; i++)
        for (int j = 0; j < aw; j++) {
            float acc = 0.0f;
            for (int v = 1; v < aw; v++) acc += t[v * aw + j] * tb->cosw[v * aw + i];
            x[i * aw + j] = (t[j] + 2.0f * acc) * tb->coef4inv;
        }
}
template <class SHR>
__device__ __noinline__ void sadctw_fwd(float* v, int aw, SHR sh, TbPtr tb) {
    float x[kBigAw], y[kBigAw];
    for (int s = 0; s < aw; s++) {
        const int n = sh.row_n[s];
        if (n == 1) v[s * aw] = v[s * aw + sh.idx[s * aw]];
        else if (n > 1) {
            for (int t = 0; t < n; t++) x[t] = v[s * aw + sh.idx[s * aw + t]];
            r10_small(x, y, n, tb);
            for (int t = 0; t < n; t++) v[s * aw + t] = y[t] * tb->cn1[n][t];
        }
    }
    for (int t = 0; t < aw; t++) {
        const int n = sh.col_n[t];
        if (n == 1) v[t] = v[sh.idx_col[t] * aw + t];
        else if (n > 1) {
            for (int s = 0; s < n; s++) x[s] = v[sh.idx_col[s * aw + t] * aw + t];
            r10_small(x, y, n, tb);
            for (int s = 0; s < n; s++) v[s * aw + t] = y[s] * tb->cn1[n][s];
        }
    }
    const float coef = 0.5f * 0.70710678118654752f;
    for (int i = 0; i < aw * aw; i++) v[i] *= (float)sh.mask_dct[i] * coef;
}
template <class SHR>
__device__ __noinline__ void sadctw_inv(float* v, int aw, SHR sh, TbPtr tb) {
    float x[kBigAw], y[kBigAw];
    const float coef = 2.0f * 1.41421356237309505f;
    for (int t = 0; t < aw; t++) {
        const int n = sh.col_n[t];
        if (n == 1) v[sh.idx_col[t] * aw + t] = v[t] * coef;
        else if (n > 1) {
            for (int s = 0; s < n; s++) x[s] = v[s * aw + t] * tb->cni1[n][s] * coef;
            r01_small(x, y, n, tb);
            for (int s = 0; s < n; s++) v[sh.idx_col[s * aw + t] * aw + t] = y[s] * tb->c1inv[n];
        }
    }
    for (int s = 0; s < aw; s++) {
        const int n = sh.row_n[s];
        if (n == 1) v[s * aw + sh.idx[s * aw]] = v[s * aw];
        else if (n > 1) {
            for (int t = 0; t < n; t++) x[t] = v[s * aw + t] * tb->cni1[n][t];
            r01_small(x, y, n, tb);
            for (int t = 0; t < n; t++) v[s * aw + sh.idx[s * aw + t]] = y[t] * tb->c1inv[n];
        }
    }
    for (int i = 0; i < aw * aw; i++) v[i] *= (float)sh.mask[i];
}

/* The shape-adaptive angular transform (core:1969-2264) as SEPARABLE passes over a stack in LDS, two pixels per lane: a thread owns
 * one row (then one column) of a match's aw x aw block -- `row` / `col` point at its first element, consecutive elements P / AW * P
 * pairs apart.  The same products in the same order as sadctw_fwd / sadctw_inv (which hold the whole block of one pixel in private
 * memory: hundreds of times slower on the 6 - 30 % shape-adaptive groups of windows of 9x9 SAIs and more).
 * A lane's row differs from its neighbour's, so everything indexed by the row -- its length, its index list, the cosine table of that
 * length -- is a per-lane access: from the group's shape record and the constant tables that is a dependent global load per TERM
 * (100 us per four-pixel slab of a 15x15 window); the workgroup therefore copies them into an LDS table first (sa_fill), laid out
 *   idx[A] | idx_col[A] | mask[A] | mask_dct[A] | row_n[AW] | col_n[AW] | cos1 of length 1, 2, ... AW packed (n^2 each) |
 *   cn1[AW+1][AW] | cni1[AW+1][AW] | c1inv[AW+1]
 * Out of line, so that the registers of this rarer path are not the kernel's; the pointers keep their LDS address space through
 * the call. */
typedef __attribute__((address_space(3))) v2f* LdsV2;
typedef const __attribute__((address_space(3))) int* SaTab;
template <int AW> struct SaLayout {
    static constexpr int A = AW * AW;
    static constexpr int idx = 0, idx_col = A, mask = 2 * A, mask_dct = 3 * A, row_n = 4 * A, col_n = 4 * A + AW, cos = 4 * A + 2 * AW;
    static constexpr int cos_words = AW * (AW + 1) * (2 * AW + 1) / 6;
    static constexpr int cn1 = cos + cos_words, cni1 = cn1 + (AW + 1) * AW, c1inv = cni1 + (AW + 1) * AW, words = c1inv + AW + 1;
    __device__ static __forceinline__ int cos_off(int n) { return cos + (n - 1) * n * (2 * n - 1) / 6; }   /* lengths 1 .. n-1 in front */
};
template <int AW, class SHR>
__device__ __forceinline__ void sa_fill(int* tab, SHR sh, TbPtr tb, int tid, int nthreads) {   /* all threads; a barrier follows */
    typedef SaLayout<AW> L;
    float* tf = reinterpret_cast<float*>(tab);
    for (int i = tid; i < L::A; i += nthreads) {
        tab[L::idx + i] = sh.idx[i]; tab[L::idx_col + i] = sh.idx_col[i]; tab[L::mask + i] = sh.mask[i]; tab[L::mask_dct + i] = sh.mask_dct[i];
    }
    for (int i = tid; i < AW; i += nthreads) { tab[L::row_n + i] = sh.row_n[i]; tab[L::col_n + i] = sh.col_n[i]; }
    for (int n = 1; n <= AW; n++)
        for (int i = tid; i < n * n; i += nthreads) tf[L::cos_off(n) + i] = tb->cos1[n][i];
    for (int i = tid; i < (AW + 1) * AW; i += nthreads) { tf[L::cn1 + i] = tb->cn1[i / AW][i % AW]; tf[L::cni1 + i] = tb->cni1[i / AW][i % AW]; }
    for (int i = tid; i <= AW; i += nthreads) tf[L::c1inv + i] = tb->c1inv[i];
}
template <int AW>
__device__ __noinline__ void sadctw_rows_fwd2(LdsV2 row, int P, int s, SaTab tab) {
    typedef SaLayout<AW> L;
    const __attribute__((address_space(3))) float* tf = reinterpret_cast<const __attribute__((address_space(3))) float*>(tab);
    const int n = tab[L::row_n + s];
    if (n == 1) { row[0] = row[tab[L::idx + s * AW] * P]; return; }
    if (n < 2) return;
    const int co = L::cos_off(n);
    v2f x[AW], y[AW];
#pragma unroll
    for (int t = 0; t < AW; t++) x[t] = t < n ? row[tab[L::idx + s * AW + t] * P] : v2f{0.0f, 0.0f};
#pragma unroll
    for (int u = 0; u < AW; u++) {
        v2f a = {0.0f, 0.0f};
#pragma unroll
        for (int j = 0; j < AW; j++) if (j < n && u < n) a += x[j] * tf[co + u * n + j];
        y[u] = 2.0f * a;
    }
#pragma unroll
    for (int t = 0; t < AW; t++) if (t < n) row[t * P] = y[t] * tf[L::cn1 + n * AW + t];
}
template <int AW>
__device__ __noinline__ void sadctw_cols_fwd2(LdsV2 col, int P, int t, SaTab tab) {   /* ... and the closing scale by mask_dct */
    typedef SaLayout<AW> L;
    const __attribute__((address_space(3))) float* tf = reinterpret_cast<const __attribute__((address_space(3))) float*>(tab);
    const int n = tab[L::col_n + t];
    if (n == 1) col[0] = col[tab[L::idx_col + t] * AW * P];
    else if (n > 1) {
        const int co = L::cos_off(n);
        v2f x[AW], y[AW];
#pragma unroll
        for (int q = 0; q < AW; q++) x[q] = q < n ? col[tab[L::idx_col + q * AW + t] * AW * P] : v2f{0.0f, 0.0f};
#pragma unroll
        for (int u = 0; u < AW; u++) {
            v2f a = {0.0f, 0.0f};
#pragma unroll
            for (int j = 0; j < AW; j++) if (j < n && u < n) a += x[j] * tf[co + u * n + j];
            y[u] = 2.0f * a;
        }
#pragma unroll
        for (int q = 0; q < AW; q++) if (q < n) col[q * AW * P] = y[q] * tf[L::cn1 + n * AW + q];
    }
    const float coef = 0.5f * 0.70710678118654752f;
#pragma unroll
    for (int q = 0; q < AW; q++) col[q * AW * P] *= (float)tab[L::mask_dct + q * AW + t] * coef;
}
template <int AW>
__device__ __noinline__ void sadctw_cols_inv2(LdsV2 col, int P, int t, SaTab tab) {
    typedef SaLayout<AW> L;
    const __attribute__((address_space(3))) float* tf = reinterpret_cast<const __attribute__((address_space(3))) float*>(tab);
    const int n = tab[L::col_n + t];
    const float coef = 2.0f * 1.41421356237309505f;
    if (n == 1) { const v2f v0 = col[0]; col[tab[L::idx_col + t] * AW * P] = v0 * coef; }
    else if (n > 1) {
        const int co = L::cos_off(n);
        v2f x[AW], y[AW];
#pragma unroll
        for (int q = 0; q < AW; q++) x[q] = q < n ? col[q * AW * P] * tf[L::cni1 + n * AW + q] * coef : v2f{0.0f, 0.0f};
#pragma unroll
        for (int j = 0; j < AW; j++) {
            v2f a = {0.0f, 0.0f};
#pragma unroll
            for (int u = 1; u < AW; u++) if (u < n && j < n) a += x[u] * tf[co + u * n + j];
            y[j] = x[0] + 2.0f * a;
        }
        const float ci = tf[L::c1inv + n];
#pragma unroll
        for (int q = 0; q < AW; q++) if (q < n) col[tab[L::idx_col + q * AW + t] * AW * P] = y[q] * ci;
    }
}
template <int AW>
__device__ __noinline__ void sadctw_rows_inv2(LdsV2 row, int P, int s, SaTab tab) {   /* ... and the closing scale by mask */
    typedef SaLayout<AW> L;
    const __attribute__((address_space(3))) float* tf = reinterpret_cast<const __attribute__((address_space(3))) float*>(tab);
    const int n = tab[L::row_n + s];
    if (n == 1) { const v2f v0 = row[0]; row[tab[L::idx + s * AW] * P] = v0; }
    else if (n > 1) {
        const int co = L::cos_off(n);
        v2f x[AW], y[AW];
#pragma unroll
        for (int t = 0; t < AW; t++) x[t] = t < n ? row[t * P] * tf[L::cni1 + n * AW + t] : v2f{0.0f, 0.0f};
#pragma unroll
        for (int j = 0; j < AW; j++) {
            v2f a = {0.0f, 0.0f};
#pragma unroll
            for (int u = 1; u < AW; u++) if (u < n && j < n) a += x[u] * tf[co + u * n + j];
            y[j] = x[0] + 2.0f * a;
        }
        const float ci = tf[L::c1inv + n];
#pragma unroll
        for (int t = 0; t < AW; t++) if (t < n) row[tab[L::idx + s * AW + t] * P] = y[t] * ci;
    }
#pragma unroll
    for (int t = 0; t < AW; t++) row[t * P] *= (float)tab[L::mask + s * AW + t];
}

/* lib_transforms.cpp:403-471 / :290-321 on a register vector of compile-time length */
template <int NS, class T> __device__ __forceinline__ void haar_fwd(T* v) {   /* T = float, or v2f: two fibres per lane */
    const float s = 0.70710678118654752f;
#pragma unroll
    for (int n = NS; n > 1; n /= 2) {
        T t[NS > 1 ? NS : 1];
#pragma unroll
        for (int i = 0; i < n / 2; i++) { t[i] = (v[2 * i] + v[2 * i + 1]) * s; t[n / 2 + i] = (v[2 * i] - v[2 * i + 1]) * s; }
#pragma unroll
        for (int i = 0; i < n; i++) v[i] = t[i];
    }
}
template <int NS, class T> __device__ __forceinline__ void haar_inv(T* v) {
    const float s = 0.70710678118654752f;
#pragma unroll
    for (int h = 1; h < NS; h *= 2) {
        T t[NS > 1 ? NS : 1];
#pragma unroll
        for (int i = 0; i < h; i++) { t[2 * i] = (v[i] + v[h + i]) * s; t[2 * i + 1] = (v[i] - v[h + i]) * s; }
#pragma unroll
        for (int i = 0; i < 2 * h; i++) v[i] = t[i];
    }
}
template <int NS, class T> __device__ __forceinline__ void hadamard(T* v) {
    /* sums to the first half, differences to the second, recurse on both: log2(NS) levels of
     * the same butterfly applied block-wise */
#pragma unroll
    for (int n = NS; n > 1; n /= 2) {
        T t[NS > 1 ? NS : 1];
#pragma unroll
        for (int b = 0; b < NS; b += n)
#pragma unroll
            for (int i = 0; i < n / 2; i++) { t[b + i] = v[b + 2 * i] + v[b + 2 * i + 1]; t[b + n / 2 + i] = v[b + 2 * i] - v[b + 2 * i + 1]; }
#pragma unroll
        for (int i = 0; i < NS; i++) v[i] = t[i];
    }
}

/* 5th-dimension filter of one (st, pq) fibre held in registers.
 * HT: core:2408-2505 / :2281-2391; Wiener: core:2826-2925 / :2706-2810. */
/* 5th-dimension DCT of a fibre (tau_5D = dct): REDFT10 * coef_norm / coef_norm_inv * REDFT01 * coef
 * (core:2546-2593, norms preProcess_5d core:3262-3276) */
template <int NS> __device__ __forceinline__ int log2c() { return NS == 1 ? 0 : NS == 2 ? 1 : NS == 4 ? 2 : NS == 8 ? 3 : NS == 16 ? 4 : 5; }
template <int NS, class T> __device__ __forceinline__ void dct5_fwd(T* v, TbPtr tb) {
    TbFloats ct = NS == 32 ? tb->cos5x : tb->cos5[NS == 32 ? 0 : log2c<NS>()];
    T y[NS];
#pragma unroll
    for (int u = 0; u < NS; u++) {
        T a = T{};
#pragma unroll
        for (int j = 0; j < NS; j++) a += v[j] * ct[u * NS + j];
        y[u] = 2.0f * a * (u == 0 ? tb->cn5_0[log2c<NS>()] : tb->cn5[log2c<NS>()]);
    }
#pragma unroll
    for (int u = 0; u < NS; u++) v[u] = y[u];
}
template <int NS, class T> __device__ __forceinline__ void dct5_inv(T* v, TbPtr tb) {
    TbFloats ct = NS == 32 ? tb->cos5x : tb->cos5[NS == 32 ? 0 : log2c<NS>()];
    T y[NS];
    const T x0 = v[0] * 1.41421356237309505f;   /* coef_norm_inv[0] = sqrt2, others 1 */
#pragma unroll
    for (int j = 0; j < NS; j++) {
        T a = T{};
#pragma unroll
        for (int u = 1; u < NS; u++) a += v[u] * ct[u * NS + j];
        y[j] = (x0 + 2.0f * a) * tb->c5inv[log2c<NS>()];
    }
#pragma unroll
    for (int j = 0; j < NS; j++) v[j] = y[j];
}

/* 5th-dimension transform + shrinkage + inverse of one (st, pq) fibre held in registers.
 * o: noisy fibre, e: pilot fibre (Wiener); the filtered fibre is returned in o (HT) / e (Wiener).
 * HT: core:2408-2505 / :2281-2391; Wiener: core:2826-2925 / :2706-2810. */
template <int NS, int STEP>
__device__ __forceinline__ void shrink_fibre(float* o, float* e, unsigned tau5, float T, float sig2,
                                             bool in_shape, float& wacc, TbPtr tb) {
    const bool haar = tau5 == 9, dct = tau5 == 5;
    if (dct) { dct5_fwd<NS>(o, tb); if (STEP == 2) dct5_fwd<NS>(e, tb); }
    else if (NS > 1) {
        if (haar) { haar_fwd<NS>(o); if (STEP == 2) haar_fwd<NS>(e); }
        else      { hadamard<NS>(o); if (STEP == 2) hadamard<NS>(e); }
    }
    if (in_shape) {
        if (STEP == 1) {
            /* T = lambda*sigma*sqrt2; Hadamard: * sqrt(nSx) (core:2306); DCT: * 2 (core:2567) */
            const float Th = haar ? T : (dct ? T * 2.0f : T * sqrtf((float)NS));
#pragma unroll
            for (int n = 0; n < NS; n++) {
                if (fabsf(o[n]) > Th) wacc += 1.0f; else o[n] = 0.0f;
            }
        } else {
            const float hc = 1.0f / (float)NS;
            const bool plain = haar || dct;
#pragma unroll
            for (int n = 0; n < NS; n++) {
                float value = plain ? e[n] * e[n] : e[n] * e[n] * hc;
                value = __fdiv_rn(value, value + sig2);
                e[n] = plain ? o[n] * value : o[n] * value * hc;
                wacc += value;
            }
        }
    }
    float* r = STEP == 1 ? o : e;
    if (dct) dct5_inv<NS>(r, tb);
    else if (NS > 1) {
        if (haar) haar_inv<NS>(r);
        else {
            hadamard<NS>(r);
            if (STEP == 1) {
                const float hc = 1.0f / (float)NS;
#pragma unroll
                for (int n = 0; n < NS; n++) r[n] *= hc;
            }
        }
    }
}

/* the same on a fibre stored in the LDS stack */
template <int NS, int STEP>
__device__ __forceinline__ void filter5(float* S0, float* S1, int base, int stride, unsigned tau5,
                                        float T, float sig2, bool in_shape, float& wacc, float& s1, float& s2,
                                        TbPtr tb) {
    float o[NS], e[NS];
#pragma unroll
    for (int n = 0; n < NS; n++) o[n] = S0[base + n * stride];
    if (STEP == 2) {
#pragma unroll
        for (int n = 0; n < NS; n++) e[n] = S1[base + n * stride];
    }
    shrink_fibre<NS, STEP>(o, e, tau5, T, sig2, in_shape, wacc, tb);
    float* r = STEP == 1 ? o : e;
    float* dst = STEP == 1 ? S0 : S1;
#pragma unroll
    for (int n = 0; n < NS; n++) { dst[base + n * stride] = r[n]; s1 += r[n]; s2 += r[n] * r[n]; }
}

__device__ __forceinline__ int per_ext(int j, int L, int N) { int m = (j - L) % N; return m < 0 ? m + N : m; }

/* ------------------------------------------------------------------------------------------
 * bior1.5 2-D transform of K x K patches (K = 8, 16; lib_transforms.cpp:46-120 forward, :135-204 inverse), fast
 * path: K threads per patch, thread = one row in the row passes and one column in the column passes, the row /
 * column held in registers, all periodic-extension indices resolved at compile time.  A patch is copied from
 * the stack into a work area laid out [patch][K][K+1] (the odd row stride makes both access directions free of
 * LDS bank conflicts), transformed there through all levels and copied back.  The K threads of a patch sit in
 * one wavefront, whose DS operations execute in order: no workgroup barrier inside.  Same taps in the same
 * order as the generic path -- identical results.
 * ------------------------------------------------------------------------------------------ */
constexpr int bior_ext(int j, int N) { return (((j - 4) % N) + N) % N; }   /* per_ext(j, 4, N) */

template <int K, int N1>
__device__ __forceinline__ void bior_fwd_level(float* Tp, int r, TbPtr tb) {
#pragma clang fp contract(off)   /* the reference's separate multiply and add: bit-identical coefficients */
    if constexpr (N1 > 1) {
        constexpr int N2 = N1 / 2, RS = K + 1;
        if (r < N1) {   /* rows: first N2 outputs low-pass, next N2 high-pass */
            float v[N1], o[N1];
#pragma unroll
            for (int c = 0; c < N1; c++) v[c] = Tp[r * RS + c];
#pragma unroll
            for (int j = 0; j < N1; j++) {
                const int jj = j < N2 ? j : j - N2;
                float acc = 0.0f;
                if (j < N2) {
#pragma unroll
                    for (int t = 0; t < 10; t++) acc += v[bior_ext(t + 2 * jj, N1)] * tb->lpd[t];
                } else {   /* the high-pass analysis filter has two taps (lib_transforms.cpp:215-277); the eight products with
                            * its zero taps only ever add a zero */
                    acc = v[bior_ext(4 + 2 * jj, N1)] * tb->hpd[4];
                    acc += v[bior_ext(5 + 2 * jj, N1)] * tb->hpd[5];
                }
                o[j] = acc;
            }
#pragma unroll
            for (int c = 0; c < N1; c++) Tp[r * RS + c] = o[c];
        }
        __builtin_amdgcn_wave_barrier();
        if (r < N1) {   /* columns (thread = column r) */
            float v[N1], o[N1];
#pragma unroll
            for (int i = 0; i < N1; i++) v[i] = Tp[i * RS + r];
#pragma unroll
            for (int i = 0; i < N1; i++) {
                const int ii = i < N2 ? i : i - N2;
                float acc = 0.0f;
                if (i < N2) {
#pragma unroll
                    for (int t = 0; t < 10; t++) acc += v[bior_ext(t + 2 * ii, N1)] * tb->lpd[t];
                } else {
                    acc = v[bior_ext(4 + 2 * ii, N1)] * tb->hpd[4];
                    acc += v[bior_ext(5 + 2 * ii, N1)] * tb->hpd[5];
                }
                o[i] = acc;
            }
#pragma unroll
            for (int i = 0; i < N1; i++) Tp[i * RS + r] = o[i];
        }
        __builtin_amdgcn_wave_barrier();
        bior_fwd_level<K, N1 / 2>(Tp, r, tb);
    }
}
template <int K, int N1>
__device__ __forceinline__ void bior_inv_level(float* Tp, int r, TbPtr tb) {
#pragma clang fp contract(off)
    if constexpr (N1 <= K) {
        constexpr int N2 = N1 / 2, RS = K + 1;
        if (r < N1) {   /* columns: out[2m] from the high-pass taps, out[2m+1] from the low-pass taps */
            float v[N1], o[N1];
#pragma unroll
            for (int i = 0; i < N1; i++) v[i] = Tp[i * RS + r];
#pragma unroll
            for (int i = 0; i < N1; i++) {
                const int m = i / 2;
                float acc = 0.0f;
                if (i & 1) {   /* the low-pass synthesis filter has two taps */
                    acc = tb->lpr[4] * v[(4 * N2 + m) % N1];
                    acc += tb->lpr[5] * v[(5 * N2 + m) % N1];
                } else {
#pragma unroll
                    for (int t = 0; t < 10; t++) acc += tb->hpr[t] * v[(t * N2 + m) % N1];
                }
                o[i] = acc;
            }
#pragma unroll
            for (int i = 0; i < N1; i++) Tp[i * RS + r] = o[i];
        }
        __builtin_amdgcn_wave_barrier();
        if (r < N1) {   /* rows */
            float v[N1], o[N1];
#pragma unroll
            for (int c = 0; c < N1; c++) v[c] = Tp[r * RS + c];
#pragma unroll
            for (int j = 0; j < N1; j++) {
                const int m = j / 2;
                float acc = 0.0f;
                if (j & 1) {
                    acc = tb->lpr[4] * v[(4 * N2 + m) % N1];
                    acc += tb->lpr[5] * v[(5 * N2 + m) % N1];
                } else {
#pragma unroll
                    for (int t = 0; t < 10; t++) acc += tb->hpr[t] * v[(t * N2 + m) % N1];
                }
                o[j] = acc;
            }
#pragma unroll
            for (int c = 0; c < N1; c++) Tp[r * RS + c] = o[c];
        }
        __builtin_amdgcn_wave_barrier();
        bior_inv_level<K, N1 * 2>(Tp, r, tb);
    }
}
/* Floats per patch of the 16x16 kernels' work area [patch][16][17]: 16 * 17 = 272 = 16 (mod 32) put every other patch of a wave on the
 * same banks -- two-way conflicts in every pass of the 16x16 level, 62 % of the LDS-active cycles in conflicts
 * (profiles/r04_a_sq_counters.txt).  280 = 8 (mod 32): the four patches of a half-wave (eight lanes each) start 0 / 8 / 16 / 24 banks
 * apart, and rows 17 r + c as well as columns r + 17 c of eight lanes then fall on 32 distinct banks; two workgroups of 72 patches
 * still fit a CU (2 x 80 640 B). */
constexpr int kT16Patch = 16 * 17 + 8;
/* The same levels for the 16x16 kernel below, two rows (then two columns) per thread as packed pairs: a level of
 * side N1 takes N1/2 threads per patch, rows r and r + N1/2 travel as one v2f (one ds_read2 / ds_write2 per
 * element: the partner sits N1/2 rows, or N1/2 floats, away), every tap is one packed multiply or add.  Same
 * taps, order and unfused arithmetic as bior_fwd_level / bior_inv_level -> identical coefficients. */
template <int N1, bool FWD>
__device__ __forceinline__ void bior_taps2(const v2f* v, v2f* o, TbPtr tb) {
#pragma clang fp contract(off)
    constexpr int N2 = N1 / 2;
    if (FWD) {
#pragma unroll
        for (int j = 0; j < N2; j++) {
            v2f acc = v[bior_ext(2 * j, N1)] * tb->lpd[0];
#pragma unroll
            for (int t = 1; t < 10; t++) acc += v[bior_ext(t + 2 * j, N1)] * tb->lpd[t];
            o[j] = acc;
            v2f hi = v[bior_ext(4 + 2 * j, N1)] * tb->hpd[4];
            hi += v[bior_ext(5 + 2 * j, N1)] * tb->hpd[5];
            o[N2 + j] = hi;
        }
    } else {
#pragma unroll
        for (int m = 0; m < N2; m++) {
            v2f acc = v[m % N1] * tb->hpr[0];
#pragma unroll
            for (int t = 1; t < 10; t++) acc += v[(t * N2 + m) % N1] * tb->hpr[t];
            o[2 * m] = acc;
            v2f lo = v[(4 * N2 + m) % N1] * tb->lpr[4];
            lo += v[(5 * N2 + m) % N1] * tb->lpr[5];
            o[2 * m + 1] = lo;
        }
    }
}
template <int N1, bool FWD, bool ROWS>
__device__ __forceinline__ void bior16_pass2(float* Tp, int r, TbPtr tb) {
    constexpr int N2 = N1 / 2, RS = 17;
    v2f v[N1], o[N1];
#pragma unroll
    for (int c = 0; c < N1; c++)
        v[c] = ROWS ? v2f{Tp[r * RS + c], Tp[(r + N2) * RS + c]} : v2f{Tp[c * RS + r], Tp[c * RS + r + N2]};
    bior_taps2<N1, FWD>(v, o, tb);
#pragma unroll
    for (int c = 0; c < N1; c++) {
        if (ROWS) { Tp[r * RS + c] = o[c].x; Tp[(r + N2) * RS + c] = o[c].y; }
        else      { Tp[c * RS + r] = o[c].x; Tp[c * RS + r + N2] = o[c].y; }
    }
}
/* one level of all NP patches of the work area [patch][16][17]; all 256 threads call it */
template <int N1, bool FWD>
__device__ __forceinline__ void bior16_level_all(float* work, int NP, int tid, TbPtr tb) {
    constexpr int TPP = N1 / 2, PPI = kThreads / TPP, PSZ = kT16Patch;   /* threads per patch (one wavefront holds them all) */
    const int slot = tid / TPP, r = tid % TPP;
    for (int p0 = 0; p0 < NP; p0 += PPI) {
        const int patch = p0 + slot;
        if (patch < NP) {
            float* Tp = work + patch * PSZ;
            bior16_pass2<N1, FWD, FWD>(Tp, r, tb);         /* forward: rows first; inverse: columns first */
            __builtin_amdgcn_wave_barrier();
            bior16_pass2<N1, FWD, !FWD>(Tp, r, tb);
        }
    }
    __syncthreads();   /* the next level deals the patches to other threads */
}
/* work area: (kThreads / K) patches of K x (K+1) floats */
template <int K> constexpr int bior_tmp_floats() { return (kThreads / K) * K * (K + 1); }
template <int K, bool FWD>
__device__ __noinline__ void bior2d_fast(float* S, float* tmp, int np, TbPtr tb) {
    constexpr int PPI = kThreads / K, RS = K + 1;
    const int tid = threadIdx.x, slot = tid / K, r = tid % K;
    float* Tp = tmp + slot * K * RS;
    for (int p0 = 0; p0 < np; p0 += PPI) {
        const int patch = p0 + slot;
        if (patch < np) {     /* uniform for the K threads of a patch */
            float* X = S + (size_t)patch * K * K + r * K;
            float x[K];
#pragma unroll
            for (int c4 = 0; c4 < K; c4 += 4) {
                const v4f q = *reinterpret_cast<const v4f*>(X + c4);
                x[c4] = q[0]; x[c4 + 1] = q[1]; x[c4 + 2] = q[2]; x[c4 + 3] = q[3];
            }
#pragma unroll
            for (int c = 0; c < K; c++) Tp[r * RS + c] = x[c];
            __builtin_amdgcn_wave_barrier();
            if (FWD) bior_fwd_level<K, K>(Tp, r, tb); else bior_inv_level<K, 2>(Tp, r, tb);
#pragma unroll
            for (int c = 0; c < K; c++) x[c] = Tp[r * RS + c];
#pragma unroll
            for (int c4 = 0; c4 < K; c4 += 4) *reinterpret_cast<v4f*>(X + c4) = v4f{x[c4], x[c4 + 1], x[c4 + 2], x[c4 + 3]};
            __builtin_amdgcn_wave_barrier();
        }
    }
    __syncthreads();
}

/* 2-D forward transform of `np` patches stored back to back at S (k*k floats each).  All threads
 * of the workgroup call this. */

/* barrier between the two passes of a patch: when a patch is exactly one wavefront (k = 8) the LDS
 * traffic of a patch stays inside that wave, whose DS operations execute in order -- no workgroup
 * barrier needed */
#define PATCH_SYNC() do { if (wave_local) __builtin_amdgcn_wave_barrier(); else __syncthreads(); } while (0)

/* 2-D DCT of all patches (bm3d.cpp:745-757; inverse bm3d.cpp:1039-1071), K threads per patch: a thread owns a ROW of its patch in the
 * first pass and a COLUMN in the second, K values in registers and K outputs each, kThreads / K patches per iteration through a
 * work area [patch][K][K+1] in LDS (the odd pitch keeps both directions free of bank conflicts).  The stack S may live in LDS or in
 * the HBM slice of k_group_big: a round of 16 / 21 / 32 patches costs two barriers and one round trip to wherever S is, where
 * round 4's thread-per-coefficient form paid them per patch (16x16) or per four (8x8).  The same sums in the same order: the same
 * results. */
template <int K> constexpr int dct_tmp_floats() { return (kThreads / K) * K * (K + 1); }
template <int K>
__device__ __noinline__ void fwd2d_dct(float* S, float* tmp, int np, TbPtr tb) {
    constexpr int K2 = K * K, PPI = kThreads / K, RS = K + 1;
    constexpr bool wave_local = (64 % K) == 0;   /* the K threads of a patch in one wavefront */
    const int tid = threadIdx.x, slot = tid / K, r = tid % K;
    float* Tp = tmp + slot * K * RS;
    for (int p0 = 0; p0 < np; p0 += PPI) {
        const int patch = p0 + slot;
        const bool on = slot < PPI && patch < np;
        float* X = S + (size_t)patch * K2;
        if (on) {
            float x[K];
#pragma unroll
            for (int t = 0; t < K; t++) x[t] = X[r * K + t];
#pragma unroll 1
            for (int j = 0; j < K; j++) {   /* (rolled: one scalar load of K cosines per output; unrolled, the K^2 constants cost 200 registers) */
                float a = 0.0f;
#pragma unroll
                for (int t = 0; t < K; t++) a += x[t] * tb->cos2[j * K + t];
                Tp[r * RS + j] = 2.0f * a;
            }
        }
        PATCH_SYNC();
        if (on) {
            float c[K];
#pragma unroll
            for (int t = 0; t < K; t++) c[t] = Tp[t * RS + r];
#pragma unroll 1
            for (int i = 0; i < K; i++) {
                float a = 0.0f;
#pragma unroll
                for (int t = 0; t < K; t++) a += c[t] * tb->cos2[i * K + t];
                X[i * K + r] = 2.0f * a * tb->cn2[i * K + r];
            }
        }
        PATCH_SYNC();
    }
    __syncthreads();
}
template <int K>
__device__ __noinline__ void inv2d_dct(float* S, float* tmp, int np, TbPtr tb) {
    constexpr int K2 = K * K, PPI = kThreads / K, RS = K + 1;
    constexpr bool wave_local = (64 % K) == 0;
    const int tid = threadIdx.x, slot = tid / K, r = tid % K;
    float* Tp = tmp + slot * K * RS;
    const float c2 = tb->coef2inv;
    for (int p0 = 0; p0 < np; p0 += PPI) {
        const int patch = p0 + slot;
        const bool on = slot < PPI && patch < np;
        float* X = S + (size_t)patch * K2;
        if (on) {   /* row r: times coef_norm_inv, REDFT01 along the row */
            float y[K];
#pragma unroll
            for (int v = 0; v < K; v++) y[v] = X[r * K + v] * tb->cni2[r * K + v];
#pragma unroll 1
            for (int j = 0; j < K; j++) {   /* (rolled: one scalar load of K cosines per output; unrolled, the K^2 constants cost 200 registers) */
                float a = 0.0f;
#pragma unroll
                for (int v = 1; v < K; v++) a += y[v] * tb->cos2[v * K + j];
                Tp[r * RS + j] = y[0] + 2.0f * a;
            }
        }
        PATCH_SYNC();
        if (on) {   /* column r */
            float c[K];
#pragma unroll
            for (int u = 0; u < K; u++) c[u] = Tp[u * RS + r];
#pragma unroll 1
            for (int i = 0; i < K; i++) {
                float a = 0.0f;
#pragma unroll
                for (int u = 1; u < K; u++) a += c[u] * tb->cos2[u * K + i];
                X[i * K + r] = c2 * (c[0] + 2.0f * a);
            }
        }
        PATCH_SYNC();
    }
    __syncthreads();
}

/* patches of more coefficients than the workgroup has threads (k > 16): one patch at a time, the threads stride over its coefficients;
 * the same sums in the same order as the general form below */
__device__ __noinline__ void fwd2d_big(float* S, float* Tm, int np, int k, unsigned tau2, TbPtr tb) {
    const int k2 = k * k, tid = threadIdx.x;
    for (int patch = 0; patch < np; patch++) {
        float* X = S + (size_t)patch * k2;
        if (tau2 == 5) {
            for (int pq = tid; pq < k2; pq += kThreads) { const int i = pq / k, j = pq % k; float a = 0.0f; for (int t = 0; t < k; t++) a += X[i * k + t] * tb->cos2[j * k + t]; Tm[pq] = 2.0f * a; }
            __syncthreads();
            for (int pq = tid; pq < k2; pq += kThreads) { const int i = pq / k, j = pq % k; float a = 0.0f; for (int t = 0; t < k; t++) a += Tm[t * k + j] * tb->cos2[i * k + t]; X[pq] = 2.0f * a * tb->cn2[pq]; }
            __syncthreads();
        } else {
            for (int N1 = k; N1 > 1; N1 /= 2) {
                const int N2 = N1 / 2;
                for (int pq = tid; pq < k2; pq += kThreads) {
                    const int i = pq / k, j = pq % k;
                    if (i < N1 && j < N1) {
                        const bool lo = j < N2; const int jj = lo ? j : j - N2;
                        TbFloats f = lo ? tb->lpd : tb->hpd;
                        float a = 0.0f;
                        for (int t = 0; t < 10; t++) a += X[i * k + per_ext(t + 2 * jj, 4, N1)] * f[t];
                        Tm[pq] = a;
                    }
                }
                __syncthreads();
                for (int pq = tid; pq < k2; pq += kThreads) {
                    const int i = pq / k, j = pq % k;
                    if (i < N1 && j < N1) {
                        const bool lo = i < N2; const int ii = lo ? i : i - N2;
                        TbFloats f = lo ? tb->lpd : tb->hpd;
                        float a = 0.0f;
                        for (int t = 0; t < 10; t++) a += Tm[per_ext(t + 2 * ii, 4, N1) * k + j] * f[t];
                        X[pq] = a;
                    }
                }
                __syncthreads();
            }
        }
    }
    __syncthreads();
}
__device__ __noinline__ void inv2d_big(float* S, float* Tm, int np, int k, unsigned tau2, TbPtr tb) {
    const int k2 = k * k, tid = threadIdx.x;
    for (int patch = 0; patch < np; patch++) {
        float* X = S + (size_t)patch * k2;
        if (tau2 == 5) {
            for (int pq = tid; pq < k2; pq += kThreads) {
                const int i = pq / k, j = pq % k;
                float a = 0.0f;
                for (int v = 1; v < k; v++) a += X[i * k + v] * tb->cni2[i * k + v] * tb->cos2[v * k + j];
                Tm[pq] = X[i * k] * tb->cni2[i * k] + 2.0f * a;
            }
            __syncthreads();
            for (int pq = tid; pq < k2; pq += kThreads) {
                const int i = pq / k, j = pq % k;
                float a = 0.0f;
                for (int u = 1; u < k; u++) a += Tm[u * k + j] * tb->cos2[u * k + i];
                X[pq] = tb->coef2inv * (Tm[j] + 2.0f * a);
            }
            __syncthreads();
        } else {
            for (int N1 = 2; N1 <= k; N1 *= 2) {
                const int N2 = N1 / 2;
                for (int pq = tid; pq < k2; pq += kThreads) {
                    const int i = pq / k, j = pq % k;
                    if (i < N1 && j < N1) {
                        const int m = i / 2; TbFloats f = (i & 1) ? tb->lpr : tb->hpr;
                        float a = 0.0f;
                        for (int t = 0; t < 10; t++) a += f[t] * X[((t * N2 + m) % N1) * k + j];
                        Tm[pq] = a;
                    }
                }
                __syncthreads();
                for (int pq = tid; pq < k2; pq += kThreads) {
                    const int i = pq / k, j = pq % k;
                    if (i < N1 && j < N1) {
                        const int m = j / 2; TbFloats f = (j & 1) ? tb->lpr : tb->hpr;
                        float a = 0.0f;
                        for (int t = 0; t < 10; t++) a += f[t] * Tm[i * k + (t * N2 + m) % N1];
                        X[pq] = a;
                    }
                }
                __syncthreads();
            }
        }
    }
    __syncthreads();
}

__device__ void fwd2d(float* S, float* tmp, int np, int k, unsigned tau2, TbPtr tb) {
    if (k * k > kThreads) return fwd2d_big(S, tmp, np, k, tau2, tb);
    if (tau2 == 5) {
        if (k == 8) return fwd2d_dct<8>(S, tmp, np, tb);
        if (k == 12) return fwd2d_dct<12>(S, tmp, np, tb);
        if (k == 16) return fwd2d_dct<16>(S, tmp, np, tb);
    }   /* (other sizes: the run-time form below) */
    if (tau2 == 7 && k == 16) return bior2d_fast<16, true>(S, tmp, np, tb);
    if (tau2 == 7 && k == 8) return bior2d_fast<8, true>(S, tmp, np, tb);
    const int k2 = k * k, tid = threadIdx.x;
    const bool wave_local = k2 == 64;
    const int ppi = kThreads / k2 > 0 ? kThreads / k2 : 1; /* patches per iteration */
    for (int p0 = 0; p0 < np; p0 += ppi) {
        const int slot = tid / k2, pq = tid % k2, patch = p0 + slot;
        const bool on = slot < ppi && patch < np;
        float* X = S + (size_t)patch * k2;
        float* Tm = tmp + slot * k2;
        const int i = pq / k, j = pq % k;
        if (tau2 == 5) { /* DCT: REDFT10 rows, REDFT10 columns, * coef_norm (bm3d.cpp:745-757) */
            if (on) { float a = 0.0f; for (int t = 0; t < k; t++) a += X[i * k + t] * tb->cos2[j * k + t]; Tm[pq] = 2.0f * a; }
            PATCH_SYNC();
            if (on) { float a = 0.0f; for (int t = 0; t < k; t++) a += Tm[t * k + j] * tb->cos2[i * k + t]; X[pq] = 2.0f * a * tb->cn2[pq]; }
            PATCH_SYNC();
        } else {         /* bior1.5 (lib_transforms.cpp:46-120) */
            for (int N1 = k; N1 > 1; N1 /= 2) {
                const int N2 = N1 / 2;
                if (on && i < N1 && j < N1) {
                    const bool lo = j < N2; const int jj = lo ? j : j - N2;
                    TbFloats f = lo ? tb->lpd : tb->hpd;
                    float a = 0.0f;
                    for (int t = 0; t < 10; t++) a += X[i * k + per_ext(t + 2 * jj, 4, N1)] * f[t];
                    Tm[pq] = a;
                }
                PATCH_SYNC();
                if (on && i < N1 && j < N1) {
                    const bool lo = i < N2; const int ii = lo ? i : i - N2;
                    TbFloats f = lo ? tb->lpd : tb->hpd;
                    float a = 0.0f;
                    for (int t = 0; t < 10; t++) a += Tm[per_ext(t + 2 * ii, 4, N1) * k + j] * f[t];
                    X[pq] = a;
                }
                PATCH_SYNC();
            }
        }
    }
    __syncthreads();
}
__device__ void inv2d(float* S, float* tmp, int np, int k, unsigned tau2, TbPtr tb) {
    if (k * k > kThreads) return inv2d_big(S, tmp, np, k, tau2, tb);
    if (tau2 == 5) {
        if (k == 8) return inv2d_dct<8>(S, tmp, np, tb);
        if (k == 12) return inv2d_dct<12>(S, tmp, np, tb);
        if (k == 16) return inv2d_dct<16>(S, tmp, np, tb);
    }
    if (tau2 == 7 && k == 16) return bior2d_fast<16, false>(S, tmp, np, tb);
    if (tau2 == 7 && k == 8) return bior2d_fast<8, false>(S, tmp, np, tb);
    const int k2 = k * k, tid = threadIdx.x;
    const bool wave_local = k2 == 64;
    const int ppi = kThreads / k2 > 0 ? kThreads / k2 : 1;
    for (int p0 = 0; p0 < np; p0 += ppi) {
        const int slot = tid / k2, pq = tid % k2, patch = p0 + slot;
        const bool on = slot < ppi && patch < np;
        float* X = S + (size_t)patch * k2;
        float* Tm = tmp + slot * k2;
        const int i = pq / k, j = pq % k;
        if (tau2 == 5) { /* bm3d.cpp:1039-1071 */
            if (on) {
                float a = 0.0f;
                for (int v = 1; v < k; v++) a += X[i * k + v] * tb->cni2[i * k + v] * tb->cos2[v * k + j];
                Tm[pq] = X[i * k] * tb->cni2[i * k] + 2.0f * a;
            }
            PATCH_SYNC();
            if (on) {
                float a = 0.0f;
                for (int u = 1; u < k; u++) a += Tm[u * k + j] * tb->cos2[u * k + i];
                X[pq] = tb->coef2inv * (Tm[j] + 2.0f * a);
            }
            PATCH_SYNC();
        } else { /* lib_transforms.cpp:135-204 */
            for (int N1 = 2; N1 <= k; N1 *= 2) {
                const int N2 = N1 / 2;
                if (on && i < N1 && j < N1) { /* columns: out[2m] = high, out[2m+1] = low */
                    const int m = i / 2; TbFloats f = (i & 1) ? tb->lpr : tb->hpr;
                    float a = 0.0f;
                    for (int t = 0; t < 10; t++) a += f[t] * X[((t * N2 + m) % N1) * k + j];
                    Tm[pq] = a;
                }
                PATCH_SYNC();
                if (on && i < N1 && j < N1) { /* rows */
                    const int m = j / 2; TbFloats f = (j & 1) ? tb->lpr : tb->hpr;
                    float a = 0.0f;
                    for (int t = 0; t < 10; t++) a += f[t] * Tm[i * k + (t * N2 + m) % N1];
                    X[pq] = a;
                }
                PATCH_SYNC();
            }
        }
    }
    __syncthreads();
}

/* SADCT bookkeeping of group g (pre-pass output) */
__device__ __forceinline__ ShRef group_shape(const GroupArgs& a, unsigned g) {
    return *reinterpret_cast<const __attribute__((address_space(4))) ShapeInfo*>(
        (const __attribute__((address_space(4))) char*)a.gshape + (size_t)g * sizeof(ShapeInfo));
}
__device__ __forceinline__ ShRefBig group_shape_big(const GroupArgs& a, unsigned g) {
    return *reinterpret_cast<const __attribute__((address_space(4))) ShapeInfoBig*>(
        (const __attribute__((address_space(4))) char*)a.gshape + (size_t)g * sizeof(ShapeInfoBig));
}

/* XCD-aware group numbering: hardware deals consecutive workgroup ids round-robin to the 8 XCDs (each with an L2 of its
 * own), so workgroup b takes group (b % 8) * per_xcd + b / 8: an XCD works its way through a contiguous band of reference
 * patches and its L2 fetches that band's window rows once instead of every XCD fetching every row.  The launch rounds
 * grid.x up to 8 * per_xcd; indices past the last group return. */
__device__ __forceinline__ unsigned xcd_group_index(const GroupArgs& a) {
    const unsigned per_xcd = (a.n_groups + 7) / 8;
    return (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
}

__device__ __forceinline__ void dct8_fwd(float* x) {
    const float a0 = 0.35355339059327376f;   /* 1/sqrt(8) */
    const float h = 0.5f;
    const float c1 = 0.98078528040323044f, c2 = 0.92387953251128674f, c3 = 0.83146961230254524f, c4 = 0.70710678118654752f,
                c5 = 0.55557023301960222f, c6 = 0.38268343236508977f, c7 = 0.19509032201612827f;
    const float s0 = x[0] + x[7], s1 = x[1] + x[6], s2 = x[2] + x[5], s3 = x[3] + x[4];
    const float d0 = x[0] - x[7], d1 = x[1] - x[6], d2 = x[2] - x[5], d3 = x[3] - x[4];
    const float p0 = s0 + s3, p1 = s1 + s2, m0 = s0 - s3, m1 = s1 - s2;
    x[0] = a0 * (p0 + p1);
    x[4] = (h * c4) * (p0 - p1);
    x[2] = h * (c2 * m0 + c6 * m1);
    x[6] = h * (c6 * m0 - c2 * m1);
    x[1] = h * (c1 * d0 + c3 * d1 + c5 * d2 + c7 * d3);
    x[3] = h * (c3 * d0 - c7 * d1 - c1 * d2 - c5 * d3);
    x[5] = h * (c5 * d0 - c1 * d1 + c7 * d2 + c3 * d3);
    x[7] = h * (c7 * d0 - c5 * d1 + c3 * d2 - c1 * d3);
}
__device__ __forceinline__ void dct8_inv(float* X) {
    const float a0 = 0.35355339059327376f;
    const float h = 0.5f;
    const float c1 = 0.98078528040323044f, c2 = 0.92387953251128674f, c3 = 0.83146961230254524f, c4 = 0.70710678118654752f,
                c5 = 0.55557023301960222f, c6 = 0.38268343236508977f, c7 = 0.19509032201612827f;
    const float e0 = a0 * X[0] + (h * c4) * X[4], e1 = a0 * X[0] - (h * c4) * X[4];
    const float f0 = h * (c2 * X[2] + c6 * X[6]), f1 = h * (c6 * X[2] - c2 * X[6]);
    const float E0 = e0 + f0, E1 = e1 + f1, E2 = e1 - f1, E3 = e0 - f0;
    const float O0 = h * (c1 * X[1] + c3 * X[3] + c5 * X[5] + c7 * X[7]);
    const float O1 = h * (c3 * X[1] - c7 * X[3] - c1 * X[5] - c5 * X[7]);
    const float O2 = h * (c5 * X[1] - c1 * X[3] + c7 * X[5] + c3 * X[7]);
    const float O3 = h * (c7 * X[1] - c5 * X[3] + c3 * X[5] - c1 * X[7]);
    X[0] = E0 + O0; X[7] = E0 - O0;
    X[1] = E1 + O1; X[6] = E1 - O1;
    X[2] = E2 + O2; X[5] = E2 - O2;
    X[3] = E3 + O3; X[4] = E3 - O3;
}

/* orthonormal 16-point DCT-II / its inverse in registers: even outputs = the 8-point transform of the folded sums
 * (scaled by 1/sqrt2), odd outputs = an 8x8 product of the folded differences with cos((2n+1)(2k+1) pi/32) */
constexpr float kCos32[16] = {1.0f, 0.99518472667219689f, 0.98078528040323044f, 0.95694033573220887f, 0.92387953251128674f,
                              0.88192126434835503f, 0.83146961230254524f, 0.77301045336273696f, 0.70710678118654752f,
                              0.63439328416364549f, 0.55557023301960222f, 0.47139673682599764f, 0.38268343236508977f,
                              0.29028467725446236f, 0.19509032201612827f, 0.09801714032956060f};
constexpr float cos32(int m) {   /* cos(m pi / 32), m odd */
    m &= 63;
    return m < 16 ? kCos32[m] : m < 32 ? -kCos32[32 - m] : m < 48 ? -kCos32[m - 32] : kCos32[64 - m];
}
__device__ __forceinline__ void dct16_fwd(float* x) {
    const float r2 = 0.70710678118654752f, h = 0.35355339059327376f;   /* sqrt(2/16) */
    float e[8], o[8];
#pragma unroll
    for (int n = 0; n < 8; n++) { e[n] = (x[n] + x[15 - n]) * r2; o[n] = (x[n] - x[15 - n]) * h; }
    dct8_fwd(e);
#pragma unroll
    for (int u = 0; u < 8; u++) {
        float acc = 0.0f;
#pragma unroll
        for (int n = 0; n < 8; n++) acc += o[n] * cos32((2 * n + 1) * (2 * u + 1));
        x[2 * u] = e[u]; x[2 * u + 1] = acc;
    }
}
__device__ __forceinline__ void dct16_inv(float* X) {
    const float r2 = 0.70710678118654752f, h = 0.35355339059327376f;
    float e[8], o[8];
#pragma unroll
    for (int u = 0; u < 8; u++) { e[u] = X[2 * u] * r2; o[u] = X[2 * u + 1] * h; }
    dct8_inv(e);
#pragma unroll
    for (int n = 0; n < 8; n++) {
        float acc = 0.0f;
#pragma unroll
        for (int u = 0; u < 8; u++) acc += o[u] * cos32((2 * n + 1) * (2 * u + 1));
        X[n] = e[n] + acc; X[15 - n] = e[n] - acc;
    }
}
} /* namespace */
} /* namespace lfbm5d */
#endif
