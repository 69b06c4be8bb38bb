/*
 * lfbm5d_oracle.cpp -- CPU restatement of the LFBM5D hot path.  TEST INFRASTRUCTURE ONLY
 * (see lfbm5d_oracle.h for the scope, the reference lines followed and the parity-pin status).
 *
 * Written from the reference's behaviour, not from its text: flat arrays instead of nested
 * vectors, per-patch 2-D transforms instead of sliding tables (same values), per-reference score
 * rows instead of whole-image candidate tables (same values), OpenMP over independent units with
 * the aggregation kept in the reference's sequential order so results do not depend on threads.
 */
#include "lfbm5d_oracle.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

const double kSqrt2 = 1.414213562373095;      /* core:33 */
const double kSqrt2Inv = 0.7071067811865475;  /* core:34 */
const double kPi = 3.14159265358979323846;

int g_threads = 0; /* 0 = OpenMP default */
std::vector<unsigned> g_last_windows; /* run API: processed SAI of every window of the last step, in order */
int g_tiles = 1; /* run API: 1 = untiled (nb_threads == 1 semantics), > 1 = the reference's OpenMP tile mode with that many tiles */
double g_time_limit = 0.0; /* run API: stop after the window that exceeds it (0 = none) */

double now_s() {
    using namespace std::chrono;
    return duration<double>(steady_clock::now().time_since_epoch()).count();
}

/* ------------------------------------------------------------------------------------------ */
/* FFTW r2r kinds restated (published definitions; see header)                                 */
/* ------------------------------------------------------------------------------------------ */
const unsigned kMaxDct = 64;
const unsigned kMaxAng = 17 * 17;   /* SAIs of the largest angular window (17 x 17: aswSize 8) */
struct CosTables {
    std::vector<double> t[kMaxDct + 1]; /* t[n][k*n + j] = cos(pi (j+1/2) k / n) */
    CosTables() {
        for (unsigned n = 1; n <= kMaxDct; n++) {
            t[n].resize(n * n);
            for (unsigned k = 0; k < n; k++)
                for (unsigned j = 0; j < n; j++) t[n][k * n + j] = std::cos(kPi * (j + 0.5) * k / n);
        }
    }
};
const CosTables g_cos;

/* y_k = 2 sum_j x_j cos(pi (j+1/2) k / n), stride-aware, double in/out */
void redft10_d(const double* x, unsigned xs, double* y, unsigned ys, unsigned n) {
    const double* c = g_cos.t[n].data();
    double tmp[kMaxDct];
    for (unsigned k = 0; k < n; k++) {
        double a = 0.0;
        for (unsigned j = 0; j < n; j++) a += x[j * xs] * c[k * n + j];
        tmp[k] = 2.0 * a;
    }
    for (unsigned k = 0; k < n; k++) y[k * ys] = tmp[k];
}
/* y_j = X_0 + 2 sum_{k>=1} X_k cos(pi k (j+1/2) / n) */
void redft01_d(const double* x, unsigned xs, double* y, unsigned ys, unsigned n) {
    const double* c = g_cos.t[n].data();
    double tmp[kMaxDct];
    for (unsigned j = 0; j < n; j++) {
        double a = 0.0;
        for (unsigned k = 1; k < n; k++) a += x[k * xs] * c[k * n + j];
        tmp[j] = x[0] + 2.0 * a;
    }
    for (unsigned j = 0; j < n; j++) y[j * ys] = tmp[j];
}
/* rank-2 plan on an n0 x n1 row-major array, in place (double) */
void r2r_2d(double* a, unsigned n0, unsigned n1, bool forward) {
    for (unsigned i = 0; i < n0; i++) {
        if (forward) redft10_d(a + i * n1, 1, a + i * n1, 1, n1);
        else         redft01_d(a + i * n1, 1, a + i * n1, 1, n1);
    }
    for (unsigned j = 0; j < n1; j++) {
        if (forward) redft10_d(a + j, n1, a + j, n1, n0);
        else         redft01_d(a + j, n1, a + j, n1, n0);
    }
}

/* ------------------------------------------------------------------------------------------ */
/* Normalisation tables                                                                        */
/* ------------------------------------------------------------------------------------------ */
/* bm3d.cpp:1148-1168 */
void dct2d_norms(unsigned k, std::vector<float>& cn, std::vector<float>& cni) {
    cn.resize(k * k); cni.resize(k * k);
    const float coef = 0.5f / ((float)k);
    for (unsigned i = 0; i < k; i++)
        for (unsigned j = 0; j < k; j++) {
            if (i == 0 && j == 0) { cn[i * k + j] = 0.5f * coef; cni[i * k + j] = 2.0f; }
            else if (i * j == 0)  { cn[i * k + j] = (float)(kSqrt2Inv * coef); cni[i * k + j] = (float)kSqrt2; }
            else                  { cn[i * k + j] = 1.0f * coef; cni[i * k + j] = 1.0f; }
        }
}
/* core:3191-3218 */
void dct4d_norms(unsigned aw, unsigned ah, std::vector<float>& cn, std::vector<float>& cni) {
    cn.resize(aw * ah); cni.resize(aw * ah);
    const float coef = 0.5f / (std::sqrt((float)aw) * std::sqrt((float)ah));
    for (unsigned i = 0; i < ah; i++)
        for (unsigned j = 0; j < aw; j++) {
            if (i == 0 && j == 0) { cn[i * aw + j] = (float)(0.5f * coef); cni[i * aw + j] = 2.0f; }
            else if (i * j == 0)  { cn[i * aw + j] = (float)(kSqrt2Inv * coef); cni[i * aw + j] = (float)kSqrt2; }
            else                  { cn[i * aw + j] = (float)(1.0f * coef); cni[i * aw + j] = 1.0f; }
        }
}
/* core:3229-3252 and :3262-3276 (same shape: 1-D DCT of length n scaled to 2x / 1x orthonormal) */
void dct1d_norms(unsigned n, float* cn, float* cni) {
    const float coef = (float)((float)kSqrt2 / std::sqrt((double)n));
    cn[0] = (float)(kSqrt2Inv * coef);
    cni[0] = (float)kSqrt2;
    for (unsigned i = 1; i < n; i++) { cn[i] = coef; cni[i] = 1.0f; }
}

/* bm3d.cpp:1101-1146 */
void kaiser_window(unsigned k, std::vector<float>& w) {
    w.assign(k * k, 1.0f);
    static const float q8[4][4] = {{0.1924f, 0.2989f, 0.3846f, 0.4325f},
                                   {0.2989f, 0.4642f, 0.5974f, 0.6717f},
                                   {0.3846f, 0.5974f, 0.7688f, 0.8644f},
                                   {0.4325f, 0.6717f, 0.8644f, 0.9718f}};
    static const float q12[6][6] = {{0.1924f, 0.2615f, 0.3251f, 0.3782f, 0.4163f, 0.4362f},
                                    {0.2615f, 0.3554f, 0.4419f, 0.5139f, 0.5657f, 0.5927f},
                                    {0.3251f, 0.4419f, 0.5494f, 0.6390f, 0.7033f, 0.7369f},
                                    {0.3782f, 0.5139f, 0.6390f, 0.7433f, 0.8181f, 0.8572f},
                                    {0.4163f, 0.5657f, 0.7033f, 0.8181f, 0.9005f, 0.9435f},
                                    {0.4362f, 0.5927f, 0.7369f, 0.8572f, 0.9435f, 0.9885f}};
    if (k != 8 && k != 12) return; /* any other size: all ones (bm3d.cpp:1144-1146) */
    const unsigned h = k / 2;
    for (unsigned i = 0; i < k; i++)
        for (unsigned j = 0; j < k; j++) {
            const unsigned a = i < h ? i : k - 1 - i, b = j < h ? j : k - 1 - j;
            w[i * k + j] = (k == 8) ? q8[a][b] : q12[a][b];
        }
}

/* ------------------------------------------------------------------------------------------ */
/* Wavelets (lib_transforms.cpp)                                                               */
/* ------------------------------------------------------------------------------------------ */
/* lib_transforms.cpp:403-433 */
void haar_fwd(float* v, unsigned n) {
    float tmp[kMaxDct];
    const float s = (float)kSqrt2Inv;
    while (n > 1) {
        const unsigned h = n / 2;
        for (unsigned k = 0; k < h; k++) {
            const float a = v[2 * k], b = v[2 * k + 1];
            tmp[k] = (a + b) * s;
            tmp[h + k] = (a - b) * s;
        }
        for (unsigned k = 0; k < n; k++) v[k] = tmp[k];
        n = h;
    }
}
/* lib_transforms.cpp:447-471 */
void haar_inv(float* v, unsigned n) {
    float tmp[kMaxDct];
    const float s = (float)kSqrt2Inv;
    for (unsigned h = 1; h < n; h *= 2) {
        for (unsigned k = 0; k < h; k++) {
            const float a = v[k], b = v[h + k];
            tmp[2 * k] = (a + b) * s;
            tmp[2 * k + 1] = (a - b) * s;
        }
        for (unsigned k = 0; k < 2 * h; k++) v[k] = tmp[k];
    }
}
/* lib_transforms.cpp:290-321 (sums to the first half, differences to the second, both recursed) */
void hadamard(float* v, unsigned n) {
    if (n <= 1) return;
    if (n == 2) { const float a = v[0], b = v[1]; v[0] = a + b; v[1] = a - b; return; }
    float tmp[kMaxDct];
    const unsigned h = n / 2;
    for (unsigned k = 0; k < h; k++) {
        const float a = v[2 * k], b = v[2 * k + 1];
        v[k] = a + b;
        tmp[k] = a - b;
    }
    for (unsigned k = 0; k < h; k++) v[h + k] = tmp[k];
    hadamard(v, h);
    hadamard(v + h, h);
}

struct Bior15 {
    float lpd[10], hpd[10], lpr[10], hpr[10];
    Bior15() { /* lib_transforms.cpp:215-277 */
        const float cn = 1.f / (std::sqrt(2.f) * 128.f);
        const float s = 1.f / std::sqrt(2.f);
        const float a[10] = {3.f, -3.f, -22.f, 22.f, 128.f, 128.f, 22.f, -22.f, -3.f, 3.f};
        const float b[10] = {3.f, 3.f, -22.f, -22.f, 128.f, -128.f, 22.f, 22.f, -3.f, -3.f};
        for (int i = 0; i < 10; i++) { lpd[i] = a[i] * cn; hpr[i] = b[i] * cn; hpd[i] = 0.f; lpr[i] = 0.f; }
        hpd[4] = -s; hpd[5] = s;
        lpr[4] = s;  lpr[5] = s;
    }
};
const Bior15 g_bior;

unsigned ilog2(unsigned n) { unsigned k = 1, r = 0; while (k < n) { k *= 2; r++; } return r; }
/* periodic extension index (lib_transforms.cpp:352-373): sample j of the extended signal */
inline unsigned per_ext(int j, int L, int N) { int m = (j - L) % N; if (m < 0) m += N; return (unsigned)m; }

/* lib_transforms.cpp:46-120; out is n x n contiguous */
void bior_fwd(const float* in, unsigned in_stride, float* out, unsigned n) {
    for (unsigned i = 0; i < n; i++)
        for (unsigned j = 0; j < n; j++) out[i * n + j] = in[i * in_stride + j];
    const unsigned levels = ilog2(n);
    unsigned N1 = n, N2 = n / 2;
    const int L = 4; /* S_1/2 - 1 */
    float ext[kMaxDct + 8];
    for (unsigned it = 0; it < levels; it++) {
        for (unsigned i = 0; i < N1; i++) { /* rows */
            for (unsigned j = 0; j < N1 + 2 * L; j++) ext[j] = out[i * n + per_ext((int)j, L, (int)N1)];
            for (unsigned j = 0; j < N2; j++) {
                float vl = 0.f, vh = 0.f;
                for (unsigned t = 0; t < 10; t++) { vl += ext[t + 2 * j] * g_bior.lpd[t]; vh += ext[t + 2 * j] * g_bior.hpd[t]; }
                out[i * n + j] = vl;
                out[i * n + j + N2] = vh;
            }
        }
        for (unsigned j = 0; j < N1; j++) { /* columns */
            for (unsigned i = 0; i < N1 + 2 * L; i++) ext[i] = out[per_ext((int)i, L, (int)N1) * n + j];
            for (unsigned i = 0; i < N2; i++) {
                float vl = 0.f, vh = 0.f;
                for (unsigned t = 0; t < 10; t++) { vl += ext[t + 2 * i] * g_bior.lpd[t]; vh += ext[t + 2 * i] * g_bior.hpd[t]; }
                out[i * n + j] = vl;
                out[(i + N2) * n + j] = vh;
            }
        }
        N1 /= 2; N2 /= 2;
    }
}
/* lib_transforms.cpp:135-204 */
void bior_inv(float* sig, unsigned n) {
    const unsigned levels = ilog2(n);
    unsigned N1 = 2, N2 = 1;
    std::vector<float> ext(5 * n);
    for (unsigned it = 0; it < levels; it++) {
        const int L = 4 * (int)N2;
        const unsigned len = N1 + 4 * N1;
        for (unsigned j = 0; j < N1; j++) { /* columns */
            for (unsigned i = 0; i < len; i++) ext[i] = sig[per_ext((int)i, L, (int)N1) * n + j];
            for (unsigned i = 0; i < N2; i++) {
                float vl = 0.f, vh = 0.f;
                for (unsigned t = 0; t < 10; t++) { vl += g_bior.lpr[t] * ext[t * N2 + i]; vh += g_bior.hpr[t] * ext[t * N2 + i]; }
                sig[(i * 2) * n + j] = vh;
                sig[(i * 2 + 1) * n + j] = vl;
            }
        }
        for (unsigned i = 0; i < N1; i++) { /* rows */
            for (unsigned j = 0; j < len; j++) ext[j] = sig[i * n + per_ext((int)j, L, (int)N1)];
            for (unsigned j = 0; j < N2; j++) {
                float vl = 0.f, vh = 0.f;
                for (unsigned t = 0; t < 10; t++) { vl += g_bior.lpr[t] * ext[t * N2 + j]; vh += g_bior.hpr[t] * ext[t * N2 + j]; }
                sig[i * n + j * 2] = vh;
                sig[i * n + j * 2 + 1] = vl;
            }
        }
        N1 *= 2; N2 *= 2;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* 2-D patch DCT, 4-D angular DCT, SADCT                                                        */
/* ------------------------------------------------------------------------------------------ */
struct Norms2D { unsigned k = 0; std::vector<float> cn, cni; };
struct Norms4D { unsigned aw = 0, ah = 0; std::vector<float> cn, cni; };

/* bm3d.cpp:745-757 / core:1791-1803: REDFT10 x REDFT10 then * coef_norm */
void dct2d_fwd(const float* in, unsigned in_stride, float* out, unsigned k, const Norms2D& nm) {
    double a[kMaxDct * kMaxDct / 4];
    for (unsigned i = 0; i < k; i++)
        for (unsigned j = 0; j < k; j++) a[i * k + j] = in[i * in_stride + j];
    r2r_2d(a, k, k, true);
    for (unsigned i = 0; i < k * k; i++) out[i] = (float)a[i] * nm.cn[i];
}
/* bm3d.cpp:1039-1071 */
void dct2d_inv(float* patch, unsigned k, const Norms2D& nm) {
    double a[kMaxDct * kMaxDct / 4];
    for (unsigned i = 0; i < k * k; i++) a[i] = (double)(patch[i] * nm.cni[i]);
    r2r_2d(a, k, k, false);
    const float coef = 1.0f / (float)(k * 2);
    for (unsigned i = 0; i < k * k; i++) patch[i] = coef * (float)a[i];
}
/* core:1862-1901 */
void dct4d_fwd(float* v, unsigned aw, unsigned ah, const Norms4D& nm) {
    double a[kMaxAng];
    for (unsigned i = 0; i < aw * ah; i++) a[i] = v[i];
    r2r_2d(a, ah, aw, true);
    for (unsigned i = 0; i < aw * ah; i++) v[i] = (float)a[i] * nm.cn[i];
}
/* core:1913-1954 (the [pq][st] -> [st][pq] transposition is a layout matter handled by callers) */
void dct4d_inv(float* v, unsigned aw, unsigned ah, const Norms4D& nm) {
    double a[kMaxAng];
    for (unsigned i = 0; i < aw * ah; i++) a[i] = (double)(v[i] * nm.cni[i]);
    r2r_2d(a, ah, aw, false);
    const float coef = 1.0f / (std::sqrt((float)aw) * std::sqrt((float)ah) * 2.0f);
    for (unsigned i = 0; i < aw * ah; i++) v[i] = (float)a[i] * coef;
}

/* Shape bookkeeping shared by the forward and inverse SADCT (core:302-323, :2036-2049, :2102-2104) */
struct Shape {
    unsigned aw, ah;
    std::vector<unsigned> mask, idx, mask_col, idx_col, mask_dct;
    unsigned size;
    void build(const unsigned* m, unsigned aw_, unsigned ah_) {
        aw = aw_; ah = ah_;
        const unsigned A = aw * ah;
        mask.assign(m, m + A);
        idx.assign(A, 0); mask_col.assign(A, 0); idx_col.assign(A, 0); mask_dct.assign(A, 0);
        size = 0;
        for (unsigned s = 0; s < ah; s++) {
            unsigned r = 0;
            for (unsigned t = 0; t < aw; t++)
                if (mask[s * aw + t]) { idx[s * aw + r++] = t; size++; }
            for (unsigned t = 0; t < r; t++) mask_col[s * aw + t] = 1;
        }
        for (unsigned t = 0; t < aw; t++) {
            unsigned r = 0;
            for (unsigned s = 0; s < ah; s++)
                if (mask_col[s * aw + t]) idx_col[(r++) * aw + t] = s;
            for (unsigned s = 0; s < r; s++) mask_dct[s * aw + t] = 1;
        }
    }
};

/* core:1969-2116 on one vector v[st] */
void sadct_fwd(float* v, const Shape& sh) {
    const unsigned aw = sh.aw, ah = sh.ah;
    float cn[kMaxDct], cni[kMaxDct], xf[kMaxDct];
    double x[kMaxDct], y[kMaxDct];
    for (unsigned s = 0; s < ah; s++) { /* rows */
        unsigned n = 0;
        for (unsigned t = 0; t < aw; t++) n += sh.mask[s * aw + t];
        if (n == 1) v[s * aw] = v[s * aw + sh.idx[s * aw]];
        else if (n > 1) {
            dct1d_norms(n, cn, cni);
            for (unsigned t = 0; t < n; t++) x[t] = v[s * aw + sh.idx[s * aw + t]];
            redft10_d(x, 1, y, 1, n);
            for (unsigned t = 0; t < n; t++) v[s * aw + t] = (float)y[t] * cn[t];
        }
    }
    for (unsigned t = 0; t < aw; t++) { /* columns */
        unsigned n = 0;
        for (unsigned s = 0; s < ah; s++) n += sh.mask_col[s * aw + t];
        if (n == 1) v[t] = v[sh.idx_col[t] * aw + t];
        if (n > 1) {
            dct1d_norms(n, cn, cni);
            for (unsigned s = 0; s < n; s++) x[s] = v[sh.idx_col[s * aw + t] * aw + t];
            redft10_d(x, 1, y, 1, n);
            for (unsigned s = 0; s < n; s++) xf[s] = (float)y[s] * cn[s];
            for (unsigned s = 0; s < n; s++) v[s * aw + t] = xf[s];
        }
    }
    const float coef = (float)(0.5 * (float)kSqrt2Inv);
    for (unsigned i = 0; i < aw * ah; i++) v[i] *= (float)sh.mask_dct[i] * coef;
}
/* core:2131-2264 on one vector */
void sadct_inv(float* v, const Shape& sh) {
    const unsigned aw = sh.aw, ah = sh.ah;
    float cn[kMaxDct], cni[kMaxDct], xf[kMaxDct];
    double x[kMaxDct], y[kMaxDct];
    for (unsigned t = 0; t < aw; t++) { /* columns first */
        unsigned n = 0;
        for (unsigned s = 0; s < ah; s++) n += sh.mask_col[s * aw + t];
        const float coef = (float)(2.0 * (float)kSqrt2);
        if (n == 1) v[sh.idx_col[t] * aw + t] = v[t] * coef;
        if (n > 1) {
            dct1d_norms(n, cn, cni);
            for (unsigned s = 0; s < n; s++) x[s] = (double)(v[s * aw + t] * cni[s] * coef);
            redft01_d(x, 1, y, 1, n);
            const float c2 = 0.5f * (float)kSqrt2Inv / std::sqrt((float)n);
            for (unsigned s = 0; s < n; s++) xf[s] = (float)y[s] * c2;
            for (unsigned s = 0; s < n; s++) v[sh.idx_col[s * aw + t] * aw + t] = xf[s];
        }
    }
    for (unsigned s = 0; s < ah; s++) { /* rows */
        unsigned n = 0;
        for (unsigned t = 0; t < aw; t++) n += sh.mask_col[s * aw + t];
        if (n == 1) v[s * aw + sh.idx[s * aw]] = v[s * aw];
        if (n > 1) {
            dct1d_norms(n, cn, cni);
            for (unsigned t = 0; t < n; t++) x[t] = (double)(v[s * aw + t] * cni[t]);
            redft01_d(x, 1, y, 1, n);
            const float c2 = 0.5f * (float)kSqrt2Inv / std::sqrt((float)n);
            for (unsigned t = 0; t < n; t++) xf[t] = (float)y[t] * c2;
            for (unsigned t = 0; t < n; t++) v[s * aw + sh.idx[s * aw + t]] = xf[t];
        }
    }
    for (unsigned i = 0; i < aw * ah; i++) v[i] *= (float)sh.mask[i];
}

/* ------------------------------------------------------------------------------------------ */
/* 5th-dimension filters.  X is one pq slab laid out [c][st][n] (core:363-369).                 */
/* ------------------------------------------------------------------------------------------ */
void fwd_1d(float* v, unsigned n, unsigned tau5, const float* cn5) {
    if (tau5 == ORC_HAAR) { if (n > 1) haar_fwd(v, n); }
    else if (tau5 == ORC_HADAMARD) { if (n > 1) hadamard(v, n); }
    else { /* DCT: core:2550-2559 */
        double x[kMaxDct], y[kMaxDct];
        for (unsigned i = 0; i < n; i++) x[i] = v[i];
        redft10_d(x, 1, y, 1, n);
        for (unsigned i = 0; i < n; i++) v[i] = (float)y[i] * cn5[i];
    }
}
void inv_1d(float* v, unsigned n, unsigned tau5, const float* cni5) {
    if (tau5 == ORC_HAAR) { if (n > 1) haar_inv(v, n); }
    else if (tau5 == ORC_HADAMARD) { if (n > 1) hadamard(v, n); }
    else { /* core:2578-2593 */
        double x[kMaxDct], y[kMaxDct];
        for (unsigned i = 0; i < n; i++) x[i] = (double)(v[i] * cni5[i]);
        redft01_d(x, 1, y, 1, n);
        const float coef = 0.5f * (float)kSqrt2Inv / std::sqrt((float)n);
        for (unsigned i = 0; i < n; i++) v[i] = (float)y[i] * coef;
    }
}

/* core:2281-2690 (plain and masked variants) */
void ht_filter_slab(float* X, unsigned nSx, unsigned A, unsigned C, const float* sigma,
                    float lambda, float* weight, const unsigned* mask_dct, unsigned tau5) {
    float cn5[kMaxDct], cni5[kMaxDct];
    if (tau5 == ORC_DCT) dct1d_norms(nSx, cn5, cni5);
    for (unsigned v = 0; v < A * C; v++) fwd_1d(X + v * nSx, nSx, tau5, cn5);
    for (unsigned c = 0; c < C; c++) {
        float T;
        if (tau5 == ORC_HAAR) T = lambda * sigma[c] * (float)kSqrt2;
        else if (tau5 == ORC_HADAMARD) T = lambda * sigma[c] * std::sqrt((float)nSx) * (float)kSqrt2;
        else T = lambda * sigma[c] * 2.0f * (float)kSqrt2;
        float* Xc = X + c * nSx * A;
        for (unsigned st = 0; st < A; st++) {
            if (mask_dct && !mask_dct[st]) continue;
            for (unsigned n = 0; n < nSx; n++) {
                if (std::fabs(Xc[st * nSx + n]) > T) weight[c]++;
                else Xc[st * nSx + n] = 0.0f;
            }
        }
    }
    for (unsigned v = 0; v < A * C; v++) inv_1d(X + v * nSx, nSx, tau5, cni5);
    if (tau5 == ORC_HADAMARD && nSx > 1) {
        const float coef = 1.0f / (float)nSx;
        for (unsigned i = 0; i < A * C * nSx; i++) X[i] *= coef;
    }
}
/* core:2706-3123 */
void wiener_filter_slab(float* Xo, float* Xe, unsigned nSx, unsigned A, unsigned C,
                        const float* sigma, float* weight, const unsigned* mask_dct, unsigned tau5) {
    float cn5[kMaxDct], cni5[kMaxDct];
    if (tau5 == ORC_DCT) dct1d_norms(nSx, cn5, cni5);
    for (unsigned v = 0; v < A * C; v++) {
        fwd_1d(Xo + v * nSx, nSx, tau5, cn5);
        fwd_1d(Xe + v * nSx, nSx, tau5, cn5);
    }
    const float hcoef = 1.0f / (float)nSx;
    for (unsigned c = 0; c < C; c++) {
        float* o = Xo + c * nSx * A;
        float* e = Xe + c * nSx * A;
        for (unsigned st = 0; st < A; st++) {
            if (mask_dct && !mask_dct[st]) continue;
            for (unsigned n = 0; n < nSx; n++) {
                const unsigned i = st * nSx + n;
                if (tau5 == ORC_HADAMARD) {
                    float value = e[i] * e[i] * hcoef;
                    value /= (value + sigma[c] * sigma[c]);
                    e[i] = o[i] * value * hcoef;
                    weight[c] += value;
                } else {
                    float value = e[i] * e[i];
                    value /= (value + sigma[c] * sigma[c]);
                    e[i] = o[i] * value;
                    weight[c] += value;
                }
            }
        }
    }
    for (unsigned v = 0; v < A * C; v++) inv_1d(Xe + v * nSx, nSx, tau5, cni5);
}

/* ------------------------------------------------------------------------------------------ */
/* Host helpers on the path                                                                     */
/* ------------------------------------------------------------------------------------------ */
/* utilities.cpp:633-684 */
int sigma_table(float sigma, unsigned C, unsigned cs, float* out) {
    if (C == 1) { out[0] = sigma; return 0; }
    if (cs == ORC_YUV) {
        out[0] = std::sqrt(0.299f * 0.299f + 0.587f * 0.587f + 0.114f * 0.114f) * sigma;
        out[1] = std::sqrt(0.14713f * 0.14713f + 0.28886f * 0.28886f + 0.436f * 0.436f) * sigma;
        out[2] = std::sqrt(0.615f * 0.615f + 0.51498f * 0.51498f + 0.10001f * 0.10001f) * sigma;
    } else if (cs == ORC_YCBCR) {
        out[0] = std::sqrt(0.299f * 0.299f + 0.587f * 0.587f + 0.114f * 0.114f) * sigma;
        out[1] = std::sqrt(0.169f * 0.169f + 0.331f * 0.331f + 0.500f * 0.500f) * sigma;
        out[2] = std::sqrt(0.500f * 0.500f + 0.419f * 0.419f + 0.081f * 0.081f) * sigma;
    } else if (cs == ORC_OPP) {
        out[0] = std::sqrt(0.333f * 0.333f + 0.333f * 0.333f + 0.333f * 0.333f) * sigma;
        out[1] = std::sqrt(0.5f * 0.5f + 0.0f * 0.0f + 0.5f * 0.5f) * sigma;
        out[2] = std::sqrt(0.25f * 0.25f + 0.5f * 0.5f + 0.25f * 0.25f) * sigma;
    } else if (cs == ORC_RGB) {
        out[0] = out[1] = out[2] = sigma;
    } else return 1;
    return 0;
}

/* utilities.cpp:697-712 */
void ind_init(std::vector<unsigned>& v, unsigned max_size, unsigned N, unsigned step) {
    v.clear();
    unsigned ind = N;
    while (ind < max_size - N) { v.push_back(ind); ind += step; }
    if (v.back() < max_size - N - 1) v.push_back(max_size - N - 1);
}

/* utilities_LF.cpp:1000-1016 on channel 0 of den */
bool patch_denoised(const float* den0, unsigned p_idx, unsigned W, unsigned k) {
    for (unsigned p = 0; p < k; p++)
        for (unsigned q = 0; q < k; q++)
            if (den0[p_idx + p * W + q] == 0.0f) return false;
    return true;
}
/* utilities_LF.cpp:1031-1099 */
void ind_init_subset(std::vector<unsigned>& rows, std::vector<std::vector<unsigned> >& cols,
                     unsigned max_h, unsigned max_w, unsigned W, unsigned N, unsigned step,
                     unsigned k, const float* den0) {
    rows.clear(); cols.clear();
    std::vector<unsigned> tmp;
    auto scan_row = [&](unsigned i) {
        tmp.clear();
        for (unsigned j = N; j < max_w - N; j += step)
            if (!patch_denoised(den0, i * W + j, W, k)) tmp.push_back(j);
        const bool border = tmp.empty() ? true : (tmp.back() < max_w - N - 1);
        if (border && !patch_denoised(den0, i * W + max_w - N - 1, W, k)) tmp.push_back(max_w - N - 1);
        if (!tmp.empty()) { rows.push_back(i); cols.push_back(tmp); }
    };
    for (unsigned i = N; i < max_h - N; i += step) scan_row(i);
    const bool row_border = rows.empty() ? true : (rows.back() < max_h - N - 1);
    if (row_border) scan_row(max_h - N - 1);
}

unsigned pow2_floor(unsigned n) { unsigned r = 1; while (r * 2 <= n) r *= 2; return r; } /* utilities.cpp:608-616 */

/* ------------------------------------------------------------------------------------------ */
/* Block matching                                                                               */
/* ------------------------------------------------------------------------------------------ */
/* Integral-image recurrence of core:3342-3389 / :3527-3573 for one displacement.
 * diff: W*H + k*W + k scratch whose entries outside [b, dim-b) must be (and stay) zero.  The k rows of slack
 *       matter when k - 1 > b: the recurrence then reads rows past the image for table rows no caller uses
 *       (the reference reads past its W*H vector there, core:3384); here those reads see zeros;
 * sum : W*H table, entries outside the written region keep whatever the caller put there.
 * Written region: rows/cols [b, dim - b - trim).                                            */
void integral_table(const float* img1, const float* img2, int dk, unsigned W, unsigned H,
                    unsigned k, unsigned b, unsigned trim, float* diff, float* sum) {
    for (unsigned i = b; i < H - b; i++) {
        unsigned q = i * W + b;
        for (unsigned j = b; j < W - b; j++, q++) {
            const float d = img2[(int)q + dk] - img1[q];
            diff[q] = d * d;
        }
    }
    const unsigned dn = b * W + b;
    float value = 0.0f;
    for (unsigned p = 0; p < k; p++) {
        unsigned pq = p * W + dn;
        for (unsigned q = 0; q < k; q++, pq++) value += diff[pq];
    }
    sum[dn] = value;
    for (unsigned j = b + 1; j < W - b - trim; j++) { /* first row */
        const unsigned ind = b * W + j - 1;
        float s = sum[ind];
        for (unsigned p = 0; p < k; p++) s += diff[ind + p * W + k] - diff[ind + p * W];
        sum[ind + 1] = s;
    }
    for (unsigned i = b + 1; i < H - b - trim; i++) {
        const unsigned ind = (i - 1) * W + b;
        float s = sum[ind];
        for (unsigned q = 0; q < k; q++) s += diff[ind + k * W + q] - diff[ind + q]; /* first column */
        sum[ind + W] = s;
        unsigned kk = i * W + b + 1;
        unsigned pq = (i + k - 1) * W + k - 1 + b + 1;
        for (unsigned j = b + 1; j < W - b - trim; j++, kk++, pq++) {
            sum[kk] = sum[kk - 1] + sum[kk - W] - sum[kk - 1 - W]
                    + diff[pq] - diff[pq - k] - diff[pq - k * W] + diff[pq - k - k * W];
        }
    }
}

struct Cand { float d; unsigned idx; unsigned order; };

/* core:3301-3461 and :3631-3788 (identical arithmetic; the second only changes the ref list) */
int bm_self(const float* img, unsigned W, unsigned H, unsigned k, unsigned N, unsigned nHW,
            unsigned nSim, float tauMatch, const unsigned* refs, unsigned n_refs,
            unsigned* out_idx, unsigned* out_cnt) {
    if (N <= 1) { /* core:3448-3460 */
        for (unsigned r = 0; r < n_refs; r++) { out_idx[r * (N ? N : 1)] = refs[r]; out_cnt[r] = 1; }
        return 0;
    }
    const unsigned Ns = 2 * nSim + 1;
    const unsigned nT = (nSim + 1) * Ns;
    const float threshold = tauMatch * k * k;
    const size_t WH = (size_t)W * H;
    /* all candidate tables, as the reference keeps them (value 2*threshold where never written) */
    std::vector<float> tables;
    try { tables.assign(nT * WH, 2 * threshold); } catch (...) { return 1; }
    #pragma omp parallel num_threads(g_threads > 0 ? g_threads : omp_get_max_threads())
    {
        std::vector<float> diff(WH + (size_t)k * W + k, 0.0f);
        #pragma omp for schedule(dynamic, 1)
        for (int ddk = 0; ddk < (int)nT; ddk++) {
            const unsigned di = ddk / Ns, dj = ddk % Ns;
            const int dk = (int)(di * W + dj) - (int)nSim;
            integral_table(img, img, dk, W, H, k, nHW, 0, diff.data(), tables.data() + (size_t)ddk * WH);
        }
    }
    #pragma omp parallel num_threads(g_threads > 0 ? g_threads : omp_get_max_threads())
    {
        std::vector<Cand> cand;
        cand.reserve(Ns * Ns);
        #pragma omp for schedule(dynamic, 16)
        for (int r = 0; r < (int)n_refs; r++) {
            const unsigned k_r = refs[r];
            cand.clear();
            unsigned order = 0;
            for (int dj = -(int)nSim; dj <= (int)nSim; dj++) { /* scan order of core:3407-3420 */
                for (int di = 0; di <= (int)nSim; di++) {
                    const float v = tables[(size_t)(dj + (int)nSim + di * (int)Ns) * WH + k_r];
                    if (v < threshold) cand.push_back({v, (unsigned)((int)k_r + di * (int)W + dj), order});
                    order++;
                }
                for (int di = -(int)nSim; di < 0; di++) {
                    const size_t t = (size_t)(-dj + (int)nSim + (-di) * (int)Ns) * WH;
                    if (tables[t + k_r] < threshold) /* tested at k_r with the mirrored table ... */
                        cand.push_back({tables[t + (size_t)((int)k_r + di * (int)W + dj)], /* ... scored at the candidate */
                                        (unsigned)((int)k_r + di * (int)W + dj), order});
                    order++;
                }
            }
            const unsigned nSx = N > cand.size() ? pow2_floor((unsigned)cand.size()) : N;
            if (nSx == 1 && cand.empty()) cand.push_back({0.0f, k_r, 0});
            /* std::partial_sort in the reference; tie order unspecified there -> scan order here */
            std::stable_sort(cand.begin(), cand.end(), [](const Cand& a, const Cand& b) { return a.d < b.d; });
            unsigned cnt = 0;
            for (unsigned n = 0; n < nSx; n++) out_idx[(size_t)r * N + cnt++] = cand[n].idx;
            if (nSx == 1) out_idx[(size_t)r * N + cnt++] = cand[0].idx; /* duplicate rule core:3443-3444 */
            out_cnt[r] = cnt;
        }
    }
    return 0;
}

/* core:3479-3611 (dense) == :3806-3945 (on demand): argmin over the (2 nDisp+1)^2 displacements */
int bm_stereo(const float* img1, const float* img2, unsigned W, unsigned H, unsigned k,
              unsigned nDisp, float tauMatch, unsigned* best, unsigned char* shape) {
    const unsigned Ns = 2 * nDisp + 1;
    const float threshold = tauMatch * k * k;
    const size_t WH = (size_t)W * H;
    std::vector<float> diff(WH + (size_t)k * W + k, 0.0f), sum(WH, 0.0f), bestd(WH, 0.0f);
    std::vector<unsigned> bestorder(WH, 0xffffffffu);
    const unsigned lo = nDisp, hi_r = H - nDisp - k + 1, hi_c = W - nDisp - k + 1;
    for (unsigned di = 0; di < Ns; di++)
        for (unsigned dj = 0; dj < Ns; dj++) {
            const int dk = (int)(di * W + dj) - (int)(nDisp * (1 + W));
            integral_table(img1, img2, dk, W, H, k, nDisp, k - 1, diff.data(), sum.data());
            const unsigned order = dj * Ns + di; /* candidate scan order: dj outer, di inner (core:3591-3596) */
            for (unsigned i = lo; i < hi_r; i++)
                for (unsigned j = lo; j < hi_c; j++) {
                    const size_t q = (size_t)i * W + j;
                    const float v = sum[q];
                    if (bestorder[q] == 0xffffffffu || v < bestd[q] || (v == bestd[q] && order < bestorder[q])) {
                        bestd[q] = v; bestorder[q] = order;
                        best[q] = (unsigned)((int)q + ((int)di - (int)nDisp) * (int)W + ((int)dj - (int)nDisp));
                    }
                }
        }
    for (unsigned i = lo; i < hi_r; i++)
        for (unsigned j = lo; j < hi_c; j++) {
            const size_t q = (size_t)i * W + j;
            shape[q] = bestd[q] < threshold ? 1 : 0;
        }
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* One core pass                                                                                */
/* ------------------------------------------------------------------------------------------ */
std::vector<float> g_last_weights;   /* aggregation weights of the last pass's groups (orc_last_weights) */

struct GroupOut {
    unsigned nSx = 0;
    bool use_sadct = false;
    std::vector<float> patches; /* [st][c][n][k*k], pixel domain */
    float w[4] = {0, 0, 0, 0};
};

struct PassCtx {
    int step;
    const orc_params* P;
    unsigned aw, ah, A, Wb, Hb, C, k, k2, N, nHW;
    float lambda;
    float sigma[4];
    const float* noisy; const float* basic;
    const unsigned* mask; unsigned pst;
    bool centre; /* pst == cst: the row-band tables of the centre path leave column Wb-k unfilled */
    Norms2D n2; Norms4D n4;
    const unsigned* self_idx; const unsigned* self_cnt;       /* per ref */
    std::vector<std::vector<unsigned> >* best;                /* [st][Wb*Hb] */
    std::vector<std::vector<unsigned char> >* shape;          /* [st][Wb*Hb] */
};

void transform_patch_2d(const PassCtx& cx, const float* img_c, unsigned pos, float* out) {
    const unsigned k = cx.k, W = cx.Wb;
    const unsigned t2 = cx.P->tau_2D;
    if (t2 == ORC_DCT) dct2d_fwd(img_c + pos, W, out, k, cx.n2);
    else if (t2 == ORC_BIOR) bior_fwd(img_c + pos, W, out, k);
    else
        for (unsigned p = 0; p < k; p++)
            for (unsigned q = 0; q < k; q++) out[p * k + q] = img_c[pos + p * W + q];
}

/* core:277-481 (HT) / :1054-1282 (Wiener) for one reference patch */
void process_group(const PassCtx& cx, unsigned ref_slot, unsigned k_r, GroupOut& go) {
    const unsigned A = cx.A, C = cx.C, k2 = cx.k2, N = cx.N;
    const size_t plane = (size_t)cx.Wb * cx.Hb;
    const unsigned nSx = cx.self_cnt[ref_slot];
    const unsigned tau4 = cx.P->tau_4D, tau5 = cx.P->tau_5D;
    go.nSx = nSx;
    const int S = cx.step == 2 ? 2 : 1;

    /* positions of every patch of the group: pos[n][st] */
    std::vector<unsigned> pos(nSx * A, 0);
    for (unsigned n = 0; n < nSx; n++) {
        const unsigned ind_pst = cx.self_idx[(size_t)ref_slot * N + n];
        for (unsigned st = 0; st < A; st++)
            if (cx.mask[st]) pos[n * A + st] = (st == cx.pst) ? ind_pst : (*cx.best)[st][ind_pst];
    }
    /* gather + 2-D transform: T[s][n][c][pq][st] (zeros for empty SAIs, core:286-299) */
    std::vector<float> T((size_t)S * nSx * C * k2 * A, 0.0f);
    std::vector<float> tmp(k2);
    for (int s = 0; s < S; s++) {
        const float* src = s == 0 ? cx.noisy : cx.basic;
        for (unsigned n = 0; n < nSx; n++)
            for (unsigned st = 0; st < A; st++) {
                if (!cx.mask[st]) continue;
                /* Centre path: the reference gathers from per-row tables filled only for columns
                 * j < Wb - k (core:1697, bm3d.cpp:737/:857): a patch at column Wb - k reads the table's
                 * zero initialisation.  Reachable when self and disparity offsets both max out.  The
                 * subset path fills a (2nHW+1)^2 block around each reference patch (core:1733-1738,
                 * :1782-1803, :1842-1848), which includes that column. */
                if (cx.centre && pos[n * A + st] % cx.Wb >= cx.Wb - cx.k) continue;
                for (unsigned c = 0; c < C; c++) {
                    transform_patch_2d(cx, src + ((size_t)st * C + c) * plane, pos[n * A + st], tmp.data());
                    float* dst = &T[(((size_t)s * nSx + n) * C + c) * k2 * A];
                    for (unsigned pq = 0; pq < k2; pq++) dst[pq * A + st] = tmp[pq];
                }
            }
    }
    /* SADCT shape (core:302-323) */
    Shape sh;
    bool use_sadct = false;
    if (tau4 == ORC_SADCT) {
        std::vector<unsigned> m(A);
        for (unsigned st = 0; st < A; st++)
            m[st] = (st == cx.pst) ? 1u : (cx.mask[st] ? (unsigned)(*cx.shape)[st][k_r] : 0u);
        sh.build(m.data(), cx.aw, cx.ah);
        use_sadct = sh.size != A;
    }
    go.use_sadct = use_sadct;
    /* 4-D forward (core:353-360) */
    const bool do_dct4 = (tau4 == ORC_DCT) || (tau4 == ORC_SADCT && !use_sadct);
    const bool do_sa4 = !do_dct4 && (tau4 == ORC_SADCT || use_sadct);
    for (size_t v = 0; v < (size_t)S * nSx * C * k2; v++) {
        if (do_dct4) dct4d_fwd(&T[v * A], cx.aw, cx.ah, cx.n4);
        else if (do_sa4) sadct_fwd(&T[v * A], sh);
    }
    /* 5-D slabs X[s][pq][c][st][n] (core:362-369), filtering (core:371-410), weights (core:412-421) */
    std::vector<float> X((size_t)S * k2 * C * A * nSx);
    for (int s = 0; s < S; s++)
        for (unsigned n = 0; n < nSx; n++)
            for (unsigned c = 0; c < C; c++)
                for (unsigned pq = 0; pq < k2; pq++)
                    for (unsigned st = 0; st < A; st++)
                        X[((((size_t)s * k2 + pq) * C + c) * A + st) * nSx + n] =
                            T[((((size_t)s * nSx + n) * C + c) * k2 + pq) * A + st];
    float weight[4] = {0, 0, 0, 0};
    const unsigned* md = use_sadct ? sh.mask_dct.data() : nullptr;
    const size_t slab = (size_t)C * A * nSx;
    for (unsigned pq = 0; pq < k2; pq++) {
        if (cx.step == 1) ht_filter_slab(&X[pq * slab], nSx, A, C, cx.sigma, cx.lambda, weight, md, tau5);
        else wiener_filter_slab(&X[pq * slab], &X[((size_t)k2 + pq) * slab], nSx, A, C, cx.sigma, weight, md, tau5);
    }
    float* F = cx.step == 1 ? X.data() : X.data() + (size_t)k2 * slab; /* filtered stack */
    if (cx.P->useSD) { /* core:3140-3173 */
        const unsigned Nn = nSx * A;
        for (unsigned c = 0; c < C; c++) {
            float mean = 0.0f, sd = 0.0f;
            for (unsigned pq = 0; pq < k2; pq++)
                for (unsigned i = 0; i < Nn; i++) {
                    const float x = F[pq * slab + c * Nn + i];
                    mean += x; sd += x * x;
                }
            const float res = (sd - mean * mean / (float)Nn) / (float)(Nn - 1);
            go.w[c] = res > 0.0f ? 1.0f / std::sqrt(res) : 0.0f;
        }
    } else {
        for (unsigned c = 0; c < C; c++)
            go.w[c] = weight[c] > 0.0f
                          ? (cx.sigma[c] > 0.0 ? 1.0f / (float)(cx.sigma[c] * cx.sigma[c] * weight[c])
                                               : 1.0f / (float)(weight[c]))
                          : 1.0f;
    }
    /* back to per-patch vectors, inverse 4-D (core:423-451), inverse 2-D (core:488-493) */
    go.patches.assign((size_t)A * C * nSx * k2, 0.0f);
    std::vector<float> v(A);
    std::vector<float> G((size_t)nSx * C * k2 * A);
    for (unsigned n = 0; n < nSx; n++)
        for (unsigned c = 0; c < C; c++)
            for (unsigned pq = 0; pq < k2; pq++) {
                for (unsigned st = 0; st < A; st++) v[st] = F[((pq * C + c) * A + st) * nSx + n];
                if (do_dct4) dct4d_inv(v.data(), cx.aw, cx.ah, cx.n4);
                else if (do_sa4) sadct_inv(v.data(), sh);
                for (unsigned st = 0; st < A; st++)
                    go.patches[(((size_t)st * C + c) * nSx + n) * k2 + pq] = v[st];
            }
    const unsigned t2 = cx.P->tau_2D;
    if (t2 == ORC_DCT || t2 == ORC_BIOR)
        for (unsigned st = 0; st < A; st++) {
            if (!cx.mask[st]) continue;
            for (unsigned c = 0; c < C; c++)
                for (unsigned n = 0; n < nSx; n++) {
                    float* p = &go.patches[(((size_t)st * C + c) * nSx + n) * k2];
                    if (t2 == ORC_DCT) dct2d_inv(p, cx.k, cx.n2);
                    else bior_inv(p, cx.k);
                }
        }
}

int pass_impl(int step, const orc_params* P, unsigned aw, unsigned ah, unsigned Wb, unsigned Hb,
              unsigned C, const float* noisy, const float* basic, float* num, float* den,
              const unsigned* mask, const unsigned* procSAI, unsigned cst, unsigned pst,
              int rb, int re, orc_stats* stats) {
    const double t0 = now_s();
    if (C > 3 || (step == 2 && !basic)) return 1;
    PassCtx cx;
    cx.step = step; cx.P = P; cx.aw = aw; cx.ah = ah; cx.A = aw * ah; cx.Wb = Wb; cx.Hb = Hb; cx.C = C;
    cx.k = P->k; cx.k2 = P->k * P->k; cx.N = P->N ? P->N : 1; cx.nHW = P->nSim + P->nDisp;
    cx.noisy = noisy; cx.basic = basic; cx.mask = mask; cx.pst = pst; cx.centre = pst == cst;
    const unsigned A = cx.A, k = cx.k, nHW = cx.nHW;
    const size_t plane = (size_t)Wb * Hb;
    if (sigma_table(P->sigma, C, P->color_space, cx.sigma)) return 1;
    /* core:146 / :915 */
    const float tauMatch = (C == 1 ? 3.f : 1.f) * (cx.sigma[0] < 35.0f ? (step == 1 ? 3000 : 2000) : 5000);
    cx.lambda = P->lambda;
    if (step == 1 && P->tau_2D == ORC_ID && P->tau_4D == ORC_DCT) cx.lambda /= (float)kSqrt2; /* core:206-207 */
    dct2d_norms(k, cx.n2.cn, cx.n2.cni); cx.n2.k = k;
    dct4d_norms(aw, ah, cx.n4.cn, cx.n4.cni); cx.n4.aw = aw; cx.n4.ah = ah;

    /* reference patch grid (core:149-165) */
    std::vector<unsigned> rows;
    std::vector<std::vector<unsigned> > cols;
    if (pst == cst) {
        std::vector<unsigned> call;
        ind_init(rows, Hb - k + 1, nHW, P->p);
        ind_init(call, Wb - k + 1, nHW, P->p);
        cols.assign(rows.size(), call);
    } else {
        ind_init_subset(rows, cols, Hb - k + 1, Wb - k + 1, Wb, nHW, P->p, k, den + (size_t)pst * C * plane);
    }
    if (rows.empty()) return 0;

    /* current estimate for matching, channel 0 only is ever read (core:167-170) */
    const float* sub = step == 1 ? noisy : basic;
    std::vector<std::vector<float> > est(A);
    for (unsigned st = 0; st < A; st++) {
        if (!mask[st]) continue;
        est[st].resize(plane);
        const size_t o = (size_t)st * C * plane;
        for (size_t i = 0; i < plane; i++) est[st][i] = den[o + i] ? num[o + i] / den[o + i] : sub[o + i];
    }

    /* block matching (core:209-236) */
    const double tb0 = now_s();
    std::vector<unsigned> refs;
    std::vector<unsigned> row_start(rows.size() + 1, 0);
    for (size_t r = 0; r < rows.size(); r++) {
        row_start[r] = (unsigned)refs.size();
        for (unsigned j : cols[r]) refs.push_back(rows[r] * Wb + j);
    }
    row_start[rows.size()] = (unsigned)refs.size();
    std::vector<unsigned> self_idx(refs.size() * cx.N), self_cnt(refs.size());
    if (bm_self(est[pst].data(), Wb, Hb, k, P->N, nHW, P->nSim, tauMatch, refs.data(), (unsigned)refs.size(),
                self_idx.data(), self_cnt.data())) return 1;
    std::vector<std::vector<unsigned> > best(A);
    std::vector<std::vector<unsigned char> > shape(A);
    #pragma omp parallel for schedule(dynamic, 1) num_threads(g_threads > 0 ? g_threads : omp_get_max_threads())
    for (int st = 0; st < (int)A; st++) {
        if ((unsigned)st == pst || !mask[st]) continue;
        best[st].assign(plane, 0);
        shape[st].assign(plane, 0);
        bm_stereo(est[pst].data(), est[st].data(), Wb, Hb, k, P->nDisp, tauMatch, best[st].data(), shape[st].data());
    }
    const double tb1 = now_s();
    cx.self_idx = self_idx.data(); cx.self_cnt = self_cnt.data(); cx.best = &best; cx.shape = &shape;

    std::vector<float> kaiser;
    kaiser_window(k, kaiser);
    const unsigned k2 = cx.k2;
    const int r0 = rb < 0 ? 0 : rb;
    const int r1 = (re < 0 || re > (int)rows.size()) ? (int)rows.size() : re;
    unsigned long long n_groups = 0, n_sadct = 0, n_stack = 0;
    for (int r = r0; r < r1; r++) {
        const unsigned ncols = (unsigned)cols[r].size();
        std::vector<GroupOut> outs(ncols);
        #pragma omp parallel for schedule(dynamic, 1) num_threads(g_threads > 0 ? g_threads : omp_get_max_threads())
        for (int jj = 0; jj < (int)ncols; jj++)
            process_group(cx, row_start[r] + jj, rows[r] * Wb + cols[r][jj], outs[jj]);
        /* aggregation in the reference's order: st, then group, then c, n, p, q (core:484-528) */
        for (unsigned st = 0; st < A; st++) {
            if (procSAI[st] || !mask[st]) continue;   /* (the schedule marks empty SAIs processed, bm5d.cpp:268-270) */
            float* num_st = num + (size_t)st * C * plane;
            float* den_st = den + (size_t)st * C * plane;
            for (unsigned jj = 0; jj < ncols; jj++) {
                const unsigned slot = row_start[r] + jj;
                const unsigned k_r = rows[r] * Wb + cols[r][jj];
                const GroupOut& go = outs[jj];
                if (!(P->tau_4D != ORC_SADCT || st == pst || shape[st][k_r])) continue;
                for (unsigned c = 0; c < C; c++)
                    for (unsigned n = 0; n < go.nSx; n++) {
                        const unsigned ind_pst = self_idx[(size_t)slot * cx.N + n];
                        const size_t ind_st = (size_t)((st == pst) ? ind_pst : best[st][ind_pst]) + c * plane;
                        const float* patch = &go.patches[(((size_t)st * C + c) * go.nSx + n) * k2];
                        for (unsigned p = 0; p < k; p++)
                            for (unsigned q = 0; q < k; q++) {
                                const size_t ind = ind_st + p * Wb + q;
                                num_st[ind] += kaiser[p * k + q] * go.w[c] * patch[p * k + q];
                                den_st[ind] += kaiser[p * k + q] * go.w[c];
                            }
                    }
            }
        }
        for (unsigned jj = 0; jj < ncols; jj++) {   /* inspection: the groups' aggregation weights, [reference patch][channel] */
            const size_t slot = row_start[r] + jj;
            if (g_last_weights.size() < (slot + 1) * C) g_last_weights.resize((slot + 1) * C, 0.0f);
            for (unsigned c = 0; c < C; c++) g_last_weights[slot * C + c] = outs[jj].w[c];
        }
        for (const GroupOut& go : outs) { n_groups++; n_sadct += go.use_sadct; n_stack += go.nSx; }
    }
    if (stats) {
        stats->groups += n_groups; stats->sadct_groups += n_sadct; stats->stack_patches += n_stack;
        stats->passes += 1;
        stats->bm_seconds += tb1 - tb0;
        stats->total_seconds += now_s() - t0;
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* Whole steps                                                                                  */
/* ------------------------------------------------------------------------------------------ */
/* utilities.cpp:215-263 */
void symetrize(const float* img, float* out, unsigned W, unsigned H, unsigned C, unsigned N) {
    const unsigned w = W + 2 * N, h = H + 2 * N;
    for (unsigned c = 0; c < C; c++) {
        const float* src = img + (size_t)c * W * H;
        float* dst = out + (size_t)c * w * h;
        for (unsigned i = 0; i < H; i++)
            for (unsigned j = 0; j < W; j++) dst[(i + N) * w + j + N] = src[i * W + j];
        for (unsigned j = 0; j < w; j++)
            for (unsigned i = 0; i < N; i++) {
                dst[i * w + j] = dst[(2 * N - i - 1) * w + j];
                dst[(h - i - 1) * w + j] = dst[(h - 2 * N + i) * w + j];
            }
        for (unsigned i = 0; i < h; i++)
            for (unsigned j = 0; j < N; j++) {
                dst[i * w + j] = dst[i * w + 2 * N - j - 1];
                dst[i * w + w - j - 1] = dst[i * w + w - 2 * N + j];
            }
    }
}
/* utilities.cpp:276-298 */
void unsymetrize(float* img, const float* sym, unsigned W, unsigned H, unsigned C, unsigned N) {
    const unsigned w = W + 2 * N, h = H + 2 * N;
    for (unsigned c = 0; c < C; c++)
        for (unsigned i = 0; i < H; i++)
            for (unsigned j = 0; j < W; j++)
                img[(size_t)c * W * H + i * W + j] = sym[(size_t)c * w * h + (i + N) * w + j + N];
}

/* utilities.cpp:482-599 */
int color_transform(float* img, unsigned cs, unsigned W, unsigned H, unsigned C, bool fwd) {
    if (C == 1 || cs == ORC_RGB) return 0;
    const size_t n = (size_t)W * H;
    float* r = img; float* g = img + n; float* b = img + 2 * n;
    for (size_t i = 0; i < n; i++) {
        const float R = r[i], G = g[i], B = b[i];
        float x, y, z;
        if (cs == ORC_YUV) {
            if (fwd) { x = 0.299f * R + 0.587f * G + 0.114f * B; y = -0.14713f * R - 0.28886f * G + 0.436f * B; z = 0.615f * R - 0.51498f * G - 0.10001f * B; }
            else     { x = R + 1.13983f * B; y = R - 0.39465f * G - 0.5806f * B; z = R + 2.03211f * G; }
        } else if (cs == ORC_YCBCR) {
            if (fwd) { x = 0.299f * R + 0.587f * G + 0.114f * B; y = -0.169f * R - 0.331f * G + 0.500f * B; z = 0.500f * R - 0.419f * G - 0.081f * B; }
            else     { x = 1.000f * R + 0.000f * G + 1.402f * B; y = 1.000f * R - 0.344f * G - 0.714f * B; z = 1.000f * R + 1.772f * G + 0.000f * B; }
        } else if (cs == ORC_OPP) {
            if (fwd) { x = 0.333f * R + 0.333f * G + 0.333f * B; y = 0.500f * R + 0.000f * G - 0.500f * B; z = 0.250f * R - 0.500f * G + 0.250f * B; }
            else     { x = 1.0f * R + 1.0f * G + 0.666f * B; y = 1.0f * R + 0.0f * G - 1.333f * B; z = 1.0f * R - 1.0f * G + 0.666f * B; }
        } else return 1;
        r[i] = x; g[i] = y; b[i] = z;
    }
    return 0;
}

/* utilities_LF.cpp:881-901 */
void search_window(int aidx, unsigned asize, unsigned an, int& c, int& mn, int& mx) {
    mn = aidx - (int)an;
    mx = aidx + (int)an;
    int shift = mn < 0 ? -mn : 0;
    mn += shift; mx += shift;
    c = (int)an - shift;
    shift = mx >= (int)asize ? ((int)asize - mx - 1) : 0;
    mn += shift; mx += shift;
    c -= shift;
}

/* utilities_LF.cpp:967-995 (counts (i,j,c) triples, divides without C: reproduce) */
float denoised_percent(const float* den, const unsigned* mask, unsigned A, unsigned W, unsigned H,
                       unsigned C, unsigned N, unsigned k) {
    const unsigned wb = W + 2 * N, hb = H + 2 * N;
    const size_t plane = (size_t)wb * hb;
    float cnt = 0.0f;
    unsigned n_mask = 0;
    for (unsigned st = 0; st < A; st++) {
        if (!mask[st]) continue;
        n_mask++;
        const float* d = den + (size_t)st * C * plane;
        for (unsigned i = 0; i < H - k + 1; i++)
            for (unsigned j = 0; j < W - k + 1; j++)
                for (unsigned c = 0; c < C; c++)
                    if (d[(size_t)N * wb + N + (size_t)i * wb + j + c * plane] > 0.0) cnt++;
    }
    return cnt * 100.0f / (float)n_mask / (float)(H - k + 1) / (float)(W - k + 1);
}

/* bm5d.cpp:165-407 (step 1) and :861-1106 (step 2), nb_threads == 1 */
/* One pass over a window in the reference's OpenMP tile mode (bm5d.cpp:411-708): every SAI of the (mirror-padded) window
 * is cut into nb_tiles sub-images with a halo of N = nSim + nDisp pixels (sub_divide, utilities.cpp:312-395: the image
 * is halved along its longer side until there are nb_tiles pieces; the last row / column of tiles takes the remainder),
 * each tile runs the core pass on its own, only the tiles' interiors are kept (undivide_LF, utilities_LF.cpp:438-515:
 * whatever a tile aggregated into its halo is DISCARDED, which is what costs the tiled mode about 0.5 dB) and the window's
 * num / den are padded again for the next pass.  w_* are the padded window buffers [Aw][C*hb*wb], updated in place.
 * pct receives the sum of the tiles' LF_denoised_percent (bm5d.cpp:666-668). */
/* exact zeros of one padded SAI [C][H + 2N][W + 2N], counted tile by tile over the tiles sub_divide would cut (halo of N) */
long tiled_zero_count(const float* den_b, unsigned W, unsigned H, unsigned C, unsigned N, int nb_tiles) {
    const unsigned hb = H + 2 * N, wb = W + 2 * N;
    unsigned w_small = W, h_small = H, nw = 1, nh = 1;
    for (int n = nb_tiles; n > 1; n /= 2) {
        if (w_small > h_small) { w_small = (unsigned)std::floor((float)w_small * 0.5f); nw *= 2; }
        else { h_small = (unsigned)std::floor((float)h_small * 0.5f); nh *= 2; }
    }
    const unsigned h_bound = nh > 1 ? H - (nh - 1) * h_small : h_small;
    const unsigned w_bound = nw > 1 ? W - (nw - 1) * w_small : w_small;
    long cnt = 0;
    for (unsigned i = 0; i < nh; i++)
        for (unsigned j = 0; j < nw; j++) {
            const unsigned h = (i == nh - 1 ? h_bound : h_small) + 2 * N, w = (j == nw - 1 ? w_bound : w_small) + 2 * N;
            for (unsigned c = 0; c < C; c++)
                for (unsigned y = 0; y < h; y++) {
                    const float* row = den_b + ((size_t)c * hb + i * h_small + y) * wb + j * w_small;
                    cnt += (long)std::count(row, row + w, 0.0f);
                }
        }
    return cnt;
}

int tiled_pass(int step, const orc_params* Pw, unsigned asw, unsigned W, unsigned H, unsigned C, unsigned N,
               const std::vector<float>& w_noisy, const std::vector<float>& w_basic, std::vector<float>& w_num,
               std::vector<float>& w_den, const std::vector<unsigned>& mask_w, const std::vector<unsigned>& proc_w,
               unsigned cst_w, unsigned pst_w, int nb_tiles, float* pct, orc_stats* stats) {
    const unsigned Aw = asw * asw, hb = H + 2 * N, wb = W + 2 * N;
    const size_t imgb = (size_t)C * wb * hb, img = (size_t)C * W * H;
    unsigned w_small = W, h_small = H, nw = 1, nh = 1;
    for (int n = nb_tiles; n > 1; n /= 2) {
        if (w_small > h_small) { w_small = (unsigned)std::floor((float)w_small * 0.5f); nw *= 2; }
        else { h_small = (unsigned)std::floor((float)h_small * 0.5f); nh *= 2; }
    }
    const unsigned h_bound = nh > 1 ? H - (nh - 1) * h_small : h_small;
    const unsigned w_bound = nw > 1 ? W - (nw - 1) * w_small : w_small;
    const int nt = (int)(nw * nh);
    std::vector<std::vector<float> > und_num(Aw, std::vector<float>(img, 0.0f)), und_den(Aw, std::vector<float>(img, 0.0f));
    std::vector<orc_stats> tstats((size_t)nt);
    std::vector<float> tpct((size_t)nt, 0.0f);
    std::vector<int> trc((size_t)nt, 0);
    for (orc_stats& ts : tstats) std::memset(&ts, 0, sizeof(ts));
    const double t0 = now_s();
    #pragma omp parallel for schedule(dynamic, 1) num_threads(g_threads > 0 ? g_threads : omp_get_max_threads())
    for (int kt = 0; kt < nt; kt++) {
        const unsigned i = (unsigned)kt / nw, j = (unsigned)kt % nw;
        const unsigned h = (i == nh - 1 ? h_bound : h_small) + 2 * N, w = (j == nw - 1 ? w_bound : w_small) + 2 * N;
        const size_t timg = (size_t)C * w * h;
        std::vector<float> t_noisy(Aw * timg), t_basic(step == 2 ? Aw * timg : 0), t_num(Aw * timg), t_den(Aw * timg);
        auto cut = [&](const std::vector<float>& src, std::vector<float>& dst) {
            for (unsigned a = 0; a < Aw; a++) {
                if (!mask_w[a]) continue;
                for (unsigned c = 0; c < C; c++)
                    for (unsigned p = 0; p < h; p++)
                        std::memcpy(&dst[a * timg + ((size_t)c * h + p) * w],
                                    &src[a * imgb + ((size_t)c * hb + i * h_small + p) * wb + j * w_small], w * sizeof(float));
            }
        };
        cut(w_noisy, t_noisy); if (step == 2) cut(w_basic, t_basic); cut(w_num, t_num); cut(w_den, t_den);
        if (h < 2 * N + Pw->k + 1 || w < 2 * N + Pw->k + 1) { trc[kt] = 2; continue; }   /* tile smaller than the search range */
        trc[kt] = pass_impl(step, Pw, asw, asw, w, h, C, t_noisy.data(), step == 2 ? t_basic.data() : nullptr, t_num.data(), t_den.data(),
                            mask_w.data(), proc_w.data(), cst_w, pst_w, 0, -1, &tstats[(size_t)kt]);
        tpct[kt] = denoised_percent(t_den.data(), mask_w.data(), Aw, w - 2 * N, h - 2 * N, C, N, Pw->k);
        for (unsigned a = 0; a < Aw; a++) {   /* interiors only */
            if (!mask_w[a]) continue;
            for (unsigned c = 0; c < C; c++)
                for (unsigned p = 0; p < h - 2 * N; p++) {
                    const size_t so = a * timg + ((size_t)c * h + N + p) * w + N;
                    const size_t dof = ((size_t)c * H + i * h_small + p) * W + j * w_small;
                    std::memcpy(&und_num[a][dof], &t_num[so], (w - 2 * N) * sizeof(float));
                    std::memcpy(&und_den[a][dof], &t_den[so], (w - 2 * N) * sizeof(float));
                }
        }
    }
    *pct = 0.0f;
    for (int kt = 0; kt < nt; kt++) {
        if (trc[kt]) return 1;
        *pct += tpct[kt];
        if (stats) { stats->groups += tstats[kt].groups; stats->sadct_groups += tstats[kt].sadct_groups; stats->stack_patches += tstats[kt].stack_patches;
                     stats->bm_seconds += tstats[kt].bm_seconds; }
    }
    if (stats) { stats->passes += 1; stats->total_seconds += now_s() - t0; }
    for (unsigned a = 0; a < Aw; a++) {
        if (!mask_w[a]) continue;
        symetrize(und_num[a].data(), &w_num[a * imgb], W, H, C, N);
        symetrize(und_den[a].data(), &w_den[a * imgb], W, H, C, N);
    }
    return 0;
}

int run_step(int step, const orc_params* P, float* LF_noisy, const unsigned* mask, float* LF_basic,
             float* LF_out, unsigned ang_major, unsigned awidth, unsigned aheight, unsigned an,
             unsigned W, unsigned H, unsigned C, int max_windows, orc_stats* stats) {
    const unsigned asize = awidth * aheight;
    const unsigned cs = aheight / 2, ct = awidth / 2;
    const unsigned cst = ang_major == ORC_ROWMAJOR ? cs * awidth + ct : cs + ct * aheight;
    const unsigned asw = 2 * an + 1;
    if (asw > aheight || asw > awidth) {
        std::printf("Wrong size of angular search window, the angular search window must be smaller than the light field angular size.\n");
        return 1;
    }
    const unsigned nHW = P->nSim + P->nDisp;
    const size_t img = (size_t)C * W * H;
    unsigned tau_4D = P->tau_4D;
    for (unsigned st = 0; st < asize; st++) {
        if (!mask[st]) continue;
        if (color_transform(LF_noisy + st * img, P->color_space, W, H, C, true)) return 1;
        if (step == 2 && color_transform(LF_basic + st * img, P->color_space, W, H, C, true)) return 1;
    }
    std::vector<float> num(asize * img, 0.0f), den(asize * img, 0.0f);
    const unsigned hb = H + 2 * nHW, wb = W + 2 * nHW;
    const size_t imgb = (size_t)C * wb * hb;
    std::vector<unsigned> proc(asize);
    for (unsigned st = 0; st < asize; st++) proc[st] = !mask[st];
    unsigned remaining = (unsigned)std::count(proc.begin(), proc.end(), 0u);
    const unsigned total = remaining;
    unsigned ps = 0, pt = 0, pst = 0;
    int windows = 0;
    const unsigned Aw = asw * asw;
    std::vector<float> w_noisy(Aw * imgb), w_basic(step == 2 ? Aw * imgb : 0), w_num(Aw * imgb), w_den(Aw * imgb);
    const double t_run0 = now_s();
    g_last_windows.clear();
    while (remaining) {
        if (max_windows > 0 && windows >= max_windows) break;
        if (g_time_limit > 0.0 && windows > 0 && now_s() - t_run0 > g_time_limit) break;   /* bounded timing samples (bench.py) */
        if (remaining == total && mask[cst]) { ps = cs; pt = ct; }
        else { /* bm5d.cpp:187-213: most exact-zero weights, last index wins ties */
            long best_cnt = -1;
            for (unsigned st = 0; st < asize; st++) {
                if (proc[st]) continue;
                const long cnt = (long)std::count(den.begin() + st * img, den.begin() + (st + 1) * img, 0.0f);
                if (cnt >= best_cnt) { pst = st; best_cnt = cnt; }
            }
            if (ang_major == ORC_ROWMAJOR) { ps = pst / awidth; pt = pst - ps * awidth; }
            else { pt = pst / aheight; ps = pst - pt * aheight; }
        }
        g_last_windows.push_back(ang_major == ORC_ROWMAJOR ? ps * awidth + pt : ps + pt * aheight);
        int cs_w, mins, maxs, ct_w, mint, maxt;
        search_window((int)ps, aheight, an, cs_w, mins, maxs);
        search_window((int)pt, awidth, an, ct_w, mint, maxt);
        const unsigned cst_w = ang_major == ORC_ROWMAJOR ? cs_w * asw + ct_w : cs_w + ct_w * asw;
        std::vector<unsigned> st_idx(Aw), mask_w(Aw), proc_w(Aw);
        for (unsigned s = 0; s < asw; s++)
            for (unsigned t = 0; t < asw; t++) {
                const unsigned S = s + mins, T = t + mint;
                if (ang_major == ORC_ROWMAJOR) st_idx[s * asw + t] = S * awidth + T;
                else st_idx[s + t * asw] = S + T * aheight;
            }
        for (unsigned i = 0; i < Aw; i++) {
            const unsigned st = st_idx[i];
            mask_w[i] = mask[st];
            if (!mask[st]) continue;
            symetrize(LF_noisy + st * img, &w_noisy[i * imgb], W, H, C, nHW);
            if (step == 2) symetrize(LF_basic + st * img, &w_basic[i * imgb], W, H, C, nHW);
            symetrize(&num[st * img], &w_num[i * imgb], W, H, C, nHW);
            symetrize(&den[st * img], &w_den[i * imgb], W, H, C, nHW);
        }
        for (unsigned i = 0; i < Aw; i++) proc_w[i] = !mask_w[i];
        unsigned rem_w = (unsigned)std::count(proc_w.begin(), proc_w.end(), 0u);
        const unsigned tot_w = rem_w;
        if (tot_w != Aw && tau_4D == ORC_DCT) tau_4D = ORC_SADCT; /* bm5d.cpp:276-280 */
        orc_params Pw = *P;
        Pw.tau_4D = tau_4D;
        unsigned ps_w = 0, pt_w = 0, pst_w = 0;
        while (rem_w) {
            if (rem_w == tot_w && mask_w[cst_w]) { ps_w = cs_w; pt_w = ct_w; pst_w = cst_w; }
            else {
                long best_cnt = -1;
                for (unsigned i = 0; i < Aw; i++) {
                    if (proc_w[i]) continue;
                    /* untiled: zeros of the padded SAI (bm5d.cpp:327); tile mode: summed over the tiles as sub_divide cuts
                     * them, halos included -- a zero under two tiles' halos counts twice (bm5d.cpp:598-600) */
                    const long cnt = g_tiles <= 1 ? (long)std::count(w_den.begin() + i * imgb, w_den.begin() + (i + 1) * imgb, 0.0f)
                                                  : tiled_zero_count(&w_den[i * imgb], W, H, C, nHW, g_tiles);
                    if (cnt >= best_cnt) { pst_w = i; best_cnt = cnt; }
                }
                if (ang_major == ORC_ROWMAJOR) { ps_w = pst_w / asw; pt_w = pst_w - ps_w * asw; }
                else { pt_w = pst_w / asw; ps_w = pst_w - pt_w * asw; }
            }
            float pct_tiles = 0.0f;
            if (g_tiles <= 1) {
                if (pass_impl(step, &Pw, asw, asw, wb, hb, C, w_noisy.data(), step == 2 ? w_basic.data() : nullptr,
                              w_num.data(), w_den.data(), mask_w.data(), proc_w.data(), cst_w, pst_w, 0, -1, stats))
                    return 1;
            } else if (tiled_pass(step, &Pw, asw, W, H, C, nHW, w_noisy, w_basic, w_num, w_den, mask_w, proc_w, cst_w, pst_w, g_tiles,
                                  &pct_tiles, stats))
                return 1;
            proc_w[pst_w] += 1;
            const unsigned st = ang_major == ORC_ROWMAJOR ? (mins + ps_w) * awidth + (mint + pt_w)
                                                          : (mins + ps_w) + (mint + pt_w) * aheight;
            proc[st] += 1;
            if (g_tiles <= 1 ? denoised_percent(w_den.data(), mask_w.data(), Aw, W, H, C, nHW, P->k) >= 100.0f
                             : pct_tiles >= 100.0f * (float)g_tiles)   /* bm5d.cpp:668-672 */
                for (unsigned i = 0; i < Aw; i++)
                    if (proc_w[i] == 0) { proc_w[i] += 1; proc[st_idx[i]] += 1; }
            rem_w = (unsigned)std::count(proc_w.begin(), proc_w.end(), 0u);
        }
        for (unsigned i = 0; i < Aw; i++) {
            const unsigned st = st_idx[i];
            if (!mask[st]) continue;
            unsymetrize(&num[st * img], &w_num[i * imgb], W, H, C, nHW);
            unsymetrize(&den[st * img], &w_den[i * imgb], W, H, C, nHW);
        }
        remaining = (unsigned)std::count(proc.begin(), proc.end(), 0u);
        windows++;
        if (stats) stats->windows += 1;
    }
    const float* sub = step == 1 ? LF_noisy : LF_basic;
    for (unsigned st = 0; st < asize; st++) {
        if (!mask[st]) continue;
        for (size_t i = 0; i < img; i++) {
            const size_t q = st * img + i;
            LF_out[q] = den[q] ? num[q] / den[q] : sub[q];
        }
    }
    for (unsigned st = 0; st < asize; st++) {
        if (!mask[st]) continue;
        if (step == 2) {
            if (color_transform(LF_out + st * img, P->color_space, W, H, C, false)) return 1;
            if (color_transform(LF_basic + st * img, P->color_space, W, H, C, false)) return 1;
        } else if (color_transform(LF_out + st * img, P->color_space, W, H, C, false)) return 1;
        if (color_transform(LF_noisy + st * img, P->color_space, W, H, C, false)) return 1;
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* Per-SAI BM3D (LFBM3Ddenoising): bm3d.cpp:86-690, bm3d_LF.cpp:75-125, nb_threads == 1          */
/* ------------------------------------------------------------------------------------------ */
/* One step on a mirror-padded image [C][Hb][Wb] (bm3d_1st_step bm3d.cpp:315-505, bm3d_2nd_step :507-690):
 * out = numerator / denominator over the whole padded image.  tau_2D in {DCT, BIOR}; the third
 * dimension is always the Hadamard transform (ht_filtering_hadamard :914-966,
 * wiener_filtering_hadamard :980-1027).  Groups of one row of reference patches are independent. */
int bm3d_step(int step, float sigma, float lambda3D, const float* noisy, const float* basic, float* out,
              unsigned Wb, unsigned Hb, unsigned C, unsigned nHW, unsigned k, unsigned N, unsigned p,
              unsigned useSD, unsigned color_space, unsigned tau_2D, orc_stats* stats) {
    if (C > 3 || (step == 2 && !basic) || (tau_2D != ORC_DCT && tau_2D != ORC_BIOR)) return 1;
    float sig[4];
    if (sigma_table(sigma, C, color_space, sig)) return 1;
    const float tauMatch = step == 1 ? (C == 1 ? 3.f : 1.f) * (sig[0] < 35.0f ? 2500 : 5000)   /* bm3d.cpp:339 */
                                     : (sig[0] < 35.0f ? 400 : 3500);                         /* bm3d.cpp:531 */
    const unsigned k2 = k * k;
    const size_t plane = (size_t)Wb * Hb;
    std::vector<unsigned> rows, cols;
    ind_init(rows, Hb - k + 1, nHW, p);
    ind_init(cols, Wb - k + 1, nHW, p);
    std::vector<float> kaiser;
    kaiser_window(k, kaiser);
    Norms2D n2; n2.k = k; dct2d_norms(k, n2.cn, n2.cni);
    /* block matching on channel 0 of the noisy image (step 1) / the basic estimate (step 2): bm3d.cpp:1187-1343
     * is the routine core:3301-3461 was derived from (same tables, same scan order, same duplicate rule) --
     * except that N == 1 has no short cut here: the single match is stored twice like any nSx_r == 1 */
    std::vector<unsigned> refs;
    for (unsigned i : rows) for (unsigned j : cols) refs.push_back(i * Wb + j);
    const unsigned Nst = N > 2 ? N : 2;
    std::vector<unsigned> self_idx(refs.size() * Nst), self_cnt(refs.size());
    {
        /* bm_self stores N entries per reference; run it with N >= 2 slots so that the duplicate fits */
        const double t0 = now_s();
        if (N >= 2) { if (bm_self(step == 1 ? noisy : basic, Wb, Hb, k, N, nHW, nHW, tauMatch, refs.data(), (unsigned)refs.size(), self_idx.data(), self_cnt.data())) return 1; }
        else {
            std::vector<unsigned> i2(refs.size() * 2), c2(refs.size());
            if (bm_self(step == 1 ? noisy : basic, Wb, Hb, k, 2, nHW, nHW, tauMatch, refs.data(), (unsigned)refs.size(), i2.data(), c2.data())) return 1;
            for (size_t r = 0; r < refs.size(); r++) { self_idx[2 * r] = i2[2 * r]; self_idx[2 * r + 1] = i2[2 * r]; self_cnt[r] = 2; }
        }
        if (stats) stats->bm_seconds += now_s() - t0;
    }
    std::vector<float> num(C * plane, 0.0f), den(C * plane, 0.0f);
    struct G3 { unsigned nSx; std::vector<float> patches; float w[4]; };
    for (size_t ri = 0; ri < rows.size(); ri++) {
        std::vector<G3> outs(cols.size());
        #pragma omp parallel for schedule(dynamic, 1) num_threads(g_threads > 0 ? g_threads : omp_get_max_threads())
        for (int jj = 0; jj < (int)cols.size(); jj++) {
            const size_t slot = ri * cols.size() + jj;
            const unsigned nSx = self_cnt[slot];
            G3& go = outs[jj];
            go.nSx = nSx;
            const int S = step == 2 ? 2 : 1;
            /* group_3D[s][c][pq][n] (bm3d.cpp:393-401 / :594-606); patches at column Wb - k read the tables'
             * zero initialisation (bm3d.cpp:737, :857) */
            std::vector<float> X((size_t)S * C * k2 * nSx, 0.0f), tmp(k2);
            for (int s2 = 0; s2 < S; s2++) {
                const float* src = (step == 2 && s2 == 1) ? basic : noisy;   /* s2 = 0: img, 1: est */
                for (unsigned c = 0; c < C; c++)
                    for (unsigned n = 0; n < nSx; n++) {
                        const unsigned pos = self_idx[slot * Nst + n];
                        if (pos % Wb >= Wb - k) continue;
                        if (tau_2D == ORC_DCT) dct2d_fwd(src + c * plane + pos, Wb, tmp.data(), k, n2);
                        else bior_fwd(src + c * plane + pos, Wb, tmp.data(), k);
                        for (unsigned pq = 0; pq < k2; pq++) X[(((size_t)s2 * C + c) * k2 + pq) * nSx + n] = tmp[pq];
                    }
            }
            float weight[4] = {0, 0, 0, 0};
            const float coef = 1.0f / (float)nSx;
            float* img = X.data();
            float* est = X.data() + (size_t)C * k2 * nSx;
            float* F;
            if (step == 1) {   /* bm3d.cpp:914-966 */
                const float coef_norm = std::sqrt((float)nSx);
                for (unsigned v = 0; v < k2 * C; v++) hadamard(img + v * nSx, nSx);
                for (unsigned c = 0; c < C; c++) {
                    const float T = lambda3D * sig[c] * coef_norm;
                    float* g = img + (size_t)c * nSx * k2;
                    for (unsigned i = 0; i < k2 * nSx; i++) { if (std::fabs(g[i]) > T) weight[c]++; else g[i] = 0.0f; }
                }
                for (unsigned v = 0; v < k2 * C; v++) hadamard(img + v * nSx, nSx);
                for (size_t i = 0; i < (size_t)C * k2 * nSx; i++) img[i] *= coef;
                F = img;
            } else {           /* bm3d.cpp:980-1027 */
                for (unsigned v = 0; v < k2 * C; v++) { hadamard(img + v * nSx, nSx); hadamard(est + v * nSx, nSx); }
                for (unsigned c = 0; c < C; c++) {
                    const size_t dc = (size_t)c * nSx * k2;
                    for (unsigned i = 0; i < k2 * nSx; i++) {
                        float value = est[dc + i] * est[dc + i] * coef;
                        value /= (value + sig[c] * sig[c]);
                        est[dc + i] = img[dc + i] * value * coef;
                        weight[c] += value;
                    }
                }
                for (unsigned v = 0; v < k2 * C; v++) hadamard(est + v * nSx, nSx);
                F = est;
            }
            if (!useSD)
                for (unsigned c = 0; c < C; c++)
                    go.w[c] = weight[c] > 0.0f ? 1.0f / (float)(sig[c] * sig[c] * weight[c]) : 1.0f;
            else {             /* sd_weighting bm3d.cpp:1345-1373: reads the first nSx*k2 entries -- channel 0 -- for every channel */
                const unsigned Nn = nSx * k2;
                for (unsigned c = 0; c < C; c++) {
                    float mean = 0.0f, sd = 0.0f;
                    for (unsigned i = 0; i < Nn; i++) { mean += F[i]; sd += F[i] * F[i]; }
                    const float res = (sd - mean * mean / (float)Nn) / (float)(Nn - 1);
                    go.w[c] = res > 0.0f ? 1.0f / std::sqrt(res) : 0.0f;
                }
            }
            /* inverse 2-D of every filtered patch (bm3d.cpp:428-432 / :639-643) */
            go.patches.assign((size_t)C * nSx * k2, 0.0f);
            for (unsigned c = 0; c < C; c++)
                for (unsigned n = 0; n < nSx; n++) {
                    float* pp = &go.patches[((size_t)c * nSx + n) * k2];
                    for (unsigned pq = 0; pq < k2; pq++) pp[pq] = F[((size_t)c * k2 + pq) * nSx + n];
                    if (tau_2D == ORC_DCT) dct2d_inv(pp, k, n2); else bior_inv(pp, k);
                }
        }
        /* aggregation in the reference's order (bm3d.cpp:434-463 / :645-674) */
        for (size_t jj = 0; jj < cols.size(); jj++) {
            const size_t slot = ri * cols.size() + jj;
            const G3& go = outs[jj];
            for (unsigned c = 0; c < C; c++)
                for (unsigned n = 0; n < go.nSx; n++) {
                    const size_t base = self_idx[slot * Nst + n] + c * plane;
                    const float* pp = &go.patches[((size_t)c * go.nSx + n) * k2];
                    for (unsigned a = 0; a < k; a++)
                        for (unsigned b = 0; b < k; b++) {
                            num[base + a * Wb + b] += kaiser[a * k + b] * go.w[c] * pp[a * k + b];
                            den[base + a * Wb + b] += kaiser[a * k + b] * go.w[c];
                        }
                }
            if (stats) { stats->groups += 1; stats->stack_patches += go.nSx; }
        }
    }
    for (size_t i = 0; i < C * plane; i++) out[i] = num[i] / den[i];   /* bm3d.cpp:467-468 / :678-679 */
    if (stats) stats->passes += 1;
    return 0;
}

/* run_bm3d (bm3d.cpp:86-300, nb_threads == 1 branch) for every SAI of the mask (bm3d_LF.cpp:111-121) */
int run_bm3d_lf(float sigma, float* LF_noisy, const unsigned* mask, float* LF_basic, float* LF_denoised,
                unsigned asize, unsigned W, unsigned H, unsigned C, unsigned nHard, unsigned nWien,
                unsigned kHard, unsigned kWien, unsigned NHard, unsigned NWien, unsigned pHard, unsigned pWien,
                unsigned useSD_h, unsigned useSD_w, unsigned tau_2D_hard, unsigned tau_2D_wien, float lambda3D,
                unsigned color_space, orc_stats* stats) {
    const double t0 = now_s();
    const size_t img = (size_t)C * W * H;
    const unsigned wb = W + 2 * nHard, hb = H + 2 * nHard;   /* both steps pad by nHard (bm3d.cpp:126-127, :158) */
    std::vector<float> sym_noisy, sym_basic((size_t)C * wb * hb), sym_out((size_t)C * wb * hb);
    for (unsigned st = 0; st < asize; st++) {
        if (!mask[st]) continue;
        float* noisy = LF_noisy + st * img; float* basic = LF_basic + st * img; float* deno = LF_denoised + st * img;
        if (color_transform(noisy, color_space, W, H, C, true)) return 1;
        sym_noisy.assign((size_t)C * wb * hb, 0.0f);
        symetrize(noisy, sym_noisy.data(), W, H, C, nHard);
        if (bm3d_step(1, sigma, lambda3D, sym_noisy.data(), nullptr, sym_out.data(), wb, hb, C, nHard, kHard, NHard, pHard,
                      useSD_h, color_space, tau_2D_hard, stats)) return 1;
        unsymetrize(basic, sym_out.data(), W, H, C, nHard);      /* crop bm3d.cpp:148-156 */
        symetrize(basic, sym_basic.data(), W, H, C, nHard);
        if (bm3d_step(2, sigma, lambda3D, sym_noisy.data(), sym_basic.data(), sym_out.data(), wb, hb, C, nWien, kWien, NWien,
                      pWien, useSD_w, color_space, tau_2D_wien, stats)) return 1;
        /* the crop of the second step uses nWien as the offset into the nHard-padded image (bm3d.cpp:181-189):
         * only right when the two are equal, which every documented parameter set satisfies */
        for (unsigned c = 0; c < C; c++)
            for (unsigned i = 0; i < H; i++)
                for (unsigned j = 0; j < W; j++)
                    deno[((size_t)c * H + i) * W + j] = sym_out[(size_t)c * wb * hb + (size_t)(nWien + i) * wb + nWien + j];
        if (color_transform(deno, color_space, W, H, C, false)) return 1;
        if (color_transform(noisy, color_space, W, H, C, false)) return 1;
        if (color_transform(basic, color_space, W, H, C, false)) return 1;
    }
    if (stats) stats->total_seconds += now_s() - t0;
    return 0;
}

/* MT19937 (mt19937ar.c) */
struct MT {
    unsigned long mt[624]; int mti = 625;
    void seed(unsigned long s) {
        mt[0] = s & 0xffffffffUL;
        for (mti = 1; mti < 624; mti++) {
            mt[mti] = (1812433253UL * (mt[mti - 1] ^ (mt[mti - 1] >> 30)) + (unsigned long)mti);
            mt[mti] &= 0xffffffffUL;
        }
    }
    unsigned long next() {
        static const unsigned long mag01[2] = {0x0UL, 0x9908b0dfUL};
        unsigned long y;
        if (mti >= 624) {
            if (mti == 625) seed(5489UL);
            int kk;
            for (kk = 0; kk < 624 - 397; kk++) {
                y = (mt[kk] & 0x80000000UL) | (mt[kk + 1] & 0x7fffffffUL);
                mt[kk] = mt[kk + 397] ^ (y >> 1) ^ mag01[y & 0x1UL];
            }
            for (; kk < 623; kk++) {
                y = (mt[kk] & 0x80000000UL) | (mt[kk + 1] & 0x7fffffffUL);
                mt[kk] = mt[kk + (397 - 624)] ^ (y >> 1) ^ mag01[y & 0x1UL];
            }
            y = (mt[623] & 0x80000000UL) | (mt[0] & 0x7fffffffUL);
            mt[623] = mt[396] ^ (y >> 1) ^ mag01[y & 0x1UL];
            mti = 0;
        }
        y = mt[mti++];
        y ^= (y >> 11);
        y ^= (y << 7) & 0x9d2c5680UL;
        y ^= (y << 15) & 0xefc60000UL;
        y ^= (y >> 18);
        return y & 0xffffffffUL;
    }
    double res53() {
        const unsigned long a = next() >> 5, b = next() >> 6;
        return (a * 67108864.0 + b) * (1.0 / 9007199254740992.0);
    }
};
MT g_mt;

} /* namespace */

/* ============================================================================================ */
/* C API                                                                                        */
/* ============================================================================================ */
extern "C" {

int orc_bm3d_step(int step, float sigma, float lambda3D, const float* noisy, const float* basic, float* out,
                  unsigned Wb, unsigned Hb, unsigned C, unsigned nHW, unsigned k, unsigned N, unsigned p,
                  unsigned useSD, unsigned color_space, unsigned tau_2D, orc_stats* stats) {
    return bm3d_step(step, sigma, lambda3D, noisy, basic, out, Wb, Hb, C, nHW, k, N, p, useSD, color_space, tau_2D, stats);
}
int orc_run_bm3d_lf(float sigma, float* LF_noisy, const unsigned* mask, float* LF_basic, float* LF_denoised,
                    unsigned asize, unsigned W, unsigned H, unsigned C, unsigned nHard, unsigned nWien,
                    unsigned kHard, unsigned kWien, unsigned NHard, unsigned NWien, unsigned pHard, unsigned pWien,
                    unsigned useSD_h, unsigned useSD_w, unsigned tau_2D_hard, unsigned tau_2D_wien, float lambda3D,
                    unsigned color_space, orc_stats* stats) {
    return run_bm3d_lf(sigma, LF_noisy, mask, LF_basic, LF_denoised, asize, W, H, C, nHard, nWien, kHard, kWien, NHard, NWien,
                       pHard, pWien, useSD_h, useSD_w, tau_2D_hard, tau_2D_wien, lambda3D, color_space, stats);
}
void orc_haar_forward(float* v, unsigned n) { haar_fwd(v, n); }
void orc_haar_inverse(float* v, unsigned n) { haar_inv(v, n); }
void orc_hadamard(float* v, unsigned n) { hadamard(v, n); }
void orc_bior_forward(const float* in, unsigned in_stride, float* out, unsigned n) { bior_fwd(in, in_stride, out, n); }
void orc_bior_inverse(float* patch, unsigned n) { bior_inv(patch, n); }
void orc_redft10(const float* x, float* y, unsigned n) {
    double a[kMaxDct], b[kMaxDct];
    for (unsigned i = 0; i < n; i++) a[i] = x[i];
    redft10_d(a, 1, b, 1, n);
    for (unsigned i = 0; i < n; i++) y[i] = (float)b[i];
}
void orc_redft01(const float* x, float* y, unsigned n) {
    double a[kMaxDct], b[kMaxDct];
    for (unsigned i = 0; i < n; i++) a[i] = x[i];
    redft01_d(a, 1, b, 1, n);
    for (unsigned i = 0; i < n; i++) y[i] = (float)b[i];
}
void orc_dct2d_forward(const float* in, unsigned in_stride, float* out, unsigned k) {
    Norms2D nm; dct2d_norms(k, nm.cn, nm.cni); dct2d_fwd(in, in_stride, out, k, nm);
}
void orc_dct2d_inverse(float* patch, unsigned k) {
    Norms2D nm; dct2d_norms(k, nm.cn, nm.cni); dct2d_inv(patch, k, nm);
}
void orc_dct4d_forward(float* v, unsigned aw, unsigned ah) {
    Norms4D nm; dct4d_norms(aw, ah, nm.cn, nm.cni); dct4d_fwd(v, aw, ah, nm);
}
void orc_dct4d_inverse(float* v, unsigned aw, unsigned ah) {
    Norms4D nm; dct4d_norms(aw, ah, nm.cn, nm.cni); dct4d_inv(v, aw, ah, nm);
}
void orc_sadct_forward(float* v, const unsigned* mask, unsigned aw, unsigned ah, unsigned* mask_dct) {
    Shape sh; sh.build(mask, aw, ah); sadct_fwd(v, sh);
    if (mask_dct) for (unsigned i = 0; i < aw * ah; i++) mask_dct[i] = sh.mask_dct[i];
}
void orc_sadct_inverse(float* v, const unsigned* mask, unsigned aw, unsigned ah) {
    Shape sh; sh.build(mask, aw, ah); sadct_inv(v, sh);
}
void orc_ht_filter_slab(float* X, unsigned nSx, unsigned A, unsigned C, const float* sigma, float lambda, float* weight,
                        const unsigned* mask_dct, unsigned tau5) { ht_filter_slab(X, nSx, A, C, sigma, lambda, weight, mask_dct, tau5); }
void orc_wiener_filter_slab(float* Xo, float* Xe, unsigned nSx, unsigned A, unsigned C, const float* sigma, float* weight,
                            const unsigned* mask_dct, unsigned tau5) { wiener_filter_slab(Xo, Xe, nSx, A, C, sigma, weight, mask_dct, tau5); }
void orc_kaiser_window(float* out, unsigned k) {
    std::vector<float> w; kaiser_window(k, w); std::memcpy(out, w.data(), sizeof(float) * k * k);
}

int orc_bm_self(const float* img, unsigned W, unsigned H, unsigned k, unsigned N, unsigned nHW,
                unsigned nSim, float tauMatch, const unsigned* refs, unsigned n_refs,
                unsigned* out_idx, unsigned* out_cnt) {
    return bm_self(img, W, H, k, N, nHW, nSim, tauMatch, refs, n_refs, out_idx, out_cnt);
}
int orc_bm_stereo(const float* img1, const float* img2, unsigned W, unsigned H, unsigned k,
                  unsigned nDisp, float tauMatch, unsigned* best, unsigned char* shape) {
    return bm_stereo(img1, img2, W, H, k, nDisp, tauMatch, best, shape);
}

int orc_pass(int step, const orc_params* P, unsigned aw, unsigned ah, unsigned Wb, unsigned Hb,
             unsigned C, const float* noisy, const float* basic, float* num, float* den,
             const unsigned* mask, const unsigned* procSAI, unsigned cst, unsigned pst,
             int ref_row_begin, int ref_row_end, orc_stats* stats) {
    return pass_impl(step, P, aw, ah, Wb, Hb, C, noisy, basic, num, den, mask, procSAI, cst, pst,
                     ref_row_begin, ref_row_end, stats);
}

int orc_run_step1(const orc_params* P, float* LF_noisy, const unsigned* mask, float* LF_basic,
                  unsigned ang_major, unsigned awidth, unsigned aheight, unsigned an, unsigned W,
                  unsigned H, unsigned C, int max_windows, orc_stats* stats) {
    return run_step(1, P, LF_noisy, mask, nullptr, LF_basic, ang_major, awidth, aheight, an, W, H, C, max_windows, stats);
}
int orc_run_step2(const orc_params* P, float* LF_noisy, const unsigned* mask, float* LF_basic,
                  float* LF_denoised, unsigned ang_major, unsigned awidth, unsigned aheight,
                  unsigned an, unsigned W, unsigned H, unsigned C, int max_windows, orc_stats* stats) {
    return run_step(2, P, LF_noisy, mask, LF_basic, LF_denoised, ang_major, awidth, aheight, an, W, H, C, max_windows, stats);
}

void orc_mt_seed(unsigned long s) { g_mt.seed(s); }
unsigned long orc_mt_int32(void) { return g_mt.next(); }
double orc_mt_res53(void) { return g_mt.res53(); }
void orc_add_noise(const float* img, float* out, unsigned long long n, float sigma) {
    for (unsigned long long i = 0; i < n; i++) { /* utilities.cpp:176-183 */
        const double a = g_mt.res53();
        const double b = g_mt.res53();
        const double z = (double)sigma * std::sqrt(-2.0 * std::log(a)) * std::cos(2.0 * kPi * b);
        out[i] = img[i] + (float)z;
    }
}
void orc_symetrize(const float* img, float* out, unsigned W, unsigned H, unsigned C, unsigned N) { symetrize(img, out, W, H, C, N); }
void orc_unsymetrize(float* img, const float* sym, unsigned W, unsigned H, unsigned C, unsigned N) { unsymetrize(img, sym, W, H, C, N); }
int orc_color_transform(float* img, unsigned cs, unsigned W, unsigned H, unsigned C, int forward) { return color_transform(img, cs, W, H, C, forward != 0); }
int orc_sigma_table(float sigma, unsigned C, unsigned cs, float* out) { return sigma_table(sigma, C, cs, out); }
unsigned orc_last_weights(float* out, unsigned cap) {
    const unsigned n = (unsigned)g_last_weights.size();
    for (unsigned i = 0; i < n && i < cap; i++) out[i] = g_last_weights[i];
    return n;
}
unsigned orc_ind_initialize(unsigned max_size, unsigned N, unsigned step, unsigned* out) {
    std::vector<unsigned> v; ind_init(v, max_size, N, step);
    if (out) std::memcpy(out, v.data(), v.size() * sizeof(unsigned));
    return (unsigned)v.size();
}
void orc_search_window(int aidx, unsigned asize, unsigned an, int* c, int* mn, int* mx) { search_window(aidx, asize, an, *c, *mn, *mx); }
float orc_denoised_percent(const float* den, const unsigned* mask, unsigned A, unsigned W, unsigned H,
                           unsigned C, unsigned N, unsigned k) { return denoised_percent(den, mask, A, W, H, C, N, k); }
void orc_psnr(const float* a, const float* b, unsigned long long n, float* psnr, float* rmse) {
    float tmp = 0.0f; /* utilities.cpp:412-435: float accumulation, reproduce */
    for (unsigned long long i = 0; i < n; i++) tmp += (a[i] - b[i]) * (a[i] - b[i]);
    *rmse = std::sqrt(tmp / (float)n);
    *psnr = 20.0f * std::log10(255.0f / *rmse);
}
void orc_set_threads(int n) { g_threads = n; }
int orc_last_windows(unsigned* out, unsigned cap) {
    for (size_t i = 0; i < g_last_windows.size() && i < cap && out; i++) out[i] = g_last_windows[i];
    return (int)g_last_windows.size();
}
void orc_set_time_limit(double seconds) { g_time_limit = seconds; }
void orc_set_tiles(int n) {   /* floored to a power of two like main.cpp:101-102 does with nb_threads */
    int t = 1;
    while (t * 2 <= n) t *= 2;
    g_tiles = n > 1 ? t : 1;
}
int orc_get_threads(void) {
#ifdef _OPENMP
    return g_threads > 0 ? g_threads : omp_get_max_threads();
#else
    return 1;
#endif
}

} /* extern "C" */
