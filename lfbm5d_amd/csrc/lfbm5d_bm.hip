/*
 * lfbm5d_bm.hip -- block matching on gfx950, index-exact with the reference's arithmetic.
 *
 * The reference scores candidate patches with an integral-image recurrence in float32
 * (precompute_BM core:3301-3461, precompute_BM_stereo core:3479-3611):
 *     S[i][j] = S[i][j-1] + S[i-1][j] - S[i-1][j-1] + D(br) - D(bl) - D(tr) + D(tl)
 * whose rounding depends on the evaluation order, and its quirks (zero band outside
 * [nHW, dim-nHW), never-written entries = 2*threshold, mirrored test vs scored distance) are all
 * consequences of that construction.  To select the SAME patches the kernels below evaluate the
 * same recurrence in the same order -- one wavefront per displacement table, rows skewed across
 * the 64 lanes (lane l works on row r0+l, one column behind lane l-1) so the three neighbours come
 * from the lane's own previous value and a one-lane shift -- instead of re-deriving distances by
 * direct SSD.  No FMA contraction in this file: it would change the rounding.
 *
 * Bound: latency of the dependent add chain (about 8 dependent VALU ops per step, ~(cols+64)*rows/64
 * steps per table); every table is an independent wave, so a pass keeps ~2000 waves in flight.
 */
#include "lfbm5d_kernels.h"

#pragma clang fp contract(off)

namespace lfbm5d {

namespace {

__device__ __forceinline__ float sq_diff(const float* __restrict__ i1, const float* __restrict__ i2,
                                         int q, int dk) {
    const float d = i2[q + dk] - i1[q];
    return d * d;
}

/* D(y,x): squared difference image of the current displacement, zero outside the band the
 * reference fills (core:3335-3340 / :3520-3525). */
struct DiffImg {
    const float* i1; const float* i2; int dk; int W, H, b;
    __device__ __forceinline__ float operator()(int y, int x) const {
        if (x < b || x >= W - b || y < b || y >= H - b) return 0.0f;
        return sq_diff(i1, i2, y * W + x, dk);
    }
};

template <int K>
__global__ __launch_bounds__(64) void k_bm_scan(ScanArgs a) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x;
    const int W = a.W, H = a.H, b = a.b;
    const int Ns = 2 * (int)a.half + 1;
    const int ncand = Ns * Ns;
    const int nrows = H - 2 * b - (int)a.trim, ncols = W - 2 * b - (int)a.trim;
    float* prow = lds;            /* [ncols] previous block's last row */
    float* ebuf = lds + ncols;    /* [64*K] staging of difference terms */

    int di, dj, slot = 0;
    DiffImg D;
    const size_t WH = (size_t)W * H;
    if (a.stereo) {
        slot = blockIdx.x / ncand;
        const int ddk = blockIdx.x % ncand;
        di = ddk / Ns; dj = ddk % Ns;
        D.i1 = a.est + (size_t)a.pst * WH;
        D.i2 = a.est + (size_t)a.st_of_slot[slot] * WH;
        D.dk = di * W + dj - (int)a.half * (1 + W);
    } else {
        di = blockIdx.x / Ns; dj = blockIdx.x % Ns;
        D.i1 = D.i2 = a.est + (size_t)a.pst * WH;
        D.dk = di * W + dj - (int)a.half;
    }
    D.W = W; D.H = H; D.b = b;
    float* table = a.stereo ? a.tables + (size_t)blockIdx.x * WH : nullptr;
    const int djs = dj - (int)a.half;
    const int nSim = (int)a.half;
    const int ord_fwd = dj * Ns + di;
    const int ord_bwd = (-djs + nSim) * Ns + (nSim + 1) + (nSim - di);

    auto emit = [&](int y, int x, float S) {
        const int pos = y * W + x;
        if (a.stereo) { table[pos] = S; return; }
        const int r = a.refmap[pos];
        if (r >= 0) a.scores[(size_t)r * ncand + ord_fwd] = S;
        if (di > 0) { /* this entry is also the score of the mirrored candidate of ref pos + d */
            const int yy = y + di, xx = x + djs;
            if (yy < H && xx >= 0 && xx < W) {
                const int r2 = a.refmap[yy * W + xx];
                if (r2 >= 0) a.scores[(size_t)r2 * ncand + ord_bwd] = S;
            }
        }
    };

    /* ---- corner (core:3344-3352): K*K terms summed sequentially ---- */
    for (int e = lane; e < K * K; e += 64) ebuf[e] = D(b + e / K, b + e % K);
    __syncthreads();
    float v = 0.0f;
    for (int e = 0; e < K * K; e++) v += ebuf[e];
    __syncthreads();
    if (lane == 0) { prow[0] = v; emit(b, b, v); }

    /* ---- first row (core:3354-3362): S[b][j] = S[b][j-1] + sum_p (D(b+p, j-1+K) - D(b+p, j-1)) ---- */
    for (int c0 = 1; c0 < ncols; c0 += 64) {
        const int c = c0 + lane;
        if (c < ncols)
            for (int p = 0; p < K; p++) ebuf[lane * K + p] = D(b + p, b + c - 1 + K) - D(b + p, b + c - 1);
        __syncthreads();
        const int m = min(64, ncols - c0);
        float mine = 0.0f;
        for (int l = 0; l < m; l++) {
            for (int p = 0; p < K; p++) v += ebuf[l * K + p];
            if (l == lane) mine = v;
        }
        __syncthreads();
        if (c < ncols) { prow[c] = mine; emit(b, b + c, mine); }
    }
    __syncthreads();

    /* ---- remaining rows, 64 at a time, skewed one column per lane ---- */
    for (int r0 = 1; r0 < nrows; r0 += 64) {
        const int r = r0 + lane;
        const bool row_ok = r < nrows;
        const int y = b + r;
        const int last_lane = min(63, nrows - 1 - r0);
        float prevS = 0.0f;   /* S[r][c-1] */
        float up_prev = 0.0f; /* S[r-1][c-1] */
        const int nsteps = ncols + last_lane;
        for (int t = 0; t < nsteps; t++) {
            const int c = t - lane;
            /* value the lane above produced in the previous step == S[r-1][c] */
            float up = __shfl_up(prevS, 1);
            if (lane == 0) up = (t < ncols) ? prow[t] : 0.0f;
            const bool act = row_ok && c >= 0 && c < ncols;
            float S = prevS;
            if (act) {
                const int x = b + c;
                if (c == 0) { /* first column (core:3367-3372) */
                    S = up;
                    for (int q = 0; q < K; q++) S += D(y - 1 + K, b + q) - D(y - 1, b + q);
                } else {      /* general case (core:3377-3387), same association */
                    S = prevS + up;
                    S = S - up_prev;
                    S = S + D(y + K - 1, x + K - 1);
                    S = S - D(y + K - 1, x - 1);
                    S = S - D(y - 1, x + K - 1);
                    S = S + D(y - 1, x - 1);
                }
                emit(y, x, S);
                if (lane == last_lane) prow[c] = S; /* hand-off row for the next block */
            }
            if (c >= 0) up_prev = up;
            prevS = S;
        }
        __syncthreads();
    }
}

/* order-preserving float -> uint map (scores can be slightly negative after cancellation) */
__device__ __forceinline__ unsigned f2ord(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

/* Candidate selection of precompute_BM (core:3397-3445), one wave per reference patch. */
__global__ __launch_bounds__(64) void k_self_select(const float* __restrict__ scores,
                                                     const unsigned* __restrict__ refs, unsigned n_refs,
                                                     int W, int nSim, int N, float thr,
                                                     unsigned* __restrict__ self_idx,
                                                     unsigned* __restrict__ self_cnt) {
    extern __shared__ unsigned long long keys[];
    const int lane = threadIdx.x;
    const unsigned ref = blockIdx.x;
    if (ref >= n_refs) return;
    const int Ns = 2 * nSim + 1, ncand = Ns * Ns;
    const float* sc = scores + (size_t)ref * ncand;
    const int k_r = (int)refs[ref];
    int cnt = 0;
    for (int e = lane; e < ncand; e += 64) {
        const int djp = e / Ns - nSim, r = e % Ns;
        const bool fwd = r <= nSim;
        const int dip = fwd ? r : r - 2 * nSim - 1;
        const float score = sc[e];
        /* backward half: tested with the mirrored table at k_r, scored at the candidate (core:3415-3419) */
        const float test = fwd ? score : sc[(-djp + nSim) * Ns + (-dip)];
        const bool pass = test < thr;
        keys[e] = pass ? (((unsigned long long)f2ord(score) << 32) | (unsigned)e) : ~0ull;
        cnt += pass ? 1 : 0;
    }
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
    __syncthreads();
    unsigned nSx;
    if (N > cnt) { nSx = 1; while (nSx * 2 <= (unsigned)cnt) nSx *= 2; } else nSx = N;
    unsigned* out = self_idx + (size_t)ref * N;
    if (cnt == 0) { /* core:3429-3432 */
        if (lane == 0) { out[0] = k_r; out[1] = k_r; self_cnt[ref] = 2; }
        return;
    }
    unsigned long long last = 0;
    bool first = true;
    for (unsigned n = 0; n < nSx; n++) { /* n-th smallest (distance, scan order) key */
        unsigned long long best = ~0ull;
        for (int e = lane; e < ncand; e += 64) {
            const unsigned long long kk = keys[e];
            if ((first || kk > last) && kk < best) best = kk;
        }
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned long long oth = __shfl_xor(best, o);
            best = oth < best ? oth : best;
        }
        last = best; first = false;
        if (lane == 0) {
            const int e = (int)(best & 0xffffffffu);
            const int djp = e / Ns - nSim, r = e % Ns;
            const int dip = (r <= nSim) ? r : r - 2 * nSim - 1;
            out[n] = (unsigned)(k_r + dip * W + djp);
            if (nSx == 1) out[1] = out[0]; /* duplicate rule core:3443-3444 */
        }
    }
    if (lane == 0) self_cnt[ref] = nSx == 1 ? 2 : nSx;
}

__global__ void k_self_trivial(const unsigned* __restrict__ refs, unsigned n_refs,
                               unsigned* __restrict__ self_idx, unsigned* __restrict__ self_cnt) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_refs) { self_idx[i] = refs[i]; self_cnt[i] = 1; } /* core:3448-3460 */
}

/* argmin over the (2 nDisp+1)^2 displacement tables (core:3581-3608); ties keep scan order
 * (dj outer, di inner), the order the reference pushes candidates in. */
__global__ void k_stereo_argmin(const float* __restrict__ tables, unsigned slot, unsigned st, int W,
                                int H, int k, int nDisp, float thr, unsigned* __restrict__ best,
                                unsigned char* __restrict__ shape) {
    const int span_c = W - 2 * nDisp - k + 1, span_r = H - 2 * nDisp - k + 1;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= span_c * span_r) return;
    const int y = nDisp + i / span_c, x = nDisp + i % span_c;
    const int Ns = 2 * nDisp + 1, ncand = Ns * Ns;
    const size_t WH = (size_t)W * H;
    const int pos = y * W + x;
    const float* t = tables + (size_t)slot * ncand * WH + pos;
    float bv = 0.0f; int bo = -1, bi = 0;
    for (int ddk = 0; ddk < ncand; ddk++) {
        const float v = t[(size_t)ddk * WH];
        const int di = ddk / Ns, dj = ddk % Ns;
        const int order = dj * Ns + di;
        if (bo < 0 || v < bv || (v == bv && order < bo)) {
            bv = v; bo = order;
            bi = pos + (di - nDisp) * W + (dj - nDisp);
        }
    }
    best[(size_t)st * WH + pos] = (unsigned)bi;
    shape[(size_t)st * WH + pos] = bv < thr ? 1 : 0;
}

__global__ void k_refmap(const unsigned* __restrict__ refs, unsigned n_refs, int* __restrict__ refmap) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_refs) refmap[refs[i]] = (int)i;
}

} /* namespace */

hipError_t launch_refmap(hipStream_t s, const unsigned* refs, unsigned n_refs, int* refmap) {
    hipLaunchKernelGGL(k_refmap, dim3((n_refs + 255) / 256), dim3(256), 0, s, refs, n_refs, refmap);
    return hipGetLastError();
}

hipError_t launch_bm_scan(hipStream_t s, const ScanArgs& a) {
    const unsigned ncols = a.W - 2 * a.b - a.trim;
    const size_t lds = (size_t)(ncols + 64 * a.k) * sizeof(float);
    switch (a.k) {
        case 8:  hipLaunchKernelGGL(k_bm_scan<8>,  dim3(a.n_tables), dim3(64), lds, s, a); break;
        case 12: hipLaunchKernelGGL(k_bm_scan<12>, dim3(a.n_tables), dim3(64), lds, s, a); break;
        case 16: hipLaunchKernelGGL(k_bm_scan<16>, dim3(a.n_tables), dim3(64), lds, s, a); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_self_select(hipStream_t s, const float* scores, const unsigned* refs, unsigned n_refs,
                              unsigned W, unsigned nSim, unsigned N, float thr, unsigned* self_idx,
                              unsigned* self_cnt) {
    const unsigned Ns = 2 * nSim + 1;
    hipLaunchKernelGGL(k_self_select, dim3(n_refs), dim3(64), (size_t)Ns * Ns * sizeof(unsigned long long), s,
                       scores, refs, n_refs, (int)W, (int)nSim, (int)N, thr, self_idx, self_cnt);
    return hipGetLastError();
}

hipError_t launch_self_trivial(hipStream_t s, const unsigned* refs, unsigned n_refs, unsigned* self_idx,
                               unsigned* self_cnt) {
    hipLaunchKernelGGL(k_self_trivial, dim3((n_refs + 255) / 256), dim3(256), 0, s, refs, n_refs, self_idx, self_cnt);
    return hipGetLastError();
}

hipError_t launch_stereo_argmin(hipStream_t s, const float* tables, unsigned slot, unsigned st,
                                unsigned W, unsigned H, unsigned k, unsigned nDisp, float thr,
                                unsigned* best, unsigned char* shape) {
    const unsigned n = (W - 2 * nDisp - k + 1) * (H - 2 * nDisp - k + 1);
    hipLaunchKernelGGL(k_stereo_argmin, dim3((n + 255) / 256), dim3(256), 0, s, tables, slot, st, (int)W,
                       (int)H, (int)k, (int)nDisp, thr, best, shape);
    return hipGetLastError();
}

} /* namespace lfbm5d */
