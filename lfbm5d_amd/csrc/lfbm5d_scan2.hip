/*
 * lfbm5d_scan2.hip -- block-matching distance tables, second generation (round 3).
 *
 * Same arithmetic as lfbm5d_bm.hip (the reference's integral-image recurrence in the reference's association order,
 * precompute_BM core:3301-3461, precompute_BM_stereo core:3479-3611 -- bit-identical tables), different machine mapping:
 *
 *   - a WORKGROUP of eight wavefronts walks eight displacement tables of one image pair in lockstep (one barrier per
 *     chunk of eight steps).  The eight tables read the same image rows, only shifted by their displacement, so the
 *     workgroup keeps ONE ring of raw rows per image in LDS (round 2: one ring of squared differences per table:
 *     26.8 KiB per wave, six waves per CU) and every wave forms its squared differences on the fly (two packed
 *     subtract / multiply pairs per step);
 *   - the rings are stored TRANSPOSED, [column][row slot], with a column pitch = 1 (mod 4) rows: the four values a lane
 *     needs on four consecutive steps (one column, four consecutive rows) are one aligned ds_read_b128 -- one LDS
 *     instruction per step instead of four -- and the 16 lanes of an LDS lane group fall on 16 different bank quads.
 *     The 16-byte alignment must hold for every lane of every table at once, which it does for the first image by
 *     construction and for the second when (di + dj) mod 4 is the same for all tables of a workgroup: tables are grouped
 *     by that class, and the class sets the row phase of the second ring;
 *   - the two operands of the band's upper edge come from a 2K-register FIFO like round 2;
 *   - the four results of four steps leave as ONE 16-byte store per lane (table layout [strip][step / 4][lane][4]),
 *     the hand-off column to the next strip as one 16-byte store of the strip's last lane into a per-table scratch row
 *     in global memory, and comes back as uniform 16-byte loads -- no LDS traffic besides the ring reads;
 *   - the row loads are dealt to all 512 threads (at most one 16-byte load per thread per chunk) two chunks ahead.
 *
 * No FMA contraction in this file (it would change the rounding).
 */
#include "lfbm5d_kernels.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

#pragma clang fp contract(off)

namespace lfbm5d {

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef int v4i __attribute__((ext_vector_type(4)));

constexpr unsigned kRsrcFlags = 0x00020000u;
constexpr int kLeadBytes = 32;   /* the image resource starts 8 floats in front of the first plane (the estimate buffer has 64 of slack) */

template <int K> struct S2Geom {
    static constexpr int CW1 = 64 + K;                       /* ring 1: columns cb-1 .. cb+62+K */
    static constexpr int LAG = (63 - K + 7) / 8;             /* blocks of eight rows a chunk still reads behind its own */
    static constexpr int LEAD1 = (K + 7) / 8;                /* ... and ahead */
    static constexpr int RR1 = 8 * (LAG + LEAD1 + 2);        /* + the block being written */
    static constexpr int RRp1 = RR1 + 13;                    /* column pitch: 7 mirror rows, pitch = 5 (mod 8) */
    __host__ __device__ static int lead2(int rh) { return (7 + K + rh) / 8; }
    __host__ __device__ static int rr2(int rh) { return 8 * (LAG + lead2(rh) + 2); }
    __host__ __device__ static int cw2(int ch) { return (64 + K + ch + 3) & ~3; }
};

/* Reference-grid slot of a coordinate (utilities.cpp:697-712: nHW + i*p, plus a forced last index) */
__device__ __forceinline__ int s2_grid_index(int v, int n, int last, int nHW, int p) {
    if (v == last) return n - 1;
    const int d = v - nHW;
    if (d < 0 || d % p) return -1;
    const int i = d / p;
    return i < n - 1 ? i : -1;
}

__device__ __forceinline__ void lds_barrier() {
    /* LDS writes of this wave done, then the workgroup barrier; vector-memory operations stay in flight */
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <int K, bool STEREO>
__device__ __forceinline__ void scan2_body(const ScanArgs& a, const Scan2Wg& g, float* lds) {
    typedef S2Geom<K> G;
    constexpr int CW1 = G::CW1, RR1 = G::RR1, RRp1 = G::RRp1, LEAD1 = G::LEAD1;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);   /* wave-uniform: everything derived from the wave's table stays scalar */
    const int W = a.W, H = a.H;
    const int half = STEREO ? (int)a.nDisp : (int)a.nSim;
    const int b = STEREO ? (int)a.nDisp : (int)a.nHW;
    const int trim = STEREO ? K - 1 : 0;
    const int Ns = 2 * half + 1, ncand = Ns * Ns;
    const int nrows = H - 2 * b - trim, ncols = W - 2 * b - trim, band_rows = H - 2 * b;
    const size_t WH = (size_t)W * H;
    const int tab_w = g.tab[w];
    const bool live = tab_w >= 0;
    const int tb = live ? tab_w : g.tab[0];
    const int di = tb / Ns, dj = tb % Ns;
    const int dioff = STEREO ? di - half : di, djoff = dj - half;       /* second image: row / column offset */
    const int r2lo = g.r2lo, c2lo = g.c2lo;
    const int CW2 = G::cw2(g.ch), LEAD2 = G::lead2(g.rh), RR2 = G::rr2(g.rh), RRp2 = RR2 + 13;
    /* row phase of ring 2: (djoff - c2lo) + (dioff - r2lo) + r0 = 0 (mod 4), the same for every table of the workgroup */
    int r0;
    {
        const int t0 = g.tab[0], di0 = t0 / Ns, dj0 = t0 % Ns;
        r0 = (4 - (((dj0 - half) - c2lo + (STEREO ? di0 - half : di0) - r2lo) & 3)) & 3;
    }
    float* ring1 = lds;
    float* ring2 = lds + CW1 * RRp1;
    short* rs = reinterpret_cast<short*>(ring2 + CW2 * RRp2);   /* self search: reference-grid row slot of every image row (+ 64 of padding) */

    const unsigned pl1 = a.pst, pl2 = STEREO ? a.st_of_slot[g.slot] : a.pst;
    const float* img1 = a.est + (size_t)pl1 * WH;
    const float* img2 = a.est + (size_t)pl2 * WH;
    const int dk = dioff * W + djoff;
    /* one resource over all planes for the ring loads (a thread loads for either image), one per image for the pre-pass */
    const __amdgpu_buffer_rsrc_t rE = __builtin_amdgcn_make_buffer_rsrc((void*)(a.est - kLeadBytes / 4), 0, (int)(a.est_planes * WH * 4 + kLeadBytes + 1024), kRsrcFlags);
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc((void*)img1, 0, (int)(WH * 4 + 1024), kRsrcFlags);
    const __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc((void*)(img2 + dk), 0, (int)(WH * 4 + 1024), kRsrcFlags);
    /* outputs: waves without a table of their own (the last workgroup of a class) run along with stores that go nowhere */
    const int tglob = STEREO ? (int)a.n_self + g.slot * ncand + tb : tb;
    float* lcolT = a.lcol + (size_t)tglob * a.lcol_stride;
    const __amdgpu_buffer_rsrc_t rL = __builtin_amdgcn_make_buffer_rsrc((void*)lcolT, 0, live ? (int)(a.lcol_stride * 4) : 0, kRsrcFlags);
    const size_t tstride = STEREO ? stereo_table_stride2(a.W, a.H, a.k, a.nDisp) : 0;
    float* table = STEREO ? a.tables + (size_t)(g.slot * ncand + tb) * tstride : nullptr;
    const __amdgpu_buffer_rsrc_t rT = __builtin_amdgcn_make_buffer_rsrc((void*)table, 0, (STEREO && live) ? (int)(tstride * 4) : 0, kRsrcFlags);
    const __amdgpu_buffer_rsrc_t rS = __builtin_amdgcn_make_buffer_rsrc((void*)a.scores, 0, (!STEREO && live) ? (int)a.scores_bytes : 0, kRsrcFlags);
    const int SRq = (int)stereo_table_srq(a.H, a.k, a.nDisp);
    const int nstrips = (ncols - 1 + 63) / 64;      /* strips of 64 columns, starting at column 1 of the table */
    const int col0_off = nstrips * SRq * 256;       /* disparity tables: column 0 sits behind the strips */

    /* self search: where a table value goes in `scores` (core:3407-3420) */
    const int gR = a.n_ref_rows, gC = a.n_ref_cols, gP = a.p, gN = a.nHW;
    const int lastR = H - K - gN, lastC = W - K - gN;
    const int djs = dj - half;
    const int ord_fwd = dj * Ns + di;
    const int ord_bwd = (-djs + half) * Ns + (half + 1) + (half - di);
    const int row_bytes = gC * ncand * 4;

    /* ---- corner (core:3344-3352) and first column (core:3367-3372) -> the hand-off row of this table ---- */
    float corner = 0.0f;
    {
        float* scr = ring1 + w * 256;   /* K*K <= 256 floats of this wave's own */
        for (int e = lane; e < K * K; e += 64) {
            const int q = (b + e / K) * W + b + e % K;
            const float d = img2[q + dk] - img1[q];
            scr[e] = d * d;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (int e = 0; e < K * K; e++) corner += scr[e];
        /* column 0 of the table does not belong to a strip: it is the hand-off column of the first strip, and its values
         * go straight to their place (disparity search: a column area behind the strips; self search: `scores`) */
        auto emit0 = [&](int i, float v, bool on) {
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), rL, on ? (64 + i) * 4 : -1, 0, 0);
            if (STEREO) {
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), rT, on ? (col0_off + i) * 4 : -1, 0, 0);
            } else {
                const int cxa = s2_grid_index(b, gC, lastC, gN, gP);
                const int cxb = di > 0 ? s2_grid_index(b + djs, gC, lastC, gN, gP) : -1;
                const int yy = min(b + i, H - 1);
                const int ra = a.rslot[yy], rb = a.rslot[yy + di];
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), rS, (on && (ra | cxa) >= 0) ? ((ra * gC + cxa) * ncand + ord_fwd) * 4 : -1, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), rS, (on && (rb | cxb) >= 0) ? ((rb * gC + cxb) * ncand + ord_bwd) * 4 : -1, 0, 0);
            }
        };
        emit0(0, corner, lane == 0);
        /* First column: S[i][b] = S[i-1][b] + sum_q (D[i-1+K][b+q] - D[i-1][b+q]), the K terms added one after the other.
         * A lane owns a row. */
        constexpr int K4 = K / 4;
        auto load_e = [&](int i, v4f* lo1, v4f* lo2, v4f* hi1, v4f* hi2) {   /* row i-1 and row i-1+K of both images */
            const int ra = min(b + i - 1, H - 1), rb = min(b + i - 1 + K, H - 1);
#pragma unroll
            for (int j = 0; j < K4; j++) {
                lo1[j] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rs1, (ra * W + b + 4 * j) * 4, 0, 0));
                lo2[j] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rs2, (ra * W + b + 4 * j) * 4, 0, 0));
                hi1[j] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rs1, (rb * W + b + 4 * j) * 4, 0, 0));
                hi2[j] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rs2, (rb * W + b + 4 * j) * 4, 0, 0));
            }
        };
        float carry = corner;
        for (int i0 = 1; i0 < nrows; i0 += 64) {
            const int i = i0 + lane;
            v4f lo1[K4], lo2[K4], hi1[K4], hi2[K4];
            load_e(i, lo1, lo2, hi1, hi2);
            float e[K];
            const bool lo_in = b + i - 1 < H - b, hi_in = b + i - 1 + K < H - b;   /* rows outside [b, H-b) read as zero */
#pragma unroll
            for (int j = 0; j < K4; j++)
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const float dl = lo2[j][q] - lo1[j][q], dh = hi2[j][q] - hi1[j][q];
                    e[4 * j + q] = i < nrows ? (hi_in ? dh * dh : 0.0f) - (lo_in ? dl * dl : 0.0f) : 0.0f;
                }
            const int m = min(64, nrows - i0);
            float mine = 0.0f;
            for (int l = 0; l < m; l++) {
                float cand = carry;
#pragma unroll
                for (int q = 0; q < K; q++) cand += e[q];
                carry = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cand), l));
                if (lane == l) mine = carry;
            }
            emit0(i, mine, i < nrows);
        }
    }
    if (!STEREO)
        for (int i = tid; i < H + 64; i += blockDim.x) rs[i] = (short)a.rslot[i];
    __syncthreads();

    /* ---- ring loads: thread -> (image, row of the block, group of four columns) ---- */
    const int n1 = 2 * CW1, n2 = 2 * CW2;                 /* 16-byte pieces of a block of eight rows */
    const bool ld1 = tid < n1, ld2 = !ld1 && tid - n1 < n2;
    const bool loader = ld1 || ld2;
    const int item = ld1 ? tid : tid - n1;
    const int lr = item & 7, lquad = item >> 3;
    const int l_RR = ld1 ? RR1 : RR2, l_RRp = ld1 ? RRp1 : RRp2, l_lead = ld1 ? LEAD1 : LEAD2;
    const int l_y0 = (ld1 ? b : b + r2lo) + lr;                          /* image row of the item in block 0 */
    const int l_plane = (int)((ld1 ? pl1 : pl2) * WH) * 4 + kLeadBytes;
    float* const l_dst = (ld1 ? ring1 : ring2) + 4 * lquad * l_RRp;
    const int l_r0 = ld1 ? 0 : r0;

    float row0_left = corner;   /* S[b][cb-1] */
    for (int strip = 0; strip < nstrips; strip++) {
        const int cb = b + 1 + 64 * strip;
        const int x = cb + lane;
        const bool col_ok = x < b + ncols;
        const int last_lane = min(63, ncols - 2 - 64 * strip);
        const int lane_eff = col_ok ? lane : 0x40000000;   /* lanes past the last column are never active */
        /* squared differences outside the band [b, dim - b) are zeros (core:3335-3340); only the self search reads there:
         * columns through the operands at x + K - 1 of the last strip, rows in the ramp-down chunks */
        const int cm1 = (STEREO || x + K - 1 < W - b) ? -1 : 0;

        /* the hand-off column of the previous strip (first strip: the first column) must have landed */
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

        const int l_x = cb - 1 + (ld1 ? 0 : c2lo) + 4 * lquad;
        auto ring_load = [&](int j) -> v4f {
            const int y = min(max(l_y0 + 8 * j, 0), H - 1);
            return __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rE, loader ? l_plane + (y * W + l_x) * 4 : -1, 0, 0));
        };
        int wslot = (lr + l_r0) % l_RR;   /* row slot of this thread's piece in the next block to write */
        auto ring_write = [&](const v4f v) {
            if (loader) {
                const int slot = wslot;
                wslot += 8; wslot = wslot >= l_RR ? wslot - l_RR : wslot;
                float* d = l_dst + slot;
                d[0] = v[0]; d[l_RRp] = v[1]; d[2 * l_RRp] = v[2]; d[3 * l_RRp] = v[3];
                if (slot < 7) { d += l_RR; d[0] = v[0]; d[l_RRp] = v[1]; d[2 * l_RRp] = v[2]; d[3 * l_RRp] = v[3]; }
            }
        };
        v4f stg[2];
        {
            for (int j = 0; j <= l_lead; j++) ring_write(ring_load(j));
            stg[0] = ring_load(l_lead + 1);
            stg[1] = ring_load(l_lead + 2);
        }
        lds_barrier();

        /* squared difference at ring row rho (relative to the band's first row), ring-1 column c */
        auto Draw = [&](int rho, int c) -> float {
            const float i1 = ring1[c * RRp1 + rho];
            const float i2 = ring2[(c + djoff - c2lo) * RRp2 + rho + dioff - r2lo + r0];
            const float d = i2 - i1;
            return d * d;
        };

        /* ---- first row of the strip (core:3354-3362): chain across the lanes ---- */
        float S0 = 0.0f;
        {
            float e[K];
#pragma unroll
            for (int p = 0; p < K; p++) {
                float dr = Draw(p, lane + K);
                if (!STEREO) dr = __int_as_float(__float_as_int(dr) & cm1);
                e[p] = dr - Draw(p, lane);
            }
            float carry = row0_left;
            for (int l = 0; l <= last_lane; l++) {
                float cand = carry;
#pragma unroll
                for (int p = 0; p < K; p++) cand += e[p];
                carry = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cand), l));
                if (lane == l) S0 = carry;
            }
        }
        /* column slots of this lane in `scores` (self search): constant over the strip */
        const int cx = STEREO ? -1 : s2_grid_index(x, gC, lastC, gN, gP);
        const int cx2 = (STEREO || di == 0) ? -1 : s2_grid_index(x + djs, gC, lastC, gN, gP);
        const int base_fwd = (col_ok && cx >= 0) ? (cx * ncand + ord_fwd) * 4 : -1;
        const int base_bwd = (col_ok && cx2 >= 0 && di > 0) ? (cx2 * ncand + ord_bwd) * 4 : -1;
        /* row 0 of the table */
        if (STEREO) {
            /* lanes 1.. : through the main loop (a lane's "result" of the step before its first is its row-0 value) */
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, S0), rT, lane == 0 ? (strip * SRq * 256 + 3) * 4 : -1, 0, 0);
        } else {
            const int ry = s2_grid_index(b, gR, lastR, gN, gP);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, S0), rS, (ry >= 0 && base_fwd >= 0) ? ry * row_bytes + base_fwd : -1, 0, 0);
            const int ry2 = di > 0 ? s2_grid_index(b + di, gR, lastR, gN, gP) : -1;
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, S0), rS, (ry2 >= 0 && base_bwd >= 0) ? ry2 * row_bytes + base_bwd : -1, 0, 0);
        }

        /* ---- the remaining rows.  Lane l works on table row 1 + t - l at step t: active for t in [l, nrows-2+l].
         * Operands of the band's upper edge (rows t - l): the lower edge's of K steps ago, from a register FIFO. ---- */
        float F1[K], F2[K];
#pragma unroll
        for (int m = 0; m < K; m++) {
            const int row = max(m - lane, 0);
            float dr = Draw(row, lane + K);
            if (!STEREO) dr = __int_as_float(__float_as_int(dr) & cm1);
            F1[m] = dr; F2[m] = Draw(row, lane);
        }
        float curS = S0;
        float left_prev = row0_left;    /* lane 0: S[0][cb-1]; other lanes: overwritten before use */
        /* ring positions of step 0: rows K - lane of both images, in row slots; advanced by eight per chunk */
        unsigned slot1 = (unsigned)((K - lane + RR1) % RR1);
        unsigned slot2 = (unsigned)(((K - lane + dioff - r2lo + r0) % RR2 + RR2) % RR2);
        const float* const colA = ring1 + (lane + K) * RRp1;
        const float* const colB = ring2 + (lane + K + djoff - c2lo) * RRp2;
        const int KB2 = K * RRp2;
        /* outputs: running byte offsets.  Table: [strip][step / 4 + 1][lane][4]; hand-off: the strip's last lane, rows 1 + t - 63 */
        int voffT = ((strip * SRq + 1) * 64 + lane) * 16;
        int voffL = lane == 63 ? (64 + 1 - 63) * 4 : 0x70000000;
        /* hand-off column values of steps 0..15 (rows 1..16 of the column left of the strip) */
        v4f lcr[2][2];
#pragma unroll
        for (int c = 0; c < 2; c++)
#pragma unroll
            for (int h = 0; h < 2; h++)
                lcr[c][h] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rL, 0, (65 + 8 * c + 4 * h) * 4, 0));
        int jw = 0;   /* chunk counter of the loader */

        /* FL 0: steady -- every lane active on every step, every row in the band; FL 1: edge (ramp-up, ramp-down, short tables) */
        auto body16 = [&](auto fl_tag, const int t0) {
            constexpr bool EDGE = decltype(fl_tag)::value;
            /* per-lane step numbers relative to this group of sixteen (compared with small constants below) */
            const int rel_start = lane_eff - t0;                 /* the lane's first step */
            const int rel_band = band_rows - K + lane - t0;      /* first step whose lower-edge row lies past the band */
#pragma unroll
            for (int ch = 0; ch < 2; ch++) {
                const int tc = t0 + 8 * ch;
                /* rows for the next chunk go into the ring, the loads for the one after the next start */
                ring_write(stg[ch]);
                stg[ch] = ring_load(jw + l_lead + 3);
                jw++;
                const v4f lc[2] = {lcr[ch][0], lcr[ch][1]};
                lcr[ch][0] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rL, 0, (65 + tc + 16) * 4, 0));
                lcr[ch][1] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rL, 0, (65 + tc + 20) * 4, 0));
                const float* pA = colA + slot1;
                const float* pB = colB + slot2;
#pragma unroll
                for (int gq = 0; gq < 2; gq++) {
                    const int tg = tc + 4 * gq;
                    const v4f a1 = *reinterpret_cast<const v4f*>(pA + 4 * gq);
                    const v4f a2 = *reinterpret_cast<const v4f*>(pA - K * RRp1 + 4 * gq);
                    const v4f b1 = *reinterpret_cast<const v4f*>(pB + 4 * gq);
                    const v4f b2 = *reinterpret_cast<const v4f*>(pB - KB2 + 4 * gq);
                    v4f e1 = b1 - a1, e2 = b2 - a2;
                    e1 = e1 * e1; e2 = e2 * e2;
                    if (!STEREO) {
#pragma unroll
                        for (int s = 0; s < 4; s++) {
                            int m = cm1;
                            if (EDGE) m = (8 * ch + 4 * gq + s < rel_band) ? m : 0;
                            e1[s] = __int_as_float(__float_as_int(e1[s]) & m);
                            if (EDGE) e2[s] = (8 * ch + 4 * gq + s < rel_band) ? e2[s] : 0.0f;
                        }
                    }
                    v4f out;
#pragma unroll
                    for (int s = 0; s < 4; s++) {
                        const int fs = (8 * ch + 4 * gq + s) % K;
                        /* left neighbour's value of the previous step; lane 0 takes the hand-off column */
                        const float left = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(lc[gq][s]), __float_as_int(curS),
                                                                                      0x138 /* wave_shr:1 */, 0xf, 0xf, false));
                        float S = left + curS;             /* core:3379-3386, same association */
                        S = S - left_prev;
                        S = S + e1[s];
                        S = S - e2[s];
                        S = S - F1[fs];
                        S = S + F2[fs];
                        F1[fs] = e1[s]; F2[fs] = e2[s];
                        if (EDGE) S = (rel_start == 8 * ch + 4 * gq + s + 1) ? S0 : S; /* the step before a lane's first leaves its row-0 value */
                        out[s] = S;
                        curS = S;
                        left_prev = left;
                    }
                    if (STEREO) {
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4i, out), rT, voffT, 0, 0);
                        voffT += 1024;
                    } else {
#pragma unroll
                        for (int s = 0; s < 4; s++) {
                            const int t = tg + s;
                            const bool act = (unsigned)(t - lane_eff) <= (unsigned)(nrows - 2);
                            const int y = min(max(b + 1 + t - lane, 0), H - 1);
                            const int r1 = rs[y], r2 = rs[y + di];                  /* -1: not a grid row */
                            const int v1 = (act && (r1 | base_fwd) >= 0) ? r1 * row_bytes + base_fwd : -1;
                            const int v2 = (act && (r2 | base_bwd) >= 0) ? r2 * row_bytes + base_bwd : -1;
                            /* a store no lane takes part in is skipped (on the regular grid three steps in four) */
                            const float ov = out[s];   /* (a bit_cast of the vector element itself picks element 0) */
                            if (__builtin_amdgcn_ballot_w64(v1 != -1)) __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(ov), rS, v1, 0, 0);
                            if (__builtin_amdgcn_ballot_w64(v2 != -1)) __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(ov), rS, v2, 0, 0);
                        }
                    }
                    /* hand-off column for the next strip: rows 1 + t - 63 of this strip's last column */
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4i, out), rL, voffL, 0, 0);
                    voffL += 16;
                }
                slot1 += 8; slot1 = min(slot1, slot1 - (unsigned)RR1);
                slot2 += 8; slot2 = min(slot2, slot2 - (unsigned)RR2);
                lds_barrier();
            }
        };

        const int nsteps = (nrows - 1) + last_lane;
        const int tS1 = (min(nrows - 1, band_rows - K) / 16) * 16;   /* steady chunks end before lane 0 stops / the band ends */
        {
            int t0 = 0;
            for (; t0 < nsteps && t0 < 64; t0 += 16) body16(std::true_type{}, t0);
            for (; t0 < tS1; t0 += 16) body16(std::false_type{}, t0);
            for (; t0 < nsteps; t0 += 16) body16(std::true_type{}, t0);
        }
        row0_left = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(S0), last_lane));
    }
}

template <int K>
__global__ __launch_bounds__(512) void k_bm_scan2(ScanArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds2[];
    const Scan2Wg& g = a.wgs[blockIdx.x];
#ifdef LFBM5D_SCAN2_ONLY_STEREO
    scan2_body<K, true>(a, g, lds2);
#elif defined(LFBM5D_SCAN2_ONLY_SELF)
    scan2_body<K, false>(a, g, lds2);
#else
    if (g.slot >= 0) scan2_body<K, true>(a, g, lds2);
    else scan2_body<K, false>(a, g, lds2);
#endif
}

/* argmin over the (2 nDisp+1)^2 displacement tables (core:3581-3608) in the second-generation layout
 * [strip][Q / 4][lane][Q % 4], Q = table row + lane + 3: a thread takes the four entries of one lane (one 16-byte load per
 * table) = four consecutive rows of one column; ties keep scan order (dj outer, di inner). */
struct Argmin2Args { const float* tables; size_t tstride; unsigned st_of_slot[kMaxA]; int W, H, k, nDisp, SRq; float thr; unsigned* best; unsigned char* shape; };
__global__ __launch_bounds__(256) void k_stereo_argmin2(Argmin2Args a) {
    const int W = a.W, H = a.H, nDisp = a.nDisp;
    const int span_c = W - 2 * nDisp - a.k + 1, span_r = H - 2 * nDisp - a.k + 1;
    const int nstrips = (span_c - 1 + 63) / 64;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int n_strip_items = nstrips * a.SRq * 64;
    if (i >= n_strip_items + a.SRq) return;
    const unsigned slot = blockIdx.y, st = a.st_of_slot[slot];
    int r0, col;
    if (i < n_strip_items) {
        const int l = i & 63, qg = (i >> 6) % a.SRq, strip = (i >> 6) / a.SRq;
        r0 = 4 * qg - l - 3; col = 1 + 64 * strip + l;
    } else { r0 = 4 * (i - n_strip_items); col = 0; }     /* column 0: four consecutive rows per thread */
    if (r0 + 3 < 0 || r0 >= span_r || col >= span_c) return;
    const int Ns = 2 * nDisp + 1, ncand = Ns * Ns;
    const size_t WH = (size_t)W * H;
    const float* t = a.tables + (size_t)slot * ncand * a.tstride + (size_t)i * 4;
    float bv[4]; int bo[4], bd[4];
    {
        const v4f v0 = *reinterpret_cast<const v4f*>(t);
#pragma unroll
        for (int e = 0; e < 4; e++) { bv[e] = v0[e]; bo[e] = 0; bd[e] = 0; }
    }
    for (int d0 = 0; d0 < ncand; d0 += 8) {
        v4f v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = *reinterpret_cast<const v4f*>(t + (size_t)min(d0 + u, ncand - 1) * a.tstride);
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int ddk = d0 + u;
            if (ddk < ncand) {
                const int di = ddk / Ns, dj = ddk - di * Ns;
                const int order = dj * Ns + di;
#pragma unroll
                for (int e = 0; e < 4; e++)
                    if (v[u][e] < bv[e] || (v[u][e] == bv[e] && order < bo[e])) { bv[e] = v[u][e]; bo[e] = order; bd[e] = ddk; }
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 4; e++) {
        const int r = r0 + e;
        if (r < 0 || r >= span_r) continue;            /* entries never written hold garbage */
        const int pos = (nDisp + r) * W + nDisp + col;
        const int di = bd[e] / Ns, dj = bd[e] % Ns;
        a.best[(size_t)st * WH + pos] = (unsigned)(pos + (di - nDisp) * W + (dj - nDisp));
        a.shape[(size_t)st * WH + pos] = bv[e] < a.thr ? 1 : 0;
    }
}

template <int K> size_t scan2_lds_bytes(const ScanArgs& a, int rh_max, int ch_max) {
    typedef S2Geom<K> G;
    const int RRp2 = G::rr2(rh_max) + 13;
    return (size_t)(G::CW1 * G::RRp1 + G::cw2(ch_max) * RRp2) * sizeof(float) + (a.n_self ? (size_t)(a.H + 64) * sizeof(short) : 0) + 16;
}

} /* namespace */

/* Which kernel generation a pass's distance tables are built with (and hence the layout the arg-min reads):
 * 2 unless the configuration is outside what the ring-sharing kernel covers (12x12 patches, irregular reference lists,
 * search windows whose rings do not fit) or LFBM5D_SCAN_V1 asks for round 2's kernel. */
int bm_scan_version(const ScanArgs& a) {
    if (const char* e = std::getenv("LFBM5D_SCAN_V1")) if (e[0] && e[0] != '0') return 1;
    if (a.k != 8 && a.k != 16) return 1;
    if (a.n_self && a.refmap) return 1;
    std::vector<Scan2Wg> wgs; size_t lds = 0;
    if (!scan2_plan(a, wgs, &lds)) return 1;
    return 2;
}

/* Workgroups of a launch: up to eight tables of one image pair and one class (di + dj) mod 4. */
bool scan2_plan(const ScanArgs& a, std::vector<Scan2Wg>& wgs, size_t* lds_bytes) {
    wgs.clear();
    int rh_max = 0, ch_max = 0;
    auto add = [&](int slot, const std::vector<int>& tabs, int r2lo, int c2lo, int rh, int ch) {
        for (size_t i = 0; i < tabs.size(); i += 8) {
            Scan2Wg g;
            std::memset(&g, 0, sizeof(g));
            for (int w = 0; w < 8; w++) g.tab[w] = i + w < tabs.size() ? (short)tabs[i + w] : (short)-1;
            g.slot = (short)slot; g.r2lo = (short)r2lo; g.c2lo = (short)c2lo; g.rh = (short)rh; g.ch = (short)ch;
            wgs.push_back(g);
        }
        rh_max = std::max(rh_max, rh); ch_max = std::max(ch_max, ch);
    };
    if (a.n_self) {
        /* self search: di in [0, nSim], dj in [0, 2 nSim]; tiles of 4 x 8 displacements, one workgroup per class of a tile
         * (eight tables): the second ring then spans 3 more rows and 7 more columns than the first */
        const int nSim = (int)a.nSim, Ns = 2 * nSim + 1;
        for (int d0 = 0; d0 <= nSim; d0 += 4)
            for (int j0 = 0; j0 < Ns; j0 += 8)
                for (int cls = 0; cls < 4; cls++) {
                    std::vector<int> tabs;
                    for (int di = d0; di < std::min(d0 + 4, nSim + 1); di++)
                        for (int dj = j0; dj < std::min(j0 + 8, Ns); dj++)
                            if (((di + dj) & 3) == cls) tabs.push_back(di * Ns + dj);
                    if (!tabs.empty()) add(-1, tabs, d0, j0 - nSim, 3, 7);
                }
    }
    if (a.n_stereo) {
        const int nD = (int)a.nDisp, Ns = 2 * nD + 1, ncand = Ns * Ns;
        const int n_slots = (int)a.n_stereo / ncand;
        for (int slot = 0; slot < n_slots; slot++)
            for (int cls = 0; cls < 4; cls++) {
                std::vector<int> tabs;
                for (int ddk = 0; ddk < ncand; ddk++)
                    if ((((ddk / Ns) + (ddk % Ns)) & 3) == cls) tabs.push_back(ddk);
                if (!tabs.empty()) add(slot, tabs, -nD, -nD, 2 * nD, 2 * nD);
            }
    }
    size_t lds = a.k == 8 ? scan2_lds_bytes<8>(a, rh_max, ch_max) : scan2_lds_bytes<16>(a, rh_max, ch_max);
    if (lds_bytes) *lds_bytes = lds;
    /* one 16-byte piece per thread and chunk; the rings within the CU's LDS */
    const int n1 = 2 * (64 + (int)a.k), n2 = 2 * ((64 + (int)a.k + ch_max + 3) & ~3);
    if (n1 + n2 > 512 || lds > 160 * 1024) return false;
    return !wgs.empty();
}

unsigned scan2_lcol_stride(const ScanArgs& a) {
    const unsigned rows_self = a.n_self ? a.H - 2 * a.nHW : 0, rows_st = a.n_stereo ? a.H - 2 * a.nDisp - (a.k - 1) : 0;
    return ((std::max(rows_self, rows_st) + 64 + 160 + 63) / 64) * 64;
}

hipError_t launch_bm_scan2(hipStream_t s, const ScanArgs& a, size_t lds) {
    if (!a.n_wgs) return hipSuccess;
    static bool prepared = false;
    if (!prepared) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bm_scan2<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bm_scan2<16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        prepared = true;
    }
    if (a.k == 8) hipLaunchKernelGGL((k_bm_scan2<8>), dim3(a.n_wgs), dim3(512), lds, s, a);
    else if (a.k == 16) hipLaunchKernelGGL((k_bm_scan2<16>), dim3(a.n_wgs), dim3(512), lds, s, a);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t launch_stereo_argmin2(hipStream_t s, const float* tables, const unsigned* st_of_slot, unsigned n_slots,
                                 unsigned W, unsigned H, unsigned k, unsigned nDisp, float thr,
                                 unsigned* best, unsigned char* shape) {
    const unsigned span_c = W - 2 * nDisp - k + 1;
    Argmin2Args a;
    a.tables = tables; a.tstride = stereo_table_stride2(W, H, k, nDisp); a.W = (int)W; a.H = (int)H; a.k = (int)k; a.nDisp = (int)nDisp; a.thr = thr; a.best = best; a.shape = shape;
    a.SRq = (int)stereo_table_srq(H, k, nDisp);
    const unsigned n = ((span_c - 1 + 63) / 64) * a.SRq * 64 + a.SRq;
    for (unsigned i = 0; i < n_slots && i < (unsigned)kMaxA; i++) a.st_of_slot[i] = st_of_slot[i];
    hipLaunchKernelGGL(k_stereo_argmin2, dim3((n + 255) / 256, n_slots), dim3(256), 0, s, a);
    return hipGetLastError();
}

} /* namespace lfbm5d */
