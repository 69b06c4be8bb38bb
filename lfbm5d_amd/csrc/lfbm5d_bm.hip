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
 * Bound: one wave per SIMD (LDS), about 45 instructions per step (~(cols+64)*rows/64 steps per table) of which
 * 58 % of the wave's cycles are issue and a third waits; plus the HBM write of the disparity tables.  Every
 * table is an independent wave: a pass keeps ~2000 of them in two rounds of four per CU.
 */
#include "lfbm5d_kernels.h"

#include <type_traits>
#include <algorithm>
#include <cstdlib>

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

/* Reference-grid slot of a coordinate (utilities.cpp:697-712: nHW + i*p, plus a forced last index) */
__device__ __forceinline__ int grid_index(int v, int n, int last, int nHW, int p) {
    if (v == last) return n - 1;
    const int d = v - nHW;
    if (d < 0 || d % p) return -1;
    const int i = d / p;
    return i < n - 1 ? i : -1;
}

/*
 * One wavefront per displacement table.  Lanes own 64 consecutive COLUMNS (a strip) and walk down
 * the rows one step behind their left neighbour, so that
 *   left    = the neighbour's value of the previous step (DPP wave shift),
 *   upleft  = the neighbour's value of two steps ago,
 *   up      = the lane's own previous value,
 * and every global load is a coalesced row segment: the squared-difference image D of the strip is
 * streamed through an LDS ring (RR rows x 64+K columns) that also provides the skew.  The first
 * column of the table (a K-term chain per row) is computed up front; strips hand their last
 * column to the next strip through `lcol`.
 */
#ifndef LFBM5D_SCAN_DEPTH
#define LFBM5D_SCAN_DEPTH(T) 4   /* with two waves per SIMD a shallower pipeline suffices, and it keeps the kernel under 256 VGPRs */
#endif
typedef float v4f __attribute__((ext_vector_type(4)));
template <int K, int MODE, bool WIDE>   /* MODE 0: self search on the regular grid, 1: self search on an irregular list, 2: disparity */
__device__ __forceinline__ void scan_body(const ScanArgs& a, const int bid, float* lds) {
    /* T steps per chunk; the ring holds 63 (skew) + K + 2T + 1 rows (+ T-1 mirror rows).  With the hand-off column,
     * the row-slot table and the straightening buffer a wave takes just under 40 KiB at 560-wide windows: four
     * waves per CU, one per SIMD */
    /* FIFO: the operands of the band's upper edge (rows t - lane) are the lower edge's of K steps ago, kept in 2K registers
     * instead of K more ring rows -- half the ring reads and a ring of 64 + 2T rows (the pipeline depth must be a multiple of K) */
    constexpr bool FIFO = K == 8 || K == 16;
    constexpr int T = 4, RR = FIFO ? 64 + 2 * T : 64 + K + 2 * T, CW = 64 + K;
    constexpr int DEP = LFBM5D_SCAN_DEPTH(T);   /* row-load pipeline depth in chunks */
    constexpr bool stereo = MODE == 2;
    constexpr bool irregular = MODE == 1;
    float* ring = lds;              /* [RR][CW] D rows of the current strip; ring col 0 <-> x = cb-1 */
    /* rows RR .. RR+T-2 mirror rows 0 .. T-2, so T consecutive ring rows can be read without wrapping */
    float* lcol = lds + (RR + T - 1) * CW;   /* [nrows + T + 1] column left of the current strip (strip 0: first column) */
    const int lane = threadIdx.x;
    /* A chunk's T new D rows (T x CW entries, row-major) are fetched with as few memory instructions as
     * possible -- vmcnt is an in-order counter of at most 63 loads AND stores, so the fewer operations a
     * chunk issues, the further ahead of the (slow) table stores the loads can run: NA 16-byte loads of
     * four consecutive entries per lane (CW is a multiple of 4, a quad never straddles rows) and one
     * 4-byte load for the REM <= 64 entries left. */
    constexpr int E = T * CW, NA = E / 256, REM = E - 256 * NA;
    static_assert(CW % 4 == 0 && REM >= 0 && REM <= 64 && RR % T == 0 && (K + T) % T == 0, "chunk geometry");
    static_assert(!FIFO || (DEP * T) % K == 0, "the register FIFO rotates once per group of chunks");
    int qrow[NA], qcol[NA];
#pragma unroll
    for (int q = 0; q < NA; q++) { qrow[q] = (256 * q + 4 * lane) / CW; qcol[q] = (256 * q + 4 * lane) % CW; }
    const int xrow = (256 * NA + lane) / CW, xcol = (256 * NA + lane) % CW;
    const bool hasB = lane < REM;
    const int half = stereo ? (int)a.nDisp : (int)a.nSim;
    const int trim = stereo ? (int)a.k - 1 : 0;
    const int W = a.W, H = a.H, b = stereo ? (int)a.nDisp : (int)a.nHW;
    const int Ns = 2 * half + 1;
    const int ncand = Ns * Ns;
    const int nrows = H - 2 * b - trim, ncols = W - 2 * b - trim;

    int di, dj;
    DiffImg D;
    const size_t WH = (size_t)W * H;
    if (stereo) {
        const int slot = bid / ncand, ddk = bid % ncand;
        di = ddk / Ns; dj = ddk % Ns;
        D.i1 = a.est + (size_t)a.pst * WH;
        D.i2 = a.est + (size_t)a.st_of_slot[slot] * WH;
        D.dk = di * W + dj - half * (1 + W);
    } else {
        di = bid / Ns; dj = bid % Ns;
        D.i1 = D.i2 = a.est + (size_t)a.pst * WH;
        D.dk = di * W + dj - half;
    }
    D.W = W; D.H = H; D.b = b;
    const size_t tstride = stereo ? stereo_table_stride(a.W, a.H, a.k, a.nDisp) : 0;
    float* table = stereo ? a.tables + (size_t)bid * tstride : nullptr;   /* [strip][row + lane][64]: skewed, see the store stage */
    const int SR = nrows + 63;   /* rows of a strip in the skewed layout */
    const int djs = dj - half;
    const int nSim = half;
    const int ord_fwd = dj * Ns + di;
    const int ord_bwd = (-djs + nSim) * Ns + (nSim + 1) + (nSim - di);
    /* buffer resources: 32-bit offsets (scalar row base + per-lane column offset) instead of 64-bit
     * flat addressing, and stores that are dropped by setting the lane offset out of range */
    const unsigned kRsrcFlags = 0x00020000u;
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc((void*)D.i1, 0, (int)(WH * 4 + 1024), kRsrcFlags);
    const __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc((void*)(D.i2 + D.dk), 0, (int)(WH * 4 + 1024), kRsrcFlags);
    const __amdgpu_buffer_rsrc_t rsS = __builtin_amdgcn_make_buffer_rsrc((void*)a.scores, 0, (int)a.scores_bytes, kRsrcFlags);
    const __amdgpu_buffer_rsrc_t rsM = __builtin_amdgcn_make_buffer_rsrc((void*)a.refmap, 0, irregular ? (int)(WH * 4) : 0, kRsrcFlags);
    const int gR = a.n_ref_rows, gC = a.n_ref_cols, gP = a.p, gN = a.nHW;
    const int lastR = H - (int)a.k - gN, lastC = W - (int)a.k - gN;

    auto emit = [&](int y, int x, float S) {
        if (stereo) { table[((size_t)((x - b) / 64) * SR + (y - b) + (x - b) % 64) * 64 + (x - b) % 64] = S; return; }
        if (irregular) { /* irregular reference list (subset path, core:3631-3788): slots come from a position map */
            const int r = a.refmap[y * W + x];
            if (r >= 0) a.scores[(size_t)r * ncand + ord_fwd] = S;
            const int yy = y + di, xx = x + djs;
            if (di > 0 && yy < H && xx >= 0 && xx < W) {
                const int r2 = a.refmap[yy * W + xx];
                if (r2 >= 0) a.scores[(size_t)r2 * ncand + ord_bwd] = S;
            }
            return;
        }
        /* forward candidate of the reference patch at (y,x) (core:3410-3413) */
        const int cx = grid_index(x, gC, lastC, gN, gP);
        if (cx >= 0) {
            const int ry = grid_index(y, gR, lastR, gN, gP);
            if (ry >= 0) a.scores[(size_t)(ry * gC + cx) * ncand + ord_fwd] = S;
        }
        if (di > 0) { /* ... and the score of the mirrored candidate of the reference at (y,x)+d (core:3416-3419) */
            const int cx2 = grid_index(x + djs, gC, lastC, gN, gP);
            if (cx2 >= 0) {
                const int ry2 = grid_index(y + di, gR, lastR, gN, gP);
                if (ry2 >= 0) a.scores[(size_t)(ry2 * gC + cx2) * ncand + ord_bwd] = S;
            }
        }
    };

#ifdef LFBM5D_PHASE_TIMING
    long long tk[6] = {0, 0, 0, 0, 0, 0};
    long long tlast = (long long)__builtin_readcyclecounter();
#define SCAN_MARK(i) do { const long long tn = (long long)__builtin_readcyclecounter(); tk[i] += tn - tlast; tlast = tn; } while (0)
#else
#define SCAN_MARK(i) do {} while (0)
#endif
    /* ---- corner (core:3344-3352) and first column (core:3367-3372) -> lcol ---- */
    for (int e = lane; e < K * K; e += 64) ring[e] = D(b + e / K, b + e % K);
    __syncthreads();
    float corner = 0.0f;
    for (int e = 0; e < K * K; e++) corner += ring[e];
    __syncthreads();
    if (lane == 0) lcol[0] = corner;
    {
        /* First column (core:3367-3372): S[i][b] = S[i-1][b] + sum_q (D[i-1+K][b+q] - D[i-1][b+q]), the K terms added
         * one after the other.  A lane owns a row: its K differences come from four 16-byte loads per image row (the
         * columns b .. b+K-1 are always inside the band; rows past it are zeros), the loads of the next 64 rows are in
         * flight while the serial chain of the current 64 runs. */
        constexpr int K4 = K / 4;
        auto load_e = [&](int i, v4f* lo1, v4f* lo2, v4f* hi1, v4f* hi2) {   /* row i-1 and row i-1+K of both images */
            const int r0 = min(b + i - 1, H - 1), r1 = min(b + i - 1 + K, H - 1);
#pragma unroll
            for (int j = 0; j < K4; j++) {
                lo1[j] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rs1, (r0 * W + b + 4 * j) * 4, 0, 0));
                lo2[j] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rs2, (r0 * W + b + 4 * j) * 4, 0, 0));
                hi1[j] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rs1, (r1 * W + b + 4 * j) * 4, 0, 0));
                hi2[j] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rs2, (r1 * W + b + 4 * j) * 4, 0, 0));
            }
        };
        v4f lo1[K4], lo2[K4], hi1[K4], hi2[K4], nlo1[K4], nlo2[K4], nhi1[K4], nhi2[K4];
        load_e(1 + lane, lo1, lo2, hi1, hi2);
        float carry = corner;
        for (int i0 = 1; i0 < nrows; i0 += 64) {
            const int i = i0 + lane;
            if (i0 + 64 < nrows) load_e(i + 64, nlo1, nlo2, nhi1, nhi2);
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
            if (i < nrows) lcol[i] = mine;
#pragma unroll
            for (int j = 0; j < K4; j++) { lo1[j] = nlo1[j]; lo2[j] = nlo2[j]; hi1[j] = nhi1[j]; hi2[j] = nhi2[j]; }
        }
    }
    __syncthreads();

    for (int i = nrows + lane; i < nrows + T + 1; i += 64) lcol[i] = 0.0f;
    /* regular grid: row -> reference-grid row slot (-1: none) as 16-bit entries in LDS (a global lookup per
     * step would make every step wait on vmcnt, i.e. on all the loads and stores in flight); like the global
     * table it has 64 entries of -1 padding behind row H-1 for the rows y + di of the mirrored candidate */
    typedef typename std::conditional<WIDE, short, signed char>::type slot_t;   /* WIDE: more than 127 reference rows */
    slot_t* rs16 = reinterpret_cast<slot_t*>(lcol + nrows + T + 1);
    if (MODE == 0)
        for (int i = lane; i < H + 64; i += 64) rs16[i] = (slot_t)a.rslot[i];
    __syncthreads();

    SCAN_MARK(0);
    float row0_left = corner; /* S[b][cb-1] */
    const int nstrips = (ncols + 63) / 64;
    for (int strip = 0; strip < nstrips; strip++) {
        const int cb = b + 64 * strip;
        const int x = cb + lane;
        const bool col_ok = x < b + ncols;
        const int last_lane = min(63, ncols - 1 - 64 * strip);
        const bool first_col = strip == 0 && lane == 0;
        /* column slots of this lane: constant over the strip */
        const int cx = stereo ? -1 : grid_index(x, gC, lastC, gN, gP);
        const int cx2 = (stereo || di == 0) ? -1 : grid_index(x + djs, gC, lastC, gN, gP);
        /* byte offset of (grid column of this lane, this table's candidate) inside a grid row of `scores`; negative:
         * this lane never stores.  A slot's offset is then row_slot * row_bytes + base (one multiply-add per step) */
        const int base_fwd = (col_ok && cx >= 0) ? (cx * ncand + ord_fwd) * 4 : -1;
        const int base_bwd = (col_ok && cx2 >= 0 && di > 0) ? (cx2 * ncand + ord_bwd) * 4 : -1;
        const int row_bytes = gC * ncand * 4;

        /* D rows of the strip: ring column 0 <-> x = cb-1 */
        /* per-lane pieces of the chunk loads: byte offset inside the chunk's first row, in-band column masks */
        int vA[NA], mA[NA];
#pragma unroll
        for (int q = 0; q < NA; q++) {
            vA[q] = (qrow[q] * W + cb - 1 + qcol[q]) * 4;
            mA[q] = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) { const int xx = cb - 1 + qcol[q] + j; mA[q] |= (xx >= b && xx < W - b) ? 1 << j : 0; }
        }
        const int vB = (xrow * W + cb - 1 + xcol) * 4;
        const bool mB = hasB && cb - 1 + xcol >= b && cb - 1 + xcol < W - b;
        /* chunk loads.  steady: one scalar row offset; edge: every lane clamps its own row into the image
         * (rows past the band are zeroed when they are written to the ring) */
        auto load_chunk = [&](auto edge_tag, int R0, v4f* a1, v4f* a2, float& b1, float& b2) {
            constexpr bool EDGE = decltype(edge_tag)::value;
#pragma unroll
            for (int q = 0; q < NA; q++) {
                const int vo = EDGE ? (min(b + R0 + qrow[q], H - 1) * W + cb - 1 + qcol[q]) * 4 : vA[q];
                const int so = EDGE ? 0 : (b + R0) * W * 4;
                a1[q] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rs1, vo, so, 0));
                a2[q] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rs2, vo, so, 0));
            }
            if (REM > 0) {
                const int vo = hasB ? (EDGE ? (min(b + R0 + xrow, H - 1) * W + cb - 1 + xcol) * 4 : vB) : -1;   /* -1: reads 0 */
                const int so = EDGE ? 0 : (b + R0) * W * 4;
                b1 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs1, vo, so, 0));
                b2 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs2, vo, so, 0));
            }
        };
        /* squared differences of T rows (first row R0, a multiple of T) into ring rows wr .. wr+T-1 (wr a multiple
         * of T like RR: no wrap inside a chunk) and into the mirror rows behind the ring when wr == 0 */
        auto write_chunk = [&](auto edge_tag, int R0, int wr, const v4f* sa1, const v4f* sa2, float sb1, float sb2) {
            constexpr bool EDGE = decltype(edge_tag)::value;
            float* rr = ring + wr * CW;       /* uniform */
#pragma unroll
            for (int q = 0; q < NA; q++) {
                const v4f d = sa2[q] - sa1[q];
                v4f v = d * d;
                const bool rin = !EDGE || b + R0 + qrow[q] < H - b;   /* rows past the band are zeros */
#pragma unroll
                for (int j = 0; j < 4; j++) v[j] = (rin && ((mA[q] >> j) & 1)) ? v[j] : 0.0f;
                *reinterpret_cast<v4f*>(rr + qrow[q] * CW + qcol[q]) = v;
                if (wr == 0 && qrow[q] < T - 1) *reinterpret_cast<v4f*>(rr + (RR + qrow[q]) * CW + qcol[q]) = v;   /* mirror rows */
            }
            if (REM > 0) {
                const float d = sb2 - sb1;
                const bool rin = !EDGE || b + R0 + xrow < H - b;
                const float v = (rin && mB) ? d * d : 0.0f;
                if (hasB) {
                    rr[xrow * CW + xcol] = v;
                    if (wr == 0 && xrow < T - 1) rr[(RR + xrow) * CW + xcol] = v;
                }
            }
        };
        /* rows 0 .. K+T-1 go straight to the ring, the next (DEP-1)*T rows wait in registers (row loads run
         * DEP-1 chunks ahead of the ring writes); all of these loads are in flight together */
        constexpr int NPRE = (K + T) / T;
        v4f A1[DEP][NA], A2[DEP][NA];
        float Bq1[DEP], Bq2[DEP];
        {
            v4f P1[NPRE][NA], P2[NPRE][NA];
            float Pb1[NPRE], Pb2[NPRE];
#pragma unroll
            for (int j = 0; j < NPRE; j++) load_chunk(std::true_type{}, j * T, P1[j], P2[j], Pb1[j], Pb2[j]);
#pragma unroll
            for (int j = 0; j < DEP - 1; j++) load_chunk(std::true_type{}, K + T + j * T, A1[j], A2[j], Bq1[j], Bq2[j]);
#pragma unroll
            for (int j = 0; j < NPRE; j++) write_chunk(std::true_type{}, j * T, j * T, P1[j], P2[j], Pb1[j], Pb2[j]);
        }
        SCAN_MARK(1);
        int filled = K + T;   /* rows [0, filled) are in the ring; rows [filled, filled + (DEP-1)T) wait in registers */
        __syncthreads();

        /* first row of the strip (core:3354-3362): chain across the lanes */
        float S0 = 0.0f;
        {
            float e[K];
#pragma unroll
            for (int p = 0; p < K; p++) e[p] = ring[p * CW + lane + K] - ring[p * CW + lane];
            float carry = row0_left;
            int l0 = 0;
            if (strip == 0) { if (lane == 0) S0 = corner; l0 = 1; }
            for (int l = l0; l <= last_lane; l++) {
                float cand = carry;
#pragma unroll
                for (int p = 0; p < K; p++) cand += e[p];
                carry = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cand), l));
                if (lane == l) S0 = carry;
            }
            if (col_ok) emit(b, x, S0);
        }

        SCAN_MARK(2);
        /* remaining rows: T steps per chunk -- loads, then the register-only chain, then stores.
         * Lane l works on table row i = 1 + t - l at step t, i.e. it is active for t in [l, nrows-2+l].
         * Two flavours of the same chunk:
         *   steady: every lane 0..last_lane is active for all T steps and every loaded row lies in the band,
         *           so nothing is predicated;
         *   edge:   ramp-up / ramp-down (and the whole strip of small windows): stores of inactive lanes are
         *           dropped through an out-of-range offset, a lane's row-0 value is injected on the step it
         *           becomes active, row loads are clamped to the band and rows past it written as zeros.
         * Inactive lanes compute on garbage that no active lane ever reads.  Ring rows are addressed through
         * running pointers (mirror rows make T consecutive rows wrap-free); chunks are issued in pairs with
         * the two sets of staging registers swapped instead of copied. */
        float F1[FIFO ? K : 1], F2[FIFO ? K : 1];   /* D[t - lane][lane + K], D[t - lane][lane] of the next K steps */
        if (FIFO) {
#pragma unroll
            for (int m = 0; m < K; m++) {
                const int row = max(m - lane, 0);   /* rows 0 .. K-1 are in the ring; lanes that start later read garbage-free junk */
                F1[m] = ring[row * CW + lane + K]; F2[m] = ring[row * CW + lane];
            }
        }
        float curS = S0;
        float left_prev = row0_left;    /* lane 0: S[0][cb-1]; other lanes: overwritten before use */
        const int nsteps = (nrows - 1) + last_lane;
        const int lane_eff = col_ok ? lane : 0x40000000;   /* lanes past the last column are never active */
        /* Disparity tables are stored SKEWED: the value lane l computes at step t -- row 1 + t - l of column l -- goes to
         * row t + 1 = (table row) + l of the strip, column l.  Every step then stores one contiguous 256-byte row with a
         * single 4-byte buffer store (lane offset constant, row offset scalar): no straightening buffer in LDS (round 1
         * kept 8 KiB of it per wave, and its 16-byte stores are where the store-data hazard of DESIGN.md section 3 lived).
         * The arg-min kernel reads the tables in the same skewed coordinates (it works position by position, the layout
         * is irrelevant to it) and un-skews only its own small output. */
        const __amdgpu_buffer_rsrc_t rsTb = __builtin_amdgcn_make_buffer_rsrc((void*)table, 0, (int)(tstride * 4), kRsrcFlags);
        /* running LDS offsets (floats) of ring rows (t + K - lane) and (t - lane), column lane */
        int oA = ((K - lane + 64 * RR) % RR) * CW + lane, oB = ((64 * RR - lane) % RR) * CW + lane;
        int wrow = filled % RR;   /* a multiple of T, like RR: the T rows of a chunk never wrap inside the ring */
        /* fast steady flavour (regular grid): next step with a forward / mirrored store (uniform), the lanes' byte offsets
         * at that step (-1: the lane never stores) and their increment per grid row */
        int nextF = 0x7fffffff, nextB = 0x7fffffff, offF = -1, offB = -1, incF = 0, incB = 0;
        auto chunk = [&](auto edge_tag, int t0, const int jj,   /* jj: index of the chunk in its group (a constant after unrolling) */
                         v4f* la1, v4f* la2, float& lb1, float& lb2,                   /* receive rows filled+(DEP-1)T .. */
                         const v4f* sa1, const v4f* sa2, float sb1, float sb2) {       /* rows filled .. go to the ring */
            /* flavour 0: steady; 1: ramp-up (lanes start one after the other, every row and index still in range);
             * 2: general edge (ramp-down, rows past the band, short tables); 3: tail of a disparity table's ramp-down:
             * every row of the band is in the ring already and no lane starts any more -- the steady chain without row
             * loads and ring writes, only the hand-off column indices are range-checked */
            /* 4: steady chunk of a self-similarity table on the regular grid, rows and columns all on the grid's pattern:
             * the lanes that store are the same 64 / p lanes on one step in p (see fast_init below), so the slot
             * arithmetic of the steady flavour shrinks to two scalar compares per step */
            constexpr int FL = decltype(edge_tag)::value;
            constexpr bool FAST = FL == 4;
            constexpr bool TAIL = FL == 3;
            constexpr bool EDGE = FL == 1 || FL == 2;   /* lane predicates */
            constexpr bool REDGE = FL == 2;             /* row / index range handling */
            if (!TAIL) load_chunk(std::integral_constant<bool, REDGE>{}, filled + (DEP - 1) * T, la1, la2, lb1, lb2);
            const float* pa = ring + oA;
            const float* pb = ring + oB;
            float d1[T], d2[T], d3[T], d4[T], lc[T], Sout[T];
#pragma unroll
            for (int s = 0; s < T; s++) {
                d1[s] = pa[s * CW + K]; d2[s] = pa[s * CW];
                if (FIFO) {
                    const int slot = (jj * T + s) % K;
                    d3[s] = F1[slot]; d4[s] = F2[slot];
                    F1[slot] = d1[s]; F2[slot] = d2[s];
                } else { d3[s] = pb[s * CW + K]; d4[s] = pb[s * CW]; }
                lc[s] = lcol[(REDGE || TAIL) ? min(1 + t0 + s, nrows + T) : 1 + t0 + s];   /* uniform address: lane 0's left neighbour */
            }
            oA += T * CW; oA = oA >= RR * CW ? oA - RR * CW : oA;
            if (!FIFO) { oB += T * CW; oB = oB >= RR * CW ? oB - RR * CW : oB; }
#pragma unroll
            for (int s = 0; s < T; s++) {
                if (EDGE) curS = (lane_eff == t0 + s) ? S0 : curS;      /* becomes active: start from its row-0 value */
                /* left neighbour's value of the previous step; lane 0 takes the hand-off column */
                const float left = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(lc[s]), __float_as_int(curS),
                                                                              0x138 /* wave_shr:1 */, 0xf, 0xf, false));
                float S = left + curS;             /* core:3379-3386, same association */
                S = S - left_prev;
                S = S + d1[s];
                S = S - d2[s];
                S = S - d3[s];
                S = S + d4[s];
                S = first_col ? left : S;          /* first column was computed up front */
                Sout[s] = S;
                curS = S;
                left_prev = left;
            }
#pragma unroll
            for (int s = 0; s < T; s++) {
                const int t = t0 + s;
                const bool act = EDGE ? (unsigned)(t - lane_eff) <= (unsigned)(nrows - 2) : col_ok;
                if (stereo) {
                    const bool on = (EDGE || TAIL) ? (unsigned)(t - lane_eff) <= (unsigned)(nrows - 2) : col_ok;
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, Sout[s]), rsTb, on ? lane * 4 : -1, (strip * SR + t + 1) * 256, 0);
                } else if (FAST) {
                    if (t == nextF) { __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, Sout[s]), rsS, offF, 0, 0); offF += incF; nextF += gP; }
                    if (t == nextB) { __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, Sout[s]), rsS, offB, 0, 0); offB += incB; nextB += gP; }
                } else {
                    int v1, v2;
                    if (irregular) {   /* irregular list: whole-slot lookups (out-of-range offsets read 0, masked by act) */
                        const int q = (b + 1 + t - lane) * W + x;
                        const int r1 = __builtin_amdgcn_raw_buffer_load_b32(rsM, q * 4, 0, 0);
                        const int r2 = (di > 0 && x + djs >= 0 && x + djs < W) ? __builtin_amdgcn_raw_buffer_load_b32(rsM, (q + di * W + djs) * 4, 0, 0) : -1;
                        v1 = (act && r1 >= 0) ? (int)(((unsigned)r1 * (unsigned)ncand + (unsigned)ord_fwd) * 4u) : -1;
                        v2 = (act && r2 >= 0) ? (int)(((unsigned)r2 * (unsigned)ncand + (unsigned)ord_bwd) * 4u) : -1;
                    } else {
                        /* rows of lanes that have not started / have finished are clamped into the table (masked by act) */
                        const slot_t* rp = rs16 + min(max(b + 1 + t0 - lane, 0), H - T);
                        const int r1 = rp[s], r2 = rp[s + di];                  /* -1: not a grid row */
                        const bool ok1 = (r1 | base_fwd) >= 0, ok2 = (r2 | base_bwd) >= 0;
                        v1 = (ok1 && (!EDGE || act)) ? r1 * row_bytes + base_fwd : -1;
                        v2 = (ok2 && (!EDGE || act)) ? r2 * row_bytes + base_bwd : -1;
                    }
                    /* a store no lane takes part in is skipped (on the regular grid three steps in four) */
                    if (__builtin_amdgcn_ballot_w64(v1 != -1)) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, Sout[s]), rsS, v1, 0, 0);
                    if (__builtin_amdgcn_ballot_w64(v2 != -1)) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, Sout[s]), rsS, v2, 0, 0);
                }
                /* hand-off column for the next strip (uniform address and value) */
                const int il = 1 + t - last_lane;
                const float hv = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(Sout[s]), last_lane));
                lcol[(REDGE || TAIL) ? ((il >= 1 && il < nrows) ? il : nrows + T) : FL == 1 ? max(il, 0) : il] = hv;   /* ramp-up: rows < 1 land in slot 0... */
            }
            if (!TAIL) {
                write_chunk(std::integral_constant<bool, REDGE>{}, filled, wrow, sa1, sa2, sb1, sb2);
                wrow = wrow + T == RR ? 0 : wrow + T;
                filled += T;
            }
        };
        {
            /* [0, tS0): ramp-up, [tS0, tS1): steady, [tS1, nsteps): ramp-down; all in groups of DEP chunks (the
             * register buffers rotate with the chunk index).  The edge flavour is valid anywhere, so the
             * ranges are simply rounded to whole groups. */
            const int band_rows = H - 2 * b;
            constexpr int G = DEP * T;
            int tS0 = ((last_lane + 1 + G - 1) / G) * G;       /* step last_lane (that lane's start) is still an edge step */
            int tS1 = min(nrows - 1, band_rows - K - DEP * T);   /* chunks [t0, t0+T) with t0+T <= tS1 are steady */
            tS1 = tS1 > tS0 ? tS0 + ((tS1 - tS0) / G) * G : tS0;
            /* ramp-up chunks may use the light flavour when every row they load, write and hand off is in range */
            const bool light_up = tS0 + K + (DEP + 1) * T <= band_rows && tS0 + T < nrows;
            int t0 = 0;
            auto group = [&](auto tag, int tg) {
#pragma unroll
                for (int j = 0; j < DEP; j++)
                    chunk(tag, tg + j * T, j, A1[(j + DEP - 1) % DEP], A2[(j + DEP - 1) % DEP], Bq1[(j + DEP - 1) % DEP], Bq2[(j + DEP - 1) % DEP],
                          A1[j], A2[j], Bq1[j], Bq2[j]);
            };
            SCAN_MARK(2);
            if (light_up) for (; t0 < min(tS0, nsteps); t0 += G) group(std::integral_constant<int, 1>{}, t0);
            else          for (; t0 < min(tS0, nsteps); t0 += G) group(std::integral_constant<int, 2>{}, t0);
            SCAN_MARK(3);
            if (MODE == 0) {
                /* A reference row / column is "on the pattern" when (coordinate - nHW) % p == 0 and its slot is below the
                 * forced last one.  If every grid column of this strip is (all but a forced last column that is not), the
                 * storing lanes share l mod p, hence also the residue mod p of the steps on which their row 1 + t - l is a
                 * grid row: one store step in p, the same lanes every time, slot = row / p.  Groups whose rows (plus di for
                 * the mirrored candidate) stay below the forced last row take that flavour; the rest of the steady range and
                 * strips with an off-pattern column keep the table look-ups. */
                const bool offpat = (base_fwd >= 0 && (x - gN) % gP != 0) || (base_bwd >= 0 && (x + djs - gN) % gP != 0);
                const unsigned long long mF = __builtin_amdgcn_ballot_w64(base_fwd >= 0), mB = __builtin_amdgcn_ballot_w64(base_bwd >= 0);
                int tF1 = (gR - 1) * gP - 1 - G - di;                       /* last group start whose rows are all regular */
                tF1 = tF1 >= t0 ? t0 + ((tF1 - t0) / G + 1) * G : t0;       /* -> end of the fast range (exclusive) */
                tF1 = min(tF1, tS1);
                if (!__builtin_amdgcn_ballot_w64(offpat) && t0 < tF1) {
                    nextF = nextB = 0x7fffffff; offF = offB = -1; incF = incB = 0;
                    if (mF) {
                        const int l1 = (int)__builtin_ctzll(mF);
                        nextF = t0 + (((l1 - 1 - t0) % gP) + gP) % gP;          /* 1 + t - l1 = 0 (mod p) */
                        if (base_fwd >= 0) { offF = ((1 + nextF - lane) / gP) * row_bytes + base_fwd; incF = row_bytes; }
                    }
                    if (mB) {
                        const int l2 = (int)__builtin_ctzll(mB);
                        nextB = t0 + (((l2 - 1 - di - t0) % gP) + gP) % gP;     /* 1 + t - l2 + di = 0 (mod p) */
                        if (base_bwd >= 0) { offB = ((1 + nextB - lane + di) / gP) * row_bytes + base_bwd; incB = row_bytes; }
                    }
                    for (; t0 < tF1; t0 += G) group(std::integral_constant<int, 4>{}, t0);
                }
            }
            for (; t0 < tS1; t0 += G) group(std::integral_constant<int, 0>{}, t0);
            SCAN_MARK(4);
            const int t_end = nsteps;
            if (stereo) {
                /* a disparity table never reads past the band (nrows = band_rows - K + 1): once the band is in the ring and
                 * the last lane has started, the rest of the ramp-down needs neither loads nor ring writes */
                for (; t0 < t_end && (filled < band_rows || t0 <= last_lane); t0 += G) group(std::integral_constant<int, 2>{}, t0);
                for (; t0 < t_end; t0 += G) group(std::integral_constant<int, 3>{}, t0);
            } else
                for (; t0 < t_end; t0 += G) group(std::integral_constant<int, 2>{}, t0);
        }
        SCAN_MARK(3);
        row0_left = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(S0), last_lane));
        __syncthreads();
    }
#ifdef LFBM5D_PHASE_TIMING
    if (lane == 0 && a.dbg) {
        const int base = stereo ? 0 : 6;
        for (int i = 0; i < 5; i++) atomicAdd(&a.dbg[base + i], (unsigned long long)tk[i]);
        atomicAdd(&a.dbg[base + 5], 1ull);
    }
#endif
}

/* Two waves per SIMD: the scan is issue-bound with long dependent chains (0.61 of a lone wave's cycles issue), so a
 * second wave on the SIMD is worth a quarter of the kernel's time -- if it fits: 256 VGPRs (round 1 used 298) and an
 * LDS footprint of at most a sixth of the CU's 160 KiB (round 1: 40 KiB, a quarter). */
#ifndef LFBM5D_SCAN_WAVES_ATTR
#define LFBM5D_SCAN_WAVES_ATTR __attribute__((amdgpu_waves_per_eu(2, 2)))
#endif
template <int K, bool WIDE>
__global__ __launch_bounds__(64) LFBM5D_SCAN_WAVES_ATTR void k_bm_scan(ScanArgs a) {
    extern __shared__ float lds[];
    /* one launch covers both searches: blocks [0, n_self) are self-similarity tables, the rest
     * disparity tables (fewer, fuller rounds of resident waves than two launches; dispatching the disparity tables first,
     * or one self table per two disparity tables, measured 9 % slower: 2.98 / 2.99 vs 2.75 ms of block matching per pass) */
    if (blockIdx.x >= a.n_self) scan_body<K, 2, WIDE>(a, (int)(blockIdx.x - a.n_self), lds);
    else if (a.refmap) scan_body<K, 1, WIDE>(a, (int)blockIdx.x, lds);
    else scan_body<K, 0, WIDE>(a, (int)blockIdx.x, lds);
}

/* ---- any patch size (round 5) ----
 * The table kernels above are instantiated for 8-, 12- and 16-pixel patches (register FIFOs and ring geometry are compile-time);
 * the reference takes any kHard / kWien (utilities_LF.cpp:1214, :1255).  Every other patch size runs this plain form of the same
 * mapping: one wavefront per displacement table, lane = column of a 64-column strip, rows skewed across the lanes so that the
 * recurrence's neighbours are the lane's own previous value and a one-lane shift -- the SAME operations in the SAME order, hence
 * the same tables bit for bit -- with the four squared differences of a step read straight from the estimate (no ring, no FIFO:
 * eight uncoalesced loads per step).  About 2 ms per 560 x 560 pass where the dedicated kernels take 0.8: a fallback, not a
 * tuned path.  Outputs in the first-generation layout (skewed disparity tables, score stores through the grid / position map). */
template <int MODE>
__device__ __forceinline__ void scan_any_body(const ScanArgs& a, const int bid, float* lds) {
    constexpr bool stereo = MODE == 2;
    constexpr bool irregular = MODE == 1;
    const int lane = threadIdx.x;
    const int K = (int)a.k;
    const int half = stereo ? (int)a.nDisp : (int)a.nSim;
    const int trim = stereo ? K - 1 : 0;
    const int W = a.W, H = a.H, b = stereo ? (int)a.nDisp : (int)a.nHW;
    const int Ns = 2 * half + 1, ncand = Ns * Ns;
    const int nrows = H - 2 * b - trim, ncols = W - 2 * b - trim;
    int di, dj;
    DiffImg D;
    const size_t WH = (size_t)W * H;
    if (stereo) {
        const int slot = bid / ncand, ddk = bid % ncand;
        di = ddk / Ns; dj = ddk % Ns;
        D.i1 = a.est + (size_t)a.pst * WH;
        D.i2 = a.est + (size_t)a.st_of_slot[slot] * WH;
        D.dk = di * W + dj - half * (1 + W);
    } else {
        di = bid / Ns; dj = bid % Ns;
        D.i1 = D.i2 = a.est + (size_t)a.pst * WH;
        D.dk = di * W + dj - half;
    }
    D.W = W; D.H = H; D.b = b;
    const size_t tstride = stereo ? stereo_table_stride(a.W, a.H, a.k, a.nDisp) : 0;
    float* table = stereo ? a.tables + (size_t)bid * tstride : nullptr;
    const int SR = nrows + 63;
    const int djs = dj - half;
    const int ord_fwd = dj * Ns + di;
    const int ord_bwd = (-djs + half) * Ns + (half + 1) + (half - di);
    const int gR = a.n_ref_rows, gC = a.n_ref_cols, gP = a.p, gN = a.nHW;
    const int lastR = H - K - gN, lastC = W - K - gN;
    auto emit = [&](int y, int x, float S) {
        if (stereo) { table[((size_t)((x - b) / 64) * SR + (y - b) + (x - b) % 64) * 64 + (x - b) % 64] = S; return; }
        if (irregular) {
            const int r = a.refmap[y * W + x];
            if (r >= 0) a.scores[(size_t)r * ncand + ord_fwd] = S;
            const int yy = y + di, xx = x + djs;
            if (di > 0 && yy < H && xx >= 0 && xx < W) {
                const int r2 = a.refmap[yy * W + xx];
                if (r2 >= 0) a.scores[(size_t)r2 * ncand + ord_bwd] = S;
            }
            return;
        }
        const int cx = grid_index(x, gC, lastC, gN, gP);
        if (cx >= 0) {
            const int ry = grid_index(y, gR, lastR, gN, gP);
            if (ry >= 0) a.scores[(size_t)(ry * gC + cx) * ncand + ord_fwd] = S;
        }
        if (di > 0) {
            const int cx2 = grid_index(x + djs, gC, lastC, gN, gP);
            if (cx2 >= 0) {
                const int ry2 = grid_index(y + di, gR, lastR, gN, gP);
                if (ry2 >= 0) a.scores[(size_t)(ry2 * gC + cx2) * ncand + ord_bwd] = S;
            }
        }
    };
    float* Lprev = lds;                   /* [nrows] column left of the current strip */
    float* Lnext = lds + nrows;           /* [nrows] last column of the current strip */
    float* tbuf = lds + 2 * nrows;        /* [64][K] terms of the first-row / first-column chains; the corner's K x K differences */

    /* corner (core:3344-3352): K x K terms added one after the other */
    for (int e = lane; e < K * K; e += 64) tbuf[e] = D(b + e / K, b + e % K);
    __syncthreads();
    float corner = 0.0f;
    for (int e = 0; e < K * K; e++) corner += tbuf[e];
    __syncthreads();
    if (lane == 0) { Lnext[0] = corner; emit(b, b, corner); }
    /* first column (core:3367-3372): S[i][b] = S[i-1][b] + sum_q (D[i-1+K][b+q] - D[i-1][b+q]), the K terms one after the other */
    {
        float carry = corner;
        for (int i0 = 1; i0 < nrows; i0 += 64) {
            const int i = i0 + lane;
            for (int q = 0; q < K; q++) tbuf[lane * K + q] = i < nrows ? D(b + i - 1 + K, b + q) - D(b + i - 1, b + q) : 0.0f;
            __syncthreads();
            const int m = min(64, nrows - i0);
            float mine = 0.0f;
            for (int l = 0; l < m; l++) {
                float cand = carry;
                for (int q = 0; q < K; q++) cand += tbuf[l * K + q];
                carry = cand;
                if (lane == l) mine = carry;
            }
            if (i < nrows) { Lnext[i] = mine; emit(b + i, b, mine); }
            __syncthreads();
        }
    }
    /* strips of 64 columns from table column 1 on; the column left of a strip: the first column, then the previous strip's last */
    const int nstrips = (ncols - 1 + 63) / 64;
    for (int strip = 0; strip < nstrips; strip++) {
        { float* t = Lprev; Lprev = Lnext; Lnext = t; }
        __syncthreads();
        const int cb = b + 1 + 64 * strip;
        const int x = cb + lane;
        const bool col_ok = x < b + ncols;
        const int last_lane = min(63, ncols - 2 - 64 * strip);
        /* first row (core:3354-3362): S[b][x] = S[b][x-1] + sum_p (D[b+p][x-1+K] - D[b+p][x-1]), chained across the lanes */
        for (int p = 0; p < K; p++) tbuf[lane * K + p] = col_ok ? D(b + p, x - 1 + K) - D(b + p, x - 1) : 0.0f;
        __syncthreads();
        float S0 = 0.0f;
        {
            float carry = Lprev[0];
            for (int l = 0; l <= last_lane; l++) {
                float cand = carry;
                for (int p = 0; p < K; p++) cand += tbuf[l * K + p];
                carry = cand;
                if (lane == l) S0 = carry;
            }
        }
        if (col_ok) emit(b, x, S0);
        if (lane == last_lane) Lnext[0] = S0;
        /* the remaining rows: lane l works on table row 1 + t - l at step t (core:3375-3383, same association) */
        float cur = S0, leftprev = Lprev[0];
        const int nsteps = (nrows - 1) + last_lane;
        for (int t = 0; t < nsteps; t++) {
            const int r = 1 + t - lane;
            float nb = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(cur), 0x138 /* wave_shr:1 */, 0xf, 0xf, false));
            if (lane == 0) nb = Lprev[min(max(r, 0), nrows - 1)];
            if (col_ok && r >= 1 && r <= nrows - 1) {
                const int y = b + r;
                float S = nb + cur;
                S = S - leftprev;
                S = S + D(y + K - 1, x + K - 1);
                S = S - D(y + K - 1, x - 1);
                S = S - D(y - 1, x + K - 1);
                S = S + D(y - 1, x - 1);
                emit(y, x, S);
                cur = S;
                if (lane == last_lane) Lnext[r] = S;
            }
            leftprev = nb;
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(64) void k_bm_scan_any(ScanArgs a) {
    extern __shared__ float lds[];
    if (blockIdx.x >= a.n_self) scan_any_body<2>(a, (int)(blockIdx.x - a.n_self), lds);
    else if (a.refmap) scan_any_body<1>(a, (int)blockIdx.x, lds);
    else scan_any_body<0>(a, (int)blockIdx.x, lds);
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
                                                     unsigned* __restrict__ self_cnt,
                                                     int grid_cols, int nHW, int p, int last_r, int last_c) {
    extern __shared__ unsigned long long keys[];
    const int lane = threadIdx.x;
    const unsigned ref = blockIdx.x;
    if (ref >= n_refs) return;
    const int Ns = 2 * nSim + 1, ncand = Ns * Ns;
    /* e / Ns by multiplication: exact for e < 2^20 / Ns, i.e. for every candidate index (Ns <= 127) */
    const unsigned div_m = ((1u << 20) + (unsigned)Ns - 1) / (unsigned)Ns;
    const int k_r = (int)refs[ref];
    size_t row = ref;
    if (grid_cols) {   /* the score table is the regular grid's: this reference's row in it */
        const int y = k_r / W, x = k_r - y * W;
        const int gi = y == last_r ? (last_r - nHW + p - 1) / p : (y - nHW) / p;   /* the forced last index sits behind the stepped ones */
        const int gj = x == last_c ? (last_c - nHW + p - 1) / p : (x - nHW) / p;
        row = (size_t)gi * grid_cols + gj;
    }
    const float* sc = scores + row * ncand;
    /* keys of the candidates that pass the threshold, compacted in scan order (ballot prefix): the selection
     * rounds below then only walk those */
    int cnt = 0;
    unsigned long long mine = ~0ull;   /* smallest key this lane has appended (any split of the keys into 64 sets serves the pruning bound) */
    constexpr int kB = 8;    /* 64-candidate chunks whose score loads are in flight together */
    for (int e0 = 0; e0 < ncand; e0 += 64 * kB) {
        float score[kB], test[kB];
#pragma unroll
        for (int u = 0; u < kB; u++) {
            const int e = e0 + u * 64 + lane;
            score[u] = 0.0f; test[u] = 0.0f;
            if (e < ncand) {
                const int q = (int)(((unsigned)e * div_m) >> 20);
                const int djp = q - nSim, r = e - q * Ns;
                const bool fwd = r <= nSim;
                const int dip = fwd ? r : r - 2 * nSim - 1;
                score[u] = sc[e];
                /* backward half: tested with the mirrored table at k_r, scored at the candidate (core:3415-3419) */
                test[u] = fwd ? score[u] : sc[(-djp + nSim) * Ns + (-dip)];
            }
        }
#pragma unroll
        for (int u = 0; u < kB; u++) {
            const int e = e0 + u * 64 + lane;
            const bool pass = e < ncand && test[u] < thr;
            const unsigned long long bal = __ballot(pass);
            if (pass) {
                const unsigned long long kk = ((unsigned long long)f2ord(score[u]) << 32) | (unsigned)e;
                keys[cnt + __popcll(bal & ((1ull << lane) - 1ull))] = kk;
                mine = kk < mine ? kk : mine;
            }
            cnt += __popcll(bal);
        }
    }
    __syncthreads();
    unsigned nSx;
    if (N > cnt) { nSx = 1; while (nSx * 2 <= (unsigned)cnt) nSx *= 2; } else nSx = N;
    unsigned* out = self_idx + (size_t)ref * N;
    if (cnt == 0) { /* core:3429-3432 */
        if (lane == 0) { out[0] = k_r; out[1] = k_r; self_cnt[ref] = 2; }
        return;
    }
    /* minimum of a 64-bit key over the wavefront by DPP row shifts / row broadcasts (a `__shfl_xor` butterfly is six dependent
     * pairs of ds_bpermute: with 2 nSx of these reductions per reference patch that was half the kernel) */
    auto wave_min = [](unsigned long long v) {
        auto step = [&](auto ctrl, auto rmask, auto bmask, unsigned long long src) {
            constexpr int C = decltype(ctrl)::value, R = decltype(rmask)::value, B = decltype(bmask)::value;
            const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp((int)(unsigned)v, (int)(unsigned)src, C, R, B, false);
            const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp((int)(unsigned)(v >> 32), (int)(unsigned)(src >> 32), C, R, B, false);
            const unsigned long long oth = ((unsigned long long)hi << 32) | lo;   /* lanes the pattern does not write keep their own key */
            v = oth < v ? oth : v;
        };
        using std::integral_constant;
        const unsigned long long v0 = v;
        step(integral_constant<int, 0x111>{}, integral_constant<int, 0xf>{}, integral_constant<int, 0xf>{}, v0);   /* row_shr:1 */
        step(integral_constant<int, 0x112>{}, integral_constant<int, 0xf>{}, integral_constant<int, 0xf>{}, v0);   /* row_shr:2 */
        step(integral_constant<int, 0x113>{}, integral_constant<int, 0xf>{}, integral_constant<int, 0xf>{}, v0);   /* row_shr:3 */
        step(integral_constant<int, 0x114>{}, integral_constant<int, 0xf>{}, integral_constant<int, 0xe>{}, v);    /* row_shr:4 */
        step(integral_constant<int, 0x118>{}, integral_constant<int, 0xf>{}, integral_constant<int, 0xc>{}, v);    /* row_shr:8 */
        step(integral_constant<int, 0x142>{}, integral_constant<int, 0xa>{}, integral_constant<int, 0xf>{}, v);    /* row_bcast:15 */
        step(integral_constant<int, 0x143>{}, integral_constant<int, 0xc>{}, integral_constant<int, 0xf>{}, v);    /* row_bcast:31 */
        const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, 63), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), 63);
        return ((unsigned long long)hi << 32) | lo;
    };
    /* number of lanes whose key is smaller than this lane's (equal keys -- only the sentinels -- by lane number): 64
     * broadcast-compare steps, no dependent chain (nSx rounds of a wave minimum are nSx x 7 dependent DPP steps) */
    auto wave_rank = [&](unsigned long long v) {
        unsigned r = 0;
#pragma unroll
        for (int j = 0; j < 64; j++) {
            const unsigned long long o = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), j) << 32)
                                       | (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, j);
            r += (o < v || (o == v && j < lane)) ? 1u : 0u;
        }
        return r;
    };
    if (cnt > 128) {
        /* Prune before selecting: the nSx-th smallest of the 64 per-lane minima bounds the nSx-th smallest key overall
         * (there are nSx distinct keys at or below it), so only keys up to that bound can be selected.  They are
         * compacted in place -- the write index never passes the read index, and a wavefront's reads of a round
         * precede its writes -- and the selection rounds below walk a list of typically a few dozen keys. */
        /* the lane whose minimum has nSx - 1 smaller ones holds the bound (fewer than nSx lanes with a key: the bound is the
         * sentinel and nothing is pruned) */
        const unsigned rk = wave_rank(mine);
        const unsigned long long hb = __ballot(rk == nSx - 1);
        const int src = __ffsll((long long)hb) - 1;
        const unsigned long long bound = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(mine >> 32), src) << 32)
                                       | (unsigned)__builtin_amdgcn_readlane((int)(unsigned)mine, src);
        int c2 = 0;
        for (int e0 = 0; e0 < cnt; e0 += 64) {
            const int e = e0 + lane;
            const unsigned long long kk = e < cnt ? keys[e] : ~0ull;
            const bool keep = kk <= bound;
            const unsigned long long bal = __ballot(keep);
            if (keep) keys[c2 + __popcll(bal & ((1ull << lane) - 1ull))] = kk;
            c2 += __popcll(bal);
        }
        cnt = c2;
        __builtin_amdgcn_wave_barrier();
    }
    if (cnt <= 64) {
        /* one key per lane: its rank is its place in the selection (core:3437-3441: partial_sort by distance, ties in scan order) */
        const unsigned long long kk = lane < cnt ? keys[lane] : ~0ull;
        const unsigned rk = wave_rank(kk);
        if (lane < cnt && rk < nSx) {
            const int e = (int)(kk & 0xffffffffu);
            const int djp = e / Ns - nSim, r = e % Ns;
            const int dip = (r <= nSim) ? r : r - 2 * nSim - 1;
            out[rk] = (unsigned)(k_r + dip * W + djp);
            if (nSx == 1) out[1] = out[0]; /* duplicate rule core:3443-3444 */
        }
        if (lane == 0) self_cnt[ref] = nSx == 1 ? 2 : nSx;
        return;
    }
    unsigned long long last = 0;
    bool first = true;
    for (unsigned n = 0; n < nSx; n++) { /* n-th smallest (distance, scan order) key */
        unsigned long long best = ~0ull;
        for (int e = lane; e < cnt; e += 64) {
            const unsigned long long kk = keys[e];
            if ((first || kk > last) && kk < best) best = kk;
        }
        best = wave_min(best);
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
 * (dj outer, di inner), the order the reference pushes candidates in.  grid.y = table slot. */
struct ArgminArgs { const float* tables; size_t tstride; unsigned st_of_slot[kBigA]; int W, H, k, nDisp; float thr; unsigned* best; unsigned char* shape; };
__global__ __launch_bounds__(256) void k_stereo_argmin(ArgminArgs a) {
    /* The tables are skewed (scan kernel, store stage): entry [strip][q][l] holds table row q - l of column 64 strip + l.
     * A thread takes four consecutive lanes l0 .. l0+3 of one skewed row q (one 16-byte load per table), eight tables in
     * flight; the arg-min is taken position by position, so the skew only matters when the result is written. */
    const int W = a.W, H = a.H, nDisp = a.nDisp;
    const int span_c = W - 2 * nDisp - a.k + 1, span_r = H - 2 * nDisp - a.k + 1;
    const int SR = span_r + 63, nstrips = (span_c + 63) / 64;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nstrips * SR * 16) return;
    const unsigned slot = blockIdx.y, st = a.st_of_slot[slot];
    const int l0 = 4 * (i & 15), q = (i >> 4) % SR, strip = (i >> 4) / SR;
    /* rows of the four entries: q - l0 - e; all outside the table -> nothing to do (the corners of the skew) */
    if (q - l0 < 0 || q - l0 - 3 >= span_r) return;
    const int Ns = 2 * nDisp + 1, ncand = Ns * Ns;
    const size_t WH = (size_t)W * H;
    const float* t = a.tables + (size_t)slot * ncand * a.tstride + ((size_t)strip * SR + q) * 64 + l0;
    typedef float f4 __attribute__((ext_vector_type(4)));
    float bv[4]; int bo[4], bd[4];
    {
        const f4 v0 = *reinterpret_cast<const f4*>(t);
#pragma unroll
        for (int e = 0; e < 4; e++) { bv[e] = v0[e]; bo[e] = 0; bd[e] = 0; }
    }
    for (int d0 = 0; d0 < ncand; d0 += 8) {
        f4 v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = *reinterpret_cast<const f4*>(t + (size_t)min(d0 + u, ncand - 1) * a.tstride);
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
        const int r = q - l0 - e, col = 64 * strip + l0 + e;
        if (r < 0 || r >= span_r || col >= span_c) continue;            /* entries never written hold garbage */
        const int pos = (nDisp + r) * W + nDisp + col;
        const int di = bd[e] / Ns, dj = bd[e] % Ns;
        a.best[(size_t)st * WH + pos] = (unsigned)(pos + (di - nDisp) * W + (dj - nDisp));
        a.shape[(size_t)st * WH + pos] = bv[e] < a.thr ? 1 : 0;
    }
}

__global__ void k_refmap(const unsigned* __restrict__ refs, unsigned n_refs, int* __restrict__ refmap) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_refs) refmap[refs[i]] = (int)i;
}

/* The reference list of a subset pass (core:157-158, utilities_LF.cpp:1000-1099) on the device: the patches of the regular grid
 * whose k x k footprint still holds an exactly-zero weight in den0 (channel 0 of den[pst]), in raster order.  k_subset_flags: one
 * wavefront per grid patch; k_subset_compact: one workgroup, ordered compaction (every thread a run of consecutive patches, an
 * exclusive scan of the runs' counts through LDS). */
__global__ __launch_bounds__(256) void k_subset_flags(const float* __restrict__ den0, int Wb, int k, int nHW, int p, int n_rows, int n_cols,
                                                       int last_r, int last_c, unsigned char* __restrict__ flags) {
    const int ref = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (ref >= n_rows * n_cols) return;
    const int gi = ref / n_cols, gj = ref - gi * n_cols;
    const int y = gi == n_rows - 1 ? last_r : nHW + gi * p, x = gj == n_cols - 1 ? last_c : nHW + gj * p;
    bool zero = false;
    for (int e = lane; e < k * k; e += 64) zero = zero || den0[(size_t)(y + e / k) * Wb + x + e % k] == 0.0f;
    const unsigned long long any = __ballot(zero);
    if (lane == 0) flags[ref] = any ? 1 : 0;
}
__global__ __launch_bounds__(1024) void k_subset_compact(const unsigned char* __restrict__ flags, int Wb, int nHW, int p, int n_rows, int n_cols,
                                                          int last_r, int last_c, unsigned* __restrict__ refs, unsigned* __restrict__ count) {
    __shared__ unsigned part[1024];
    const int R = n_rows * n_cols, tid = threadIdx.x, per = (R + 1023) / 1024, b = tid * per, e = min(b + per, R);
    unsigned n = 0;
    for (int i = b; i < e; i++) n += flags[i];
    part[tid] = n;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {   /* inclusive scan */
        const unsigned v = tid >= o ? part[tid - o] : 0u;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    unsigned w = part[tid] - n;
    for (int i = b; i < e; i++)
        if (flags[i]) {
            const int gi = i / n_cols, gj = i - gi * n_cols;
            const int y = gi == n_rows - 1 ? last_r : nHW + gi * p, x = gj == n_cols - 1 ? last_c : nHW + gj * p;
            refs[w++] = (unsigned)(y * Wb + x);
        }
    if (tid == 1023) *count = part[1023];
}

} /* namespace */

hipError_t launch_subset_list(hipStream_t s, const float* den0, unsigned Wb, unsigned k, unsigned nHW, unsigned p, unsigned n_rows,
                              unsigned n_cols, unsigned last_r, unsigned last_c, unsigned char* flags, unsigned* refs, unsigned* count) {
    const unsigned R = n_rows * n_cols;
    hipLaunchKernelGGL(k_subset_flags, dim3((R + 3) / 4), dim3(256), 0, s, den0, (int)Wb, (int)k, (int)nHW, (int)p, (int)n_rows, (int)n_cols,
                       (int)last_r, (int)last_c, flags);
    hipLaunchKernelGGL(k_subset_compact, dim3(1), dim3(1024), 0, s, flags, (int)Wb, (int)nHW, (int)p, (int)n_rows, (int)n_cols, (int)last_r,
                       (int)last_c, refs, count);
    return hipGetLastError();
}

hipError_t launch_refmap(hipStream_t s, const unsigned* refs, unsigned n_refs, int* refmap) {
    hipLaunchKernelGGL(k_refmap, dim3((n_refs + 255) / 256), dim3(256), 0, s, refs, n_refs, refmap);
    return hipGetLastError();
}

hipError_t launch_bm_scan(hipStream_t s, const ScanArgs& a) {
    const unsigned rows_self = a.n_self ? a.H - 2 * a.nHW : 0, rows_st = a.n_stereo ? a.H - 2 * a.nDisp - (a.k - 1) : 0;
    const unsigned nrows = rows_self > rows_st ? rows_self : rows_st;
    const unsigned T = 4;
    const bool wide = a.n_ref_rows > 127;   /* row-slot table entries: bytes unless the grid has more than 127 rows */
    const unsigned ring_rows = (a.k == 8 || a.k == 16) ? 64 + 2 * T : 64 + a.k + 2 * T;   /* scan_body: FIFO */
    size_t lds = (size_t)((ring_rows + T - 1) * (64 + a.k) + nrows + T + 1) * sizeof(float)
               + (a.n_self && !a.refmap ? (size_t)(a.H + 64 + 8) * (wide ? 2 : 1) : 0);     /* row-slot table of the regular grid (+ alignment) */
    const unsigned n = a.n_self + a.n_stereo;
    if (!n) return hipSuccess;
#ifdef LFBM5D_SCAN_LDS_EXPERIMENT   /* development builds: timing at a smaller LDS footprint (results are garbage) */
    if (a.lds_cap) lds = std::min<size_t>(lds, (size_t)a.lds_cap);   /* option scan_lds_cap */
#endif
#define LFBM5D_SCAN(K_) do { if (wide) hipLaunchKernelGGL((k_bm_scan<K_, true>), dim3(n), dim3(64), lds, s, a); \
                             else      hipLaunchKernelGGL((k_bm_scan<K_, false>), dim3(n), dim3(64), lds, s, a); } while (0)
    /* option scan_any: test hook, the plain any-patch-size kernel for 8 / 12 / 16 too (compared bit for bit with the dedicated ones) */
    switch ((a.opt & kOptScanAny) ? 0u : a.k) {
        case 8:  LFBM5D_SCAN(8); break;
        case 12: LFBM5D_SCAN(12); break;
        case 16: LFBM5D_SCAN(16); break;
        default: {   /* any other patch size: the plain form (k_bm_scan_any) */
            const size_t lds_any = (size_t)(2 * nrows + 64 * std::max(a.k, 16u) + 64) * sizeof(float);
            hipLaunchKernelGGL(k_bm_scan_any, dim3(n), dim3(64), lds_any, s, a);
            break;
        }
    }
#undef LFBM5D_SCAN
    return hipGetLastError();
}

hipError_t launch_self_select(hipStream_t s, const float* scores, const unsigned* refs, unsigned n_refs,
                              unsigned W, unsigned nSim, unsigned N, float thr, unsigned* self_idx,
                              unsigned* self_cnt, unsigned grid_cols, unsigned nHW, unsigned p, unsigned last_r, unsigned last_c) {
    const unsigned Ns = 2 * nSim + 1;
    hipLaunchKernelGGL(k_self_select, dim3(n_refs), dim3(64), (size_t)Ns * Ns * sizeof(unsigned long long), s,
                       scores, refs, n_refs, (int)W, (int)nSim, (int)N, thr, self_idx, self_cnt,
                       (int)grid_cols, (int)nHW, (int)p, (int)last_r, (int)last_c);
    return hipGetLastError();
}

hipError_t launch_self_trivial(hipStream_t s, const unsigned* refs, unsigned n_refs, unsigned* self_idx,
                               unsigned* self_cnt) {
    hipLaunchKernelGGL(k_self_trivial, dim3((n_refs + 255) / 256), dim3(256), 0, s, refs, n_refs, self_idx, self_cnt);
    return hipGetLastError();
}

hipError_t launch_stereo_argmin(hipStream_t s, const float* tables, const unsigned* st_of_slot, unsigned n_slots,
                                unsigned W, unsigned H, unsigned k, unsigned nDisp, float thr,
                                unsigned* best, unsigned char* shape) {
    const unsigned span_c = W - 2 * nDisp - k + 1, span_r = H - 2 * nDisp - k + 1;
    const unsigned n = ((span_c + 63) / 64) * (span_r + 63) * 16;   /* groups of four lanes of every skewed row */
    ArgminArgs a;
    a.tables = tables; a.tstride = stereo_table_stride(W, H, k, nDisp); a.W = (int)W; a.H = (int)H; a.k = (int)k; a.nDisp = (int)nDisp; a.thr = thr; a.best = best; a.shape = shape;
    for (unsigned i = 0; i < n_slots && i < (unsigned)kBigA; i++) a.st_of_slot[i] = st_of_slot[i];
    hipLaunchKernelGGL(k_stereo_argmin, dim3((n + 255) / 256, n_slots), dim3(256), 0, s, a);
    return hipGetLastError();
}

} /* namespace lfbm5d */
