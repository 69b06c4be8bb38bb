/*
 * lfbm5d_scan2.hip -- block-matching distance tables, second generation (round 3).
 *
 * Same arithmetic as lfbm5d_bm.hip (the reference's integral-image recurrence in the reference's association order,
 * precompute_BM core:3301-3461, precompute_BM_stereo core:3479-3611 -- bit-identical tables), different machine mapping:
 *
 *   - a WORKGROUP of seven table waves and one loader wave walks seven displacement tables of one image pair in lockstep
 *     (one barrier per chunk of eight steps).  The eight tables read the same image rows, only shifted by their displacement, so the
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
 *     in global memory;
 *   - the table waves issue stores only: every load -- ring rows two chunks ahead, the hand-off column values of the
 *     next chunks into a small staging area -- belongs to the loader wave, whose vmcnt (an in-order counter of loads
 *     AND stores) therefore never waits for a table store to be acknowledged.
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
/* development builds (tools/build_variant.sh ... -DLFBM5D_S2_EXP=bits): timing experiments, results are garbage.
 * 1: table stores go nowhere; 2: no hand-off column traffic; 4: no ring loads; 8 / 16: no self / no disparity tables */
#ifndef LFBM5D_S2_EXP
#define LFBM5D_S2_EXP 0
#endif
/* cache policy of the (value, order) pairs: written once here, read once by the arg-min -- non-temporal on both sides (round 6:
 * block matching 1-2 % shorter per pass, profiles/r06_a_nt_knobs_ab.txt; the estimate rows every workgroup's loader re-reads stay in L2) */
#ifndef LFBM5D_PAIR_STORE_AUX
#define LFBM5D_PAIR_STORE_AUX 2
#endif
#ifndef LFBM5D_PAIR_LOAD_NT
#define LFBM5D_PAIR_LOAD_NT 1
#endif
#ifndef LFBM5D_S2_HAND_AUX
#define LFBM5D_S2_HAND_AUX 17   /* sc0 sc1 */
#endif
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

#ifndef LFBM5D_S2_NW
#define LFBM5D_S2_NW 11
#endif
constexpr int kS2NW = LFBM5D_S2_NW;         /* tables (table waves) of a workgroup */
constexpr int kS2NL = 1;          /* ... and its loader wave: twelve waves, three per SIMD (the loader's instruction count per chunk is about a
                                   * table wave's, so every SIMD carries the same load and no wave idles at the chunk barrier), one workgroup per CU */
constexpr int kS2NI = 6;          /* 16-byte pieces of a block of eight ring rows per loader lane: 64 * kS2NI >= 2 (CW1 + CW2) */
constexpr int kS2LD = 2;          /* blocks the loader wave has in flight */

/* ---- the loader wave: image rows into the two rings, hand-off column values into their staging area ---- */
template <int K, bool STEREO, int NW, int NL>
__device__ __forceinline__ void scan2_loader(const ScanArgs& a, const Scan2Wg& g, float* lds, int r0, const int lw) {
    static_assert(NL == 1, "one loader wave");
    /* the loader wave is the youngest of the workgroup and would get the issue slots the table waves leave over, with all
     * of them waiting at the chunk barrier for the rows: it goes first */
    __builtin_amdgcn_s_setprio(3);
    typedef S2Geom<K> G;
    constexpr int CW1 = G::CW1, RR1 = G::RR1, RRp1 = G::RRp1, LEAD1 = G::LEAD1;
    const int lane = threadIdx.x & 63;
    const int W = a.W, H = a.H;
    const int half = STEREO ? (int)a.nDisp : (int)a.nSim;
    const int b = STEREO ? (int)a.nDisp : (int)a.nHW;
    const int trim = STEREO ? K - 1 : 0;
    const int Ns = 2 * half + 1, ncand = Ns * Ns;
    const int nrows = H - 2 * b - trim, ncols = W - 2 * b - trim;
    const size_t WH = (size_t)W * H;
    const int CW2 = G::cw2(g.ch), LEAD2 = G::lead2(g.rh), RR2 = G::rr2(g.rh), RRp2 = RR2 + 13;
    const int LEADM = LEAD2 > LEAD1 ? LEAD2 : LEAD1;
    float* ring1 = lds;
    float* ring2 = lds + CW1 * RRp1;
    float* lcst = ring2 + CW2 * RRp2;            /* [2][NW][8] hand-off column values of the current / next chunk */
    const unsigned pl1 = a.pst, pl2 = STEREO ? a.st_of_slot[g.slot] : a.pst;
    const __amdgpu_buffer_rsrc_t rE = __builtin_amdgcn_make_buffer_rsrc((void*)(a.est - kLeadBytes / 4), 0, (int)(a.est_planes * WH * 4 + kLeadBytes + 1024), kRsrcFlags);
    const __amdgpu_buffer_rsrc_t rLA = __builtin_amdgcn_make_buffer_rsrc((void*)a.lcol, 0, (int)((a.n_self + a.n_stereo) * a.lcol_stride * 4), kRsrcFlags);
    /* piece k of a block <-> (ring, row of the block, four columns).  Every lane issues all kS2NI loads of a block, pieces past
     * the last one at an out-of-range offset: with loads under a uniform branch the compiler can no longer count them and
     * waits for vmcnt(0), i.e. for the block it has just asked for */
    const int n1 = 2 * CW1, nit = n1 + 2 * CW2;
    bool pv[kS2NI], p1[kS2NI];
    int py0[kS2NI], pgo[kS2NI], pslot[kS2NI];
    float* pdst[kS2NI];
#pragma unroll
    for (int k = 0; k < kS2NI; k++) {
        const int item = k * 64 + lane;
        pv[k] = item < nit; p1[k] = item < n1;
        const int it = p1[k] ? item : item - n1;
        const int r = it & 7, quad = it >> 3;
        py0[k] = (p1[k] ? b : b + g.r2lo) + r;
        /* byte offset of the piece in row 0 of its plane, relative to the strip's first column */
        pgo[k] = (int)((p1[k] ? pl1 : pl2) * WH) * 4 + kLeadBytes + ((p1[k] ? 0 : g.c2lo) + 4 * quad - 1) * 4;
        pdst[k] = (p1[k] ? ring1 : ring2) + 4 * quad * (p1[k] ? RRp1 : RRp2);
        pslot[k] = (r + (p1[k] ? 0 : r0)) % (p1[k] ? RR1 : RR2);
    }
    /* hand-off values: lane -> (table, value of a chunk of eight steps) */
    static_assert(NW <= 16, "two hand-off values per loader lane");
    (void)lw;
    const int he = lane & 7;
    int hbase[2];
    float* hdst[2];
#pragma unroll
    for (int q = 0; q < 2; q++) {
        const int hw = (q * 64 + lane) >> 3;
        hbase[q] = -1;
        if (hw < NW) {
            const int tw = g.tab[hw];
            if (tw >= 0) hbase[q] = ((STEREO ? (int)a.n_self + g.slot * ncand + tw : tw) * (int)a.lcol_stride + 65 + he) * 4;
        }
        hdst[q] = lcst + (hw < NW ? hw : 0) * 8 + he;
    }
    const bool hok[2] = {(lane >> 3) < NW, ((64 + lane) >> 3) < NW};

    const int nstrips = (ncols - 1 + 63) / 64;
    for (int strip = 0; strip < nstrips; strip++) {
        const int cb = b + 1 + 64 * strip;
        const int last_lane = min(63, ncols - 2 - 64 * strip);
        const int nchunks = (((nrows - 1) + last_lane + 15) >> 4) * 2;
        lds_barrier();   /* E: the table waves are done with the rings and their hand-off stores have landed */
        int ws[kS2NI];
#pragma unroll
        for (int k = 0; k < kS2NI; k++) ws[k] = pslot[k];
        auto piece_load = [&](int k, int j) -> v4f {
            const int y = min(max(py0[k] + 8 * j, 0), H - 1);
            return __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rE, pv[k] ? pgo[k] + (y * W + cb) * 4 : -1, 0, 0));
        };
        auto piece_write = [&](int k, const v4f v) {
            if (pv[k]) {
                const int RRk = p1[k] ? RR1 : RR2, RRpk = p1[k] ? RRp1 : RRp2;
                const int slot = ws[k];
                ws[k] += 8; ws[k] = ws[k] >= RRk ? ws[k] - RRk : ws[k];
                float* d = pdst[k] + slot;
                d[0] = v[0]; d[RRpk] = v[1]; d[2 * RRpk] = v[2]; d[3 * RRpk] = v[3];
                if (slot < 7) { d += RRk; d[0] = v[0]; d[RRpk] = v[1]; d[2 * RRpk] = v[2]; d[3 * RRpk] = v[3]; }
            }
        };
        auto hand_load = [&](int q, int c) -> float {
            /* sc0 sc1: served by L2, never by this CU's L1 (the values were stored by the table waves of this workgroup) */
            return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rLA, hbase[q] >= 0 ? hbase[q] + 32 * c : -1, 0, LFBM5D_S2_HAND_AUX));
        };
        /* blocks 0 .. lead of either ring, the hand-off values of chunk 0; then kS2LD blocks / chunks in flight */
        for (int j = 0; j <= LEADM; j++) {
            v4f v[kS2NI];
#pragma unroll
            for (int k = 0; k < kS2NI; k++) v[k] = piece_load(k, j);
#pragma unroll
            for (int k = 0; k < kS2NI; k++) if (j <= (p1[k] ? LEAD1 : LEAD2)) piece_write(k, v[k]);
        }
        v4f stg[kS2LD][kS2NI];
#pragma unroll
        for (int d = 0; d < kS2LD; d++)
#pragma unroll
            for (int k = 0; k < kS2NI; k++) stg[d][k] = piece_load(k, (p1[k] ? LEAD1 : LEAD2) + 1 + d);
#pragma unroll
        for (int q = 0; q < 2; q++) { const float h0 = hand_load(q, 0); if (hok[q]) hdst[q][0] = h0; }
        float hv[kS2LD][2];
#pragma unroll
        for (int d = 0; d < kS2LD; d++)
#pragma unroll
            for (int q = 0; q < 2; q++) hv[d][q] = hand_load(q, 1 + d);
        lds_barrier();   /* A */
#ifdef LFBM5D_PHASE_TIMING
        long long lk[2] = {0, 0};
        long long llast = (long long)__builtin_readcyclecounter();
#endif
        /* chunk cc: block cc + lead + 1 goes into the ring, the loads of block cc + lead + 1 + kS2LD start */
        auto chunk = [&](const int d, const int cc) {
            if (LFBM5D_S2_EXP & 32) { lds_barrier(); return; }   /* experiment: a loader that only keeps the barrier count */
#pragma unroll
            for (int k = 0; k < kS2NI; k++) {
                piece_write(k, stg[d][k]);
                if (!(LFBM5D_S2_EXP & 4)) stg[d][k] = piece_load(k, cc + (p1[k] ? LEAD1 : LEAD2) + 1 + kS2LD);
            }
#pragma unroll
            for (int q = 0; q < 2; q++) {
                if (hok[q]) hdst[q][((cc + 1) & 1) * NW * 8] = hv[d][q];
                hv[d][q] = hand_load(q, cc + 1 + kS2LD);
            }
#ifdef LFBM5D_PHASE_TIMING
            { const long long tn = (long long)__builtin_readcyclecounter(); lk[0] += tn - llast; llast = tn; }
#endif
            lds_barrier();
#ifdef LFBM5D_PHASE_TIMING
            { const long long tn = (long long)__builtin_readcyclecounter(); lk[1] += tn - llast; llast = tn; }
#endif
        };
        static_assert(kS2LD == 2, "the chunk count is even");
        for (int c = 0; c < nchunks; c += 2) { chunk(0, c); chunk(1, c + 1); }
#ifdef LFBM5D_PHASE_TIMING
        if (lane == 0 && a.dbg && STEREO) { atomicAdd(&a.dbg[5], (unsigned long long)lk[0]); atomicAdd(&a.dbg[11], (unsigned long long)lk[1]); }
#endif
    }
    lds_barrier();   /* the table waves' last E */
}

/* ---- a table wave ----
 * COMB (disparity search): the table values do not leave the workgroup.  The table waves hand the eight values of a chunk
 * to an exchange area in LDS, and during the next chunk every wave takes a share of the chunk's 512 positions and keeps,
 * per position, the smallest value of the workgroup's tables and its displacement (ties: the earlier in the reference's
 * scan order, core:3581-3593 -- the tables of a workgroup are sorted by it): one 8-byte (value, order) pair per position
 * and workgroup goes to memory instead of one value per position and table, and the arg-min kernel reduces over the
 * workgroups of a SAI (lfbm5d_kernels.h, stereo_part_*).  Column 0 and the row-0 entry of every strip's first column never
 * pass through the chunk loop: they go to a small per-table edge array. */
template <int K, bool STEREO, int NW, bool COMB>
__device__ __forceinline__ void scan2_table(const ScanArgs& a, const Scan2Wg& g, float* lds, const int w, const int r0) {
    typedef S2Geom<K> G;
    constexpr int CW1 = G::CW1, RR1 = G::RR1, RRp1 = G::RRp1;
    const int lane = threadIdx.x & 63;
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
    const int CW2 = G::cw2(g.ch), RR2 = G::rr2(g.rh), RRp2 = RR2 + 13;
    float* ring1 = lds;
    float* ring2 = lds + CW1 * RRp1;
    const float* lcst = ring2 + CW2 * RRp2;
    const short* rs = reinterpret_cast<const short*>(lcst + 2 * NW * 8);   /* self search: reference-grid row slot of every image row (+ 64 of padding) */
    float* const xch = const_cast<float*>(lcst) + 2 * NW * 8;              /* COMB: [2][NW][512] values of the current / previous chunk */
    if (STEREO && COMB && !live) {   /* its rows of the exchange area: +inf once, never written again (eight selects per chunk for every wave otherwise) */
        const v4f inf4 = {__builtin_inff(), __builtin_inff(), __builtin_inff(), __builtin_inff()};
#pragma unroll
        for (int ch = 0; ch < 2; ch++) {
            *reinterpret_cast<v4f*>(xch + (ch * NW + w) * 512 + lane * 8) = inf4;
            *reinterpret_cast<v4f*>(xch + (ch * NW + w) * 512 + lane * 8 + 4) = inf4;
        }
    }

    const unsigned pl1 = a.pst, pl2 = STEREO ? a.st_of_slot[g.slot] : a.pst;
    const float* img1 = a.est + (size_t)pl1 * WH;
    const float* img2 = a.est + (size_t)pl2 * WH;
    const int dk = dioff * W + djoff;
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc((void*)img1, 0, (int)(WH * 4 + 1024), kRsrcFlags);
    const __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc((void*)(img2 + dk), 0, (int)(WH * 4 + 1024), kRsrcFlags);
    /* outputs: waves without a table of their own (the last workgroup of a class) run along with stores that go nowhere */
    const int tglob = STEREO ? (int)a.n_self + g.slot * ncand + tb : tb;
    float* lcolT = a.lcol + (size_t)tglob * a.lcol_stride;
    const __amdgpu_buffer_rsrc_t rL = __builtin_amdgcn_make_buffer_rsrc((void*)lcolT, 0, live ? (int)(a.lcol_stride * 4) : 0, kRsrcFlags);
    const size_t tstride = (STEREO && !COMB) ? stereo_table_stride2(a.W, a.H, a.k, a.nDisp) : 0;
    float* table = (STEREO && !COMB) ? a.tables + (size_t)(g.slot * ncand + tb) * tstride : nullptr;
    const __amdgpu_buffer_rsrc_t rT = __builtin_amdgcn_make_buffer_rsrc((void*)table, 0, (STEREO && !COMB && live && !(LFBM5D_S2_EXP & 1)) ? (int)(tstride * 4) : 0, kRsrcFlags);
    /* COMB: the workgroup's (value, order) pairs [strip][chunk][512] and this table's edge array [rows][strips] */
    const unsigned NCH = stereo_part_chunks(a.H, a.k, a.nDisp);
    const size_t pstride = COMB ? stereo_part_stride(a.W, a.H, a.k, a.nDisp) : 0;
    const size_t estride = COMB ? stereo_edge_stride(a.W, a.H, a.k, a.nDisp) : 0;
    const unsigned n_slots = STEREO ? a.n_stereo / (unsigned)ncand : 0;
    float* part = COMB ? a.tables + ((size_t)g.slot * a.nwg_slot + g.wgj) * pstride * 2 : nullptr;
    float* edge = COMB ? a.tables + (size_t)n_slots * a.nwg_slot * pstride * 2 + (size_t)(g.slot * ncand + tb) * estride : nullptr;
    const __amdgpu_buffer_rsrc_t rP = __builtin_amdgcn_make_buffer_rsrc((void*)part, 0, (COMB && !(LFBM5D_S2_EXP & 1)) ? (int)(pstride * 8) : 0, kRsrcFlags);
    const __amdgpu_buffer_rsrc_t rEd = __builtin_amdgcn_make_buffer_rsrc((void*)edge, 0, (COMB && live) ? (int)(estride * 4) : 0, kRsrcFlags);
    /* COMB: the scan order dj * Ns + di of the workgroup's tables (packed from wave 0) comes with the descriptor: scalar loads
     * (an array computed here ends up in scratch, indexed by a chain of selects) */
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
            if (STEREO && COMB) {
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), rEd, on ? i * 4 : -1, 0, 0);
            } else if (STEREO) {
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

        /* E: the hand-off column of the previous strip (first strip: the first column) has landed; the loader wave may
         * refill the rings */
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lds_barrier();
        lds_barrier();   /* A: rows 0 .. K + 7 of the strip are in the rings, the hand-off values of chunk 0 staged */

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
        if (STEREO && COMB) {
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, S0), rEd, lane == 0 ? (nrows + strip) * 4 : -1, 0, 0);
        } else if (STEREO) {
            /* lanes 1.. : through the main loop (a lane's "result" of the step before its first is its row-0 value) */
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, S0), rT, lane == 0 ? (strip * SRq * 256 + 3) * 4 : -1, 0, 0);
        } else {
            const int ry = s2_grid_index(b, gR, lastR, gN, gP);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, S0), rS, (ry >= 0 && base_fwd >= 0) ? ry * row_bytes + base_fwd : -1, 0, 0);
            const int ry2 = di > 0 ? s2_grid_index(b + di, gR, lastR, gN, gP) : -1;
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, S0), rS, (ry2 >= 0 && base_bwd >= 0) ? ry2 * row_bytes + base_bwd : -1, 0, 0);
        }
        /* Self search on the regular grid: a lane's column is a grid column or not for the whole strip, and the lanes on grid
         * columns share lane mod p, hence also the steps on which their row 1 + t - lane is a grid row: ONE store step in p,
         * the same lanes every time, the row slot one further each time.  That holds when every grid column of the strip
         * and the forced last row lie on the pattern nHW + i p (else: per-step table look-ups, the general form below). */
        bool fast = false;
        int nfF = 0x40000000, nfB = 0x40000000, offF = -1, offB = -1, incF = 0, incB = 0;
        unsigned boundF = 0, boundB = 0;
        if (!STEREO) {
            const bool offpat = (base_fwd >= 0 && (x - gN) % gP != 0) || (base_bwd >= 0 && (x + djs - gN) % gP != 0);
            fast = !__builtin_amdgcn_ballot_w64(offpat) && (lastR - gN) % gP == 0;
            if (fast) {
                const int tF0 = ((-(2 + 64 * strip)) % gP + gP) % gP;              /* rows 1 + t - lane of the forward lanes: grid rows */
                const int tB0 = ((-(2 + 64 * strip + djs + di)) % gP + gP) % gP;   /* rows 1 + t - lane + di of the mirrored lanes */
                nfF = tF0; nfB = di > 0 ? tB0 : 0x40000000;
                if (base_fwd >= 0) { offF = ((1 + tF0 - lane) / gP) * row_bytes + base_fwd; incF = row_bytes; }
                if (base_bwd >= 0) { offB = ((1 + tB0 - lane + di) / gP) * row_bytes + base_bwd; incB = row_bytes; }
                /* a lane stores while its row is in the table and (plus di) not past the last grid row */
                boundF = (unsigned)max(min(nrows - 2, lastR - b - 1), -1);
                boundB = (unsigned)max(min(nrows - 2, lastR - b - 1 - di), -1);
            }
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
        const float* const lcw = lcst + w * 8;
        /* COMB: this wave's share of a chunk's positions p = lane-of-the-table * 8 + step-in-chunk */
        constexpr int kShare = (512 + NW - 1) / NW;
        /* the scan orders of the workgroup's tables, lane q of a register holding table q's: the reduction tracks the INDEX of the winning
         * table (selects of inline constants) and fetches its order with one ds_bpermute.  Written as `bo = lt ? g.ord[q] : bo`, hipcc
         * turns the selects of loads into ONE vector load at a selected address -- a global (or, from a local copy, scratch) load and an
         * s_waitcnt vmcnt(0), which also waits for the wave's stores, per chunk and in front of the chunk barrier. */
        const int ordv = (int)g.ord[lane & 15];
        const int rp = w * kShare + lane;
        const bool rok = lane < kShare && rp < 512;
        auto reduce_chunk = [&](const int cprev) {
            const float* src = xch + (cprev & 1) * NW * 512 + (rok ? rp : 0);
            /* all NW rows at once (waves without a table of their own hand over +inf, which never wins the strict comparison, so
             * their rows need no test -- a test per row is a branch, a wait and an exec-masked move each, right after the
             * barrier where every wave of the CU does the same) */
            float v[NW];
#pragma unroll
            for (int q = 0; q < NW; q++) v[q] = src[q * 512];
            /* smallest value (three-operand minima), then the FIRST table that holds it: the same (value, table) as the sequential
             * `if (v[q] < best)` scan -- ties to the earlier table -- with a dependent chain of 5 + NW instead of 2 NW instructions */
            float best = v[0];
#pragma unroll
            for (int q = 1; q + 1 < NW; q += 2) best = __builtin_fminf(__builtin_fminf(best, v[q]), v[q + 1]);
            if ((NW & 1) == 0) best = __builtin_fminf(best, v[NW - 1]);
            int bq = NW - 1;
#pragma unroll
            for (int q = NW - 2; q >= 0; q--) bq = v[q] == best ? q : bq;
            const int bo = __builtin_amdgcn_ds_bpermute(bq << 2, ordv);
            typedef int v2i __attribute__((ext_vector_type(2)));
            const v2i pr = {__float_as_int(best), bo};
            __builtin_amdgcn_raw_buffer_store_b64(pr, rP, rok ? (int)(((strip * NCH + cprev) * 512 + rp) * 8) : -1, 0, LFBM5D_PAIR_STORE_AUX);
        };

        /* chain flavours: steady -- every lane active on every step, every row in the band; edge (ramp-up, ramp-down, short
         * tables).  Store flavours of the self search: pattern (above) or look-ups */
#ifdef LFBM5D_PHASE_TIMING
        long long tk[4] = {0, 0, 0, 0};
        long long tlast = (long long)__builtin_readcyclecounter();
#define S2_MARK(i) do { const long long tn = (long long)__builtin_readcyclecounter(); tk[i] += tn - tlast; tlast = tn; } while (0)
#else
#define S2_MARK(i) do {} while (0)
#endif
#if defined(LFBM5D_PHASE_TIMING) && LFBM5D_PHASE_TIMING >= 2   /* finer: inside a steady chunk of the disparity search (reported in the self search's slots) */
        long long tq[4] = {0, 0, 0, 0};
        long long qlast = 0;
#define S2_Q0() do { asm volatile("" ::: "memory"); qlast = (long long)__builtin_readcyclecounter(); } while (0)
#define S2_QMARK(i) do { asm volatile("" ::: "memory"); const long long tn = (long long)__builtin_readcyclecounter(); tq[i] += tn - qlast; qlast = tn; } while (0)
#else
#define S2_Q0() do {} while (0)
#define S2_QMARK(i) do {} while (0)
#endif
        auto body16 = [&](auto fl_tag, const int t0) {
            constexpr bool EDGE = decltype(fl_tag)::value;
            /* per-lane step numbers relative to this group of sixteen (compared with small constants below) */
            const int rel_start = lane_eff - t0;                 /* the lane's first step */
            const int rel_band = band_rows - K + lane - t0;      /* first step whose lower-edge row lies past the band */
#pragma unroll
            for (int ch = 0; ch < 2; ch++) {
                const int tc = t0 + 8 * ch;
                const v4f lc[2] = {*reinterpret_cast<const v4f*>(lcw + ch * NW * 8), *reinterpret_cast<const v4f*>(lcw + ch * NW * 8 + 4)};
                if (!EDGE) S2_Q0();
                if (COMB && tc > 0) reduce_chunk((tc >> 3) - 1);
                if (!EDGE) S2_QMARK(0);
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
                    if (STEREO && COMB) {
                        /* (a wave without a table of its own walks the first table again, without its hand-off column, and hands +inf to the reduction) */
                        if (live) *reinterpret_cast<v4f*>(xch + (ch * NW + w) * 512 + lane * 8 + 4 * gq) = out;   /* (uniform: its rows hold +inf from the start otherwise) */
                    } else if (STEREO) {
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4i, out), rT, voffT, 0, 0);
                        voffT += 1024;
                    } else if (fast) {
#pragma unroll
                        for (int s = 0; s < 4; s++) {
                            const int u = 8 * ch + 4 * gq + s;
                            const float ov = out[s];   /* (a bit_cast of the vector element itself picks element 0) */
                            if (u == nfF) {
                                __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(ov), rS, (unsigned)(u - rel_start) <= boundF ? offF : -1, 0, 0);
                                offF += incF; nfF += gP;
                            }
                            if (u == nfB) {
                                __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(ov), rS, (unsigned)(u - rel_start) <= boundB ? offB : -1, 0, 0);
                                offB += incB; nfB += gP;
                            }
                        }
                    } else {
#pragma unroll
                        for (int s = 0; s < 4; s++) {
                            const float ov = out[s];
                            {
                                const int t = tg + s;
                                const bool act = (unsigned)(t - lane_eff) <= (unsigned)(nrows - 2);
                                const int y = min(max(b + 1 + t - lane, 0), H - 1);
                                const int r1 = rs[y], r2 = rs[y + di];                  /* -1: not a grid row */
                                const int v1 = (act && (r1 | base_fwd) >= 0) ? r1 * row_bytes + base_fwd : -1;
                                const int v2 = (act && (r2 | base_bwd) >= 0) ? r2 * row_bytes + base_bwd : -1;
                                /* a store no lane takes part in is skipped (on the regular grid three steps in four) */
                                if (__builtin_amdgcn_ballot_w64(v1 != -1)) __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(ov), rS, v1, 0, 0);
                                if (__builtin_amdgcn_ballot_w64(v2 != -1)) __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(ov), rS, v2, 0, 0);
                            }
                        }
                    }
                    /* hand-off column for the next strip: rows 1 + t - 63 of this strip's last column */
                    if (!(LFBM5D_S2_EXP & 2)) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4i, out), rL, voffL, 0, 0);
                    voffL += 16;
                    if (!EDGE) S2_QMARK(1 + gq);
                }
                slot1 += 8; slot1 = min(slot1, slot1 - (unsigned)RR1);
                slot2 += 8; slot2 = min(slot2, slot2 - (unsigned)RR2);
                S2_MARK(EDGE ? 1 : 0);
                lds_barrier();
                S2_MARK(2);
            }
            if (!STEREO) { nfF -= 16; nfB -= 16; }
        };

        const int nsteps = (nrows - 1) + last_lane;
        const int tS1 = (min(nrows - 1, band_rows - K) / 16) * 16;   /* steady chunks end before lane 0 stops / the band ends */
        auto run = [&]() {
            int t0 = 0;
            for (; t0 < nsteps && t0 < 64; t0 += 16) body16(std::true_type{}, t0);
            for (; t0 < tS1; t0 += 16) body16(std::false_type{}, t0);
            for (; t0 < nsteps; t0 += 16) body16(std::true_type{}, t0);
        };
        S2_MARK(3);   /* strip prologue: barriers E / A, first row, FIFO */
        run();
        if (COMB && nsteps > 0) reduce_chunk(((nsteps + 15) >> 4) * 2 - 1);   /* the last chunk's values (the barrier behind it has been passed) */
        row0_left = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(S0), last_lane));
#ifdef LFBM5D_PHASE_TIMING
#if LFBM5D_PHASE_TIMING >= 2
        if (lane == 0 && a.dbg && live && STEREO) {
            for (int i = 0; i < 4; i++) atomicAdd(&a.dbg[i], (unsigned long long)tk[i]);
            atomicAdd(&a.dbg[4], 1ull);
            for (int i = 0; i < 3; i++) atomicAdd(&a.dbg[6 + i], (unsigned long long)tq[i]);
        }
#else
        if (lane == 0 && a.dbg && live) {
            const int base = STEREO ? 0 : 6;
            for (int i = 0; i < 4; i++) atomicAdd(&a.dbg[base + i], (unsigned long long)tk[i]);
            atomicAdd(&a.dbg[base + 4], 1ull);
        }
#endif
#endif
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();   /* the loader wave's last E */
}

template <int K, bool STEREO, int NW, int NL, bool COMB>
__device__ __forceinline__ void scan2_body(const ScanArgs& a, const Scan2Wg& g, float* lds) {
    typedef S2Geom<K> G;
    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);   /* wave-uniform: everything derived from the wave's table stays scalar */
    const int half = STEREO ? (int)a.nDisp : (int)a.nSim;
    const int Ns = 2 * half + 1;
    /* row phase of ring 2: (djoff - c2lo) + (dioff - r2lo) + r0 = 0 (mod 4), the same for every table of the workgroup */
    int r0;
    {
        const int t0 = g.tab[0], di0 = t0 / Ns, dj0 = t0 % Ns;
        r0 = (4 - (((dj0 - half) - g.c2lo + (STEREO ? di0 - half : di0) - g.r2lo) & 3)) & 3;
    }
    if (!STEREO) {
        float* ring2 = lds + G::CW1 * G::RRp1;
        short* rs = reinterpret_cast<short*>(ring2 + G::cw2(g.ch) * (G::rr2(g.rh) + 13) + 2 * NW * 8);
        for (int i = tid; i < (int)a.H + 64; i += blockDim.x) rs[i] = (short)a.rslot[i];
    }
    if (w >= NW) scan2_loader<K, STEREO, NW, NL>(a, g, lds, r0, w - NW);
    else scan2_table<K, STEREO, NW, COMB>(a, g, lds, w, r0);
}

template <int K, bool COMB>
__global__ __launch_bounds__((kS2NW + kS2NL) * 64) void k_bm_scan2(ScanArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds2[];
    const Scan2Wg& g = a.wgs[blockIdx.x];
    if ((LFBM5D_S2_EXP & 8) && g.slot < 0) return;     /* experiments: disparity tables only / self tables only */
    if ((LFBM5D_S2_EXP & 16) && g.slot >= 0) return;
#ifdef LFBM5D_SCAN2_ONLY_STEREO
    scan2_body<K, true, kS2NW, kS2NL, COMB>(a, g, lds2);
#elif defined(LFBM5D_SCAN2_ONLY_SELF)
    scan2_body<K, false, kS2NW, kS2NL, false>(a, g, lds2);
#else
    if (g.slot >= 0) scan2_body<K, true, kS2NW, kS2NL, COMB>(a, g, lds2);
    else scan2_body<K, false, kS2NW, kS2NL, false>(a, g, lds2);
#endif
}

/* argmin over the (2 nDisp+1)^2 displacement tables (core:3581-3608) in the second-generation layout
 * [strip][Q / 4][lane][Q % 4], Q = table row + lane + 3: a thread takes the four entries of one lane (one 16-byte load per
 * table) = four consecutive rows of one column; ties keep scan order (dj outer, di inner). */
struct Argmin2Args { const float* tables; size_t tstride; unsigned st_of_slot[kBigA]; int W, H, k, nDisp, SRq; float thr; unsigned* best; unsigned char* shape; };
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
    const size_t tail = std::max<size_t>(a.n_self ? (size_t)(a.H + 64) * sizeof(short) : 0, a.n_stereo ? (size_t)2 * kS2NW * 512 * sizeof(float) : 0);
    return (size_t)(G::CW1 * G::RRp1 + G::cw2(ch_max) * RRp2 + 2 * kS2NW * 8) * sizeof(float) + tail + 16;
}

} /* namespace */

/* Which kernel generation a pass's distance tables are built with (and hence the layout the arg-min reads):
 * 2 unless the configuration is outside what the ring-sharing kernel covers (12x12 patches, irregular reference lists,
 * search windows whose rings do not fit) or the option scan_v1 asks for round 2's kernel. */
int bm_scan_version(const ScanArgs& a) {
    if (a.opt & (kOptScanV1 | kOptScanAny)) return 1;   /* (test hooks: round 2's kernel / the any-patch-size kernel, first-generation layout) */
    if (a.k != 8 && a.k != 16) return 1;
    if (a.n_self && a.refmap) return 1;
    /* the loader wave addresses all planes of the estimate through ONE buffer resource with 32-bit offsets */
    if ((size_t)a.est_planes * a.W * a.H * 4 + kLeadBytes + 1024 > 0x7fffffffull) return 1;
    std::vector<Scan2Wg> wgs; size_t lds = 0;
    if (!scan2_plan(a, wgs, &lds)) return 1;
    if (a.opt & kOptScanFullTables) return 2;
    return 3;
}

size_t scan_tables_floats(const ScanArgs& a, int version, unsigned n_slots, unsigned nwg_slot) {
    const size_t ncand = (size_t)(2 * a.nDisp + 1) * (2 * a.nDisp + 1);
    if (version == 1) return (size_t)n_slots * ncand * stereo_table_stride(a.W, a.H, a.k, a.nDisp);
    if (version == 2) return (size_t)n_slots * ncand * stereo_table_stride2(a.W, a.H, a.k, a.nDisp);
    return (size_t)n_slots * nwg_slot * stereo_part_stride(a.W, a.H, a.k, a.nDisp) * 2 + (size_t)n_slots * ncand * stereo_edge_stride(a.W, a.H, a.k, a.nDisp);
}

/* Workgroups of a launch: up to eight tables of one image pair and one class (di + dj) mod 4. */
bool scan2_plan(const ScanArgs& a, std::vector<Scan2Wg>& wgs, size_t* lds_bytes, unsigned* nwg_slot) {
    wgs.clear();
    int rh_max = 0, ch_max = 0;
    int wgj = 0;
    auto add = [&](int slot, const std::vector<int>& tabs, int r2lo, int c2lo, int rh, int ch) {
        for (size_t i = 0; i < tabs.size(); i += kS2NW) {
            Scan2Wg g;
            std::memset(&g, 0, sizeof(g));
            for (int w = 0; w < 16; w++) g.tab[w] = (w < kS2NW && i + w < tabs.size()) ? (short)tabs[i + w] : (short)-1;
            if (slot >= 0) { const int Ns = 2 * (int)a.nDisp + 1; for (int w = 0; w < 16; w++) g.ord[w] = g.tab[w] >= 0 ? (short)((g.tab[w] % Ns) * Ns + g.tab[w] / Ns) : (short)0; }
            g.slot = (short)slot; g.r2lo = (short)r2lo; g.c2lo = (short)c2lo; g.rh = (short)rh; g.ch = (short)ch; g.wgj = (short)wgj++;
            wgs.push_back(g);
        }
        rh_max = std::max(rh_max, rh); ch_max = std::max(ch_max, ch);
    };
    if (nwg_slot) *nwg_slot = 0;
    if (a.n_self) {
        /* self search: di in [0, nSim], dj in [0, 2 nSim]; tiles of 4 x kS2NW displacements, one workgroup per class of a
         * tile (kS2NW tables): the second ring then spans 3 more rows and kS2NW - 1 more columns than the first */
        const int nSim = (int)a.nSim, Ns = 2 * nSim + 1;
        for (int d0 = 0; d0 <= nSim; d0 += 4)
            for (int j0 = 0; j0 < Ns; j0 += kS2NW)
                for (int cls = 0; cls < 4; cls++) {
                    std::vector<int> tabs;
                    for (int di = d0; di < std::min(d0 + 4, nSim + 1); di++)
                        for (int dj = j0; dj < std::min(j0 + kS2NW, Ns); dj++)
                            if (((di + dj) & 3) == cls) tabs.push_back(di * Ns + dj);
                    if (!tabs.empty()) add(-1, tabs, d0, j0 - nSim, 3, kS2NW - 1);
                }
    }
    if (a.n_stereo) {
        const int nD = (int)a.nDisp, Ns = 2 * nD + 1, ncand = Ns * Ns;
        const int n_slots = (int)a.n_stereo / ncand;
        for (int slot = 0; slot < n_slots; slot++) {
            wgj = 0;
            for (int cls = 0; cls < 4; cls++) {
                /* in the reference's scan order dj * Ns + di (core:3581-3593): the combined form resolves ties inside a
                 * workgroup by position in this list */
                std::vector<int> tabs;
                for (int ord = 0; ord < ncand; ord++) {
                    const int dj = ord / Ns, di = ord % Ns;
                    if (((di + dj) & 3) == cls) tabs.push_back(di * Ns + dj);
                }
                if (!tabs.empty()) add(slot, tabs, -nD, -nD, 2 * nD, 2 * nD);
            }
            if (nwg_slot) *nwg_slot = (unsigned)wgj;
        }
    }
    size_t lds = a.k == 8 ? scan2_lds_bytes<8>(a, rh_max, ch_max) : scan2_lds_bytes<16>(a, rh_max, ch_max);
    if (lds_bytes) *lds_bytes = lds;
    /* at most kS2NI 16-byte pieces per loader lane and chunk; the rings within the CU's LDS */
    const int n1 = 2 * (64 + (int)a.k), n2 = 2 * ((64 + (int)a.k + ch_max + 3) & ~3);
    if (n1 + n2 > 64 * kS2NI || lds > 160 * 1024) return false;
    return !wgs.empty();
}

unsigned scan2_lcol_stride(const ScanArgs& a) {
    const unsigned rows_self = a.n_self ? a.H - 2 * a.nHW : 0, rows_st = a.n_stereo ? a.H - 2 * a.nDisp - (a.k - 1) : 0;
    return ((std::max(rows_self, rows_st) + 64 + 160 + 63) / 64) * 64;
}

/* The dynamic-LDS limit is a per-DEVICE function attribute: raised for the current device, by every context creation
 * (lfbm5d_create after hipSetDevice, and the lane contexts) -- not once per process. */
hipError_t prepare_scan2_kernels() {
    const void* fns[] = {reinterpret_cast<const void*>(&k_bm_scan2<8, false>), reinterpret_cast<const void*>(&k_bm_scan2<16, false>),
                         reinterpret_cast<const void*>(&k_bm_scan2<8, true>), reinterpret_cast<const void*>(&k_bm_scan2<16, true>)};
    for (const void* f : fns) {
        const hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

hipError_t launch_bm_scan2(hipStream_t s, const ScanArgs& a, size_t lds, bool combined) {
    if (!a.n_wgs) return hipSuccess;
    const dim3 blk((kS2NW + kS2NL) * 64);
    if (a.k == 8 && combined) hipLaunchKernelGGL((k_bm_scan2<8, true>), dim3(a.n_wgs), blk, lds, s, a);
    else if (a.k == 8) hipLaunchKernelGGL((k_bm_scan2<8, false>), dim3(a.n_wgs), blk, lds, s, a);
    else if (a.k == 16 && combined) hipLaunchKernelGGL((k_bm_scan2<16, true>), dim3(a.n_wgs), blk, lds, s, a);
    else if (a.k == 16) hipLaunchKernelGGL((k_bm_scan2<16, false>), dim3(a.n_wgs), blk, lds, s, a);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

/* arg-min of the combined form: over the (value, order) pairs of a slot's workgroups, position by position; the edge
 * positions (column 0, row 0 of every strip's first column) over the per-table edge arrays.  Ties: the smaller scan order
 * (core:3581-3608). */
struct Argmin3Args { const float* tables; unsigned st_of_slot[kBigA]; int W, H, k, nDisp, n_slots, nwg_slot; float thr; unsigned* best; unsigned char* shape; };
__global__ __launch_bounds__(256) void k_stereo_argmin3(Argmin3Args a) {
    const int W = a.W, H = a.H, nDisp = a.nDisp;
    const int span_c = W - 2 * nDisp - a.k + 1, span_r = H - 2 * nDisp - a.k + 1;
    const int nstrips = (span_c - 1 + 63) / 64;
    const int NCH = (int)stereo_part_chunks(H, a.k, nDisp);
    const size_t pstride = stereo_part_stride(W, H, a.k, nDisp), estride = stereo_edge_stride(W, H, a.k, nDisp);
    const int Ns = 2 * nDisp + 1, ncand = Ns * Ns;
    const unsigned slot = blockIdx.y, st = a.st_of_slot[slot];
    const size_t WH = (size_t)W * H;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int n_main = nstrips * NCH * 256, n_edge = span_r + nstrips;      /* main: two consecutive steps of a lane per thread (16-byte loads) */
    if (i >= n_main + 16 * n_edge) return;
    auto put = [&](int row, int col, float bv, int bo) {
        const int pos = (nDisp + row) * W + nDisp + col;
        const int dj = bo / Ns, di = bo % Ns;
        a.best[(size_t)st * WH + pos] = (unsigned)(pos + (di - nDisp) * W + (dj - nDisp));
        a.shape[(size_t)st * WH + pos] = bv < a.thr ? 1 : 0;
    };
    if (i < n_main) {
        const int p = 2 * (i & 255), c = (i >> 8) % NCH, strip = (i >> 8) / NCH;
        const int l = p >> 3, t = 8 * c + (p & 7);
        const int row = 1 + t - l, col = 1 + 64 * strip + l;      /* rows `row` and `row + 1` */
        if (row + 1 < 0 || row >= span_r || col >= span_c) return;
        const v4f* src = reinterpret_cast<const v4f*>(a.tables) + ((size_t)slot * a.nwg_slot * pstride >> 1) + i;
        float bv0 = __builtin_inff(), bv1 = __builtin_inff(); int bo0 = 0x7fffffff, bo1 = 0x7fffffff;
        /* sixteen loads in flight -- all of them at nDisp = 6 (a workgroup past the last repeats the last: a no-op for the minimum) */
        for (int j0 = 0; j0 < a.nwg_slot; j0 += 16) {
            v4f w8[16];
#pragma unroll
            for (int u = 0; u < 16; u++) {
#if LFBM5D_PAIR_LOAD_NT
                w8[u] = __builtin_nontemporal_load(src + (size_t)min(j0 + u, a.nwg_slot - 1) * (pstride >> 1));
#else
                w8[u] = src[(size_t)min(j0 + u, a.nwg_slot - 1) * (pstride >> 1)];
#endif
            }
#pragma unroll
            for (int u = 0; u < 16; u++) {
                const int o0 = __float_as_int(w8[u][1]), o1 = __float_as_int(w8[u][3]);
                if (w8[u][0] < bv0 || (w8[u][0] == bv0 && o0 < bo0)) { bv0 = w8[u][0]; bo0 = o0; }
                if (w8[u][2] < bv1 || (w8[u][2] == bv1 && o1 < bo1)) { bv1 = w8[u][2]; bo1 = o1; }
            }
        }
        if (row >= 0 && !(row == 0 && l == 0)) put(row, col, bv0, bo0);
        if (row + 1 < span_r && row + 1 >= 0 && !(row + 1 == 0 && l == 0)) put(row + 1, col, bv1, bo1);
    } else {
        /* edge positions: sixteen lanes per position, each over every sixteenth table, then a 16-lane minimum */
        const int e = (i - n_main) >> 4, sub = (i - n_main) & 15;
        int row, col;
        if (e < span_r) { row = e; col = 0; } else { row = 0; col = 1 + 64 * (e - span_r); }
        const bool ok = e < n_edge && col < span_c;
        const float* ed = a.tables + (size_t)a.n_slots * a.nwg_slot * pstride * 2 + (size_t)slot * ncand * estride + (ok ? e : 0);
        float bv = __builtin_inff(); int bo = 0x7fffffff;
        for (int d0 = sub; d0 < ncand; d0 += 16 * 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = ed[(size_t)min(d0 + 16 * u, ncand - 1) * estride];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int ddk = d0 + 16 * u;
                const int o = (ddk % Ns) * Ns + ddk / Ns;
                if (ddk < ncand && (v[u] < bv || (v[u] == bv && o < bo))) { bv = v[u]; bo = o; }
            }
        }
#pragma unroll
        for (int m = 8; m > 0; m >>= 1) {
            const float ov = __shfl_xor(bv, m, 16); const int oo = __shfl_xor(bo, m, 16);
            if (ov < bv || (ov == bv && oo < bo)) { bv = ov; bo = oo; }
        }
        if (ok && sub == 0) put(row, col, bv, bo);
    }
}

hipError_t launch_stereo_argmin3(hipStream_t s, const float* tables, const unsigned* st_of_slot, unsigned n_slots, unsigned nwg_slot,
                                 unsigned W, unsigned H, unsigned k, unsigned nDisp, float thr,
                                 unsigned* best, unsigned char* shape) {
    Argmin3Args a;
    a.tables = tables; a.W = (int)W; a.H = (int)H; a.k = (int)k; a.nDisp = (int)nDisp; a.n_slots = (int)n_slots; a.nwg_slot = (int)nwg_slot; a.thr = thr; a.best = best; a.shape = shape;
    for (unsigned i = 0; i < n_slots && i < (unsigned)kBigA; i++) a.st_of_slot[i] = st_of_slot[i];
    const unsigned span_c = W - 2 * nDisp - k + 1, span_r = H - 2 * nDisp - k + 1;
    const unsigned nstrips = (span_c - 1 + 63) / 64;
    const unsigned n = nstrips * stereo_part_chunks(H, k, nDisp) * 256 + 16 * (span_r + nstrips);
    hipLaunchKernelGGL(k_stereo_argmin3, dim3((n + 255) / 256, n_slots), dim3(256), 0, s, a);
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
    for (unsigned i = 0; i < n_slots && i < (unsigned)kBigA; i++) a.st_of_slot[i] = st_of_slot[i];
    hipLaunchKernelGGL(k_stereo_argmin2, dim3((n + 255) / 256, n_slots), dim3(256), 0, s, a);
    return hipGetLastError();
}

} /* namespace lfbm5d */
