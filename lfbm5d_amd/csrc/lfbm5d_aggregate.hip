/*
 * lfbm5d_aggregate.hip -- the aggregation stage of a core pass (core:484-528) for gfx950: a gather, not a scatter -- one
 * wavefront per tile of 64 pixels of one SAI adds every filtered patch that overlaps the tile, in the reference's own order
 * (reference patches in raster order, then match index), so num / den are reproducible run to run and need no float atomics.
 * Split from lfbm5d_kernels.hip in round 5.
 */
#include "lfbm5d_kernels.h"

#include <algorithm>
#include <cstdlib>

namespace lfbm5d {

namespace {

/* ================================ aggregation kernel ====================================== */


/* Gather form of core:484-528.  One wavefront = one tile of 64 pixels of one SAI (thread = pixel).
 * Candidates are the patch instances (reference patch in raster order, then match index n) of the
 * reference patches whose search range can reach the tile; every pixel adds its contributions in
 * that order -- the reference's order for that pixel -- starting from the value already in num/den.
 * Everything is wave-synchronous and built to keep many loads in flight (the kernel is a chain of
 * dependent gathers, so latency, not bandwidth, is what has to be hidden):
 *   scan     kAggPF chunks of 64 candidates at a time: their aggregation positions are loaded
 *            together, then the group weights of the hits, then the hits are appended in candidate
 *            order (ballot prefix) to a hit list in LDS: position, patch offset in filt, weights;
 *   consume  once the list holds enough hits (or is full, or at the end) all lanes walk it in order, kAggU hits
 *            per round: the loads of a round are issued together, the adds stay in list order.
 * Workgroups are renumbered so that the tiles one XCD works on at a time are neighbours: the
 * filtered patches they share are then fetched into that XCD's L2 once. */
constexpr int kAggFlush = 64;   /* hits that make a consume phase worth starting */
/* WINDOWED: Kaiser window (k = 8, 12); any other size has an all-ones window (bm3d.cpp:1144-1146).
 * TW x TH: tile shape (64 pixels).  A filtered patch row is k floats, so wide flat tiles read longer
 * contiguous runs of it: 16x4 for k >= 12 (64-byte rows), 8x8 for k = 8 (a whole patch is two cache lines).
 * kAggPF chunks per scan round and kAggU hits per load round trade latency hiding against registers and LDS
 * (occupancy): the k = 8 pass has few hits per candidate and wants occupancy, the k = 16 pass deeper rounds.
 * BIG: filt is 4 GiB or more (windows far beyond 560 x 560): 64-bit gather addresses instead of a buffer resource. */
template <bool WINDOWED, int TW, int TH, int kAggPF, int kAggU, bool BIG, bool VEC4>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu((VEC4 && TW == 8 && !BIG) ? 8 : 1))) void k_aggregate(AggArgs a) {
    constexpr int kAggCap = kAggFlush + kAggPF * 64 + kAggU;   /* + padding of the last round */
    __shared__ uint4 hit_a[kAggCap];       /* (py << 16) | px, offset of the patch in filt, weights of channels 0 and 1 */
    __shared__ float hit_w2[kAggCap];      /* weight of channel 2 */
    __shared__ float kai[WINDOWED ? 256 : 1];   /* Kaiser windows exist for 8x8 and 12x12 patches only (bm3d.cpp:1101-1146) */
    const int lane = threadIdx.x;
    /* XCD-aware renumbering: hardware deals consecutive workgroup ids round-robin to the 8 XCDs */
    const unsigned gx = (a.Wb + TW - 1) / TW, gy = (a.Hb + TH - 1) / TH, total_wg = gx * gy * a.A;
    const unsigned per_xcd = gridDim.x / 8;          /* the launch is rounded up to a multiple of 8 workgroups */
    const unsigned lin2 = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
    if (lin2 >= total_wg) return;
    const int st = (int)(lin2 / (gx * gy));
    const int tile_y = (int)((lin2 / gx) % gy), tile_x = (int)(lin2 % gx);
    if (a.proc_bits.test((unsigned)st)) return;      /* procSAI[st] != 0: skipped (core:486) */
    if (!a.mask_bits.test((unsigned)st)) return;
    const int tx0 = tile_x * TW, ty0 = tile_y * TH;
    const int x = tx0 + lane % TW, y = ty0 + lane / TW;
    const bool inside = x < (int)a.Wb && y < (int)a.Hb;
    const int k = a.k, k2 = k * k, C = a.C, N = a.N, A = a.A;
    const int logN = 31 - __builtin_clz((unsigned)N);   /* N is a power of two (lfbm5d_api.hip validate) */
    const size_t plane = (size_t)a.Wb * a.Hb;
    const int reach = (st == (int)a.pst) ? (int)a.nSim : (int)a.nHW;
    if (WINDOWED) for (int i = lane; i < k2; i += 64) kai[i] = a.tb->kaiser[i];

    /* reference-grid index ranges that can reach this tile (grid = nHW + i*p, plus a forced last
     * index, utilities.cpp:697-712) */
    const int last_r = (int)a.Hb - k - (int)a.nHW, last_c = (int)a.Wb - k - (int)a.nHW;
    auto lo_idx = [&](int v) { int d = v - (int)a.nHW; return d <= 0 ? 0 : (d + (int)a.p - 1) / (int)a.p; };
    auto hi_idx = [&](int v, int n, int lastv) { /* largest index whose coordinate <= v */
        if (v >= lastv) return n - 1;
        int d = v - (int)a.nHW; if (d < 0) return -1;
        int i = d / (int)a.p; return i > n - 2 ? n - 2 : i;
    };
    int r_lo = lo_idx(ty0 - k + 1 - reach), r_hi = hi_idx(ty0 + TH - 1 + reach, (int)a.n_ref_rows, last_r);
    int c_lo = lo_idx(tx0 - k + 1 - reach), c_hi = hi_idx(tx0 + TW - 1 + reach, (int)a.n_ref_cols, last_c);
    if (r_lo > (int)a.n_ref_rows - 1) r_lo = (int)a.n_ref_rows - 1; /* the forced last index may sit closer than p */
    if (c_lo > (int)a.n_ref_cols - 1) c_lo = (int)a.n_ref_cols - 1;
    /* a launch covers the groups [ref_begin, ref_begin + n_groups): whole rows of the reference grid (a band of a pass processed band by
     * band, a rank's share of a row-sharded pass) -- only those rows' candidates are enumerated, and a tile none of them can reach
     * returns before it has touched num / den.  Bands launched in raster order add up in the reference's order (core:484-528): the sums
     * do not depend on how a pass is cut. */
    if (!a.irregular && a.n_ref_cols) {
        r_lo = max(r_lo, (int)(a.ref_begin / a.n_ref_cols));
        r_hi = min(r_hi, (int)((a.ref_begin + a.n_groups - 1) / a.n_ref_cols));
    }
    const int ncols_span = c_hi - c_lo + 1;
    /* (an irregular list -- subset passes -- is in raster order too: the launch's slice of it) */
    const int n_rr = a.irregular ? (int)a.n_groups : ((r_hi >= r_lo && c_hi >= c_lo) ? (r_hi - r_lo + 1) * ncols_span : 0);
    const int n_cand = n_rr << logN;
    if (n_cand == 0) return;

    float accn[3] = {0, 0, 0}, accd[3] = {0, 0, 0};
    const size_t pix = (size_t)st * C * plane + (size_t)y * a.Wb + x;
    if (inside) for (int c = 0; c < C; c++) { accn[c] = a.num[pix + c * plane]; accd[c] = a.den[pix + c * plane]; }

    /* rr / ncols_span by multiplication: exact while rr < 2^20 / ncols_span (rr is a few hundred) */
    const bool mul_div = !a.irregular && (long long)n_rr * ncols_span < (1 << 20);
    const unsigned div_m = ((1u << 20) + (unsigned)max(ncols_span, 1) - 1) / (unsigned)max(ncols_span, 1);
    const unsigned g_end = a.ref_begin + a.n_groups;
    /* every patch index (g N + n) A + st below 2^24: full-rate 24-bit multiplies for the per-candidate index arithmetic */
    /* SAI-major filt (wide windows, kernels.h filt_patch): this SAI's patches are a filt of their own, [g][n][c][k2] */
    const bool sai_major = a.filt_sai_stride != 0;
    const unsigned Af = sai_major ? 1u : (unsigned)A, stf = sai_major ? 0u : (unsigned)st;
    const float* const filt = a.filt + (sai_major ? (size_t)st * a.filt_sai_stride : (size_t)0);
    const unsigned long long filt_bytes = sai_major ? a.filt_sai_stride * sizeof(float) : a.filt_bytes;
    const bool small24 = (((unsigned long long)a.n_refs_total << logN) + 1) * Af < (1ull << 24);
    const unsigned* apos = a.aggpos + (size_t)st * a.n_refs_total * N;
    /* channel stride inside a filtered patch, bytes; greyscale: all three loads read channel 0 (its weights are 0) */
    const unsigned cstride = C > 1 ? (unsigned)k2 * 4u : 0u;
    const __amdgpu_buffer_rsrc_t rs_filt = __builtin_amdgcn_make_buffer_rsrc((void*)filt, 0, BIG ? 0 : (int)(unsigned)filt_bytes, 0x00020000u);
    unsigned nh = 0;   /* hits in the list (uniform) */

    auto consume = [&]() {
        /* pad the list to whole rounds with entries no pixel is covered by (position 0xffff, 0xffff; first patch; weight 0) */
        const unsigned nh_pad = (nh + kAggU - 1) / kAggU * kAggU;
        if (nh + lane < nh_pad) { hit_a[nh + lane] = make_uint4(0xffffffffu, 0u, 0u, 0u); hit_w2[nh + lane] = 0.0f; }
        __builtin_amdgcn_wave_barrier();
        for (unsigned h0 = 0; h0 < nh_pad; h0 += kAggU) {
            float val[kAggU][3], kw[kAggU][3];
#pragma unroll
            for (int u = 0; u < kAggU; u++) {
                const uint4 ha = hit_a[h0 + u];
                const float w2 = hit_w2[h0 + u];
                const int dy = y - (int)(ha.x >> 16), dx = x - (int)(ha.x & 0xffffu);
                const bool on = (unsigned)dy < (unsigned)k && (unsigned)dx < (unsigned)k;
                /* pixels the patch does not cover read the nearest pixel it does cover -- a pixel of this tile,
                 * so no extra cache line is touched ... */
                const unsigned o = __umul24((unsigned)min(max(dy, 0), k - 1), (unsigned)k) + (unsigned)min(max(dx, 0), k - 1);   /* v_mad_u32_u24: a 32-bit multiply is quarter rate */
                const float kz = WINDOWED ? kai[o] : 1.0f;
                if (BIG) {
                    const char* fp = reinterpret_cast<const char*>(filt) + ((size_t)ha.y + o) * 4;
                    val[u][0] = *reinterpret_cast<const float*>(fp);
                    val[u][1] = *reinterpret_cast<const float*>(fp + cstride);
                    val[u][2] = *reinterpret_cast<const float*>(fp + 2 * cstride);
                } else {
                    const int vo = (int)((ha.y + o) * 4u);
#if defined(LFBM5D_AGG_EXP) && LFBM5D_AGG_EXP == 1     /* timing experiment (results garbage): one gather per hit instead of three */
                    val[u][0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_filt, vo, 0, 0));
                    val[u][1] = val[u][0]; val[u][2] = val[u][0];
#elif defined(LFBM5D_AGG_EXP) && LFBM5D_AGG_EXP == 2   /* timing experiment: one 12-byte gather per hit (as if the channels were interleaved) */
                    { typedef float v3f __attribute__((ext_vector_type(3)));
                      const v3f t3 = __builtin_bit_cast(v3f, __builtin_amdgcn_raw_buffer_load_b96(rs_filt, (int)(ha.y * 4u + o * 12u), 0, 0));
                      val[u][0] = t3[0]; val[u][1] = t3[1]; val[u][2] = t3[2]; }
#else
                    /* (LFBM5D_FILT_LOAD_AUX = 2, non-temporal: 0.97 -> 1.5 ms per HT pass -- the neighbouring tiles' second reads of a row must find it in L2 / the Infinity Cache) */
                    val[u][0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_filt, vo, 0, LFBM5D_FILT_LOAD_AUX));
                    val[u][1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_filt, vo, (int)cstride, LFBM5D_FILT_LOAD_AUX));
                    val[u][2] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_filt, vo, (int)(2 * cstride), LFBM5D_FILT_LOAD_AUX));
#endif
                }
                kw[u][0] = on ? kz * __uint_as_float(ha.z) : 0.0f;   /* ... and add it with weight zero */
                kw[u][1] = on ? kz * __uint_as_float(ha.w) : 0.0f;
                kw[u][2] = on ? kz * w2 : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < kAggU; u++)
#pragma unroll
                for (int c = 0; c < 3; c++) {
#pragma clang fp contract(off)
                    accn[c] += kw[u][c] * val[u][c];   /* core:516-520 */
                    accd[c] += kw[u][c];
                }
        }
        nh = 0;
        __builtin_amdgcn_wave_barrier();
    };

    __builtin_amdgcn_wave_barrier();
    if (VEC4) {
        /* N a multiple of four (round 4): a lane takes FOUR consecutive candidates -- matches n0 .. n0 + 3 of ONE reference patch -- with
         * one 16-byte load of their aggregation positions; the group index, the weights and the patch offset are per lane instead of per
         * candidate, the tile test is two unsigned range checks per candidate.  Hits are appended in candidate order (lane-major: an
         * exclusive prefix of the lanes' hit counts from three ballots) -- the list the consume phase walks is the same as before, entry
         * for entry.  A super-chunk of 256 candidates can hold more hits than the list has room for: lanes are then taken in
         * runs that fit, with a consume phase in between. */
        constexpr unsigned cap_hits = (unsigned)(kAggCap - kAggU);
        const unsigned lo_y = (unsigned)(ty0 - k + 1), lo_x = (unsigned)(tx0 - k + 1);          /* (wrap around for tiles at the border: the */
        const unsigned span_y = (unsigned)(TH + k - 1), span_x = (unsigned)(TW + k - 1);        /*  unsigned test below still means lo <= v < lo + span) */
        const unsigned pstep = Af * (unsigned)(C * k2);                                          /* filt offset from match n to n + 1 */
        /* the candidates c0 + 4 lane .. + 3 of the lanes [l0, l1): test, and append the hits if the list has room (else: false, nothing
         * appended).  Nothing computed here is alive across a consume phase -- that is what keeps the kernel at its wave count. */
        auto try_append = [&](const int c0, const unsigned l0, const unsigned l1) -> bool {
            const int e = c0 + lane * 4;
            unsigned g = 0, n0 = 0;
            uint4 p4 = make_uint4(0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu);
            if (e < n_cand && (unsigned)lane >= l0 && (unsigned)lane < l1) {
                n0 = (unsigned)e & (unsigned)(N - 1);
                const unsigned rr = (unsigned)e >> logN;
                if (a.irregular) g = a.ref_begin + rr;
                else {
                    unsigned q = __umul24(rr, div_m) >> 20;
                    if (!mul_div) { asm volatile("" ::: "memory"); q = rr / (unsigned)ncols_span; }
                    g = __umul24((unsigned)r_lo + q, a.n_ref_cols) + (unsigned)c_lo + (rr - __umul24(q, (unsigned)ncols_span));
                }
                if (g >= a.ref_begin && g < g_end) p4 = *reinterpret_cast<const uint4*>(apos + ((size_t)g << logN) + n0);
            }
            const unsigned pj[4] = {p4.x, p4.y, p4.z, p4.w};
            unsigned m4 = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) {   /* an absent patch (0xffffffff) is at row / column 65535: outside every tile's range */
                const bool h = ((pj[j] >> 16) - lo_y) < span_y && ((pj[j] & 0xffffu) - lo_x) < span_x;
                m4 |= h ? (1u << j) : 0u;
            }
            const unsigned cnt = (unsigned)__popc(m4);
            /* exclusive prefix of cnt (0 .. 4) over the lanes below this one */
            const unsigned long long b0 = __ballot(cnt & 1u), b1 = __ballot(cnt & 2u), b2 = __ballot(cnt & 4u);
            auto below = [&](unsigned long long b) { return __builtin_amdgcn_mbcnt_hi((unsigned)(b >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)b, 0u)); };
            const unsigned total = (unsigned)__popcll(b0) + 2u * (unsigned)__popcll(b1) + 4u * (unsigned)__popcll(b2);
            if (nh + total > cap_hits) return false;
            if (m4) {
                unsigned slot = nh + below(b0) + 2u * below(b1) + 4u * below(b2);
                size_t wbase; unsigned off;
                const unsigned gl = g - a.ref_begin;   /* filt holds the launch's groups: [g - ref_begin][n][st][c][k2] */
                if (small24) { wbase = __umul24(g, (unsigned)C); off = __umul24(__umul24((gl << logN) + n0, Af) + stf, (unsigned)(C * k2)); }
                else { asm volatile("" ::: "memory"); wbase = (size_t)g * C; off = (((gl << logN) + n0) * Af + stf) * C * k2; }
                float w[3];
#pragma unroll
                for (int c = 0; c < 3; c++) w[c] = c < C ? a.wgt[wbase + (a.wchan0 ? 0 : c)] : 0.0f;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    if (m4 & (1u << j)) {
                        hit_a[slot] = make_uint4(pj[j], off, __float_as_uint(w[0]), __float_as_uint(w[1]));
                        hit_w2[slot] = w[2];
                        slot++;
                    }
                    off += pstep;
                }
            }
            nh += total;
            __builtin_amdgcn_wave_barrier();
            return true;
        };
        for (int c0 = 0; c0 < n_cand; c0 += 256) {
            if (!try_append(c0, 0u, 64u))                     /* more hits than the list has room for: sixteen lanes (<= 64 hits) at a time */
                for (unsigned l0 = 0; l0 < 64; l0 += 16) { consume(); (void)try_append(c0, l0, l0 + 16); }
            if (nh >= kAggFlush) consume();
        }
    } else
    for (int c0 = 0; c0 < n_cand; c0 += 64 * kAggPF) {
        unsigned g[kAggPF], p[kAggPF], nn[kAggPF];
#pragma unroll
        for (int u = 0; u < kAggPF; u++) {
            const int e = c0 + u * 64 + lane;
            p[u] = 0xffffffffu; g[u] = 0; nn[u] = 0;
            if (e < n_cand) {
                const unsigned n = (unsigned)e & (unsigned)(N - 1), rr = (unsigned)e >> logN;
                nn[u] = n;
                if (a.irregular) g[u] = a.ref_begin + rr;   /* the list is in raster order too (row lists, then columns) */
                else {
                    /* 24-bit multiplies (full rate): rr, q < 2^20 and div_m <= 2^20 under mul_div; grid rows / columns < 2^16 */
                    unsigned q = __umul24(rr, div_m) >> 20;
                    if (!mul_div) { asm volatile("" ::: "memory"); q = rr / (unsigned)ncols_span; }   /* kept a branch: the division's 32-bit multiplies are quarter rate */
                    g[u] = __umul24((unsigned)r_lo + q, a.n_ref_cols) + (unsigned)c_lo + (rr - __umul24(q, (unsigned)ncols_span));
                }
                if (g[u] >= a.ref_begin && g[u] < g_end) p[u] = apos[((size_t)g[u] << logN) + n];
            }
        }
        bool hit[kAggPF];
        float w[kAggPF][3];
        size_t wbase[kAggPF];
        if (small24) {
#pragma unroll
            for (int u = 0; u < kAggPF; u++) wbase[u] = __umul24(g[u], (unsigned)C);
        } else {
            asm volatile("" ::: "memory");
#pragma unroll
            for (int u = 0; u < kAggPF; u++) wbase[u] = (size_t)g[u] * C;
        }
#pragma unroll
        for (int u = 0; u < kAggPF; u++) {
            const int py = (int)(p[u] >> 16), px = (int)(p[u] & 0xffffu);
            hit[u] = p[u] != 0xffffffffu && py < ty0 + TH && py + k > ty0 && px < tx0 + TW && px + k > tx0;
#pragma unroll
            for (int c = 0; c < 3; c++) w[u][c] = (hit[u] && c < C) ? a.wgt[wbase[u] + (a.wchan0 ? 0 : c)] : 0.0f;
        }
        /* ordered append of the hits of these chunks */
#pragma unroll
        for (int u = 0; u < kAggPF; u++) {
            const unsigned long long bal = __ballot(hit[u]);
            if (hit[u]) {
                const unsigned slot = nh + __popcll(bal & ((1ull << lane) - 1ull));
                unsigned off;
                const unsigned gl = g[u] - a.ref_begin;
                if (small24) off = __umul24(__umul24((gl << logN) + nn[u], Af) + stf, (unsigned)(C * k2));
                else { asm volatile("" ::: "memory"); off = (((gl << logN) + nn[u]) * Af + stf) * C * k2; }
                hit_a[slot] = make_uint4(p[u], off, __float_as_uint(w[u][0]), __float_as_uint(w[u][1]));
                hit_w2[slot] = w[u][2];
            }
            nh += __popcll(bal);
        }
        __builtin_amdgcn_wave_barrier();
        if (nh >= kAggFlush) consume();
    }
    if (nh) consume();
    if (inside) for (int c = 0; c < C; c++) { a.num[pix + c * plane] = accn[c]; a.den[pix + c * plane] = accd[c]; }
}

} /* namespace */

hipError_t launch_aggregate(hipStream_t s, const AggArgs& a) {
    const bool wide = a.k >= 12;
#ifndef LFBM5D_AGG16_TW
#define LFBM5D_AGG16_TW 16
#define LFBM5D_AGG16_TH 4
#endif
#ifndef LFBM5D_AGG8_TW
#define LFBM5D_AGG8_TW 8
#define LFBM5D_AGG8_TH 8
#endif
    const unsigned tw = wide ? (a.k == 12 ? 16 : LFBM5D_AGG16_TW) : (a.k == 8 ? LFBM5D_AGG8_TW : 8), th = wide ? (a.k == 12 ? 4 : LFBM5D_AGG16_TH) : (a.k == 8 ? LFBM5D_AGG8_TH : 8);
    const unsigned tiles = ((a.Wb + tw - 1) / tw) * ((a.Hb + th - 1) / th) * a.A;
    const dim3 grid(((tiles + 7) / 8) * 8), block(64);
    const bool big = (a.filt_sai_stride ? a.filt_sai_stride * sizeof(float) : a.filt_bytes) > 0xfffff000ull || (a.opt & kOptAgg64Bit);   /* option agg_64bit: exercise the 64-bit path in tests */
    /* four candidates per lane and 16-byte position loads when a reference patch's N matches come in fours (option agg_scalar_scan: the
     * one-candidate-per-lane scan of rounds 1-3, for A/B runs; N = 1, 2 always take it) */
    /* Measured at the headline window (same box, rounds of tools/pass_time.py; instruction counts: tools/pmc_agg_ab.sh): VALU instructions
     * -14 % (k = 8) / -11 % (k = 16), scalar -51 % / -43 %, time 0.648 against 0.658 ms (k = 8), 0.973 against 0.964 (k = 16) -- the
     * kernel's time is its consume phase, not the scan -- so only the 8 x 8 tiles take it. */
    const bool vec4 = a.N % 4 == 0 && !wide && !(a.opt & kOptAggScalarScan);
#define LFBM5D_AGG(W_, TW_, TH_, PF_, U_) \
    do { if (big && vec4) hipLaunchKernelGGL((k_aggregate<W_, TW_, TH_, PF_, U_, true, true>), grid, block, 0, s, a); \
         else if (big)    hipLaunchKernelGGL((k_aggregate<W_, TW_, TH_, PF_, U_, true, false>), grid, block, 0, s, a); \
         else if (vec4)   hipLaunchKernelGGL((k_aggregate<W_, TW_, TH_, PF_, U_, false, true>), grid, block, 0, s, a); \
         else             hipLaunchKernelGGL((k_aggregate<W_, TW_, TH_, PF_, U_, false, false>), grid, block, 0, s, a); } while (0)
    /* scan / consume depths (candidate chunks per scan round, hits per load round): re-swept in round 6, profiles/r06_n_agg_depths.txt */
#ifndef LFBM5D_AGG16_PF
#define LFBM5D_AGG16_PF 3
#endif
#ifndef LFBM5D_AGG16_U
#define LFBM5D_AGG16_U 12
#endif
#ifndef LFBM5D_AGG8_PF
#define LFBM5D_AGG8_PF 3
#endif
#ifndef LFBM5D_AGG8_U
#define LFBM5D_AGG8_U 6
#endif
    if (a.k == 12)      LFBM5D_AGG(true, 16, 4, 3, 12);
    else if (a.k == 8)  LFBM5D_AGG(true, LFBM5D_AGG8_TW, LFBM5D_AGG8_TH, LFBM5D_AGG8_PF, LFBM5D_AGG8_U);
    else if (wide)      LFBM5D_AGG(false, LFBM5D_AGG16_TW, LFBM5D_AGG16_TH, LFBM5D_AGG16_PF, LFBM5D_AGG16_U);
    else                LFBM5D_AGG(false, 8, 8, 2, 6);
#undef LFBM5D_AGG
    return hipGetLastError();
}

} /* namespace lfbm5d */
