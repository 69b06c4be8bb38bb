/*
 * lfbm5d_kernels.h -- launchers of the HIP kernels behind the C-ABI (include/lfbm5d.h).
 * Internal to liblfbm5d_hip.so.  All launchers enqueue on the given stream and return
 * hipGetLastError(); nothing here synchronises or allocates.
 */
#ifndef LFBM5D_KERNELS_H
#define LFBM5D_KERNELS_H

#include <hip/hip_runtime.h>
#include <vector>

#include "lfbm5d_options.h"

/* Cache policy of the `filt` stream (written once by the group kernels, read once or twice by the aggregation): the aux operand
 * of the buffer instructions -- 0 default, 2 = nt (streaming), 1 = sc0, 16 = sc1.  Round 6, measured on the headline window
 * (profiles/r06_a_nt_filt_ab.txt): non-temporal STORES take k_group_id_haar from 0.93-1.06 to 0.81-0.85 ms per pass and k_group_dct8w3
 * and both aggregation kernels 3 % down (the write stream no longer evicts the window rows the gathers of the same XCD re-read from
 * its L2); non-temporal LOADS in the aggregation cost it half its speed (a filtered row is read by two neighbouring tiles).
 * Build-time knobs for A/B runs (tools/build_variant_files.sh). */
#ifndef LFBM5D_FILT_STORE_AUX
#define LFBM5D_FILT_STORE_AUX 2
#endif
#ifndef LFBM5D_FILT_LOAD_AUX
#define LFBM5D_FILT_LOAD_AUX 0
#endif

#if defined(__HIPCC__)
/* Stores of filtered patches through global pointers.  The non-temporal form ONLY where a wave-instruction (or two back to back)
 * writes whole lines -- the register-resident HT kernel (256 contiguous bytes per instruction; it passes the aux operand to its buffer
 * stores itself) and k_group_dct8w3 (eight lanes write a patch's 256 bytes) --: measured on the 16 x 16 wavelet kernel, whose
 * instructions scatter 16-byte pieces 64 bytes apart and complete a line over four of them, non-temporal stores cost 2.2 -> 4.8 ms
 * per pass (partial-line writes at the memory side; profiles/r06_g_nt_partial_lines.txt).  Everything else stores plainly. */
typedef float filt_v4f __attribute__((ext_vector_type(4)));
typedef float filt_v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void filt_put(float* p, const float v) { *p = v; }
__device__ __forceinline__ void filt_put2(filt_v2f* p, const filt_v2f v) { *p = v; }
__device__ __forceinline__ void filt_put4(filt_v4f* p, const filt_v4f v) { *p = v; }
__device__ __forceinline__ void filt_put4(float4* p, const float4 v) { *p = v; }
__device__ __forceinline__ void filt_put4_nt(filt_v4f* p, const filt_v4f v) {
#if LFBM5D_FILT_STORE_AUX == 2
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}
#endif

namespace lfbm5d {

constexpr int kMaxK = 32;       /* largest patch side (dedicated kernels: 8, 12, 16; the general forms take any) */
constexpr int kMaxA = 49;       /* SAIs per angular window of the register forms: 3x3 (an = 1, every dedicated kernel), 5x5 or 7x7 (an = 2, 3: generic kernel) */
constexpr int kA3 = 9;          /* ... of the 3x3 window the dedicated kernels are written for */
constexpr int kMaxAw = 7;       /* side of the largest such window */
/* larger windows (aswSize 4 .. 8: 9x9 .. 17x17 SAIs; the reference takes any window that fits the light field, bm5d.cpp:119-124): the
 * general forms -- run-time transform sizes, vectors in scratch memory, stacks in HBM -- slow, but no refusal */
constexpr int kBigAw = 17;
constexpr int kBigA = kBigAw * kBigAw;
constexpr int kMaskWords = (kBigA + 63) / 64;
/* one bit per SAI of an angular window */
struct SaiMask {
    unsigned long long w[kMaskWords];
    __host__ __device__ bool test(unsigned i) const { return (w[i >> 6] >> (i & 63)) & 1ull; }
    __host__ __device__ void set(unsigned i) { w[i >> 6] |= 1ull << (i & 63); }
    __host__ __device__ unsigned count() const { unsigned n = 0; for (int i = 0; i < kMaskWords; i++) n += (unsigned)__builtin_popcountll(w[i]); return n; }
    __host__ __device__ bool holds_all(unsigned n) const { for (unsigned i = 0; i < n; i++) if (!test(i)) return false; return true; }
};
inline SaiMask sai_mask_none() { SaiMask m; for (int i = 0; i < kMaskWords; i++) m.w[i] = 0; return m; }
constexpr int kMaxN = 16;      /* max similar patches (power of two) of the light-field core and its dedicated kernels */
constexpr int kMaxN3 = 32;     /* ... of the per-SAI BM3D flavour (generic group kernel only) */

/* Normalisation tables and filter taps, computed on the host exactly as the reference's
 * preProcess / preProcess_4d / preProcess_4d_sadct / bior15_coef do (bm3d.cpp:1101-1169,
 * core:3191-3252, lib_transforms.cpp:215-277), uploaded once per pass configuration. */
struct GroupTables {
    float kaiser[kMaxK * kMaxK];
    float cn2[kMaxK * kMaxK];   /* coef_norm      (2-D patch DCT) */
    float cni2[kMaxK * kMaxK];  /* coef_norm_inv                  */
    float cos2[kMaxK * kMaxK];  /* cos(pi (j+1/2) u / k) at [u*k + j] */
    float cn4[kBigA];           /* coef_norm_4d */
    float cni4[kBigA];
    float cos3[9];              /* cos(pi (j+1/2) u / 3) */
    float cosw[kBigA];          /* cos(pi (j+1/2) u / aw) at [u*aw + j] for the window side aw (general angular DCT) */
    float cos1[kBigAw + 1][kBigA]; /* SADCT: cos(pi (j+1/2) u / n) at [n][u*n + j], n = 1..aw */
    float cn1[kBigAw + 1][kBigAw]; /* SADCT 1-D norms for length n (core:3229-3252) */
    float cni1[kBigAw + 1][kBigAw];
    float c1inv[kBigAw + 1];    /* 0.5 * SQRT2_INV / sqrt(n) (core:2190) */
    float cos5[5][256];         /* 5th-dimension DCT: cos(pi (j+1/2) u / n) at [log2 n][u*n + j], n = 1..16 */
    float cos5x[1024];          /* ... and for n = 32 (generic kernel only) */
    float cn5_0[6], cn5[6];     /* coef_norm of preProcess_5d (core:3262-3276), [log2 n] */
    float c5inv[6];             /* 0.5 * SQRT2_INV / sqrt(n) (core:2591) */
    float lpd[10], hpd[10], lpr[10], hpr[10];
    float coef2inv;             /* 1 / (2k)                       (bm3d.cpp:1064) */
    float coef4inv;             /* 1 / (2 sqrt(aw) sqrt(ah))      (core:1945)     */
    /* 3x3 windows, hard-thresholding step with Haar fibres (round 6): constants of the unnormalised chain group_id_compute_fast
     * (lfbm5d_group_ht.hip), evaluated in double.  alpha = (2, sqrt 3, 1), gamma = (1, sqrt 3, 1). */
    float ht3_f[9];             /* F[st] = alpha_v alpha_u cn4[st]: unnormalised forward coefficient -> the reference's normalised one */
    float ht3_gf[9];            /* F[st] cni4[st] coef4inv gamma_v gamma_u: times 1 / nSx, the inverse's input scale */
};

constexpr unsigned kShapeInfoBytes = (5 * kMaxA + 2 * kMaxAw + 1) * 4;          /* per-group SADCT record, windows of up to 7x7 SAIs */
constexpr unsigned kShapeInfoBigBytes = (5 * kBigA + 2 * kBigAw + 1) * 4;      /* ... larger windows */

struct GroupArgs {
    const float* noisy;         /* [A][C][Hb][Wb] */
    const float* basic;         /* step 2 only */
    float* num;
    float* den;
    const unsigned* refs;       /* [R] flat positions, raster order */
    const unsigned* self_idx;   /* [R][N] */
    const unsigned* self_cnt;   /* [R] */
    const unsigned* best;       /* [A][Wb*Hb] disparity match */
    const unsigned char* shape; /* [A][Wb*Hb] */
    const GroupTables* tb;
    float* filt;                /* [R][N][A][C][k2] filtered patches (pixel domain), indexed by the ABSOLUTE group number: a launch over the groups
                                 * [ref_begin, ref_begin + n_groups) whose buffer holds only those passes (buffer - ref_begin * N*A*C*k2) */
    float* wgt;                 /* [R][C] aggregation weights */
    unsigned* aggpos;           /* [A][R][N] where each filtered patch is aggregated (0xffffffff: nowhere) */
    unsigned* gpos;             /* [R][N][A] window position of every patch of every group (pre-pass output) */
    unsigned* sa_list;          /* [1 + 4 R] round 6: [0] = number of entries, then group number | channel mask << 29 (any order; k_group_pos zeroes
                                 * the count): what the register-resident HT kernels leave to k_group_id_*_list -- groups whose angular shape is not
                                 * the whole window (k_group_shape appends them), guard-band cases of the fast chain (lfbm5d_group_ht.hip) */
    unsigned* gofs;             /* [R][N][A] the same as a byte offset into channel 0 of the patch's SAI (0 for an absent patch), and ...        */
    unsigned* gok;              /* [R][N] ... bit st set where patch (n, st) is there: what the register-resident HT kernel's scalar loads need */
    void* gshape;               /* [R] kShapeInfoBytes (A > 49: kShapeInfoBigBytes) each: SADCT bookkeeping of the group (pre-pass output) */
    unsigned n_refs_total;
    unsigned long long* counters; /* [0] sum nSx, [1] sadct groups */
    unsigned ref_begin, n_groups;
    unsigned Wb, Hb, C, A, k, N, pst;
    SaiMask mask_bits, proc_bits;   /* bit st: SAI st of the window is there / already processed */
    unsigned tau2, tau4, tau5, useSD;
    unsigned fill_quirk;        /* 1 on the centre path: patches at column Wb-k read as zeros (core:1697) */
    int step;
    float lambda;
    float sigma[3];
    float ht3_T[3][9][4];       /* group_id_compute_fast: the hard threshold of channel c for the UNNORMALISED Haar coefficients of level l at
                                 * angular frequency st: T_c / (ht3_f[st] 2^(-l/2)), evaluated in double on the host (run_pass) */
    float* scratch;             /* generic path, stacks beyond the LDS: HBM slices for k_group_big (or NULL) */
    unsigned long long scratch_floats;   /* size of scratch */
    unsigned opt;               /* kOpt* bits (lfbm5d_options.h): kernel-generation selectors of the context */
    unsigned long long filt_sai_stride;   /* 0: filt is group-major [g][n][st][c][k2]; > 0 (windows of kSaiMajorMinA SAIs and more, general kernels only):
                                           * SAI-major [st][g][n][c][k2], that many floats per SAI -- see filt_patch */
    unsigned bm3d;              /* per-SAI BM3D arithmetic (bm3d.cpp:914-1027, :1345-1373): threshold without sqrt2, SD weight over nSx*k^2 */
};

/* Where the filtered patch (group g, stack position n, SAI st) starts in filt (channel 0; channels follow at k2 floats).
 * Group-major [g][n][st][c][k2] is what the 3x3-window kernels write: a group's patches in one piece, and the aggregation of SAI st
 * finds its patches A C k2 floats apart.  On wide angular windows that stride is the problem: at 13x13 SAIs consecutive patches of
 * one SAI lie 519 KB apart, every gather of the aggregation lands on another 2 MB page and the pass is bound by address
 * translation (profiles/r05_e, r06_l).  SAI-major [st][g][n][c][k2] (filt_sai_stride > 0) keeps the patches one SAI's aggregation reads
 * together -- N C k2 floats per group, contiguous over the groups.  GroupArgs::filt is biased by the launch's first group in either layout. */
constexpr unsigned kSaiMajorMinA = 121;   /* 11 x 11 SAIs and more (measured: profiles/r06_l_wide_windows.txt) */
__device__ __forceinline__ size_t filt_patch(const GroupArgs& a, unsigned g, unsigned n, unsigned st, unsigned k2) {
    return a.filt_sai_stride ? (size_t)st * a.filt_sai_stride + ((size_t)g * a.N + n) * a.C * k2
                             : (((size_t)g * a.N + n) * a.A + st) * a.C * k2;
}


struct AggArgs {
    float* num;
    float* den;
    const float* filt;               /* the launch's groups only: [g - ref_begin][N][A][C][k2], or SAI-major [A][g - ref_begin][N][C][k2] */
    unsigned long long filt_sai_stride;   /* 0: group-major; > 0: SAI-major, floats per SAI (GroupArgs::filt_sai_stride) */
    unsigned long long filt_bytes;   /* size of filt: below 4 GiB the gathers go through a buffer resource */
    const float* wgt;
    const unsigned* aggpos;     /* [A][R][N] */
    unsigned n_refs_total;
    const unsigned* refs;
    const unsigned* self_idx;
    const unsigned* self_cnt;
    const unsigned* best;
    const unsigned char* shape;
    const GroupTables* tb;
    unsigned ref_begin, n_groups;   /* groups [ref_begin, ref_begin + n_groups) of this rank */
    unsigned n_ref_rows, n_ref_cols;
    unsigned Wb, Hb, C, A, k, N, pst, p, nHW, nSim, nDisp;
    SaiMask mask_bits, proc_bits; unsigned tau4;
    unsigned irregular;         /* reference list is not the regular grid (subset path): scan every reference */
    unsigned opt;               /* kOpt* bits (lfbm5d_options.h) */
    unsigned wchan0;            /* every channel uses channel 0's group weight (sd_weighting of bm3d.cpp:1345-1373) */
};

/* Disparity score tables are laid out strip-major and SKEWED -- [strip of 64 columns][table row + lane][64] -- so that
 * the scan kernel, whose lane l works on a row l steps behind lane 0's, writes the 64 values of a step as one contiguous
 * 256-byte row, one contiguous stream per table.  A strip has nrows + 63 such rows (the corners hold nothing). */
__host__ __device__ inline size_t stereo_table_stride(unsigned W, unsigned H, unsigned k, unsigned nDisp) {
    const unsigned ncols = W - 2 * nDisp - (k - 1), nrows = H - 2 * nDisp - (k - 1);
    return (size_t)((ncols + 63) / 64) * 64 * (nrows + 63);
}

/* Second-generation scan (lfbm5d_scan2.hip): the four values of four consecutive steps leave as one 16-byte store, so a
 * table is [strip][Q / 4][lane][Q % 4] with Q = table row + lane + 3 (step t of a strip writes Q = t + 4; the row-0 value
 * of lane l sits at Q = l + 3); steps are rounded up to whole groups of sixteen.  Strips start at column 1: column 0 is
 * computed up front (the hand-off column of the first strip) and stored as a plain column behind the strips. */
__host__ __device__ inline unsigned stereo_table_srq(unsigned H, unsigned k, unsigned nDisp) {
    const unsigned nrows = H - 2 * nDisp - (k - 1);
    return ((nrows - 1 + 63 + 15) / 16) * 4 + 1;
}
__host__ __device__ inline size_t stereo_table_stride2(unsigned W, unsigned H, unsigned k, unsigned nDisp) {
    const unsigned ncols = W - 2 * nDisp - (k - 1);
    return (size_t)((ncols - 1 + 63) / 64) * stereo_table_srq(H, k, nDisp) * 256 + stereo_table_srq(H, k, nDisp) * 4;
}

/* Second-generation scan, combined form (the default): the distance tables of the disparity search never reach memory.  A
 * workgroup reduces its (up to eleven) tables to one (value, scan order) pair per position -- 8 bytes -- laid out
 * [slot][workgroup of the slot][strip][chunk of eight steps][lane * 8 + step in chunk], where lane l of strip s is table column
 * 1 + 64 s + l and step t = 8 chunk + step-in-chunk is table row 1 + t - l (entries whose row falls outside the table hold
 * garbage).  Column 0 and the row-0 entry of every strip's first column are kept per table in an edge array
 * [rows][strips] behind the pairs. */
__host__ __device__ inline unsigned stereo_part_chunks(unsigned H, unsigned k, unsigned nDisp) {
    const unsigned nrows = H - 2 * nDisp - (k - 1);
    return ((nrows - 1 + 63 + 15) / 16) * 2;
}
__host__ __device__ inline size_t stereo_part_stride(unsigned W, unsigned H, unsigned k, unsigned nDisp) {   /* pairs per workgroup */
    const unsigned ncols = W - 2 * nDisp - (k - 1);
    return (size_t)((ncols - 1 + 63) / 64) * stereo_part_chunks(H, k, nDisp) * 512;
}
__host__ __device__ inline size_t stereo_edge_stride(unsigned W, unsigned H, unsigned k, unsigned nDisp) {   /* floats per table */
    const unsigned ncols = W - 2 * nDisp - (k - 1), nrows = H - 2 * nDisp - (k - 1);
    return ((size_t)nrows + (ncols - 1 + 63) / 64 + 3) & ~(size_t)3;
}

/* One workgroup of the second-generation scan: up to ten displacement tables of one image pair whose (di + dj) mod 4
 * agree (16-byte alignment of the transposed ring reads), and the extent of their displacements (size of the second ring). */
struct Scan2Wg {
    short tab[16];              /* displacement index di * Ns + dj of each table wave's table, -1: the wave has none */
    short slot;                 /* disparity search: table slot (index into st_of_slot); self search: -1 */
    short r2lo, c2lo;           /* smallest row / column offset of the second image's reads against the first's */
    short rh, ch;               /* ... and how many more rows / columns the largest needs */
    short wgj;                  /* disparity search: index of the workgroup among those of its slot */
    short pad[2];
    short ord[16];              /* disparity search: the reference's scan order dj * Ns + di of each table (ties, and the arg-min's displacement) */
};

struct ScanArgs {
    const float* est;           /* [A][Wb*Hb] channel-0 estimates (+ slack) */
    unsigned W, H, k;
    unsigned nSim, nDisp, nHW;  /* self search: band nHW, half window nSim; disparity: band and half window nDisp */
    unsigned pst;
    unsigned n_self;            /* (nSim+1)*(2nSim+1) self tables (0 when N == 1) */
    unsigned n_stereo;          /* n_slots * (2nDisp+1)^2 disparity tables */
    /* self: the regular reference grid (centre pass) */
    unsigned n_ref_rows, n_ref_cols, p;
    const int* rslot;           /* [H] reference-grid row slot of each image row, -1 off the grid */
    const int* refmap;          /* [W*H] slot of the reference patch at a position or -1: irregular lists only, else NULL */
    float* scores;              /* [R][Ns*Ns], pre-filled with 2*threshold */
    unsigned scores_bytes;
    /* stereo */
    unsigned long long* dbg;    /* development builds (LFBM5D_PHASE_TIMING): phase clocks; else unused */
    unsigned opt;               /* kOpt* bits (lfbm5d_options.h): which table-kernel generation */
    unsigned lds_cap;           /* > 0: LDS bytes the first-generation kernel may use (option scan_lds_cap) */
    float* tables;              /* [n_slots][Ns*Ns][stereo_table_stride]: strip-major [strip][row][64 columns] */
    unsigned st_of_slot[kBigA];
    /* second-generation kernel */
    unsigned est_planes;        /* SAIs in est */
    const Scan2Wg* wgs;         /* [n_wgs] workgroup descriptors (device) */
    unsigned n_wgs;
    float* lcol;                /* [n_self + n_stereo][lcol_stride] hand-off columns between the strips of a table */
    unsigned lcol_stride;
    unsigned nwg_slot;          /* workgroups per slot of the disparity search (combined form) */
};

hipError_t launch_color(hipStream_t s, float* img, unsigned cs, unsigned n_px, int forward);
/* whole light field [SAI][3][n_px] / [SAI][seg] in one launch; d_mask: device copy of the SAI mask (0 = skip) */
hipError_t launch_color_lf(hipStream_t s, float* lf, size_t sai_stride, unsigned n_sai, const unsigned* d_mask, unsigned cs,
                           unsigned n_px, int forward);
/* out = forward(inverse(in)) of every masked SAI (the colour round trip between the two steps); in == out allowed */
hipError_t launch_color_roundtrip_lf(hipStream_t s, const float* in, float* out, size_t sai_stride, unsigned n_sai, const unsigned* d_mask,
                                     unsigned cs, unsigned n_px);
hipError_t launch_estimate_lf(hipStream_t s, const float* num, const float* den, const float* sub, float* est, size_t seg,
                              unsigned n_sai, const unsigned* d_mask);
hipError_t launch_symetrize(hipStream_t s, const float* src, float* dst, unsigned W, unsigned H,
                            unsigned C, unsigned N);
hipError_t launch_unsymetrize(hipStream_t s, float* dst, const float* src, unsigned W, unsigned H,
                              unsigned C, unsigned N);
/* W x H crop at offset (off, off) of an image padded by N (off == N: unsymetrize) */
hipError_t launch_crop(hipStream_t s, float* dst, const float* src, unsigned W, unsigned H, unsigned C, unsigned N, unsigned off);
/* est = den ? num/den : sub on `n` elements */
hipError_t launch_estimate(hipStream_t s, const float* num, const float* den, const float* sub,
                           float* est, size_t n);
/* all SAIs of an angular window in one launch: slot i of the window <-> SAI L.st[i] of the light field */
struct SaiList { unsigned st[kBigA]; unsigned n; };
/* two-step jobs: basic[st] = forward(inverse(den ? num / den : sub)) for the light-field SAIs L.st[0 .. L.n) (colour = 0: no
 * colour round trip); light fields with three channels of n_px pixels */
hipError_t launch_finalize_multi(hipStream_t s, const float* num, const float* den, const float* sub, float* basic, size_t sai_stride,
                                 const SaiList& L, unsigned cs, unsigned n_px, int colour);
/* streamed host seam: the outputs of the light-field SAIs L.st[0 .. L.n) whose sums are final -- out = inverse(den ? num / den : sub),
 * basic = inverse(basic) in place (NULL: the step has no pilot), noisy_dst = inverse(noisy_src) (colour = 0: no transforms); three
 * channels of n_px pixels; sub may be basic, noisy_src may be noisy_dst */
hipError_t launch_output_multi(hipStream_t s, const float* num, const float* den, const float* sub, float* out, float* basic,
                               const float* noisy_src, float* noisy_dst, size_t sai_stride, const SaiList& L, unsigned cs, unsigned n_px,
                               int colour);
/* the gating words of the two-processes-on-one-GPU transport (k_ipc_set / k_ipc_wait in lfbm5d_window.hip) */
hipError_t launch_ipc_set(hipStream_t s, unsigned* p, unsigned v);
hipError_t launch_ipc_wait(hipStream_t s, const unsigned* p, unsigned want, unsigned* err, double timeout_s);
hipError_t launch_symetrize_multi(hipStream_t s, const float* src, size_t src_stride, float* dst, size_t dst_stride,
                                  const SaiList& L, unsigned W, unsigned H, unsigned C, unsigned N);
hipError_t launch_unsymetrize_multi(hipStream_t s, float* dst, size_t dst_stride, const float* src, size_t src_stride,
                                    const SaiList& L, unsigned W, unsigned H, unsigned C, unsigned N);
/* The two ends of a window of the graph form, one launch each.  begin: mirror-padded noisy (+ basic, NULL in step 1) + num + den of the
 * window's SAIs and the channel-0 matching estimate from the padded sums (= launch_symetrize_multi x 3..4 + launch_estimate_multi),
 * zero[0 .. kWinCounters) cleared;  end: the window's sums back into the light field and the pass's coverage count added to
 * count[0 .. kWinCounters) -- partial counters, the count is their sum -- (= launch_unsymetrize_multi x 2 + launch_count_denoised).
 * lf_stride / w_stride: floats per SAI of the light field / window. */
constexpr unsigned kWinCounters = 32;
hipError_t launch_window_begin(hipStream_t s, const float* noisy, const float* basic, const float* num, const float* den, size_t lf_stride,
                               float* w_noisy, float* w_basic, float* w_num, float* w_den, float* est, size_t w_stride, const SaiList& L,
                               unsigned W, unsigned H, unsigned C, unsigned N, unsigned* zero);
hipError_t launch_window_end(hipStream_t s, float* num, float* den, size_t lf_stride, const float* w_num, const float* w_den, size_t w_stride,
                             const SaiList& L, unsigned W, unsigned H, unsigned C, unsigned N, unsigned k, unsigned* count);
hipError_t launch_estimate_multi(hipStream_t s, const float* num, const float* den, const float* sub, float* est,
                                 size_t plane, unsigned C, unsigned A, const SaiMask& mask_bits);
/* w x h rectangle of every channel of n_slots images: dst image (dW x dH, slots dst_stride apart) at (dx0, dy0) <- src image
 * (sW x sH) at (sx0, sy0).  The reference's sub_divide / undivide_LF (utilities.cpp:312-395, utilities_LF.cpp:438-515) */
hipError_t launch_copy_rect(hipStream_t s, float* dst, size_t dst_stride, unsigned dW, unsigned dH, unsigned dx0, unsigned dy0,
                            const float* src, size_t src_stride, unsigned sW, unsigned sH, unsigned sx0, unsigned sy0,
                            unsigned w, unsigned h, unsigned C, unsigned n_slots, const SaiMask& mask_bits);
hipError_t launch_copy_rows(hipStream_t s, const float* src, unsigned src_H, unsigned src_row0, float* dst, unsigned dst_H, unsigned dst_row0,
                            unsigned n_rows, unsigned W, unsigned planes);
hipError_t launch_fill_f32(hipStream_t s, float* p, float v, size_t n);
hipError_t launch_fill_i32(hipStream_t s, int* p, int v, size_t n);
hipError_t launch_add(hipStream_t s, float* dst, const float* src, size_t n);   /* dst += src */
/* counts[i] += number of exact zeros in seg i; segments are `seg` floats long, n_seg of them */
hipError_t launch_count_zeros(hipStream_t s, const float* den, size_t seg, unsigned n_seg,
                              unsigned* counts);
/* LF_denoised_percent numerator on a padded window image (utilities_LF.cpp:985-992) */
hipError_t launch_count_denoised(hipStream_t s, const float* den, size_t sai_stride, unsigned n_slots, const SaiMask& mask_bits,
                                 unsigned W, unsigned H, unsigned C, unsigned N, unsigned k, unsigned* count);
hipError_t launch_refmap(hipStream_t s, const unsigned* refs, unsigned n_refs, int* refmap);
/* subset pass: the regular grid's patches (rows / columns nHW + i p, the last forced to last_r / last_c) whose footprint holds a zero
 * of den0, in raster order -> refs, their number -> count; flags: n_rows * n_cols bytes of scratch */
hipError_t launch_subset_list(hipStream_t s, const float* den0, unsigned Wb, unsigned k, unsigned nHW, unsigned p, unsigned n_rows,
                              unsigned n_cols, unsigned last_r, unsigned last_c, unsigned char* flags, unsigned* refs, unsigned* count);
hipError_t launch_bm_scan(hipStream_t s, const ScanArgs& a);
/* second generation (lfbm5d_scan2.hip): which kernel a configuration gets (1 or 2), the workgroup list and LDS size of a
 * launch, the hand-off row length, the launch itself and the arg-min over its table layout */
int bm_scan_version(const ScanArgs& a);   /* 1: round 2's kernel; 2: second generation with full tables (LFBM5D_SCAN_FULL_TABLES=1); 3: combined form */
bool scan2_plan(const ScanArgs& a, std::vector<Scan2Wg>& wgs, size_t* lds_bytes, unsigned* nwg_slot = nullptr);
size_t scan_tables_floats(const ScanArgs& a, int version, unsigned n_slots, unsigned nwg_slot);   /* size of the `tables` buffer */
unsigned scan2_lcol_stride(const ScanArgs& a);
hipError_t prepare_scan2_kernels();   /* per device, like prepare_group_kernels: the LDS limit of the table kernel */
hipError_t launch_bm_scan2(hipStream_t s, const ScanArgs& a, size_t lds_bytes, bool combined);
hipError_t launch_stereo_argmin3(hipStream_t s, const float* tables, const unsigned* st_of_slot, unsigned n_slots, unsigned nwg_slot,
                                 unsigned W, unsigned H, unsigned k, unsigned nDisp, float thr,
                                 unsigned* best, unsigned char* shape);
hipError_t launch_stereo_argmin2(hipStream_t s, const float* tables, const unsigned* st_of_slot, unsigned n_slots,
                                 unsigned W, unsigned H, unsigned k, unsigned nDisp, float thr,
                                 unsigned* best, unsigned char* shape);
hipError_t launch_self_select(hipStream_t s, const float* scores, const unsigned* refs, unsigned n_refs,
                              unsigned W, unsigned nSim, unsigned N, float thr, unsigned* self_idx,
                              unsigned* self_cnt,
                              /* grid_cols != 0: `scores` has a row per reference patch of the REGULAR grid (rows / columns nHW + i p,
                               * the last one forced to last_r / last_c) and `refs` lists some of them: a reference's scores are
                               * taken from its place in that grid */
                              unsigned grid_cols = 0, unsigned nHW = 0, unsigned p = 1, unsigned last_r = 0, unsigned last_c = 0);
hipError_t launch_self_trivial(hipStream_t s, const unsigned* refs, unsigned n_refs, unsigned* self_idx,
                               unsigned* self_cnt);
hipError_t launch_stereo_argmin(hipStream_t s, const float* tables, const unsigned* st_of_slot, unsigned n_slots,
                                unsigned W, unsigned H, unsigned k, unsigned nDisp, float thr,
                                unsigned* best, unsigned char* shape);
hipError_t prepare_group_kernels();   /* once per device: LDS limits of the group kernels */
hipError_t launch_group(hipStream_t s, const GroupArgs& a);
size_t group_lds_bytes(const GroupArgs& a);
/* HBM scratch (bytes) launch_group needs for this configuration: 0 unless the generic path's stacks exceed the LDS */
size_t group_scratch_bytes(const GroupArgs& a);
hipError_t launch_aggregate(hipStream_t s, const AggArgs& a);

} /* namespace lfbm5d */
#endif
