/*
 * lfbm5d_api.hip -- the C-ABI of include/lfbm5d.h: context, device-resident window schedule
 * (run_bm5d_1st_step / run_bm5d_2nd_step, bm5d.cpp:165-407 / :861-1106, nb_threads == 1 semantics),
 * one core pass (bm5d_1st_step / bm5d_2nd_step) as a sequence of HIP kernels on one stream, and the
 * RCCL reduction that replaces the reference's tile merge.
 *
 * Everything stays in HBM between passes; the host only reads back the few counters the greedy
 * schedule needs (zero-weight pixel counts per SAI, coverage count per window).
 */
#include "../../include/lfbm5d.h"
#include "lfbm5d_kernels.h"
#include "lfbm5d_plan.h"
#include "lfbm5d_options.h"

#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <fstream>
#include <thread>
#include <unistd.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

using namespace lfbm5d;

namespace {

std::string g_create_error;

const double kSqrt2 = 1.414213562373095;     /* core:33 */
const double kSqrt2Inv = 0.7071067811865475; /* core:34 */
const double kPi = 3.14159265358979323846;

struct DevBuf {
    void* p = nullptr; size_t cap = 0;
    hipError_t reserve(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        if (p) { hipError_t e = hipFree(p); p = nullptr; cap = 0; if (e != hipSuccess) return e; }
        hipError_t e = hipMalloc(&p, bytes);
        if (e == hipSuccess) cap = bytes;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

struct PassEvents { hipEvent_t e[5]; bool comm; };

struct GeomCache {
    DevBuf refs, rslot, tb, scan_wgs;
    std::vector<Scan2Wg> scan_plan; unsigned scan_key[8] = {0, 0, 0, 0, 0, 0, 0, 0}; size_t scan_lds = 0;
    unsigned scan_nwg_slot = 0; int scan_version = 0;
    std::vector<unsigned> last_refs_host;
    unsigned grid_key[5] = {0, 0, 0, 0, 0};      /* cached reference grid */
    unsigned rslot_key[5] = {0, 0, 0, 0, 0};     /* geometry rslot / n_ref_rows / n_ref_cols were built for (survives a subset pass, which replaces refs) */
    unsigned tb_key[3] = {0, 0, 0};
    unsigned n_ref_rows = 0, n_ref_cols = 0;
};

} /* namespace */

/* window lanes of the graph form when LFBM5D_LANES does not say (same-box sweep at the headline workload, round 4: one lane 194,
 * two 220, three 211, four 210, five / six 215 SAI-MP/s -- the table kernel fills the register files of the CUs it runs on, so
 * a third window mostly queues) */
constexpr int kDefaultLanes = 2;
constexpr size_t kEstLead = 64;   /* floats of slack in front of the estimate planes */

/* The caller's light fields of a *_host entry point: one host pointer per SAI (ignored for empty SAIs).  `basic` is an input of
 * run_bm5d_2nd_step only; `out` is the denoised light field of the second step (unused by the first, whose result is `basic`). */
struct HostIO {
    float* const* noisy = nullptr;
    float* const* basic = nullptr;
    float* const* out = nullptr;
};

struct lfbm5d_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    int rank = 0, world = 1;
    int tiles = 1;                         /* > 1: the reference's OpenMP tile mode (lfbm5d_set_tiles) */
    ncclComm_t comm = nullptr;
    ncclComm_t comm2 = nullptr;            /* second channel of the window-graph exchange (ncclCommSplit of comm) */
    hipStream_t cs[2] = {nullptr, nullptr}; /* exchange streams, one per channel */
    /* sharding actually applied inside a core pass: rows of reference patches over pass_world ranks (direct
     * lfbm5d_pass_device calls use rank/world; whole steps on several GPUs shard by WINDOWS instead and run
     * every pass unsharded) */
    int pass_rank = 0, pass_world = 1;
    bool pass_reduce = false;
    std::vector<unsigned> last_windows;   /* processed SAI of every window of the last step, in order */
    lfbm5d_stats stats;
    /* per-pass work buffers (grow only) */
    DevBuf t_noisy, t_basic, t_tnum, t_tden, und_num, und_den;   /* tile mode: one tile of the window, the tiles' interiors */
    DevBuf scan_lcol;                      /* second-generation scan: hand-off columns */
    DevBuf sub_flags, sub_cnt;             /* subset passes: the device-side reference list's scratch and count */
    int last_scan_version = 0;
    /* what a pass derives from its geometry alone (reference grid, transform tables, the table kernel's workgroup list):
     * cached, one set per step slot so that the windows of both steps of a two-step job can alternate on a lane without
     * re-uploading (and without the stream synchronisation an upload from a stack object needs) */
    GeomCache gc[2]; int gslot = 0;
    bool est_ready = false;                /* the caller of pass_impl has formed the matching estimate in `est` already (graph form) */
    DevBuf est, refmap, scores, tables, self_idx, self_cnt, best, shape, filt, wgt, aggpos, gpos, gofs, gok, sa_list, gshape, counters, small, t_num, t_den, d_mask;
    /* step-level buffers (g_num2 / g_den2 / n2: second step of a two-step job; e_basic: an emulated rank's own basic estimate) */
    DevBuf g_num, g_den, g_num2, g_den2, n2, e_basic, w_noisy, w_basic, w_num, w_den, h2d_noisy, h2d_basic, h2d_out, d_own, gscratch;
    /* streamed host seam (lfbm5d_*_host): the caller's light fields as host pointers per SAI, set for the duration of a job; the
     * job's inputs as they arrived (what a redo of the job starts from: the streamed outputs overwrite the caller's copies SAI by
     * SAI); the streams the uploads / downloads go through */
    /* second transport of the window-graph exchange, for tests: the ranks are PROCESSES ON ONE GPU (RCCL refuses that), a message is a
     * device copy out of the peer's buffers (hipIpcMemHandle) gated by words in mapped device memory; same graph, issue order, event
     * gating and abort path as the RCCL form (lfbm5d_comm_init_ipc) */
    bool ipc = false;
    std::string ipc_dir; double ipc_timeout_s = 30.0; unsigned ipc_epoch = 0;
    DevBuf ipc_flags, ipc_out;
    struct IpcPeer { unsigned char handle[7][64]; void* ptr[7]; };   /* flags, g_num[0..1], g_den[0..1], basic, out -- as this process maps them */
    std::vector<IpcPeer> ipc_peers;
    DevBuf pristine, pristine_b;
    /* run-time options (lfbm5d_options.h): filled from the environment once at lfbm5d_create, changed by lfbm5d_set_option; lane contexts
     * point at their parent's */
    Options opt_store; Options* opt = &opt_store;
    hipStream_t io_in = nullptr, io_out = nullptr;
    unsigned* h_small = nullptr; /* pinned, 64 uints */
    /* window lanes (run_step, pipelined form): extra contexts on the same device, each with its own stream, window
     * buffers and per-pass work buffers; owned by this context */
    std::vector<lfbm5d_ctx*> lanes;
    unsigned* h_counts = nullptr; size_t h_counts_cap = 0;   /* pinned: coverage count of every window of a step */
    unsigned long long lane_windows = 0;   /* windows of the last step that ran on a lane other than the first */
    std::vector<hipEvent_t> ev_pool; size_t ev_used = 0;
    std::vector<PassEvents> pending;
    /* last pass (inspection) */
    unsigned last_n_refs = 0, last_N = 0, last_A = 0; size_t last_plane = 0; int last_gslot = 0;   /* geometry slot of that pass */
};

namespace {

#define HIPCK(ctx, call)                                                                         \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess) {                                                                  \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                      \
            return 1;                                                                            \
        }                                                                                        \
    } while (0)

int fail(lfbm5d_ctx* c, const std::string& m) { c->err = m; return 1; }

lfbm5d_ctx* new_ctx(int device, std::string& err) {
    hipError_t e;
    lfbm5d_ctx* c = new lfbm5d_ctx();
    c->device = device;
    std::memset(&c->stats, 0, sizeof(c->stats));
    if ((e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess) { err = hipGetErrorString(e); delete c; return nullptr; }
    if ((e = prepare_group_kernels()) != hipSuccess || (e = prepare_scan2_kernels()) != hipSuccess) { err = std::string("kernel LDS limits: ") + hipGetErrorString(e); (void)hipStreamDestroy(c->stream); delete c; return nullptr; }
    if ((e = hipHostMalloc((void**)&c->h_small, 64 * sizeof(unsigned))) != hipSuccess) { err = hipGetErrorString(e); (void)hipStreamDestroy(c->stream); delete c; return nullptr; }
    return c;
}


hipEvent_t get_event(lfbm5d_ctx* c) {
    if (c->ev_used == c->ev_pool.size()) {
        hipEvent_t e; (void)hipEventCreate(&e); c->ev_pool.push_back(e);
    }
    return c->ev_pool[c->ev_used++];
}

/* fold finished passes' event times into the stats (stream must be idle) */
void drain_events(lfbm5d_ctx* c) {
    for (const PassEvents& pe : c->pending) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, pe.e[0], pe.e[1]) == hipSuccess) c->stats.ms_bm += ms;
        if (hipEventElapsedTime(&ms, pe.e[1], pe.e[2]) == hipSuccess) c->stats.ms_group += ms;
        if (hipEventElapsedTime(&ms, pe.e[2], pe.e[3]) == hipSuccess) c->stats.ms_aggregate += ms;
        if (pe.comm && hipEventElapsedTime(&ms, pe.e[3], pe.e[4]) == hipSuccess) c->stats.ms_comm += ms;
    }
    c->pending.clear();
    c->ev_used = 0;
}

/* utilities.cpp:633-684 */
int sigma_table(float sigma, unsigned C, unsigned cs, float* out) {
    if (C == 1) { out[0] = sigma; return 0; }
    if (cs == LFBM5D_YUV) {
        out[0] = std::sqrt(0.299f * 0.299f + 0.587f * 0.587f + 0.114f * 0.114f) * sigma;
        out[1] = std::sqrt(0.14713f * 0.14713f + 0.28886f * 0.28886f + 0.436f * 0.436f) * sigma;
        out[2] = std::sqrt(0.615f * 0.615f + 0.51498f * 0.51498f + 0.10001f * 0.10001f) * sigma;
    } else if (cs == LFBM5D_YCBCR) {
        out[0] = std::sqrt(0.299f * 0.299f + 0.587f * 0.587f + 0.114f * 0.114f) * sigma;
        out[1] = std::sqrt(0.169f * 0.169f + 0.331f * 0.331f + 0.500f * 0.500f) * sigma;
        out[2] = std::sqrt(0.500f * 0.500f + 0.419f * 0.419f + 0.081f * 0.081f) * sigma;
    } else if (cs == LFBM5D_OPP) {
        out[0] = std::sqrt(0.333f * 0.333f + 0.333f * 0.333f + 0.333f * 0.333f) * sigma;
        out[1] = std::sqrt(0.5f * 0.5f + 0.0f * 0.0f + 0.5f * 0.5f) * sigma;
        out[2] = std::sqrt(0.25f * 0.25f + 0.5f * 0.5f + 0.25f * 0.25f) * sigma;
    } else if (cs == LFBM5D_RGB) {
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

/* bm3d.cpp:1101-1169, core:3191-3252, lib_transforms.cpp:215-277 */
void build_tables(GroupTables& t, unsigned k, unsigned aw, unsigned ah) {
    std::memset(&t, 0, sizeof(t));
    static const float q8[4][4] = {{0.1924f, 0.2989f, 0.3846f, 0.4325f}, {0.2989f, 0.4642f, 0.5974f, 0.6717f},
                                   {0.3846f, 0.5974f, 0.7688f, 0.8644f}, {0.4325f, 0.6717f, 0.8644f, 0.9718f}};
    static const float q12[6][6] = {{0.1924f, 0.2615f, 0.3251f, 0.3782f, 0.4163f, 0.4362f},
                                    {0.2615f, 0.3554f, 0.4419f, 0.5139f, 0.5657f, 0.5927f},
                                    {0.3251f, 0.4419f, 0.5494f, 0.6390f, 0.7033f, 0.7369f},
                                    {0.3782f, 0.5139f, 0.6390f, 0.7433f, 0.8181f, 0.8572f},
                                    {0.4163f, 0.5657f, 0.7033f, 0.8181f, 0.9005f, 0.9435f},
                                    {0.4362f, 0.5927f, 0.7369f, 0.8572f, 0.9435f, 0.9885f}};
    const float coef = 0.5f / (float)k;
    for (unsigned i = 0; i < k; i++)
        for (unsigned j = 0; j < k; j++) {
            const unsigned h = k / 2, a = i < h ? i : k - 1 - i, b = j < h ? j : k - 1 - j;
            t.kaiser[i * k + j] = k == 8 ? q8[a][b] : (k == 12 ? q12[a][b] : 1.0f);
            if (i == 0 && j == 0) { t.cn2[0] = 0.5f * coef; t.cni2[0] = 2.0f; }
            else if (i * j == 0)  { t.cn2[i * k + j] = (float)(kSqrt2Inv * coef); t.cni2[i * k + j] = (float)kSqrt2; }
            else                  { t.cn2[i * k + j] = coef; t.cni2[i * k + j] = 1.0f; }
            t.cos2[i * k + j] = (float)std::cos(kPi * (j + 0.5) * i / k);
        }
    const float c4 = 0.5f / (std::sqrt((float)aw) * std::sqrt((float)ah));
    for (unsigned i = 0; i < ah; i++)
        for (unsigned j = 0; j < aw; j++) {
            if (i == 0 && j == 0) { t.cn4[0] = (float)(0.5f * c4); t.cni4[0] = 2.0f; }
            else if (i * j == 0)  { t.cn4[i * aw + j] = (float)(kSqrt2Inv * c4); t.cni4[i * aw + j] = (float)kSqrt2; }
            else                  { t.cn4[i * aw + j] = c4; t.cni4[i * aw + j] = 1.0f; }
        }
    for (unsigned u = 0; u < 3; u++)
        for (unsigned j = 0; j < 3; j++) t.cos3[u * 3 + j] = (float)std::cos(kPi * (j + 0.5) * u / 3.0);
    for (unsigned u = 0; u < aw && aw <= (unsigned)kBigAw; u++)
        for (unsigned j = 0; j < aw; j++) t.cosw[u * aw + j] = (float)std::cos(kPi * (j + 0.5) * u / (double)aw);
    for (unsigned n = 1; n <= (unsigned)kBigAw; n++) {
        for (unsigned u = 0; u < n; u++)
            for (unsigned j = 0; j < n; j++) t.cos1[n][u * n + j] = (float)std::cos(kPi * (j + 0.5) * u / n);
        const float c1 = (float)((float)kSqrt2 / std::sqrt((double)n));
        t.cn1[n][0] = (float)(kSqrt2Inv * c1); t.cni1[n][0] = (float)kSqrt2;
        for (unsigned i = 1; i < n; i++) { t.cn1[n][i] = c1; t.cni1[n][i] = 1.0f; }
        t.c1inv[n] = 0.5f * (float)kSqrt2Inv / std::sqrt((float)n);
    }
    for (unsigned l = 0; l < 6; l++) {
        const unsigned n = 1u << l;
        float* ct = l < 5 ? t.cos5[l] : t.cos5x;
        for (unsigned uu = 0; uu < n; uu++)
            for (unsigned j = 0; j < n; j++) ct[uu * n + j] = (float)std::cos(kPi * (j + 0.5) * uu / n);
        const float c5 = (float)((float)kSqrt2 / std::sqrt((double)n));
        t.cn5_0[l] = (float)(kSqrt2Inv * c5); t.cn5[l] = c5;
        t.c5inv[l] = 0.5f * (float)kSqrt2Inv / std::sqrt((float)n);
    }
    const float cn = 1.f / (std::sqrt(2.f) * 128.f), s = 1.f / std::sqrt(2.f);
    const float a1[10] = {3.f, -3.f, -22.f, 22.f, 128.f, 128.f, 22.f, -22.f, -3.f, 3.f};
    const float b1[10] = {3.f, 3.f, -22.f, -22.f, 128.f, -128.f, 22.f, 22.f, -3.f, -3.f};
    for (int i = 0; i < 10; i++) { t.lpd[i] = a1[i] * cn; t.hpr[i] = b1[i] * cn; }
    t.hpd[4] = -s; t.hpd[5] = s; t.lpr[4] = s; t.lpr[5] = s;
    t.coef2inv = 1.0f / (float)(k * 2);
    t.coef4inv = 1.0f / (std::sqrt((float)aw) * std::sqrt((float)ah) * 2.0f);
    if (aw == 3 && ah == 3) {   /* group_id_compute_fast: see GroupTables */
        const double r3 = std::sqrt(3.0), alpha[3] = {2.0, r3, 1.0}, gamma[3] = {1.0, r3, 1.0};
        for (unsigned v = 0; v < 3; v++)
            for (unsigned u = 0; u < 3; u++) {
                const double F = alpha[v] * alpha[u] * (double)t.cn4[v * 3 + u];
                t.ht3_f[v * 3 + u] = (float)F;
                t.ht3_gf[v * 3 + u] = (float)(F * (double)t.cni4[v * 3 + u] * (double)t.coef4inv * gamma[v] * gamma[u]);
            }
    }
}

bool is_pow2(unsigned n) { return n && !(n & (n - 1)); }

int validate(lfbm5d_ctx* c, int step, const lfbm5d_params* P, unsigned aw, unsigned ah, unsigned C, bool bm3d = false) {
    if (bm3d) {   /* per-SAI BM3D flavour: one image, search band = search window, Hadamard along the stack */
        if (aw != 1 || ah != 1) return fail(c, "BM3D works on single images");
        if (C != 1 && C != 3) return fail(c, "unsupported: chnls must be 1 or 3");
        if (P->k < 2 || P->k > (unsigned)kMaxK) return fail(c, "unsupported: patch size k outside 2..32");
        if (P->tau_2D != LFBM5D_DCT && P->tau_2D != LFBM5D_BIOR) return fail(c, "BM3D: tau_2D must be dct or bior");
        if (P->tau_2D == LFBM5D_BIOR && !is_pow2(P->k)) return fail(c, "bior1.5 needs a power-of-two patch size");
        if (!is_pow2(P->N) || P->N < 2 || P->N > (unsigned)kMaxN3) return fail(c, "unsupported: BM3D N must be a power of two in 2..32");
        if (P->nSim < 1 || P->nSim > 48 || P->p < 1) return fail(c, "bad search window / step");
        return 0;
    }
    /* any odd window side up to 17 (aswSize 1 .. 8): 3x3 on the dedicated kernels, 5x5 and 7x7 on the generic kernel's register forms,
     * 9x9 and more on its general forms (run-time transform sizes, stacks in HBM: slow, but the reference's whole range for light
     * fields of up to 17x17 SAIs, bm5d.cpp:119-124) */
    if (aw != ah || aw < 3 || !(aw & 1) || aw > (unsigned)kBigAw) return fail(c, "unsupported: angular search window must be a square of 3 .. 17 SAIs a side (aswSize 1 to 8)");
    if (C != 1 && C != 3) return fail(c, "unsupported: chnls must be 1 or 3");
    /* any patch size the reference would run (utilities_LF.cpp:1214, :1255; Kaiser window: all ones unless k is 8 or 12, bm3d.cpp:1144-1146);
     * 8, 12 and 16 have dedicated table kernels, 8 and 16 dedicated group kernels, everything else the general forms.  32 bounds the tables */
    if (P->k < 2 || P->k > (unsigned)kMaxK) return fail(c, "unsupported: patch size k outside 2..32");
    if (P->tau_2D == LFBM5D_BIOR && !is_pow2(P->k)) return fail(c, "bior1.5 needs a power-of-two patch size");
    if (P->tau_2D != LFBM5D_ID && P->tau_2D != LFBM5D_DCT && P->tau_2D != LFBM5D_BIOR) return fail(c, "bad tau_2D");
    if (P->tau_4D != LFBM5D_ID && P->tau_4D != LFBM5D_DCT && P->tau_4D != LFBM5D_SADCT) return fail(c, "bad tau_4D");
    if (P->tau_5D != LFBM5D_HAAR && P->tau_5D != LFBM5D_HADAMARD && P->tau_5D != LFBM5D_DCT) return fail(c, "bad tau_5D");
    if (!is_pow2(P->N) || P->N > (unsigned)kMaxN3) return fail(c, "unsupported: N must be a power of two <= 32");
    if (P->nSim < 1 || P->nDisp < 1 || P->p < 1) return fail(c, "bad search window / step");
    /* kernel limits: the row-slot tables carry 64 entries of padding for rows y + di, di <= nSim; candidate
     * indices are divided by 2 nSim + 1 with a 20-bit reciprocal; displacement tables are (2 nDisp + 1)^2 per SAI */
    if (P->nSim > 48 || P->nDisp > 24) return fail(c, "unsupported: nSim > 48 or nDisp > 24");
    (void)step;
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* One core pass                                                                                */
/* ------------------------------------------------------------------------------------------ */
int pass_impl(lfbm5d_ctx* c, int step, const lfbm5d_params* P, unsigned aw, unsigned ah, unsigned Wb,
              unsigned Hb, unsigned C, const float* d_noisy, const float* d_basic, float* d_num,
              float* d_den, const unsigned* h_mask, const unsigned* h_proc, unsigned cst, unsigned pst,
              bool bm3d = false) {
    /* the graph form's "estimate already formed" flag belongs to this call only: consumed before anything can fail, so that an early
     * error return cannot leave it set for the next pass on this context */
    const bool est_ready = c->est_ready;
    c->est_ready = false;
    if (validate(c, step, P, aw, ah, C, bm3d)) return 1;
    if (step == 2 && !d_basic) return fail(c, "step 2 needs the basic estimate");
    const unsigned A = aw * ah, k = P->k, k2 = k * k, N = P->N, nHW = P->nSim + P->nDisp;
    const size_t plane = (size_t)Wb * Hb;
    hipStream_t s = c->stream;
    GeomCache& gc = c->gc[c->gslot];
    if (Hb < 2 * nHW + k + 1 || Wb < 2 * nHW + k + 1) return fail(c, "window smaller than the search range");
    if (Hb > 65535 || Wb > 65535) return fail(c, "unsupported: window larger than 65535 pixels a side");

    float sig[3] = {0, 0, 0};
    if (sigma_table(P->sigma, C, P->color_space, sig)) return fail(c, "bad color space");
    const float tauMatch = bm3d ? (step == 1 ? (C == 1 ? 3.f : 1.f) * (sig[0] < 35.0f ? 2500 : 5000)        /* bm3d.cpp:339 */
                                             : (sig[0] < 35.0f ? 400.f : 3500.f))                           /* bm3d.cpp:531 */
                                : (C == 1 ? 3.f : 1.f) * (sig[0] < 35.0f ? (step == 1 ? 3000 : 2000) : 5000); /* core:146/:915 */
    const float thr = tauMatch * k * k;                                                                  /* core:3315 */
    float lambda = P->lambda;
    if (!bm3d && step == 1 && P->tau_2D == LFBM5D_ID && P->tau_4D == LFBM5D_DCT) lambda /= (float)kSqrt2; /* core:206-207 */
    SaiMask mask_bits = sai_mask_none(), proc_bits = sai_mask_none();
    for (unsigned st = 0; st < A; st++) { if (h_mask[st]) mask_bits.set(st); if (h_proc[st]) proc_bits.set(st); }
    if (pst >= A || cst >= A) return fail(c, "cst / pst outside the angular window");
    if (!mask_bits.test(pst)) return fail(c, "processed SAI is empty");

    /* reference grid (core:149-156); cached while the geometry is unchanged */
    const bool centre = pst == cst;
    const unsigned key[5] = {Wb, Hb, k, nHW, P->p};
    if (centre && (std::memcmp(key, gc.grid_key, sizeof(key)) != 0 || gc.last_refs_host.empty())) {
        std::vector<unsigned> rows, cols;
        ind_init(rows, Hb - k + 1, nHW, P->p);
        ind_init(cols, Wb - k + 1, nHW, P->p);
        gc.n_ref_rows = (unsigned)rows.size(); gc.n_ref_cols = (unsigned)cols.size();
        gc.last_refs_host.resize(rows.size() * cols.size());
        for (size_t i = 0; i < rows.size(); i++)
            for (size_t j = 0; j < cols.size(); j++) gc.last_refs_host[i * cols.size() + j] = rows[i] * Wb + cols[j];
        std::vector<int> rslot(Hb + 64, -1);   /* 64 slots of padding: the scan reads rslot[y + di] unclamped */
        for (size_t i = 0; i < rows.size(); i++) rslot[rows[i]] = (int)i;
        HIPCK(c, gc.rslot.reserve(rslot.size() * sizeof(int)));
        HIPCK(c, hipMemcpyAsync(gc.rslot.p, rslot.data(), rslot.size() * sizeof(int), hipMemcpyHostToDevice, s));
        HIPCK(c, gc.refs.reserve(gc.last_refs_host.size() * sizeof(unsigned)));
        HIPCK(c, hipMemcpyAsync(gc.refs.p, gc.last_refs_host.data(), gc.last_refs_host.size() * sizeof(unsigned), hipMemcpyHostToDevice, s));
        HIPCK(c, hipStreamSynchronize(s));
        std::memcpy(gc.grid_key, key, sizeof(key));
        std::memcpy(gc.rslot_key, key, sizeof(key));
    }
    unsigned R = gc.n_ref_rows * gc.n_ref_cols;
    const unsigned R_full = R;
    std::vector<unsigned> row_start;   /* subset path: first reference of every listed row (+ end) */
    /* the list on the device (round 4): the flagged patches of the regular grid in raster order -- what the host loop below
     * produces, without the copy of the plane and the 4 M comparisons a pass (1 / 0.4 ms of host time, a third of a greyscale job).
     * Row shards need the rows' first entries on the host and keep the host form */
    const bool dev_list = !centre && c->pass_world == 1 && std::memcmp(key, gc.rslot_key, sizeof(key)) == 0 && !(c->opt->kernels & kOptSubsetListHost);
    if (dev_list) {
        HIPCK(c, c->sub_flags.reserve((size_t)R_full));
        HIPCK(c, c->sub_cnt.reserve(sizeof(unsigned)));
        HIPCK(c, gc.refs.reserve((size_t)R_full * sizeof(unsigned)));
        unsigned* const d_cnt = c->sub_cnt.as<unsigned>();
        HIPCK(c, launch_subset_list(s, d_den + (size_t)pst * C * plane, Wb, k, nHW, P->p, gc.n_ref_rows, gc.n_ref_cols, Hb - k - nHW, Wb - k - nHW,
                                    reinterpret_cast<unsigned char*>(c->sub_flags.p), gc.refs.as<unsigned>(), d_cnt));
        HIPCK(c, hipMemcpyAsync(&R, d_cnt, sizeof(unsigned), hipMemcpyDeviceToHost, s));
        HIPCK(c, hipStreamSynchronize(s));
        std::memset(gc.grid_key, 0, sizeof(gc.grid_key));   /* the cached regular grid is gone */
        gc.last_refs_host.resize(R);
        if (R == 0) { c->last_n_refs = 0; return 0; }   /* nothing left to denoise (core:160-165) */
        HIPCK(c, hipMemcpyAsync(gc.last_refs_host.data(), gc.refs.p, R * sizeof(unsigned), hipMemcpyDeviceToHost, s));   /* lfbm5d_last_bm */
        HIPCK(c, hipStreamSynchronize(s));
        row_start.assign({0u, R});
        if (N > 1 && (c->opt->kernels & kOptSubsetScanV1)) {   /* (the test hook's table kernel stores through the position map) */
            HIPCK(c, c->refmap.reserve(plane * sizeof(int)));
            HIPCK(c, launch_fill_i32(s, c->refmap.as<int>(), -1, plane));
            HIPCK(c, launch_refmap(s, gc.refs.as<unsigned>(), R, c->refmap.as<int>()));
        }
    } else
    if (!centre) {
        /* Subset path (core:157-158, utilities_LF.cpp:1000-1099): only reference patches whose k x k
         * footprint still holds an exactly-zero weight in channel 0 of den[pst]; one extra column /
         * row at the far border like ind_initialize.  The list is built on the host from a copy of
         * that plane (1.2 MB at 560^2; this path only runs for greyscale light fields). */
        std::vector<float> den0(plane);
        HIPCK(c, hipMemcpyAsync(den0.data(), d_den + (size_t)pst * C * plane, plane * sizeof(float), hipMemcpyDeviceToHost, s));
        HIPCK(c, hipStreamSynchronize(s));
        auto denoised = [&](unsigned p_idx) {
            for (unsigned pp = 0; pp < k; pp++)
                for (unsigned q = 0; q < k; q++)
                    if (den0[p_idx + pp * Wb + q] == 0.0f) return false;
            return true;
        };
        const unsigned max_h = Hb - k + 1, max_w = Wb - k + 1;
        std::vector<unsigned> refs, tmp;
        row_start.clear();
        auto scan_row = [&](unsigned i) {
            tmp.clear();
            for (unsigned j = nHW; j < max_w - nHW; j += P->p)
                if (!denoised(i * Wb + j)) tmp.push_back(j);
            const bool border = tmp.empty() ? true : (tmp.back() < max_w - nHW - 1);
            if (border && !denoised(i * Wb + max_w - nHW - 1)) tmp.push_back(max_w - nHW - 1);
            if (!tmp.empty()) { row_start.push_back((unsigned)refs.size()); for (unsigned j : tmp) refs.push_back(i * Wb + j); return true; }
            return false;
        };
        unsigned last_row = 0; bool any = false;
        for (unsigned i = nHW; i < max_h - nHW; i += P->p) if (scan_row(i)) { last_row = i; any = true; }
        if (!any || last_row < max_h - nHW - 1) scan_row(max_h - nHW - 1);
        row_start.push_back((unsigned)refs.size());
        gc.last_refs_host = refs;
        std::memset(gc.grid_key, 0, sizeof(gc.grid_key));   /* the cached regular grid is gone */
        R = (unsigned)refs.size();
        if (R == 0) { c->last_n_refs = 0; return 0; }   /* nothing left to denoise (core:160-165) */
        HIPCK(c, gc.refs.reserve(R * sizeof(unsigned)));
        HIPCK(c, hipMemcpyAsync(gc.refs.p, refs.data(), R * sizeof(unsigned), hipMemcpyHostToDevice, s));
        HIPCK(c, c->refmap.reserve(plane * sizeof(int)));
        HIPCK(c, launch_fill_i32(s, c->refmap.as<int>(), -1, plane));
        HIPCK(c, launch_refmap(s, gc.refs.as<unsigned>(), R, c->refmap.as<int>()));
        HIPCK(c, hipStreamSynchronize(s));   /* refs is a stack vector */
    }

    const unsigned NsS = 2 * P->nSim + 1, NsD = 2 * P->nDisp + 1;
    unsigned slots[kBigA]; unsigned n_slots = 0;
    for (unsigned st = 0; st < A; st++) if (st != pst && mask_bits.test(st)) slots[n_slots++] = st;
    const unsigned Nst = N > 1 ? N : 1;
    /* slack on both sides: the scan's 16-byte row loads start one column left of the band (one float before
     * the first plane for the left-most displacement) and overrun the last row by less than a ring row */
    HIPCK(c, c->est.reserve((kEstLead + A * plane + 256) * sizeof(float)));
    float* const est = c->est.as<float>() + kEstLead;
    /* the scan addresses the score table through a buffer resource with 32-bit offsets */
    /* Subset passes (round 4): their list is part of the regular grid (rows / columns of ind_initialize), so the table kernel runs
     * on the full grid exactly as in a centre pass -- the second-generation kernel, whose score stores follow the grid's pattern --
     * and the selection takes a reference's scores from its place in that grid.  (Before: round 2's kernel with a position map,
     * 2.4 instead of 0.8 ms per pass, five passes per window on a greyscale light field.) */
    const bool full_scan = !centre && N > 1 && std::memcmp(key, gc.rslot_key, sizeof(key)) == 0 && !(c->opt->kernels & kOptSubsetScanV1);
    const unsigned R_sc = full_scan ? R_full : R;   /* rows of the score table */
    if (N > 1 && (size_t)R_sc * NsS * NsS * sizeof(float) > 0x7fffffffull) return fail(c, "unsupported: candidate score table of 2 GiB or more (reference patches x (2 nSim + 1)^2 x 4 B)");
    if (N > 1) HIPCK(c, c->scores.reserve((size_t)R_sc * NsS * NsS * sizeof(float)));
    HIPCK(c, c->self_idx.reserve((size_t)R * Nst * sizeof(unsigned)));
    HIPCK(c, c->self_cnt.reserve((size_t)R * sizeof(unsigned)));
    HIPCK(c, c->best.reserve(A * plane * sizeof(unsigned)));
    HIPCK(c, c->shape.reserve(A * plane));
    /* The filtered patches of a pass -- R x N x A x C x k^2 floats, 3.5 GB at the headline's hard-thresholding window -- exist between the
     * group kernel and the aggregation only, and the aggregation adds up in raster order of the reference patches: a pass can be cut
     * into BANDS of reference rows, group kernel and aggregation launched band after band, with sums bit-identical to the single
     * launch and a buffer of one band.  Bands are taken when the whole buffer would pass kFiltCapBytes (large angular windows: a 9x9
     * window with 16x16 patches is 31 GB) or the aggregation's 32-bit patch offsets, or when LFBM5D_BAND_MB asks (experiments: a band
     * that stays in the 256 MB Infinity Cache between its two kernels). */
    const size_t per_group = (size_t)Nst * A * C * k2;   /* floats */
    unsigned band_groups = R;
    {
        constexpr size_t kFiltCapBytes = (size_t)12 << 30;
        size_t cap = std::min<size_t>(kFiltCapBytes, (size_t)0xfff00000ull * sizeof(float));   /* 32-bit float offsets inside a band */
        if (c->opt->band_mb > 0) cap = std::min<size_t>(cap, (size_t)c->opt->band_mb << 20);
        const size_t row_groups = centre ? gc.n_ref_cols : 1;   /* bands are whole rows of the reference grid (a list: any cut) */
        if ((size_t)R * per_group * sizeof(float) > cap) {
            const size_t rows_fit = std::max<size_t>(1, cap / (per_group * sizeof(float) * row_groups));
            band_groups = (unsigned)std::min<size_t>(R, rows_fit * row_groups);
        }
    }
    HIPCK(c, c->filt.reserve((size_t)band_groups * per_group * sizeof(float)));
    HIPCK(c, c->wgt.reserve((size_t)R * C * sizeof(float)));
    HIPCK(c, c->aggpos.reserve((size_t)A * R * Nst * sizeof(unsigned)));
    HIPCK(c, c->gpos.reserve((size_t)A * R * Nst * sizeof(unsigned)));
    HIPCK(c, c->gofs.reserve((size_t)A * R * Nst * sizeof(unsigned)));
    HIPCK(c, c->gok.reserve((size_t)R * Nst * sizeof(unsigned)));
    HIPCK(c, c->sa_list.reserve((size_t)(4 * (size_t)R + 1) * sizeof(unsigned)));   /* a group's three channels can be listed one by one, and once as a whole */
    HIPCK(c, c->gshape.reserve((size_t)R * (A > (unsigned)kMaxA ? kShapeInfoBigBytes : kShapeInfoBytes)));
    HIPCK(c, gc.tb.reserve(sizeof(GroupTables)));
    if (!c->counters.p) {   /* [step slot][16]: sum nSx, shape-adaptive groups, development clocks */
        HIPCK(c, c->counters.reserve(32 * sizeof(unsigned long long)));
        HIPCK(c, hipMemsetAsync(c->counters.p, 0, 32 * sizeof(unsigned long long), s));
    }
    unsigned long long* const d_counters = c->counters.as<unsigned long long>() + 16 * c->gslot;
    if (gc.tb_key[0] != k || gc.tb_key[1] != aw || gc.tb_key[2] != ah) {   /* constant tables: uploaded when the geometry changes */
        GroupTables tb;
        build_tables(tb, k, aw, ah);
        HIPCK(c, hipMemcpyAsync(gc.tb.p, &tb, sizeof(tb), hipMemcpyHostToDevice, s));
        HIPCK(c, hipStreamSynchronize(s)); /* tb is a stack object */
        gc.tb_key[0] = k; gc.tb_key[1] = aw; gc.tb_key[2] = ah;
    }

    PassEvents pe; pe.comm = false;
    for (int i = 0; i < 5; i++) pe.e[i] = get_event(c);

    /* current estimate for matching, channel 0 (core:167-170) */
    const float* sub = step == 1 ? d_noisy : d_basic;
    if (!est_ready) HIPCK(c, launch_estimate_multi(s, d_num, d_den, sub, est, plane, C, A, mask_bits));
    /* multi-GPU: ranks > 0 accumulate their shard into zeroed buffers; the all-reduce restores
     * base + all contributions on every rank */
    if (c->pass_world > 1 && c->pass_rank > 0) {
        HIPCK(c, hipMemsetAsync(d_num, 0, A * C * plane * sizeof(float), s));
        HIPCK(c, hipMemsetAsync(d_den, 0, A * C * plane * sizeof(float), s));
    }

    HIPCK(c, hipEventRecord(pe.e[0], s));
    /* block matching (core:209-236): all distance tables in one launch, then the two selections */
    ScanArgs sa;
    std::memset(&sa, 0, sizeof(sa));
    sa.dbg = d_counters + 4;
    sa.est = est; sa.W = Wb; sa.H = Hb; sa.k = k; sa.pst = pst;
    sa.nSim = P->nSim; sa.nDisp = P->nDisp; sa.nHW = nHW;
    sa.n_ref_rows = gc.n_ref_rows; sa.n_ref_cols = gc.n_ref_cols; sa.p = P->p;
    sa.scores = c->scores.as<float>(); sa.tables = c->tables.as<float>(); sa.rslot = gc.rslot.as<int>(); sa.refmap = (centre || full_scan) ? nullptr : c->refmap.as<int>(); sa.scores_bytes = (unsigned)((size_t)R_sc * NsS * NsS * sizeof(float));
    sa.n_self = N > 1 ? (P->nSim + 1) * NsS : 0;
    sa.n_stereo = n_slots * NsD * NsD;
    for (unsigned i = 0; i < n_slots; i++) sa.st_of_slot[i] = slots[i];
    sa.est_planes = A;
    if (N > 1) HIPCK(c, launch_fill_f32(s, c->scores.as<float>(), 2 * thr, (size_t)R_sc * NsS * NsS));
    /* which generation of the table kernel, and its workgroup list: functions of the search geometry (and of the two
     * environment switches bm_scan_version reads), cached with it */
    sa.opt = c->opt->kernels; sa.lds_cap = (unsigned)std::max(0, c->opt->scan_lds_cap);
    const bool opt_v1 = (sa.opt & (kOptScanV1 | kOptScanAny)) != 0, opt_ft = (sa.opt & kOptScanFullTables) != 0;
    const unsigned skey[8] = {sa.n_self, sa.n_stereo, sa.nSim, sa.nDisp, sa.k, Hb, Wb,
                              1u | ((centre || full_scan) ? 0u : 2u) | (opt_v1 ? 4u : 0u) | (opt_ft ? 8u : 0u) | ((sa.opt & kOptScanAny) ? 16u : 0u)};
    const bool scan_changed = std::memcmp(skey, gc.scan_key, sizeof(skey)) != 0;
    if (scan_changed) gc.scan_version = bm_scan_version(sa);
    const int scan_version = gc.scan_version;
    c->last_scan_version = scan_version;
    if (scan_version == 1) {
        HIPCK(c, c->tables.reserve(std::max<size_t>(1, scan_tables_floats(sa, 1, n_slots, 0)) * sizeof(float)));
        sa.tables = c->tables.as<float>();
    }
    if (scan_version >= 2) {
        /* ring-sharing workgroups of eight tables (lfbm5d_scan2.hip): the list depends on the search geometry only */
        if (scan_changed) {
            if (!scan2_plan(sa, gc.scan_plan, &gc.scan_lds, &gc.scan_nwg_slot)) return fail(c, "scan plan");
            HIPCK(c, gc.scan_wgs.reserve(gc.scan_plan.size() * sizeof(Scan2Wg)));
            HIPCK(c, hipMemcpyAsync(gc.scan_wgs.p, gc.scan_plan.data(), gc.scan_plan.size() * sizeof(Scan2Wg), hipMemcpyHostToDevice, s));
            HIPCK(c, hipStreamSynchronize(s));
        }
        sa.wgs = gc.scan_wgs.as<Scan2Wg>(); sa.n_wgs = (unsigned)gc.scan_plan.size();
        sa.lcol_stride = scan2_lcol_stride(sa);
        HIPCK(c, c->scan_lcol.reserve((size_t)(sa.n_self + sa.n_stereo) * sa.lcol_stride * sizeof(float)));
        sa.lcol = c->scan_lcol.as<float>();
        sa.nwg_slot = gc.scan_nwg_slot;
        HIPCK(c, c->tables.reserve(std::max<size_t>(1, scan_tables_floats(sa, scan_version, n_slots, sa.nwg_slot)) * sizeof(float)));
        sa.tables = c->tables.as<float>();
        HIPCK(c, launch_bm_scan2(s, sa, gc.scan_lds, scan_version == 3));
    } else
        HIPCK(c, launch_bm_scan(s, sa));
    std::memcpy(gc.scan_key, skey, sizeof(skey));
    if (N > 1)
        HIPCK(c, launch_self_select(s, c->scores.as<float>(), gc.refs.as<unsigned>(), R, Wb, P->nSim, N, thr,
                                    c->self_idx.as<unsigned>(), c->self_cnt.as<unsigned>(),
                                    full_scan ? gc.n_ref_cols : 0u, nHW, P->p, Hb - k - nHW, Wb - k - nHW));
    else
        HIPCK(c, launch_self_trivial(s, gc.refs.as<unsigned>(), R, c->self_idx.as<unsigned>(), c->self_cnt.as<unsigned>()));
    if (n_slots && scan_version == 3)
        HIPCK(c, launch_stereo_argmin3(s, c->tables.as<float>(), slots, n_slots, sa.nwg_slot, Wb, Hb, k, P->nDisp, thr,
                                       c->best.as<unsigned>(), c->shape.as<unsigned char>()));
    else if (n_slots && scan_version == 2)
        HIPCK(c, launch_stereo_argmin2(s, c->tables.as<float>(), slots, n_slots, Wb, Hb, k, P->nDisp, thr,
                                       c->best.as<unsigned>(), c->shape.as<unsigned char>()));
    else if (n_slots)
        HIPCK(c, launch_stereo_argmin(s, c->tables.as<float>(), slots, n_slots, Wb, Hb, k, P->nDisp, thr,
                                      c->best.as<unsigned>(), c->shape.as<unsigned char>()));
    HIPCK(c, hipEventRecord(pe.e[1], s));

    /* shard of reference-patch rows owned by this rank */
    unsigned ref_begin, n_groups;
    if (centre) {
        unsigned rb = 0, re = gc.n_ref_rows;
        lfbm5d_shard_rows(gc.n_ref_rows, c->pass_rank, c->pass_world, &rb, &re);
        ref_begin = rb * gc.n_ref_cols; n_groups = (re - rb) * gc.n_ref_cols;
    } else {
        unsigned rb = 0, re = (unsigned)row_start.size() - 1;
        lfbm5d_shard_rows((unsigned)row_start.size() - 1, c->pass_rank, c->pass_world, &rb, &re);
        ref_begin = row_start[rb]; n_groups = row_start[re] - row_start[rb];
    }

    GroupArgs ga;
    std::memset(&ga, 0, sizeof(ga));
    ga.noisy = d_noisy; ga.basic = d_basic; ga.num = d_num; ga.den = d_den;
    ga.refs = gc.refs.as<unsigned>(); ga.self_idx = c->self_idx.as<unsigned>(); ga.self_cnt = c->self_cnt.as<unsigned>();
    ga.best = c->best.as<unsigned>(); ga.shape = c->shape.as<unsigned char>(); ga.tb = gc.tb.as<GroupTables>();
    ga.wgt = c->wgt.as<float>(); ga.aggpos = c->aggpos.as<unsigned>(); ga.gpos = c->gpos.as<unsigned>(); ga.gofs = c->gofs.as<unsigned>(); ga.gok = c->gok.as<unsigned>(); ga.sa_list = c->sa_list.as<unsigned>(); ga.gshape = c->gshape.p; ga.n_refs_total = R; ga.counters = d_counters;
    ga.ref_begin = ref_begin; ga.n_groups = n_groups;
    ga.Wb = Wb; ga.Hb = Hb; ga.C = C; ga.A = A; ga.k = k; ga.N = Nst; ga.pst = pst;
    ga.mask_bits = mask_bits; ga.proc_bits = proc_bits;
    ga.tau2 = P->tau_2D; ga.tau4 = P->tau_4D; ga.tau5 = P->tau_5D; ga.useSD = P->useSD;
    ga.step = step; ga.lambda = lambda; ga.fill_quirk = centre ? 1u : 0u;
    for (int i = 0; i < 3; i++) ga.sigma[i] = sig[i];
    if (A == 9 && step == 1) {   /* thresholds of the unnormalised transform chain (3x3 windows, Haar fibres): GroupArgs::ht3_T */
        GroupTables ht;   /* (the same constants as the device table's) */
        build_tables(ht, k, 3, 3);
        for (int ch = 0; ch < 3; ch++) {
            const float T = lambda * sig[ch] * 1.41421356237309505f;   /* the kernels' own float expression (core:2431) */
            for (int st = 0; st < 9; st++)
                for (int l = 0; l < 4; l++) ga.ht3_T[ch][st][l] = (float)((double)T / ((double)ht.ht3_f[st] * std::pow(2.0, -0.5 * l)));
        }
    }
    ga.bm3d = bm3d ? 1u : 0u;
    ga.opt = c->opt->kernels;
    if (const size_t sb = group_scratch_bytes(ga)) {   /* generic path with stacks beyond the 160 KiB LDS: HBM scratch slices */
        HIPCK(c, c->gscratch.reserve(sb));
        ga.scratch = c->gscratch.as<float>(); ga.scratch_floats = sb / sizeof(float);
    }
    AggArgs aa;
    std::memset(&aa, 0, sizeof(aa));
    aa.num = d_num; aa.den = d_den; aa.wgt = ga.wgt; aa.aggpos = ga.aggpos; aa.n_refs_total = R; aa.refs = ga.refs;
    aa.self_idx = ga.self_idx; aa.self_cnt = ga.self_cnt; aa.best = ga.best; aa.shape = ga.shape; aa.tb = ga.tb;
    aa.ref_begin = ref_begin; aa.n_groups = n_groups; aa.n_ref_rows = gc.n_ref_rows; aa.n_ref_cols = gc.n_ref_cols;
    aa.Wb = Wb; aa.Hb = Hb; aa.C = C; aa.A = A; aa.k = k; aa.N = Nst; aa.pst = pst; aa.p = P->p;
    aa.nHW = nHW; aa.nSim = P->nSim; aa.nDisp = P->nDisp;
    aa.mask_bits = mask_bits; aa.proc_bits = proc_bits; aa.tau4 = P->tau_4D; aa.irregular = centre ? 0u : 1u;
    aa.wchan0 = (bm3d && P->useSD) ? 1u : 0u;
    aa.opt = c->opt->kernels;
    if (n_groups <= band_groups) {   /* the whole pass (or this rank's rows) at once */
        ga.filt = c->filt.as<float>() - (size_t)ref_begin * per_group;   /* (the group kernels index filt by absolute group number) */
        aa.filt = c->filt.as<float>(); aa.filt_bytes = (unsigned long long)n_groups * per_group * sizeof(float);
        if (n_groups) HIPCK(c, launch_group(s, ga));
        HIPCK(c, hipEventRecord(pe.e[2], s));
        if (n_groups) HIPCK(c, launch_aggregate(s, aa));
        HIPCK(c, hipEventRecord(pe.e[3], s));
    } else {
        /* band after band; the two event intervals then cover the first band's group kernel / everything behind it */
        bool first = true;
        for (unsigned b0 = ref_begin; b0 < ref_begin + n_groups; b0 += band_groups) {
            const unsigned nb = std::min(band_groups, ref_begin + n_groups - b0);
            ga.ref_begin = b0; ga.n_groups = nb; ga.filt = c->filt.as<float>() - (size_t)b0 * per_group;
            aa.ref_begin = b0; aa.n_groups = nb; aa.filt = c->filt.as<float>(); aa.filt_bytes = (unsigned long long)nb * per_group * sizeof(float);
            HIPCK(c, launch_group(s, ga));
            if (first) HIPCK(c, hipEventRecord(pe.e[2], s));
            first = false;
            HIPCK(c, launch_aggregate(s, aa));
            c->stats.launches_group += 1; c->stats.launches_aggregate += 1;
        }
        c->stats.launches_group -= 1; c->stats.launches_aggregate -= 1;   /* (one of each is counted below) */
        HIPCK(c, hipEventRecord(pe.e[3], s));
    }

    if (c->comm && c->pass_reduce) { /* sum the window's aggregation buffers over the ranks (xGMI) */
        const size_t cnt = (size_t)A * C * plane;
        if (ncclAllReduce(d_num, d_num, cnt, ncclFloat, ncclSum, c->comm, s) != ncclSuccess) return fail(c, "ncclAllReduce(num) failed");
        if (ncclAllReduce(d_den, d_den, cnt, ncclFloat, ncclSum, c->comm, s) != ncclSuccess) return fail(c, "ncclAllReduce(den) failed");
        HIPCK(c, hipEventRecord(pe.e[4], s));
        pe.comm = true;
    }
    c->pending.push_back(pe);

    c->stats.passes += 1;
    c->stats.groups += n_groups;
    c->stats.launches_group += n_groups ? 1 : 0;
    c->stats.launches_aggregate += n_groups ? 1 : 0;
    c->last_n_refs = R; c->last_N = Nst; c->last_A = A; c->last_plane = plane; c->last_gslot = c->gslot;
    return 0;
}

/* fold the device counters (sum nSx, sadct groups) into the stats; stream must be idle */
int fold_counters(lfbm5d_ctx* c, const lfbm5d_params* P, unsigned A, unsigned C, int step, int slot = 0) {
    unsigned long long h[4] = {0, 0, 0, 0};
    if (!c->counters.p) return 0;
    unsigned long long* const d_counters = c->counters.as<unsigned long long>() + 16 * slot;
    HIPCK(c, hipMemcpyAsync(h, d_counters, sizeof(h), hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));
    HIPCK(c, hipMemsetAsync(d_counters, 0, sizeof(h), c->stream));
    c->stats.stack_patches += h[0];
    c->stats.sadct_groups += h[1];
#if defined(LFBM5D_PHASE_TIMING) || defined(LFBM5D_WIDE_PHASES) || defined(LFBM5D_SLAB_PHASES)   /* kernel-internal phase clocks of development builds (tools/build_variant.sh) */
    {
        unsigned long long ph[12];
        (void)hipMemcpy(ph, d_counters + 4, sizeof(ph), hipMemcpyDeviceToHost);
        (void)hipMemset(d_counters + 4, 0, sizeof(ph));
        std::fprintf(stderr, "[phases step %d]", step);
        for (int i = 0; i < 12; i++) std::fprintf(stderr, " %.3g", (double)ph[i]);
        std::fprintf(stderr, "\n");
    }
#endif
    /* gather 4 B * S + aggregation 16 B per stacked pixel */
    c->stats.algorithmic_bytes += (double)h[0] * A * P->k * P->k * C * (4.0 * (step == 2 ? 2 : 1) + 16.0);
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* The window graph: one step, or both steps of a denoise, executed as a dependency graph     */
/* ------------------------------------------------------------------------------------------ */
using plan::search_window;
using plan::plan_windows;

/* A JOB: run_bm5d_1st_step, run_bm5d_2nd_step, or the two back to back (lfbm5d_denoise_device).  The graph (lfbm5d_plan.h) is
 * executed on LANES -- a lane = a context of its own: stream, window buffers, per-pass work buffers -- with HIP events for the
 * dependencies between lanes; on several GPUs every rank runs the windows it owns and what a window needs from a
 * window of another rank arrives as point-to-point messages (RCCL send / recv over xGMI).  Either way every window sees exactly
 * the num / den (and, in the second step of a two-step job, the basic estimate) the window-after-window order of the
 * reference would show it: the result is bit-identical to one lane on one GPU.
 *
 * The reference decides after every pass whether the window is complete (coverage count, bm5d.cpp:370-382); for colour light
 * fields one centre pass always suffices.  The graph form assumes that, copies every window's
 * count to pinned memory and checks them all at the end (*complete). */
/* the blocking form of the host seam (jobs outside the single-rank window graph): every SAI of the caller's light field(s) up
 * before the job, every output down after it */
int io_upload_all(lfbm5d_ctx* c, const HostIO* io, const unsigned* h_mask, unsigned asize, size_t img, float* d_noisy, float* d_basic_in) {
    for (unsigned st = 0; st < asize; st++) {
        if (!h_mask[st]) continue;
        HIPCK(c, hipMemcpyAsync(d_noisy + (size_t)st * img, io->noisy[st], img * sizeof(float), hipMemcpyHostToDevice, c->stream));
        if (d_basic_in) HIPCK(c, hipMemcpyAsync(d_basic_in + (size_t)st * img, io->basic[st], img * sizeof(float), hipMemcpyHostToDevice, c->stream));
    }
    HIPCK(c, hipStreamSynchronize(c->stream));
    return 0;
}
int io_download_all(lfbm5d_ctx* c, const HostIO* io, const unsigned* h_mask, unsigned asize, size_t img, const float* d_noisy,
                    const float* d_basic, const float* d_out) {
    for (unsigned st = 0; st < asize; st++) {
        if (!h_mask[st]) continue;
        HIPCK(c, hipMemcpyAsync(io->noisy[st], d_noisy + (size_t)st * img, img * sizeof(float), hipMemcpyDeviceToHost, c->stream));
        if (d_basic) HIPCK(c, hipMemcpyAsync(io->basic[st], d_basic + (size_t)st * img, img * sizeof(float), hipMemcpyDeviceToHost, c->stream));
        if (d_out) HIPCK(c, hipMemcpyAsync(io->out[st], d_out + (size_t)st * img, img * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    }
    HIPCK(c, hipStreamSynchronize(c->stream));
    return 0;
}

/* ---- rendezvous of the two-processes-on-one-GPU transport: small files in a directory both processes see ---- */
constexpr unsigned kIpcMaxMsgs = 4096;   /* gating words: ready[kIpcMaxMsgs], taken[kIpcMaxMsgs], error */
bool ipc_put(const std::string& dir, const std::string& name, const void* data, size_t bytes) {
    const std::string tmp = dir + "/." + name + ".tmp", fin = dir + "/" + name;
    { std::ofstream f(tmp, std::ios::binary); if (!f) return false; f.write(reinterpret_cast<const char*>(data), (std::streamsize)bytes); if (!f) return false; }
    return std::rename(tmp.c_str(), fin.c_str()) == 0;
}
bool ipc_get(const std::string& dir, const std::string& name, void* data, size_t bytes, double timeout_s) {
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        std::ifstream f(dir + "/" + name, std::ios::binary);
        if (f) { f.read(reinterpret_cast<char*>(data), (std::streamsize)bytes); if (f.gcount() == (std::streamsize)bytes) return true; }
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s) return false;
        std::this_thread::sleep_for(std::chrono::milliseconds(2));
    }
}
/* every rank publishes `mine`, returns everybody's (a barrier when nobody reads the values) */
int ipc_allgather(lfbm5d_ctx* c, const char* tag, int mine, std::vector<int>& all) {
    const std::string base = std::string(tag) + "." + std::to_string(c->ipc_epoch) + ".";
    if (!ipc_put(c->ipc_dir, base + std::to_string(c->rank), &mine, sizeof(int))) return fail(c, "ipc transport: cannot write to the rendezvous directory");
    all.assign((size_t)c->world, 0);
    for (int r = 0; r < c->world; r++)
        if (!ipc_get(c->ipc_dir, base + std::to_string(r), &all[(size_t)r], sizeof(int), c->ipc_timeout_s))
            return fail(c, "ipc transport: rank " + std::to_string(r) + " did not reach '" + tag + "' within the watchdog (peer gone?)");
    return 0;
}
/* publish this rank's buffers, map every peer's (re-opened only when a peer's allocation changed) */
int ipc_exchange_handles(lfbm5d_ctx* c, void* const (&mine)[7]) {
    lfbm5d_ctx::IpcPeer me;
    std::memset(&me, 0, sizeof(me));
    for (int i = 0; i < 7; i++)
        if (mine[i]) {
            hipIpcMemHandle_t h;
            HIPCK(c, hipIpcGetMemHandle(&h, mine[i]));
            static_assert(sizeof(h) <= 64, "handle size");
            std::memcpy(me.handle[i], &h, sizeof(h));
        }
    const std::string base = "handles." + std::to_string(c->ipc_epoch) + ".";
    if (!ipc_put(c->ipc_dir, base + std::to_string(c->rank), me.handle, sizeof(me.handle))) return fail(c, "ipc transport: cannot write to the rendezvous directory");
    c->ipc_peers.resize((size_t)c->world);
    for (int r = 0; r < c->world; r++) {
        if (r == c->rank) continue;
        unsigned char hs[7][64];
        if (!ipc_get(c->ipc_dir, base + std::to_string(r), hs, sizeof(hs), c->ipc_timeout_s))
            return fail(c, "ipc transport: rank " + std::to_string(r) + " did not publish its buffers within the watchdog (peer gone?)");
        lfbm5d_ctx::IpcPeer& P = c->ipc_peers[(size_t)r];
        static const unsigned char zero[64] = {0};
        for (int i = 0; i < 7; i++) {
            if (P.ptr[i] && std::memcmp(P.handle[i], hs[i], 64) == 0) continue;
            if (P.ptr[i]) { (void)hipIpcCloseMemHandle(P.ptr[i]); P.ptr[i] = nullptr; }
            std::memcpy(P.handle[i], hs[i], 64);
            if (std::memcmp(hs[i], zero, 64) == 0) continue;
            hipIpcMemHandle_t h;
            std::memcpy(&h, hs[i], sizeof(h));
            HIPCK(c, hipIpcOpenMemHandle(&P.ptr[i], h, hipIpcMemLazyEnablePeerAccess));
        }
    }
    return 0;
}

struct GraphJob {
    int n_steps = 1;
    int step[2] = {1, 2};                          /* the reference step every slot runs */
    const lfbm5d_params* P[2] = {nullptr, nullptr};
    unsigned an[2] = {1, 1};
    const float* noisy[2] = {nullptr, nullptr};    /* the (colour-transformed) light field every slot reads */
    float* d_basic = nullptr;                      /* step 2: the pilot; two-step jobs: written SAI by SAI as the first step's sums become final */
    float* g_num[2] = {nullptr, nullptr};          /* the light field's sums, zeroed by the caller */
    float* g_den[2] = {nullptr, nullptr};
    float* d_out = nullptr;                        /* several ranks: the last slot's estimate, formed per SAI by its owner and exchanged */
    const unsigned* d_mask = nullptr;
    /* streamed host seam (one rank): the caller's SAIs are uploaded in the order the windows first use them -- forward colour
     * transform (and, two-step jobs, the round trip the second step reads) per SAI behind the copy -- and every SAI's outputs leave
     * as soon as the last window on it is done; d_noisy = the light-field buffer noisy[0] points to (the in / out LF_noisy) */
    const HostIO* io = nullptr;
    float* d_noisy = nullptr;
    float* pristine = nullptr; float* pristine_b = nullptr;
    unsigned color_space = LFBM5D_RGB;
};

int run_graph(lfbm5d_ctx* c, const GraphJob& J, const plan::Graph& G, const unsigned* h_mask, unsigned awidth, unsigned aheight,
              unsigned ang_major, unsigned W, unsigned H, unsigned C, int nranks, bool emulate, int* complete_out) {
    const unsigned asize = awidth * aheight;
    const size_t img = (size_t)C * W * H;
    hipStream_t s = c->stream;
    const size_t NN = G.nodes.size();
    const bool two = J.n_steps == 2;
    *complete_out = 1;
    /* geometry of every slot */
    struct Geo { unsigned asw, Aw, nHW, wb, hb; size_t imgb; };
    Geo geo[2];
    size_t imgb_max = 0; unsigned Aw_max = 0;
    for (int sl = 0; sl < J.n_steps; sl++) {
        Geo& g = geo[sl];
        g.asw = 2 * J.an[sl] + 1; g.Aw = g.asw * g.asw; g.nHW = J.P[sl]->nSim + J.P[sl]->nDisp;
        g.wb = W + 2 * g.nHW; g.hb = H + 2 * g.nHW; g.imgb = (size_t)C * g.wb * g.hb;
        imgb_max = std::max(imgb_max, g.Aw * g.imgb); Aw_max = std::max(Aw_max, g.Aw);
    }
    const bool any_step2 = J.step[0] == 2 || (two && J.step[1] == 2);

    struct Lane { lfbm5d_ctx* x; float* w_noisy; float* w_basic; float* w_num; float* w_den; unsigned* d_small; };
    struct RankState { int rank; lfbm5d_ctx* x; float* g_num[2]; float* g_den[2]; float* basic; std::vector<Lane> lanes; };
    auto lane_buffers = [&](lfbm5d_ctx* x, Lane& L) -> int {
        HIPCK(c, x->w_noisy.reserve(imgb_max * sizeof(float)));
        if (any_step2) HIPCK(c, x->w_basic.reserve(imgb_max * sizeof(float)));
        HIPCK(c, x->w_num.reserve(imgb_max * sizeof(float)));
        HIPCK(c, x->w_den.reserve(imgb_max * sizeof(float)));
        HIPCK(c, x->small.reserve((asize + 8 + kWinCounters) * sizeof(unsigned)));
        L.x = x; L.w_noisy = x->w_noisy.as<float>(); L.w_basic = x->w_basic.as<float>();
        L.w_num = x->w_num.as<float>(); L.w_den = x->w_den.as<float>(); L.d_small = x->small.as<unsigned>();
        return 0;
    };
    /* lanes the schedule actually uses (a 3x3 light field is one window: no extra lane, no extra buffers) */
    int lanes_used = 1;
    for (const plan::Node& nd : G.nodes) lanes_used = std::max(lanes_used, nd.lane + 1);
    const int lanes_per_rank = emulate ? 1 : lanes_used;
    const size_t need_ctx = emulate ? (size_t)nranks - 1 : (size_t)lanes_used - 1;
    while (c->lanes.size() < need_ctx) {
        std::string e;
        lfbm5d_ctx* x = new_ctx(c->device, e);
        if (!x) return fail(c, "lane context: " + e);
        x->opt = c->opt;
        c->lanes.push_back(x);
    }
    /* An error return in the middle of the graph (a failed HIP call, an RCCL call that reports an error) would leave this
     * rank's queued sends / receives waiting for peers that will never get their counterparts -- and the peers waiting for this
     * rank.  With real ranks the way out is to abort the communicators: RCCL then fails the pending operations here, the peers
     * see the failure through their own RCCL error paths (or their caller's watchdog -- bench.py has one), and every later call
     * on this context reports that the communicator is gone instead of hanging.  Disarmed when the graph has run through. */
    const bool ipc = c->ipc && nranks > 1 && !emulate;   /* ranks = processes on this GPU */
    struct AbortCommsOnError {
        lfbm5d_ctx* c; bool armed;
        ~AbortCommsOnError() {
            if (!armed) return;
            if (c->comm2) { (void)ncclCommAbort(c->comm2); c->comm2 = nullptr; }
            if (c->comm) { (void)ncclCommAbort(c->comm); c->comm = nullptr; }
            (void)hipDeviceSynchronize();
            c->err += " (multi-GPU step aborted: the RCCL communicators of this context were torn down, call lfbm5d_comm_init again)";
        }
    } abort_guard{c, nranks > 1 && !emulate && !ipc};
    /* The IPC transport has no communicator to abort: its gating kernels end by their own watchdog.  After an error the ranks may
     * have stopped at different points of the issue order (and of the rendezvous epochs), so the transport of this context is
     * closed: the next job fails at once instead of waiting for peers that are out of step. */
    struct CloseIpcOnError {
        lfbm5d_ctx* c; bool armed;
        ~CloseIpcOnError() {
            if (!armed) return;
            (void)hipDeviceSynchronize();
            for (lfbm5d_ctx::IpcPeer& P : c->ipc_peers) for (void*& q : P.ptr) if (q) { (void)hipIpcCloseMemHandle(q); q = nullptr; }
            c->ipc = false;
            c->err += " (multi-process step aborted: the IPC transport of this context was closed, call lfbm5d_comm_init_ipc again)";
        }
    } ipc_guard{c, ipc};
    /* ... and where no RCCL operation can be pending (one rank, emulated ranks, the IPC transport) an error return must not leave
     * kernels of other lanes running on the caller's buffers (which the caller is free to release once the call has failed): wait
     * for whatever has been enqueued.  With real RCCL ranks the synchronisation belongs behind the abort (abort_guard does it): in
     * front of it, it would wait for sends / receives whose peers never post their counterparts. */
    struct DrainOnError { bool armed; ~DrainOnError() { if (armed) (void)hipDeviceSynchronize(); } } drain_guard{nranks == 1 || emulate || ipc};

    std::vector<RankState> states(emulate ? (size_t)nranks : 1);
    /* two-step jobs: SAIs no window of the first step touches (LFBM5D_MAX_WINDOWS) keep the first step's input as their basic
     * estimate (bm5d.cpp:405 with den == 0), i.e. what the second step reads as noisy */
    /* the streamed host seam runs on one rank (several ranks: the caller uploads first and downloads at the end) */
    const HostIO* const io = (nranks == 1 && !emulate) ? J.io : nullptr;
    if (ipc && G.xfers.size() > kIpcMaxMsgs) return fail(c, "ipc transport: too many messages");
    if (ipc) c->ipc_epoch += 1;
    const int Ls = J.n_steps - 1;   /* the slot whose sums are the job's result */
    const bool colour_io = C == 3 && J.color_space != LFBM5D_RGB;
    std::vector<unsigned> untouched_all;
    if (two) for (unsigned st = 0; st < asize; st++) if (h_mask[st] && G.last_touch[0][st] < 0) untouched_all.push_back(st);
    if (!io)
        for (unsigned st : untouched_all)
            HIPCK(c, hipMemcpyAsync(J.d_basic + (size_t)st * img, J.noisy[1] + (size_t)st * img, img * sizeof(float), hipMemcpyDeviceToDevice, s));
    hipEvent_t ev_setup = get_event(c);
    HIPCK(c, hipEventRecord(ev_setup, s));   /* the caller's colour transforms and zeroed sums */
    /* ---- streamed host seam: uploads ---- */
    std::vector<char> up(io ? asize : 0, 0);
    std::vector<hipEvent_t> ev_up(io ? asize : 0, nullptr);
    std::vector<std::vector<unsigned>> outs(io ? NN : 0);   /* per node: the SAIs whose outputs are final behind it */
    std::vector<hipEvent_t> ev_out(io ? NN : 0, nullptr);
    std::vector<unsigned> out_nodes;
    if (io) {
        if (!c->io_in) HIPCK(c, hipStreamCreateWithFlags(&c->io_in, hipStreamNonBlocking));
        if (!c->io_out) HIPCK(c, hipStreamCreateWithFlags(&c->io_out, hipStreamNonBlocking));
        HIPCK(c, hipStreamWaitEvent(c->io_in, ev_setup, 0));
        for (unsigned st = 0; st < asize; st++)
            if (h_mask[st] && G.last_touch[Ls][st] >= 0) outs[(size_t)G.last_touch[Ls][st]].push_back(st);
    }
    const bool basic_in = io && !two && J.step[0] == 2;   /* run_bm5d_2nd_step alone: LF_basic is an input */
    /* one SAI of the caller's light field(s) into HBM and into the form the windows read: what run_bm5d_* does to the whole light
     * field at entry (bm5d.cpp:133, :827-830), per SAI; the copy is from pageable memory, i.e. it returns when the data has left */
    auto upload = [&](unsigned st) -> int {
        hipStream_t xs = c->io_in;
        const size_t off = (size_t)st * img;
        float* const dn = J.d_noisy + off;
        HIPCK(c, hipMemcpyAsync(dn, io->noisy[st], img * sizeof(float), hipMemcpyHostToDevice, xs));
        HIPCK(c, hipMemcpyAsync(J.pristine + off, dn, img * sizeof(float), hipMemcpyDeviceToDevice, xs));
        if (basic_in) {
            HIPCK(c, hipMemcpyAsync(J.d_basic + off, io->basic[st], img * sizeof(float), hipMemcpyHostToDevice, xs));
            HIPCK(c, hipMemcpyAsync(J.pristine_b + off, J.d_basic + off, img * sizeof(float), hipMemcpyDeviceToDevice, xs));
        }
        if (colour_io) {
            HIPCK(c, launch_color_lf(xs, dn, img, 1, J.d_mask + st, J.color_space, W * H, 1));
            if (basic_in) HIPCK(c, launch_color_lf(xs, J.d_basic + off, img, 1, J.d_mask + st, J.color_space, W * H, 1));
            if (two) HIPCK(c, launch_color_roundtrip_lf(xs, dn, const_cast<float*>(J.noisy[1]) + off, img, 1, J.d_mask + st, J.color_space, W * H));
        }
        if (two && G.last_touch[0][st] < 0)   /* no first-step window: the basic estimate is the step's input (bm5d.cpp:405, den == 0) */
            HIPCK(c, hipMemcpyAsync(J.d_basic + off, J.noisy[1] + off, img * sizeof(float), hipMemcpyDeviceToDevice, xs));
        ev_up[st] = get_event(c);
        HIPCK(c, hipEventRecord(ev_up[st], xs));
        up[st] = 1;
        return 0;
    };
    for (size_t r = 0; r < states.size(); r++) {
        RankState& S = states[r];
        S.rank = emulate ? (int)r : c->rank;
        S.x = r == 0 ? c : c->lanes[r - 1];
        for (int sl = 0; sl < 2; sl++) { S.g_num[sl] = J.g_num[sl]; S.g_den[sl] = J.g_den[sl]; }
        S.basic = J.d_basic;
        if (r > 0) {   /* an emulated rank keeps light-field sums (and a basic estimate) of its own, like a real one */
            DevBuf* nb[2] = {&S.x->g_num, &S.x->g_num2}; DevBuf* db[2] = {&S.x->g_den, &S.x->g_den2};
            for (int sl = 0; sl < J.n_steps; sl++) {
                HIPCK(c, nb[sl]->reserve(asize * img * sizeof(float)));
                HIPCK(c, db[sl]->reserve(asize * img * sizeof(float)));
                S.g_num[sl] = nb[sl]->as<float>(); S.g_den[sl] = db[sl]->as<float>();
                HIPCK(c, hipMemsetAsync(S.g_num[sl], 0, asize * img * sizeof(float), S.x->stream));
                HIPCK(c, hipMemsetAsync(S.g_den[sl], 0, asize * img * sizeof(float), S.x->stream));
            }
            if (two) {
                HIPCK(c, S.x->e_basic.reserve(asize * img * sizeof(float)));
                S.basic = S.x->e_basic.as<float>();
                HIPCK(c, hipStreamWaitEvent(S.x->stream, ev_setup, 0));
            }
        }
        if (r > 0)
            for (unsigned st : untouched_all)
                HIPCK(c, hipMemcpyAsync(S.basic + (size_t)st * img, J.noisy[1] + (size_t)st * img, img * sizeof(float), hipMemcpyDeviceToDevice, S.x->stream));
        S.lanes.resize((size_t)lanes_per_rank);
        for (int l = 0; l < lanes_per_rank; l++) {
            lfbm5d_ctx* lx = emulate ? S.x : (l == 0 ? c : c->lanes[(size_t)l - 1]);
            if (lane_buffers(lx, S.lanes[(size_t)l])) return 1;
            if (lx != c) HIPCK(c, hipStreamWaitEvent(lx->stream, ev_setup, 0));
        }
        if (nranks > 1)
            for (int ch = 0; ch < 2; ch++) {
                if (!S.x->cs[ch]) HIPCK(c, hipStreamCreateWithFlags(&S.x->cs[ch], hipStreamNonBlocking));
                HIPCK(c, hipStreamWaitEvent(S.x->cs[ch], ev_setup, 0));
                if (r > 0) {   /* an emulated rank's own buffers are prepared on its stream */
                    hipEvent_t e = get_event(c);
                    HIPCK(c, hipEventRecord(e, S.x->stream));
                    HIPCK(c, hipStreamWaitEvent(S.x->cs[ch], e, 0));
                }
            }
    }
    auto local = [&](int r) -> RankState* { return emulate ? &states[(size_t)r] : (r == c->rank ? &states[0] : nullptr); };
    if (ipc) {
        /* what peers read lives in buffers of this context (the caller's may be slices of an allocator's blocks, which have no IPC
         * handle of their own): the basic estimate of a two-step job, the outputs formed at the end */
        RankState& S0 = states[0];
        if (two) {
            HIPCK(c, c->e_basic.reserve(asize * img * sizeof(float)));
            S0.basic = c->e_basic.as<float>();
            for (unsigned st : untouched_all)
                HIPCK(c, hipMemcpyAsync(S0.basic + (size_t)st * img, J.noisy[1] + (size_t)st * img, img * sizeof(float), hipMemcpyDeviceToDevice, s));
        }
        HIPCK(c, c->ipc_out.reserve(asize * img * sizeof(float)));
        HIPCK(c, hipMemsetAsync(c->ipc_flags.as<unsigned>() + 2 * kIpcMaxMsgs, 0, sizeof(unsigned), s));
        HIPCK(c, hipStreamSynchronize(s));
        void* const mine_bufs[7] = {c->ipc_flags.p, S0.g_num[0], two ? (void*)S0.g_num[1] : nullptr, S0.g_den[0], two ? (void*)S0.g_den[1] : nullptr,
                                    two ? (void*)S0.basic : nullptr, c->ipc_out.p};
        if (ipc_exchange_handles(c, mine_bufs)) return 1;
    }
    unsigned* const ipc_own = c->ipc_flags.as<unsigned>();
    auto ipc_peer = [&](int r, int slot) -> float* { return reinterpret_cast<float*>(c->ipc_peers[(size_t)r].ptr[slot]); };
    std::vector<size_t> ipc_sent;   /* messages this rank sent: their "taken" words are waited for before the drain */
    if (c->h_counts_cap < NN * kWinCounters) {
        if (c->h_counts) (void)hipHostFree(c->h_counts);
        c->h_counts = nullptr; c->h_counts_cap = 0;
        HIPCK(c, hipHostMalloc((void**)&c->h_counts, NN * kWinCounters * sizeof(unsigned)));
        c->h_counts_cap = NN * kWinCounters;
    }
    std::vector<hipEvent_t> done(NN, nullptr);
    std::vector<hipEvent_t> arrived(G.xfers.size(), nullptr);   /* per message: it has reached its consumer's rank */
    /* message of (producer node, SAI slot) / of (SAI, reader rank) */
    std::vector<std::vector<int>> sum_xfer(NN);
    for (size_t n = 0; n < NN; n++) sum_xfer[n].assign(G.nodes[n].sai.size(), -1);
    std::vector<std::vector<int>> basic_xfer(two ? (size_t)nranks : 0);
    for (auto& v : basic_xfer) v.assign(asize, -1);
    for (size_t xi = 0; xi < G.xfers.size(); xi++) {
        const plan::Xfer& X = G.xfers[xi];
        if (X.kind == 0) {
            const plan::Node& pn = G.nodes[X.from];
            sum_xfer[X.from][(size_t)(std::find(pn.sai.begin(), pn.sai.end(), X.sai) - pn.sai.begin())] = (int)xi;
        } else basic_xfer[(size_t)X.to_rank][X.sai] = (int)xi;
    }
    std::vector<SaiMask> win_bits(NN, sai_mask_none());
    std::vector<char> mine(NN, 0);
    ncclComm_t comms[2] = {c->comm, c->comm2 ? c->comm2 : c->comm};
    size_t xi = 0, n_msgs = 0;
    for (unsigned n : G.order) {
        const plan::Node& nd = G.nodes[n];
        const int sl = nd.s, r = nd.rank;
        const Geo& g = geo[sl];
        RankState* S = local(r);
        if (S) {
            const Lane& Lw = S->lanes[(size_t)nd.lane];
            hipStream_t ls = Lw.x->stream;
            auto wait_node = [&](int p) -> int {   /* a node of this rank: same lane = stream order */
                if (G.nodes[(size_t)p].lane != nd.lane) HIPCK(c, hipStreamWaitEvent(ls, done[(size_t)p], 0));
                return 0;
            };
            if (io)   /* the SAIs this window is the first to use: into HBM now, the window waits for them on its lane */
                for (unsigned st : nd.sai)
                    if (!up[st]) {
                        if (upload(st)) return 1;
                        HIPCK(c, hipStreamWaitEvent(ls, ev_up[st], 0));
                    }
            for (size_t i = 0; i < nd.sai.size(); i++) {
                const int pw = nd.prev[i];
                if (pw >= 0) {
                    if (G.nodes[(size_t)pw].rank == r) { if (wait_node(pw)) return 1; }
                    else {
                        const plan::Node& pn = G.nodes[(size_t)pw];
                        const size_t j = (size_t)(std::find(pn.sai.begin(), pn.sai.end(), nd.sai[i]) - pn.sai.begin());
                        HIPCK(c, hipStreamWaitEvent(ls, arrived[(size_t)sum_xfer[(size_t)pw][j]], 0));
                    }
                }
                if (two && sl == 1) {   /* the SAI's basic estimate: finalised behind the first step's last window on it */
                    const int f = G.last_touch[0][nd.sai[i]];
                    if (f >= 0) {
                        if (G.nodes[(size_t)f].rank == r) { if (wait_node(f)) return 1; }
                        else HIPCK(c, hipStreamWaitEvent(ls, arrived[(size_t)basic_xfer[(size_t)r][nd.sai[i]]], 0));
                    }
                }
            }
            /* one angular window around SAI (ps, pt): bm5d.cpp:215-402 -- padding, the centre pass, its coverage count, and
             * (optimistic completion) the window's sums back into the light field */
            int cs_w, mins, maxs, ct_w, mint, maxt;
            search_window((int)nd.ps, aheight, J.an[sl], cs_w, mins, maxs);
            search_window((int)nd.pt, awidth, J.an[sl], ct_w, mint, maxt);
            const unsigned cst_w = ang_major == LFBM5D_ROWMAJOR ? (unsigned)cs_w * g.asw + (unsigned)ct_w : (unsigned)cs_w + (unsigned)ct_w * g.asw;
            std::vector<unsigned> mask_w(g.Aw, 0), proc_w(g.Aw, 0);
            SaiList wl; wl.n = g.Aw;
            for (unsigned si = 0; si < g.asw; si++)
                for (unsigned ti = 0; ti < g.asw; ti++) {
                    const unsigned Ss = si + (unsigned)mins, T = ti + (unsigned)mint;
                    const unsigned st = ang_major == LFBM5D_ROWMAJOR ? Ss * awidth + T : Ss + T * aheight;
                    const unsigned slot = ang_major == LFBM5D_ROWMAJOR ? si * g.asw + ti : si + ti * g.asw;
                    mask_w[slot] = h_mask[st];
                    wl.st[slot] = h_mask[st] ? st : 0xffffffffu;
                    if (h_mask[st]) win_bits[n].set(slot);
                    proc_w[slot] = !h_mask[st];
                }
            const bool wien = J.step[sl] == 2;
            /* (the estimate buffer as pass_impl lays it out: slack on both sides for the table kernel's row loads) */
            HIPCK(c, Lw.x->est.reserve((kEstLead + g.Aw * (size_t)g.wb * g.hb + 256) * sizeof(float)));
            HIPCK(c, launch_window_begin(ls, J.noisy[sl], wien ? S->basic : nullptr, S->g_num[sl], S->g_den[sl], img, Lw.w_noisy, Lw.w_basic, Lw.w_num,
                                         Lw.w_den, Lw.x->est.as<float>() + kEstLead, g.imgb, wl, W, H, C, g.nHW, Lw.d_small));
            lfbm5d_params Pw = *J.P[sl];
            Pw.tau_4D = nd.tau4;
            Lw.x->gslot = sl;
            Lw.x->est_ready = true;
            const int prc = pass_impl(Lw.x, J.step[sl], &Pw, g.asw, g.asw, g.wb, g.hb, C, Lw.w_noisy, wien ? Lw.w_basic : nullptr, Lw.w_num, Lw.w_den,
                                      mask_w.data(), proc_w.data(), cst_w, cst_w);
            Lw.x->gslot = 0;
            if (prc) { if (Lw.x != c) c->err = Lw.x->err; return 1; }
            /* the window's sums back into the light field, and the coverage count of the pass (LF_denoised_percent,
             * utilities_LF.cpp:967-995) -> pinned memory */
            HIPCK(c, launch_window_end(ls, S->g_num[sl], S->g_den[sl], img, Lw.w_num, Lw.w_den, g.imgb, wl, W, H, C, g.nHW, J.P[sl]->k, Lw.d_small));
            HIPCK(c, hipMemcpyAsync(c->h_counts + (size_t)n * kWinCounters, Lw.d_small, kWinCounters * sizeof(unsigned), hipMemcpyDeviceToHost, ls));
            if (!nd.fin.empty()) {   /* two-step jobs: these SAIs' first-step sums are final -> their basic estimate as the second step reads it */
                SaiList fl; fl.n = 0;
                for (unsigned st : nd.fin) fl.st[fl.n++] = st;
                const bool colour = C == 3 && J.P[0]->color_space != LFBM5D_RGB;
                HIPCK(c, launch_finalize_multi(ls, S->g_num[0], S->g_den[0], J.noisy[0], S->basic, img, fl, J.P[0]->color_space, W * H, colour ? 1 : 0));
            }
            done[n] = get_event(c);
            HIPCK(c, hipEventRecord(done[n], ls));
            mine[n] = 1;
            if (io && !outs[n].empty()) {   /* the SAIs nobody touches after this window: their outputs, in the form the caller gets them */
                for (size_t o0 = 0; o0 < outs[n].size(); o0 += (size_t)kBigA) {
                    SaiList ol; ol.n = 0;
                    for (size_t q = o0; q < outs[n].size() && ol.n < (unsigned)kBigA; q++) ol.st[ol.n++] = outs[n][q];
                    HIPCK(c, launch_output_multi(ls, S->g_num[Ls], S->g_den[Ls], J.step[Ls] == 1 ? J.noisy[Ls] : S->basic, J.d_out,
                                                 J.step[Ls] == 2 ? S->basic : nullptr, J.noisy[Ls], J.d_noisy, img, ol, J.color_space, W * H, colour_io ? 1 : 0));
                }
                ev_out[n] = get_event(c);
                HIPCK(c, hipEventRecord(ev_out[n], ls));
                out_nodes.push_back(n);
            }
            if (Lw.x != c) { c->lane_windows += 1; c->stats.lane_windows += 1; }
        }
        /* the messages this window's result feeds, in the order every rank issues them */
        for (; xi < G.xfers.size() && G.xfers[xi].from == n; xi++) {
            const plan::Xfer& X = G.xfers[xi];
            /* one channel when the second communicator could not be created: two streams on one communicator would break the
             * common issue order the exchange relies on */
            const int ra = r, rb = X.to_rank, ch = (emulate || c->comm2 || ipc) ? X.channel : 0;
            RankState* Sa = local(ra); RankState* Sb = local(rb);
            const size_t off = (size_t)X.sai * img;
            const int xsl = nd.s;
            if (emulate) {   /* both ends live here: the message is a device copy between the two ranks' buffers */
                hipStream_t xs = Sb->x->cs[ch];
                HIPCK(c, hipStreamWaitEvent(xs, done[n], 0));
                if (X.kind == 0) {
                    HIPCK(c, hipMemcpyAsync(Sb->g_num[xsl] + off, Sa->g_num[xsl] + off, img * sizeof(float), hipMemcpyDeviceToDevice, xs));
                    HIPCK(c, hipMemcpyAsync(Sb->g_den[xsl] + off, Sa->g_den[xsl] + off, img * sizeof(float), hipMemcpyDeviceToDevice, xs));
                } else
                    HIPCK(c, hipMemcpyAsync(Sb->basic + off, Sa->basic + off, img * sizeof(float), hipMemcpyDeviceToDevice, xs));
                arrived[xi] = get_event(c);
                HIPCK(c, hipEventRecord(arrived[xi], xs));
                n_msgs++;
            } else if (ipc && (Sa || Sb)) {
                /* the same message between two processes on one GPU: the sender publishes "ready" behind its window, the receiver's
                 * exchange stream waits for the word, copies the SAI out of the sender's (mapped) buffers and publishes "taken" */
                hipStream_t xs = c->cs[ch];
                const unsigned ep = c->ipc_epoch;
                if (Sa) {
                    HIPCK(c, hipStreamWaitEvent(xs, done[n], 0));
                    HIPCK(c, launch_ipc_set(xs, ipc_own + xi, ep));
                    ipc_sent.push_back(xi);
                } else {
                    const unsigned* const pf = reinterpret_cast<const unsigned*>(c->ipc_peers[(size_t)ra].ptr[0]);
                    HIPCK(c, launch_ipc_wait(xs, pf + xi, ep, ipc_own + 2 * kIpcMaxMsgs, c->ipc_timeout_s));
                    if (X.kind == 0) {
                        HIPCK(c, hipMemcpyAsync(Sb->g_num[xsl] + off, ipc_peer(ra, 1 + xsl) + off, img * sizeof(float), hipMemcpyDeviceToDevice, xs));
                        HIPCK(c, hipMemcpyAsync(Sb->g_den[xsl] + off, ipc_peer(ra, 3 + xsl) + off, img * sizeof(float), hipMemcpyDeviceToDevice, xs));
                    } else
                        HIPCK(c, hipMemcpyAsync(Sb->basic + off, ipc_peer(ra, 5) + off, img * sizeof(float), hipMemcpyDeviceToDevice, xs));
                    HIPCK(c, launch_ipc_set(xs, ipc_own + kIpcMaxMsgs + xi, ep));
                    arrived[xi] = get_event(c);
                    HIPCK(c, hipEventRecord(arrived[xi], xs));
                }
                n_msgs++;
            } else if (Sa || Sb) {
                hipStream_t xs = c->cs[ch];
                RankState* Sm = Sa ? Sa : Sb;
                if (Sa) HIPCK(c, hipStreamWaitEvent(xs, done[n], 0));
                bool ok = ncclGroupStart() == ncclSuccess;
                if (X.kind == 0) {
                    if (Sa) ok = ok && ncclSend(Sm->g_num[xsl] + off, img, ncclFloat, rb, comms[ch], xs) == ncclSuccess
                                    && ncclSend(Sm->g_den[xsl] + off, img, ncclFloat, rb, comms[ch], xs) == ncclSuccess;
                    else    ok = ok && ncclRecv(Sm->g_num[xsl] + off, img, ncclFloat, ra, comms[ch], xs) == ncclSuccess
                                    && ncclRecv(Sm->g_den[xsl] + off, img, ncclFloat, ra, comms[ch], xs) == ncclSuccess;
                } else {
                    if (Sa) ok = ok && ncclSend(Sm->basic + off, img, ncclFloat, rb, comms[ch], xs) == ncclSuccess;
                    else    ok = ok && ncclRecv(Sm->basic + off, img, ncclFloat, ra, comms[ch], xs) == ncclSuccess;
                }
                ok = ncclGroupEnd() == ncclSuccess && ok;
                if (!ok) return fail(c, "RCCL send / recv of a window's SAI failed");
                if (Sb) { arrived[xi] = get_event(c); HIPCK(c, hipEventRecord(arrived[xi], xs)); }
                n_msgs++;
            }
        }
    }
    if (io) {
        /* ---- streamed host seam: downloads.  Everything is enqueued; this thread now delivers every SAI's outputs as the window
         * that makes them final completes (pageable destinations: the copies block, which is all this thread has left to do) */
        for (unsigned st = 0; st < asize; st++)   /* SAIs no window uses (LFBM5D_MAX_WINDOWS): still part of the result */
            if (h_mask[st] && !up[st] && upload(st)) return 1;
        auto download = [&](unsigned st) -> int {
            const size_t off = (size_t)st * img;
            HIPCK(c, hipMemcpyAsync(io->noisy[st], J.d_noisy + off, img * sizeof(float), hipMemcpyDeviceToHost, c->io_out));
            if (J.step[Ls] == 2) {
                HIPCK(c, hipMemcpyAsync(io->basic[st], J.d_basic + off, img * sizeof(float), hipMemcpyDeviceToHost, c->io_out));
                HIPCK(c, hipMemcpyAsync(io->out[st], J.d_out + off, img * sizeof(float), hipMemcpyDeviceToHost, c->io_out));
            } else
                HIPCK(c, hipMemcpyAsync(io->basic[st], J.d_out + off, img * sizeof(float), hipMemcpyDeviceToHost, c->io_out));
            return 0;
        };
        for (unsigned n : out_nodes) {
            HIPCK(c, hipEventSynchronize(ev_out[n]));
            for (unsigned st : outs[n]) if (download(st)) return 1;
        }
        HIPCK(c, hipStreamSynchronize(c->io_in));
        /* SAIs without a window in the result's step keep that step's input (bm5d.cpp:405 / :1106 with den == 0): formed once every
         * window is done (a two-step job may still finalise their basic estimate late) */
        std::vector<unsigned> rest;
        for (unsigned st = 0; st < asize; st++) if (h_mask[st] && G.last_touch[Ls][st] < 0) rest.push_back(st);
        if (!rest.empty()) {
            for (RankState& S : states) for (Lane& Lq : S.lanes) HIPCK(c, hipStreamSynchronize(Lq.x->stream));
            for (size_t o0 = 0; o0 < rest.size(); o0 += (size_t)kBigA) {
                SaiList ol; ol.n = 0;
                for (size_t q = o0; q < rest.size() && ol.n < (unsigned)kBigA; q++) ol.st[ol.n++] = rest[q];
                HIPCK(c, launch_output_multi(c->io_in, J.g_num[Ls], J.g_den[Ls], J.step[Ls] == 1 ? J.noisy[Ls] : J.d_basic, J.d_out,
                                             J.step[Ls] == 2 ? J.d_basic : nullptr, J.noisy[Ls], J.d_noisy, img, ol, J.color_space, W * H, colour_io ? 1 : 0));
            }
            HIPCK(c, hipStreamSynchronize(c->io_in));
            for (unsigned st : rest) if (download(st)) return 1;
        }
        HIPCK(c, hipStreamSynchronize(c->io_out));
    }
    if (ipc) {   /* a send is complete when the peer has taken the SAI (what an RCCL send's completion means) */
        for (size_t xs_i : ipc_sent) {
            const plan::Xfer& X = G.xfers[xs_i];
            const int ch = X.channel;
            const unsigned* const pf = reinterpret_cast<const unsigned*>(c->ipc_peers[(size_t)X.to_rank].ptr[0]);
            HIPCK(c, launch_ipc_wait(c->cs[ch], pf + kIpcMaxMsgs + xs_i, c->ipc_epoch, ipc_own + 2 * kIpcMaxMsgs, c->ipc_timeout_s));
        }
    }
    /* drain: every lane, every exchange stream */
    for (RankState& S : states) {
        for (Lane& Lq : S.lanes) HIPCK(c, hipStreamSynchronize(Lq.x->stream));
        for (int ch = 0; ch < 2; ch++) if (S.x->cs[ch]) HIPCK(c, hipStreamSynchronize(S.x->cs[ch]));
    }
    HIPCK(c, hipStreamSynchronize(s));
    if (ipc) {
        unsigned err = 0;
        HIPCK(c, hipMemcpy(&err, ipc_own + 2 * kIpcMaxMsgs, sizeof(unsigned), hipMemcpyDeviceToHost));
        if (err) return fail(c, "ipc transport: a peer did not deliver / take a message within the watchdog");
    }
    drain_guard.armed = false;
    int complete = 1;
    for (size_t n = 0; n < NN; n++) {
        if (!mine[n]) continue;
        const int sl = G.nodes[n].s;
        const unsigned n_mask = win_bits[n].count();
        unsigned covered = 0;
        for (unsigned q = 0; q < kWinCounters; q++) covered += c->h_counts[n * kWinCounters + q];
        const float pct = (float)covered * 100.0f / (float)n_mask / (float)(H - J.P[sl]->k + 1) / (float)(W - J.P[sl]->k + 1);
        if (!(pct >= 100.0f)) complete = 0;
    }
    if (c->opt->force_redo && nranks == 1) complete = 0;   /* test hook: exercise the sequential redo */
    /* fold the other lanes' / emulated ranks' counters and event times into this context */
    auto fold_all = [&](lfbm5d_ctx* x) -> int {
        drain_events(x);
        for (int sl = 0; sl < J.n_steps; sl++)
            if (fold_counters(x, J.P[sl], geo[sl].Aw, C, J.step[sl], sl)) { c->err = x->err; return 1; }
        return 0;
    };
    for (lfbm5d_ctx* x : c->lanes) {
        if (x->pending.empty() && x->stats.passes == 0) continue;
        if (fold_all(x)) return 1;
        c->stats.passes += x->stats.passes; c->stats.groups += x->stats.groups;
        c->stats.stack_patches += x->stats.stack_patches; c->stats.sadct_groups += x->stats.sadct_groups;
        c->stats.algorithmic_bytes += x->stats.algorithmic_bytes;
        c->stats.ms_bm += x->stats.ms_bm; c->stats.ms_group += x->stats.ms_group; c->stats.ms_aggregate += x->stats.ms_aggregate;
        c->stats.launches_group += x->stats.launches_group; c->stats.launches_aggregate += x->stats.launches_aggregate;
        std::memset(&x->stats, 0, sizeof(x->stats));
    }
    if (two && fold_all(c)) return 1;   /* (single steps: run_step folds slot 0 of this context itself) */
    if (ipc) {
        std::vector<int> all;
        if (ipc_allgather(c, "complete", complete, all)) return 1;
        for (int v : all) complete = std::min(complete, v);
    } else
    if (nranks > 1 && !emulate) {   /* all ranks must agree before the collective below */
        HIPCK(c, c->small.reserve((asize + 8 + kWinCounters) * sizeof(unsigned)));
        int* d_flag = reinterpret_cast<int*>(c->small.as<unsigned>());
        HIPCK(c, hipMemcpyAsync(d_flag, &complete, sizeof(int), hipMemcpyHostToDevice, s));
        if (ncclAllReduce(d_flag, d_flag, 1, ncclInt, ncclMin, c->comm, s) != ncclSuccess) return fail(c, "ncclAllReduce(flag) failed");
        HIPCK(c, hipMemcpyAsync(&complete, d_flag, sizeof(int), hipMemcpyDeviceToHost, s));
        HIPCK(c, hipStreamSynchronize(s));
    }
    abort_guard.armed = false;   /* every exchange of the graph has completed; what follows are plain collectives */
    *complete_out = complete;
    if (!complete) { ipc_guard.armed = false; return 0; }   /* (agreed on by all ranks above) */
    for (size_t n = 0; n < NN; n++) if (mine[n]) c->stats.windows += 1;
    c->stats.messages += n_msgs;
    if (nranks > 1) {
        /* Every SAI's final sums live on the rank of the last window that touched it: that rank forms the SAI's estimate
         * (bm5d.cpp:405 / :1106), then the estimates are exchanged so that every rank ends with the whole result; two-step jobs
         * do the same with the basic estimates, which live where they were finalised */
        const int ls = J.n_steps - 1;
        std::vector<unsigned> own(asize);
        for (RankState& S : states) {
            const float* sub = J.step[ls] == 1 ? J.noisy[ls] : S.basic;
            for (unsigned st = 0; st < asize; st++)
                own[st] = (h_mask[st] && G.last_touch[ls][st] >= 0 && G.nodes[(size_t)G.last_touch[ls][st]].rank == S.rank) ? 1u : 0u;
            HIPCK(c, S.x->d_own.reserve(asize * sizeof(unsigned)));
            HIPCK(c, hipMemcpyAsync(S.x->d_own.p, own.data(), asize * sizeof(unsigned), hipMemcpyHostToDevice, s));
            HIPCK(c, launch_estimate_lf(s, S.g_num[ls], S.g_den[ls], sub, ipc ? c->ipc_out.as<float>() : J.d_out, img, asize, S.x->d_own.as<unsigned>()));
            HIPCK(c, hipStreamSynchronize(s));   /* own is reused */
            if (two && emulate && S.x != c)      /* the basic estimates this emulated rank finalised: what the broadcast below moves between real ranks */
                for (unsigned st = 0; st < asize; st++)
                    if (h_mask[st] && G.last_touch[0][st] >= 0 && G.nodes[(size_t)G.last_touch[0][st]].rank == S.rank)
                        HIPCK(c, hipMemcpyAsync(J.d_basic + (size_t)st * img, S.basic + (size_t)st * img, img * sizeof(float), hipMemcpyDeviceToDevice, s));
        }
        if (ipc) {   /* every rank's outputs are formed: pull each SAI from the rank that holds it, then leave together */
            std::vector<int> all;
            if (ipc_allgather(c, "formed", 1, all)) return 1;
            for (unsigned st = 0; st < asize; st++) {
                if (!h_mask[st]) continue;
                if (G.last_touch[ls][st] >= 0) {
                    const int r = G.nodes[(size_t)G.last_touch[ls][st]].rank;
                    const float* src = r == c->rank ? c->ipc_out.as<float>() : ipc_peer(r, 6);
                    HIPCK(c, hipMemcpyAsync(J.d_out + (size_t)st * img, src + (size_t)st * img, img * sizeof(float), hipMemcpyDeviceToDevice, s));
                }
                if (two && G.last_touch[0][st] >= 0) {
                    const int r = G.nodes[(size_t)G.last_touch[0][st]].rank;
                    const float* src = r == c->rank ? states[0].basic : ipc_peer(r, 5);
                    HIPCK(c, hipMemcpyAsync(J.d_basic + (size_t)st * img, src + (size_t)st * img, img * sizeof(float), hipMemcpyDeviceToDevice, s));
                } else if (two)
                    HIPCK(c, hipMemcpyAsync(J.d_basic + (size_t)st * img, states[0].basic + (size_t)st * img, img * sizeof(float), hipMemcpyDeviceToDevice, s));
            }
            HIPCK(c, hipStreamSynchronize(s));
            if (ipc_allgather(c, "pulled", 1, all)) return 1;
        } else
        if (!emulate) {
            hipEvent_t e0 = get_event(c), e1 = get_event(c);
            HIPCK(c, hipEventRecord(e0, s));
            bool ok = ncclGroupStart() == ncclSuccess;
            for (unsigned st = 0; st < asize && ok; st++) {
                if (!h_mask[st]) continue;
                if (G.last_touch[ls][st] >= 0)
                    ok = ncclBroadcast(J.d_out + (size_t)st * img, J.d_out + (size_t)st * img, img, ncclFloat, G.nodes[(size_t)G.last_touch[ls][st]].rank, c->comm, s) == ncclSuccess;
                if (ok && two && G.last_touch[0][st] >= 0)
                    ok = ncclBroadcast(J.d_basic + (size_t)st * img, J.d_basic + (size_t)st * img, img, ncclFloat, G.nodes[(size_t)G.last_touch[0][st]].rank, c->comm, s) == ncclSuccess;
            }
            ok = ncclGroupEnd() == ncclSuccess && ok;
            if (!ok) return fail(c, "ncclBroadcast of the estimates failed");
            HIPCK(c, hipEventRecord(e1, s));
            HIPCK(c, hipStreamSynchronize(s));
            float ms = 0.0f;
            if (hipEventElapsedTime(&ms, e0, e1) == hipSuccess) c->stats.ms_comm += ms;
        }
        /* SAIs no window touched (LFBM5D_MAX_WINDOWS) keep the step's input, like the single-rank estimate */
        for (unsigned st = 0; st < asize; st++) own[st] = (h_mask[st] && G.last_touch[ls][st] < 0) ? 1u : 0u;
        if (std::count(own.begin(), own.end(), 1u)) {
            HIPCK(c, hipMemcpyAsync(c->d_own.p, own.data(), asize * sizeof(unsigned), hipMemcpyHostToDevice, s));
            HIPCK(c, launch_estimate_lf(s, J.g_num[ls], J.g_den[ls], J.step[ls] == 1 ? J.noisy[ls] : J.d_basic, J.d_out, img, asize, c->d_own.as<unsigned>()));
            HIPCK(c, hipStreamSynchronize(s));
        }
    }
    ipc_guard.armed = false;
    return 0;
}

/* bm5d.cpp:165-407 (step 1) / :861-1106 (step 2) on device-resident buffers */
int run_step(lfbm5d_ctx* c, int step, const lfbm5d_params* P, float* d_noisy, const unsigned* h_mask,
             float* d_basic, float* d_out, unsigned ang_major, unsigned awidth, unsigned aheight,
             unsigned an, unsigned W, unsigned H, unsigned C, const HostIO* io = nullptr) {
    const unsigned asize = awidth * aheight;
    const unsigned cs = aheight / 2, ct = awidth / 2;
    const unsigned cst = ang_major == LFBM5D_ROWMAJOR ? cs * awidth + ct : cs + ct * aheight;
    const unsigned asw = 2 * an + 1;
    if (asw > aheight || asw > awidth) {
        std::printf("Wrong size of angular search window, the angular search window must be smaller than the light field angular size.\n");
        return fail(c, "angular search window larger than the light field");
    }
    if (ang_major != LFBM5D_ROWMAJOR && ang_major != LFBM5D_COLMAJOR) return fail(c, "bad ang_major");
    if (validate(c, step, P, asw, asw, C)) return 1;
    hipStream_t s = c->stream;
    const unsigned nHW = P->nSim + P->nDisp;
    const size_t img = (size_t)C * W * H;
    const unsigned hb = H + 2 * nHW, wb = W + 2 * nHW;
    const size_t imgb = (size_t)C * wb * hb;
    const unsigned Aw = asw * asw;
    unsigned tau_4D = P->tau_4D;

    if (C == 3 && P->color_space != LFBM5D_RGB) {
        if (P->color_space > LFBM5D_RGB) return fail(c, "bad color space");
    }
    HIPCK(c, c->d_mask.reserve(asize * sizeof(unsigned)));
    unsigned* d_mask = c->d_mask.as<unsigned>();
    HIPCK(c, hipMemcpyAsync(d_mask, h_mask, asize * sizeof(unsigned), hipMemcpyHostToDevice, s));
    /* transformation of the whole light field(s) at entry (bm5d.cpp:133, :827-830) -- unless the window graph streams the caller's
     * SAIs in and transforms them one by one (decided below) */
    auto forward_colour = [&]() -> int {
        if (C == 3 && P->color_space != LFBM5D_RGB) {
            HIPCK(c, launch_color_lf(s, d_noisy, img, asize, d_mask, P->color_space, W * H, 1));
            if (step == 2) HIPCK(c, launch_color_lf(s, d_basic, img, asize, d_mask, P->color_space, W * H, 1));
        }
        return 0;
    };
    HIPCK(c, c->g_num.reserve(asize * img * sizeof(float)));
    HIPCK(c, c->g_den.reserve(asize * img * sizeof(float)));
    HIPCK(c, hipMemsetAsync(c->g_num.p, 0, asize * img * sizeof(float), s));
    HIPCK(c, hipMemsetAsync(c->g_den.p, 0, asize * img * sizeof(float), s));
    /* sized for run_graph's use too (its lanes add kWinCounters words): a later, larger reserve would free the block the
     * pointers below -- and lane0's -- refer to */
    HIPCK(c, c->small.reserve((asize + 8 + kWinCounters) * sizeof(unsigned)));
    float* g_num = c->g_num.as<float>(); float* g_den = c->g_den.as<float>();
    unsigned* d_small = c->small.as<unsigned>();
    std::vector<unsigned> h_cnt(asize + 8, (unsigned)img), h_tmp(asize + 8), h_one(8);   /* den starts all zero */
    std::vector<unsigned> dirty;

    std::vector<unsigned> proc(asize);
    for (unsigned st = 0; st < asize; st++) proc[st] = !h_mask[st];
    c->last_windows.clear();

    /* One angular window around SAI (ps, pt): bm5d.cpp:215-402, in two halves so that several windows can be in flight
     * on lanes of their own (a lane = a context with its stream, window buffers and per-pass work buffers; lane 0 is this
     * context).  win_begin enqueues the padding, the centre pass and its coverage count; win_finish waits for the count,
     * runs whatever further passes the window needs (greyscale light fields) and adds the window back to the light field. */
    struct Lane { lfbm5d_ctx* x; float* w_noisy; float* w_basic; float* w_num; float* w_den; unsigned* d_small; float* g_num; float* g_den; };
    struct WinState {
        unsigned ps = 0, pt = 0; int cs_w = 0, mins = 0, ct_w = 0, mint = 0; unsigned cst_w = 0, rem_w = 0, tot_w = 0, pst_w = 0; SaiMask win_bits = sai_mask_none();
        std::vector<unsigned> st_idx, mask_w, proc_w; SaiList sl; lfbm5d_params Pw; bool counted = false;
        unsigned* h_count_dst = nullptr;   /* pinned word the coverage count is copied to (default: the lane's) */
        float tile_pct = 0.0f;             /* tile mode: sum of the tiles' LF_denoised_percent of the last pass */
    };
    auto lane_buffers = [&](lfbm5d_ctx* x, Lane& L) -> int {
        HIPCK(c, x->w_noisy.reserve(Aw * imgb * sizeof(float)));
        if (step == 2) HIPCK(c, x->w_basic.reserve(Aw * imgb * sizeof(float)));
        HIPCK(c, x->w_num.reserve(Aw * imgb * sizeof(float)));
        HIPCK(c, x->w_den.reserve(Aw * imgb * sizeof(float)));
        HIPCK(c, x->small.reserve((asize + 8 + kWinCounters) * sizeof(unsigned)));
        L.x = x; L.w_noisy = x->w_noisy.as<float>(); L.w_basic = x->w_basic.as<float>();
        L.w_num = x->w_num.as<float>(); L.w_den = x->w_den.as<float>(); L.d_small = x->small.as<unsigned>();
        L.g_num = g_num; L.g_den = g_den;   /* the light field's sums this lane's windows read and update */
        return 0;
    };
    auto lane_fail = [&](const Lane& L) { if (L.x != c) c->err = L.x->err; return 1; };
    /* coverage count of the pass just enqueued -> the lane's pinned word (LF_denoised_percent, utilities_LF.cpp:967-995) */
    auto enqueue_count = [&](const Lane& L, WinState& ws) -> int {
        hipStream_t ls = L.x->stream;
        HIPCK(c, hipMemsetAsync(L.d_small, 0, sizeof(unsigned), ls));
        HIPCK(c, launch_count_denoised(ls, L.w_den, imgb, Aw, ws.win_bits, W, H, C, nHW, P->k, L.d_small));
        HIPCK(c, hipMemcpyAsync(ws.h_count_dst ? ws.h_count_dst : L.x->h_small, L.d_small, sizeof(unsigned), hipMemcpyDeviceToHost, ls));
        ws.counted = true;
        return 0;
    };
    /* The reference's OpenMP tile mode (bm5d.cpp:411-708), opt-in through lfbm5d_set_tiles: every SAI of the padded window
     * is cut into tiles with a halo of nHW pixels (sub_divide, utilities.cpp:312-395: halved along its longer side until
     * there are `tiles` pieces, the last row / column takes the remainder), each tile runs the core pass on its own, only
     * the tiles' interiors are kept (undivide_LF, utilities_LF.cpp:438-515 -- what a tile aggregated into its halo is
     * discarded, about 0.5 dB) and the window's num / den are padded again.  A compatibility mode: tile after tile. */
    const int n_tiles = c->tiles;
    unsigned tl_w = W, tl_h = H, tl_nw = 1, tl_nh = 1;
    for (int n = n_tiles; n > 1; n /= 2) {
        if (tl_w > tl_h) { tl_w = (unsigned)std::floor((float)tl_w * 0.5f); tl_nw *= 2; }
        else { tl_h = (unsigned)std::floor((float)tl_h * 0.5f); tl_nh *= 2; }
    }
    const unsigned tl_hb = tl_nh > 1 ? H - (tl_nh - 1) * tl_h : tl_h, tl_wb = tl_nw > 1 ? W - (tl_nw - 1) * tl_w : tl_w;
    auto tiled_pass = [&](const Lane& L, WinState& ws) -> int {
        lfbm5d_ctx* x = L.x;
        hipStream_t ls = x->stream;
        const unsigned hmax = std::max(tl_h, tl_hb) + 2 * nHW, wmax = std::max(tl_w, tl_wb) + 2 * nHW;
        const size_t tmax = (size_t)C * hmax * wmax;
        HIPCK(c, x->t_noisy.reserve(Aw * tmax * sizeof(float)));
        if (step == 2) HIPCK(c, x->t_basic.reserve(Aw * tmax * sizeof(float)));
        HIPCK(c, x->t_tnum.reserve(Aw * tmax * sizeof(float)));
        HIPCK(c, x->t_tden.reserve(Aw * tmax * sizeof(float)));
        HIPCK(c, x->und_num.reserve(Aw * img * sizeof(float)));
        HIPCK(c, x->und_den.reserve(Aw * img * sizeof(float)));
        float* tn = x->t_noisy.as<float>(); float* tb = x->t_basic.as<float>();
        float* tu = x->t_tnum.as<float>(); float* td = x->t_tden.as<float>();
        const unsigned n_mask = ws.win_bits.count();
        ws.tile_pct = 0.0f;
        const unsigned long long passes0 = x->stats.passes;   /* a window pass counts once, not once per tile */
        for (unsigned kt = 0; kt < tl_nw * tl_nh; kt++) {
            const unsigned i = kt / tl_nw, j = kt % tl_nw;
            const unsigned h = (i == tl_nh - 1 ? tl_hb : tl_h) + 2 * nHW, w = (j == tl_nw - 1 ? tl_wb : tl_w) + 2 * nHW;
            const size_t timg = (size_t)C * w * h;
            if (h < 2 * nHW + P->k + 1 || w < 2 * nHW + P->k + 1) return fail(c, "tile smaller than the search range");
            auto cut = [&](const float* src, float* dst) {
                return launch_copy_rect(ls, dst, timg, w, h, 0, 0, src, imgb, wb, hb, j * tl_w, i * tl_h, w, h, C, Aw, ws.win_bits);
            };
            HIPCK(c, cut(L.w_noisy, tn));
            if (step == 2) HIPCK(c, cut(L.w_basic, tb));
            HIPCK(c, cut(L.w_num, tu));
            HIPCK(c, cut(L.w_den, td));
            if (pass_impl(x, step, &ws.Pw, asw, asw, w, h, C, tn, step == 2 ? tb : nullptr, tu, td,
                          ws.mask_w.data(), ws.proc_w.data(), ws.cst_w, ws.pst_w)) return lane_fail(L);
            HIPCK(c, hipMemsetAsync(L.d_small, 0, sizeof(unsigned), ls));
            HIPCK(c, launch_count_denoised(ls, td, timg, Aw, ws.win_bits, w - 2 * nHW, h - 2 * nHW, C, nHW, P->k, L.d_small));
            HIPCK(c, hipMemcpyAsync(x->h_small, L.d_small, sizeof(unsigned), hipMemcpyDeviceToHost, ls));
            HIPCK(c, launch_copy_rect(ls, x->und_num.as<float>(), img, W, H, j * tl_w, i * tl_h, tu, timg, w, h, nHW, nHW,
                                      w - 2 * nHW, h - 2 * nHW, C, Aw, ws.win_bits));
            HIPCK(c, launch_copy_rect(ls, x->und_den.as<float>(), img, W, H, j * tl_w, i * tl_h, td, timg, w, h, nHW, nHW,
                                      w - 2 * nHW, h - 2 * nHW, C, Aw, ws.win_bits));
            HIPCK(c, hipStreamSynchronize(ls));
            ws.tile_pct += (float)x->h_small[0] * 100.0f / (float)n_mask / (float)(h - 2 * nHW - P->k + 1) / (float)(w - 2 * nHW - P->k + 1);
        }
        x->stats.passes = passes0 + 1;
        SaiList slots; slots.n = Aw;
        for (unsigned a = 0; a < Aw; a++) slots.st[a] = ws.mask_w[a] ? a : 0xffffffffu;
        HIPCK(c, launch_symetrize_multi(ls, x->und_num.as<float>(), img, L.w_num, imgb, slots, W, H, C, nHW));
        HIPCK(c, launch_symetrize_multi(ls, x->und_den.as<float>(), img, L.w_den, imgb, slots, W, H, C, nHW));
        return 0;
    };
    auto one_pass = [&](const Lane& L, WinState& ws) -> int {
        if (n_tiles > 1) {
            if (tiled_pass(L, ws)) return 1;
        } else
        if (pass_impl(L.x, step, &ws.Pw, asw, asw, wb, hb, C, L.w_noisy, step == 2 ? L.w_basic : nullptr, L.w_num, L.w_den,
                      ws.mask_w.data(), ws.proc_w.data(), ws.cst_w, ws.pst_w)) return lane_fail(L);
        ws.proc_w[ws.pst_w] += 1;
        const unsigned ps_w = ang_major == LFBM5D_ROWMAJOR ? ws.pst_w / asw : ws.pst_w % asw;
        const unsigned pt_w = ang_major == LFBM5D_ROWMAJOR ? ws.pst_w % asw : ws.pst_w / asw;
        const unsigned st = ang_major == LFBM5D_ROWMAJOR ? (ws.mins + ps_w) * awidth + (ws.mint + pt_w)
                                                         : (ws.mins + ps_w) + (ws.mint + pt_w) * aheight;
        proc[st] += 1;
        if (n_tiles > 1) { ws.counted = true; return 0; }
        return enqueue_count(L, ws);
    };
    auto win_begin = [&](const Lane& L, unsigned ps, unsigned pt, unsigned tau4, WinState& ws) -> int {
        hipStream_t ls = L.x->stream;
        ws.ps = ps; ws.pt = pt; ws.counted = false;
        int maxs, maxt;
        search_window((int)ps, aheight, an, ws.cs_w, ws.mins, maxs);
        search_window((int)pt, awidth, an, ws.ct_w, ws.mint, maxt);
        ws.cst_w = ang_major == LFBM5D_ROWMAJOR ? ws.cs_w * asw + ws.ct_w : ws.cs_w + ws.ct_w * asw;
        ws.st_idx.assign(Aw, 0); ws.mask_w.assign(Aw, 0); ws.proc_w.assign(Aw, 0);
        for (unsigned si = 0; si < asw; si++)
            for (unsigned ti = 0; ti < asw; ti++) {
                const unsigned S = si + ws.mins, T = ti + ws.mint;
                if (ang_major == LFBM5D_ROWMAJOR) ws.st_idx[si * asw + ti] = S * awidth + T;
                else ws.st_idx[si + ti * asw] = S + T * aheight;
            }
        ws.sl.n = Aw;
        ws.win_bits = sai_mask_none();
        for (unsigned i = 0; i < Aw; i++) {
            ws.mask_w[i] = h_mask[ws.st_idx[i]];
            ws.sl.st[i] = ws.mask_w[i] ? ws.st_idx[i] : 0xffffffffu;
            if (ws.mask_w[i]) ws.win_bits.set(i);
        }
        HIPCK(c, launch_symetrize_multi(ls, d_noisy, img, L.w_noisy, imgb, ws.sl, W, H, C, nHW));
        if (step == 2) HIPCK(c, launch_symetrize_multi(ls, d_basic, img, L.w_basic, imgb, ws.sl, W, H, C, nHW));
        HIPCK(c, launch_symetrize_multi(ls, L.g_num, img, L.w_num, imgb, ws.sl, W, H, C, nHW));
        HIPCK(c, launch_symetrize_multi(ls, L.g_den, img, L.w_den, imgb, ws.sl, W, H, C, nHW));
        for (unsigned i = 0; i < Aw; i++) ws.proc_w[i] = !ws.mask_w[i];
        ws.rem_w = (unsigned)std::count(ws.proc_w.begin(), ws.proc_w.end(), 0u);
        ws.tot_w = ws.rem_w;
        ws.Pw = *P;
        ws.Pw.tau_4D = tau4;
        if (ws.rem_w && ws.mask_w[ws.cst_w]) {   /* the centre pass needs no device data to be chosen: enqueue it now */
            ws.pst_w = ws.cst_w;
            if (one_pass(L, ws)) return 1;
        }
        return 0;
    };
    auto win_finish = [&](const Lane& L, WinState& ws) -> int {
        hipStream_t ls = L.x->stream;
        std::vector<unsigned> h_tmp_w(Aw);
        while (ws.rem_w) {
            if (!ws.counted) {   /* choose the next SAI of the window from the zero-weight counts (bm5d.cpp:299-327) and process it */
                HIPCK(c, hipMemsetAsync(L.d_small, 0, Aw * sizeof(unsigned), ls));
                if (n_tiles > 1) {
                    /* tile mode: the reference counts the zeros tile by tile over the tiles sub_divide cuts from the merged
                     * window, halos included (bm5d.cpp:598-600) -- a zero under two halos counts twice */
                    lfbm5d_ctx* x = L.x;
                    const unsigned hmax = std::max(tl_h, tl_hb) + 2 * nHW, wmax = std::max(tl_w, tl_wb) + 2 * nHW;
                    HIPCK(c, x->t_tden.reserve(Aw * (size_t)C * hmax * wmax * sizeof(float)));
                    for (unsigned kt = 0; kt < tl_nw * tl_nh; kt++) {
                        const unsigned i = kt / tl_nw, j = kt % tl_nw;
                        const unsigned h = (i == tl_nh - 1 ? tl_hb : tl_h) + 2 * nHW, w = (j == tl_nw - 1 ? tl_wb : tl_w) + 2 * nHW;
                        const size_t timg = (size_t)C * w * h;
                        HIPCK(c, launch_copy_rect(ls, x->t_tden.as<float>(), timg, w, h, 0, 0, L.w_den, imgb, wb, hb, j * tl_w, i * tl_h, w, h, C, Aw, ws.win_bits));
                        HIPCK(c, launch_count_zeros(ls, x->t_tden.as<float>(), timg, Aw, L.d_small));
                    }
                } else
                HIPCK(c, launch_count_zeros(ls, L.w_den, imgb, Aw, L.d_small));
                HIPCK(c, hipMemcpyAsync(h_tmp_w.data(), L.d_small, Aw * sizeof(unsigned), hipMemcpyDeviceToHost, ls));
                HIPCK(c, hipStreamSynchronize(ls));
                long best_cnt = -1;
                for (unsigned i = 0; i < Aw; i++) {
                    if (ws.proc_w[i]) continue;
                    if ((long)h_tmp_w[i] >= best_cnt) { ws.pst_w = i; best_cnt = (long)h_tmp_w[i]; }
                }
                if (one_pass(L, ws)) return 1;
            }
            HIPCK(c, hipStreamSynchronize(ls));
            ws.counted = false;
            /* LF_denoised_percent (utilities_LF.cpp:967-995): counts (i,j,c) triples, divides without C */
            const unsigned n_mask = ws.win_bits.count();
            const float pct = (float)L.x->h_small[0] * 100.0f / (float)n_mask / (float)(H - P->k + 1) / (float)(W - P->k + 1);
            if (n_tiles > 1 ? ws.tile_pct >= 100.0f * (float)(tl_nw * tl_nh) /* bm5d.cpp:668-672 */ : pct >= 100.0f)
                for (unsigned i = 0; i < Aw; i++)
                    if (ws.proc_w[i] == 0) { ws.proc_w[i] += 1; proc[ws.st_idx[i]] += 1; }
            ws.rem_w = (unsigned)std::count(ws.proc_w.begin(), ws.proc_w.end(), 0u);
        }
        HIPCK(c, launch_unsymetrize_multi(ls, L.g_num, img, L.w_num, imgb, ws.sl, W, H, C, nHW));
        HIPCK(c, launch_unsymetrize_multi(ls, L.g_den, img, L.w_den, imgb, ws.sl, W, H, C, nHW));
        for (unsigned i = 0; i < Aw; i++) if (ws.mask_w[i]) dirty.push_back(ws.st_idx[i]);
        c->stats.windows += 1;
        return 0;
    };
    Lane lane0;
    if (lane_buffers(c, lane0)) return 1;
    /* sequential form: one window after the other on this context's stream */
    auto do_window = [&](unsigned ps, unsigned pt) -> int {
        /* the reference switches tau_4D from DCT to SADCT for good once a window holds an empty SAI (bm5d.cpp:276-280) */
        unsigned n_in = 0;
        {
            int cs_w, mins, maxs, ct_w, mint, maxt;
            search_window((int)ps, aheight, an, cs_w, mins, maxs);
            search_window((int)pt, awidth, an, ct_w, mint, maxt);
            for (unsigned si = 0; si < asw; si++)
                for (unsigned ti = 0; ti < asw; ti++)
                    n_in += h_mask[ang_major == LFBM5D_ROWMAJOR ? (si + mins) * awidth + (ti + mint) : (si + mins) + (ti + mint) * aheight] ? 1u : 0u;
        }
        if (n_in != Aw && tau_4D == LFBM5D_DCT) tau_4D = LFBM5D_SADCT;
        WinState ws;
        if (win_begin(lane0, ps, pt, tau_4D, ws)) return 1;
        if (win_finish(lane0, ws)) return 1;
        c->last_windows.push_back(ang_major == LFBM5D_ROWMAJOR ? ps * awidth + pt : ps + pt * aheight);
        return 0;
    };

    /* Window schedule.  The reference picks the unprocessed SAI with the most exact-zero weights, last
     * index winning ties (bm5d.cpp:187-213).  A window always ends with all of its SAIs processed
     * (bm5d.cpp:283-402), so an unprocessed SAI has never been aggregated into: all candidates tie and
     * the sequence of windows is a pure function of the mask -- plan_windows() (tests check it against the
     * data-driven selection, which stays available).  Several GPUs (and the lanes of one GPU) run the planned
     * sequence as a dependency graph: windows interact only through num / den of the SAIs they share, chains of
     * windows go to ranks, and what a window needs from another rank's window travels as one send / recv per SAI --
     * bit-identical to one GPU for any rank count (lfbm5d_plan.h).
     * LFBM5D_STEP_SHARDING selects the alternatives: "rows" (every core pass sharded by reference-patch rows, exact,
     * barely scales) and "blocks" (round 1: one contiguous block of windows per rank + one all-reduce per step; a rank's
     * block matching then only sees its own earlier windows: -0.01 / -0.03 / -0.07 dB at 2 / 4 / 8 ranks). */
    const int emu = c->opt->emulate_world;                           /* test hook: play all ranks on this GPU */
    /* "rows": keep the reference's window-after-window order on several GPUs too and shard every core pass by
     * reference-patch rows (bit-for-bit the single-GPU schedule, two all-reduces per pass, little speed-up) */
    const bool by_rows = c->world > 1 && c->opt->step_sharding == 1;
    /* "blocks": the round-1 scheme -- the planned sequence cut into one contiguous block of windows per rank, ONE
     * all-reduce of num / den per step.  It scales with the rank count but is NOT the reference's result: a rank's block
     * matching only sees its own earlier windows' estimates (-0.01 / -0.03 / -0.07 dB at 2 / 4 / 8 ranks).  Opt-in. */
    const bool by_blocks = (c->world > 1 || emu > 1) && c->opt->step_sharding == 2;
    /* LFBM5D_MAX_WINDOWS: stop after that many windows of the planned sequence (for
     * bisecting a multi-window difference, bounded timing samples); the estimate is still formed */
    const int max_windows = c->opt->max_windows;
    const int n_lanes = std::max(1, std::min(8, c->opt->lanes));
    /* LFBM5D_DATA_DRIVEN_SCHEDULE: select every window from the zero-weight counts like the reference does (one
     * device round trip per window); the default takes the same sequence from plan_windows() */
    const bool planned = (c->world > 1 && !by_rows) || emu > 1 || !c->opt->data_driven_schedule;   /* several ranks always plan */
    struct PassShard {   /* restores the unsharded default whatever way the function returns */
        lfbm5d_ctx* c;
        PassShard(lfbm5d_ctx* cc, bool on) : c(cc) { if (on) { c->pass_rank = c->rank; c->pass_world = c->world; c->pass_reduce = c->comm != nullptr; } }
        ~PassShard() { c->pass_rank = 0; c->pass_world = 1; c->pass_reduce = false; }
    } pass_shard(c, by_rows);
    if (by_rows && !c->comm) return fail(c, "whole steps on several ranks need lfbm5d_comm_init");

    /* ---- Graph form (colour light fields; the default on one GPU and on several): the planned windows as a dependency
     * graph (lfbm5d_plan.h) executed by run_graph above.  If a window would have needed another pass, a single-GPU step is
     * redone in the sequential form (never observed) and a multi-GPU step fails with a message; greyscale light fields,
     * where further passes are the rule, take the sequential / row-sharded forms directly. */
    const int nranks = emu > 1 ? emu : c->world;
    c->lane_windows = 0;
    plan::Graph G;
    if (c->tiles > 1 && nranks > 1) return fail(c, "the tile mode runs on one GPU");
    bool graph_mode = planned && !by_rows && !by_blocks && C == 3 && (n_lanes > 1 || nranks > 1) && c->tiles <= 1;
    if (graph_mode) {
        const plan::StepDesc sd = {an, tau_4D, 1u};
        plan::build(h_mask, awidth, aheight, ang_major, &sd, 1, nranks, emu > 1 ? 1 : n_lanes, max_windows, G);
        if (!G.centre_ok) graph_mode = false;   /* empty window centre: the first pass is chosen from device data */
    }
    if (!graph_mode && nranks > 1 && !by_rows && !by_blocks)
        return fail(c, "whole steps on several ranks: this light field needs data-driven passes (greyscale, or an empty SAI at a "
                       "window centre); set LFBM5D_STEP_SHARDING=rows");
    if (graph_mode && c->world > 1 && emu <= 1 && !c->comm && !c->ipc) return fail(c, "whole steps on several ranks need lfbm5d_comm_init");
    /* host seam: the single-rank graph takes the caller's SAIs in and out as its windows need and finish them; every other form
     * gets the whole light field(s) first */
    const bool streamable = io && graph_mode && nranks == 1 && !c->opt->host_blocking;
    bool streamed_out = false;
    if (streamable) {
        HIPCK(c, c->pristine.reserve(asize * img * sizeof(float)));
        if (step == 2) HIPCK(c, c->pristine_b.reserve(asize * img * sizeof(float)));
    } else {
        if (io && io_upload_all(c, io, h_mask, asize, img, d_noisy, step == 2 ? d_basic : nullptr)) return 1;
        if (forward_colour()) return 1;
    }
    bool graph_done = false;
    if (graph_mode) {
        GraphJob J;
        J.n_steps = 1; J.step[0] = step; J.P[0] = P; J.an[0] = an; J.noisy[0] = d_noisy; J.d_basic = d_basic;
        J.g_num[0] = g_num; J.g_den[0] = g_den; J.d_out = d_out; J.d_mask = d_mask;
        J.io = streamable ? io : nullptr; J.d_noisy = d_noisy; J.pristine = c->pristine.as<float>(); J.pristine_b = c->pristine_b.as<float>();
        J.color_space = P->color_space;
        int complete = 1;
        if (run_graph(c, J, G, h_mask, awidth, aheight, ang_major, W, H, C, nranks, emu > 1, &complete)) return 1;
        if (complete) {
            for (const plan::Node& nd : G.nodes) c->last_windows.push_back(nd.pst);
            graph_done = true;
            streamed_out = streamable;
        } else if (nranks > 1) {
            return fail(c, "a window needed more than its centre pass: set LFBM5D_STEP_SHARDING=rows for this light field");
        } else {
            /* some window needed more than its centre pass: redo the step window after window */
            if (streamable) {   /* ... from the light field(s) as they arrived: the streamed form has transformed them back SAI by SAI */
                HIPCK(c, hipMemcpyAsync(d_noisy, c->pristine.p, asize * img * sizeof(float), hipMemcpyDeviceToDevice, s));
                if (step == 2) HIPCK(c, hipMemcpyAsync(d_basic, c->pristine_b.p, asize * img * sizeof(float), hipMemcpyDeviceToDevice, s));
                if (forward_colour()) return 1;
            }
            HIPCK(c, hipMemsetAsync(g_num, 0, asize * img * sizeof(float), s));
            HIPCK(c, hipMemsetAsync(g_den, 0, asize * img * sizeof(float), s));
            for (unsigned st = 0; st < asize; st++) proc[st] = !h_mask[st];
        }
    }
    const bool pipelined = graph_done;
    if (pipelined) { /* done above */ } else
    if (!planned) {
        unsigned remaining = (unsigned)std::count(proc.begin(), proc.end(), 0u);
        const unsigned total = remaining;
        unsigned ps = 0, pt = 0, pst = 0;
        while (remaining) {
            if (remaining == total && h_mask[cst]) { ps = cs; pt = ct; }
            else { /* counts only change for the SAIs of the window just processed: recount those */
                if (!dirty.empty()) {
                    HIPCK(c, hipMemsetAsync(d_small, 0, asize * sizeof(unsigned), s));
                    for (unsigned st : dirty) HIPCK(c, launch_count_zeros(s, g_den + st * img, img, 1, d_small + st));
                    HIPCK(c, hipMemcpyAsync(h_tmp.data(), d_small, asize * sizeof(unsigned), hipMemcpyDeviceToHost, s));
                    HIPCK(c, hipStreamSynchronize(s));
                    for (unsigned st : dirty) h_cnt[st] = h_tmp[st];
                    dirty.clear();
                }
                long best_cnt = -1;
                for (unsigned st = 0; st < asize; st++) {
                    if (proc[st]) continue;
                    if ((long)h_cnt[st] >= best_cnt) { pst = st; best_cnt = (long)h_cnt[st]; }
                }
                if (ang_major == LFBM5D_ROWMAJOR) { ps = pst / awidth; pt = pst - ps * awidth; }
                else { pt = pst / aheight; ps = pst - pt * aheight; }
            }
            if (do_window(ps, pt)) return 1;
            remaining = (unsigned)std::count(proc.begin(), proc.end(), 0u);
        }
    } else {
        std::vector<unsigned> plan;
        plan_windows(h_mask, awidth, aheight, an, ang_major, plan);
        if (max_windows > 0 && plan.size() > (size_t)max_windows) plan.resize((size_t)max_windows);
        /* one rank (sequential planned form), or the opt-in window blocks */
        const bool emu_b = by_blocks && emu > 1;
        const int nb = by_blocks ? (emu > 1 ? emu : c->world) : 1, rb = by_blocks && !emu_b ? c->rank : 0;
        float* t_num = nullptr; float* t_den = nullptr;
        if (emu_b) {
            HIPCK(c, c->t_num.reserve(asize * img * sizeof(float)));
            HIPCK(c, c->t_den.reserve(asize * img * sizeof(float)));
            t_num = c->t_num.as<float>(); t_den = c->t_den.as<float>();
            HIPCK(c, hipMemsetAsync(t_num, 0, asize * img * sizeof(float), s));
            HIPCK(c, hipMemsetAsync(t_den, 0, asize * img * sizeof(float), s));
        }
        for (int r = (emu_b ? 0 : rb); r < (emu_b ? emu : rb + 1); r++) {
            /* contiguous blocks of the sequence: consecutive windows overlap, so most of a window's already
             * processed SAIs (whose running estimate the matching uses) were processed by the same rank */
            const size_t w_begin = plan.size() * (size_t)r / (size_t)nb, w_end = plan.size() * (size_t)(r + 1) / (size_t)nb;
            for (size_t wi = w_begin; wi < w_end; wi++) {
                const unsigned pst = plan[wi];
                const unsigned ps = ang_major == LFBM5D_ROWMAJOR ? pst / awidth : pst % aheight;
                const unsigned pt = ang_major == LFBM5D_ROWMAJOR ? pst % awidth : pst / aheight;
                if (do_window(ps, pt)) return 1;
            }
            if (emu_b) { /* what the all-reduce does, rank by rank */
                HIPCK(c, launch_add(s, t_num, g_num, asize * img));
                HIPCK(c, launch_add(s, t_den, g_den, asize * img));
                HIPCK(c, hipMemsetAsync(g_num, 0, asize * img * sizeof(float), s));
                HIPCK(c, hipMemsetAsync(g_den, 0, asize * img * sizeof(float), s));
            }
        }
        if (emu_b) {
            HIPCK(c, hipMemcpyAsync(g_num, t_num, asize * img * sizeof(float), hipMemcpyDeviceToDevice, s));
            HIPCK(c, hipMemcpyAsync(g_den, t_den, asize * img * sizeof(float), hipMemcpyDeviceToDevice, s));
        } else if (by_blocks && c->comm) {
            hipEvent_t e0, e1;
            HIPCK(c, hipEventCreate(&e0)); HIPCK(c, hipEventCreate(&e1));
            HIPCK(c, hipEventRecord(e0, s));
            if (ncclAllReduce(g_num, g_num, asize * img, ncclFloat, ncclSum, c->comm, s) != ncclSuccess) return fail(c, "ncclAllReduce(num) failed");
            if (ncclAllReduce(g_den, g_den, asize * img, ncclFloat, ncclSum, c->comm, s) != ncclSuccess) return fail(c, "ncclAllReduce(den) failed");
            HIPCK(c, hipEventRecord(e1, s));
            HIPCK(c, hipStreamSynchronize(s));
            float ms = 0.0f;
            if (hipEventElapsedTime(&ms, e0, e1) == hipSuccess) c->stats.ms_comm += ms;
            (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
        } else if (by_blocks && c->world > 1) {
            return fail(c, "whole steps on several ranks need lfbm5d_comm_init (lfbm5d_set_shard only shards core passes)");
        }
    }
    /* final estimate (bm5d.cpp:405) and inverse colour transforms (bm5d.cpp:711-714 / :1414-1418) */
    const float* sub = step == 1 ? d_noisy : d_basic;
    if (!streamed_out) {   /* (the streamed host seam has formed, transformed and delivered every SAI's outputs already) */
        if (!(graph_done && nranks > 1))   /* (the multi-rank graph form has formed and exchanged the estimates already) */
            HIPCK(c, launch_estimate_lf(s, g_num, g_den, sub, d_out, img, asize, d_mask));
        if (C == 3 && P->color_space != LFBM5D_RGB) {
            HIPCK(c, launch_color_lf(s, d_out, img, asize, d_mask, P->color_space, W * H, 0));
            if (step == 2) HIPCK(c, launch_color_lf(s, d_basic, img, asize, d_mask, P->color_space, W * H, 0));
            HIPCK(c, launch_color_lf(s, d_noisy, img, asize, d_mask, P->color_space, W * H, 0));
        }
    }
    HIPCK(c, hipStreamSynchronize(s));
    if (io && !streamed_out && io_download_all(c, io, h_mask, asize, img, d_noisy, step == 2 ? d_basic : d_out, step == 2 ? d_out : nullptr)) return 1;
    drain_events(c);
    return fold_counters(c, P, Aw, C, step);
}

/* run_bm5d_1st_step followed by run_bm5d_2nd_step (main.cpp:195, :242) as ONE job: the windows of both steps form one
 * dependency graph (lfbm5d_plan.h) -- a window of the second step starts when the basic estimate of each of its SAIs is final,
 * not when the whole first step is -- and what the reference does between the two calls (estimate, inverse colour transform,
 * forward colour transform: bm5d.cpp:405, :711-714, :827-830) happens SAI by SAI.  Bit-identical to the two calls.  Light fields
 * the graph form does not cover (greyscale, an empty SAI at a window centre, tile mode, the data-driven schedule, the
 * alternative multi-GPU schemes) take the two calls. */
int run_denoise(lfbm5d_ctx* c, const lfbm5d_params* P1, const lfbm5d_params* P2, float* d_noisy, const unsigned* h_mask, float* d_basic,
                float* d_out, unsigned ang_major, unsigned awidth, unsigned aheight, unsigned an1, unsigned an2, unsigned W, unsigned H,
                unsigned C, const HostIO* io = nullptr) {
    const unsigned asize = awidth * aheight;
    const int emu = c->opt->emulate_world;
    const int n_lanes = std::max(1, std::min(8, c->opt->lanes));
    const int max_windows = c->opt->max_windows;
    const int nranks = emu > 1 ? emu : c->world;
    bool fused = C == 3 && c->tiles <= 1 && !c->opt->step_sharding && !c->opt->data_driven_schedule &&
                 c->opt->fused != 0 && P1->color_space == P2->color_space &&
                 2 * an1 + 1 <= std::min(awidth, aheight) && 2 * an2 + 1 <= std::min(awidth, aheight);
    plan::Graph G;
    if (fused) {
        if (ang_major != LFBM5D_ROWMAJOR && ang_major != LFBM5D_COLMAJOR) return fail(c, "bad ang_major");
        if (validate(c, 1, P1, 2 * an1 + 1, 2 * an1 + 1, C) || validate(c, 2, P2, 2 * an2 + 1, 2 * an2 + 1, C)) return 1;
        if (P1->color_space > LFBM5D_RGB) return fail(c, "bad color space");
        /* relative cost of a window pass of either step (scheduling model only; measured on the README configuration) */
        const plan::StepDesc sd[2] = {{an1, P1->tau_4D, 10u}, {an2, P2->tau_4D, 9u}};
        plan::build(h_mask, awidth, aheight, ang_major, sd, 2, nranks, emu > 1 ? 1 : n_lanes, max_windows, G);
        if (!G.centre_ok || G.nodes.empty()) fused = false;
    }
    const size_t img = (size_t)C * W * H;
    /* the two calls one after the other, on light fields that are in HBM as a whole */
    auto two_calls = [&]() -> int {
        if (run_step(c, 1, P1, d_noisy, h_mask, nullptr, d_basic, ang_major, awidth, aheight, an1, W, H, C)) return 1;
        if (run_step(c, 2, P2, d_noisy, h_mask, d_basic, d_out, ang_major, awidth, aheight, an2, W, H, C)) return 1;
        return io ? io_download_all(c, io, h_mask, asize, img, d_noisy, d_basic, d_out) : 0;
    };
    if (!fused) {
        if (io && io_upload_all(c, io, h_mask, asize, img, d_noisy, nullptr)) return 1;
        return two_calls();
    }
    if (c->world > 1 && emu <= 1 && !c->comm && !c->ipc) return fail(c, "whole steps on several ranks need lfbm5d_comm_init");
    hipStream_t s = c->stream;
    const bool colour = P1->color_space != LFBM5D_RGB;
    HIPCK(c, c->d_mask.reserve(asize * sizeof(unsigned)));
    unsigned* d_mask = c->d_mask.as<unsigned>();
    HIPCK(c, hipMemcpyAsync(d_mask, h_mask, asize * sizeof(unsigned), hipMemcpyHostToDevice, s));
    GraphJob J;
    J.n_steps = 2; J.step[0] = 1; J.step[1] = 2; J.P[0] = P1; J.P[1] = P2; J.an[0] = an1; J.an[1] = an2;
    J.d_basic = d_basic; J.d_out = d_out; J.d_mask = d_mask;
    /* host seam: the single-rank graph takes the caller's SAIs in and out as its windows need and finish them */
    const bool streamable = io && nranks == 1 && !c->opt->host_blocking;
    if (io && !streamable && io_upload_all(c, io, h_mask, asize, img, d_noisy, nullptr)) return 1;
    /* the light field as it arrived: what the fallback below starts from (on one rank) */
    if (nranks == 1) {
        HIPCK(c, c->pristine.reserve(asize * img * sizeof(float)));
        if (!streamable) HIPCK(c, hipMemcpyAsync(c->pristine.p, d_noisy, asize * img * sizeof(float), hipMemcpyDeviceToDevice, s));
    }
    J.io = streamable ? io : nullptr; J.d_noisy = d_noisy; J.pristine = c->pristine.as<float>(); J.color_space = P1->color_space;
    /* what the first step reads: forward(noisy) (bm5d.cpp:133); what the second step reads: forward(inverse(that))
     * (bm5d.cpp:713, :827): both live for the whole job, the second in a buffer of its own */
    J.noisy[0] = d_noisy; J.noisy[1] = d_noisy;
    if (colour) {
        HIPCK(c, c->n2.reserve(asize * img * sizeof(float)));
        if (!streamable) {   /* (the streamed form does this SAI by SAI behind every upload) */
            HIPCK(c, launch_color_lf(s, d_noisy, img, asize, d_mask, P1->color_space, W * H, 1));
            HIPCK(c, hipMemcpyAsync(c->n2.p, d_noisy, asize * img * sizeof(float), hipMemcpyDeviceToDevice, s));   /* (empty SAIs too) */
            HIPCK(c, launch_color_roundtrip_lf(s, d_noisy, c->n2.as<float>(), img, asize, d_mask, P1->color_space, W * H));
        }
        J.noisy[1] = c->n2.as<float>();
    }
    DevBuf* nb[2] = {&c->g_num, &c->g_num2}; DevBuf* db[2] = {&c->g_den, &c->g_den2};
    for (int sl = 0; sl < 2; sl++) {
        HIPCK(c, nb[sl]->reserve(asize * img * sizeof(float)));
        HIPCK(c, db[sl]->reserve(asize * img * sizeof(float)));
        HIPCK(c, hipMemsetAsync(nb[sl]->p, 0, asize * img * sizeof(float), s));
        HIPCK(c, hipMemsetAsync(db[sl]->p, 0, asize * img * sizeof(float), s));
        J.g_num[sl] = nb[sl]->as<float>(); J.g_den[sl] = db[sl]->as<float>();
    }
    c->lane_windows = 0;
    c->last_windows.clear();
    int complete = 1;
    if (run_graph(c, J, G, h_mask, awidth, aheight, ang_major, W, H, C, nranks, emu > 1, &complete)) return 1;
    if (!complete) {
        /* some window needed more than its centre pass (the graph form assumes one): on one rank the job is redone as the two
         * calls, window after window, from the light field as it arrived */
        if (nranks > 1) return fail(c, "a window needed more than its centre pass: run the two steps one after the other (LFBM5D_FUSED=0)");
        HIPCK(c, hipMemcpyAsync(d_noisy, c->pristine.p, asize * img * sizeof(float), hipMemcpyDeviceToDevice, s));
        return two_calls();
    }
    for (const plan::Node& nd : G.nodes) c->last_windows.push_back(nd.pst);
    if (streamable) { HIPCK(c, hipStreamSynchronize(s)); return 0; }   /* (every SAI's outputs have been formed and delivered) */
    /* final estimate (bm5d.cpp:1106) and the closing inverse colour transforms of both steps' outputs (bm5d.cpp:1414-1418) */
    if (nranks == 1) HIPCK(c, launch_estimate_lf(s, J.g_num[1], J.g_den[1], d_basic, d_out, img, asize, d_mask));
    if (colour) {
        HIPCK(c, launch_color_lf(s, d_out, img, asize, d_mask, P1->color_space, W * H, 0));
        HIPCK(c, launch_color_lf(s, d_basic, img, asize, d_mask, P1->color_space, W * H, 0));
        HIPCK(c, hipMemcpyAsync(d_noisy, c->n2.p, asize * img * sizeof(float), hipMemcpyDeviceToDevice, s));
        HIPCK(c, launch_color_lf(s, d_noisy, img, asize, d_mask, P1->color_space, W * H, 0));
    }
    HIPCK(c, hipStreamSynchronize(s));
    return io ? io_download_all(c, io, h_mask, asize, img, d_noisy, d_basic, d_out) : 0;
}

} /* namespace */

/* ============================================================================================ */
/* C API                                                                                        */
/* ============================================================================================ */
extern "C" {

int lfbm5d_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int lfbm5d_create(lfbm5d_ctx** out, int device) {
    if (!out) return 1;
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        g_create_error = "no HIP device available (liblfbm5d_hip has no CPU fallback)";
        return 1;
    }
    if (device < 0 || device >= n) { g_create_error = "device index out of range"; return 1; }
    if ((e = hipSetDevice(device)) != hipSuccess) { g_create_error = hipGetErrorString(e); return 1; }
    lfbm5d_ctx* c = new_ctx(device, g_create_error);
    if (c) options_from_env(c->opt_store);   /* the library's only read of the environment (lfbm5d_options.h) */
    if (!c) return 1;
    *out = c;
    return 0;
}

int lfbm5d_set_option(lfbm5d_ctx* c, const char* key, const char* value) {
    if (!c || !key) return 1;
    if (!option_set(c->opt_store, key, value)) return fail(c, std::string("unknown option: ") + key);
    return 0;
}
int lfbm5d_get_option(lfbm5d_ctx* c, const char* key, char* value, unsigned long long size) {
    if (!c || !key || !value || size == 0) return 1;
    std::string v;
    if (!option_get(c->opt_store, key, v)) return fail(c, std::string("unknown option: ") + key);
    std::snprintf(value, (size_t)size, "%s", v.c_str());
    return 0;
}

void lfbm5d_destroy(lfbm5d_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    for (lfbm5d_ctx* x : c->lanes) lfbm5d_destroy(x);
    c->lanes.clear();
    if (c->h_counts) (void)hipHostFree(c->h_counts);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (lfbm5d_ctx::IpcPeer& P : c->ipc_peers) for (void* q : P.ptr) if (q) (void)hipIpcCloseMemHandle(q);
    c->ipc_flags.release(); c->ipc_out.release(); c->pristine.release(); c->pristine_b.release();
    if (c->io_in) (void)hipStreamDestroy(c->io_in);
    if (c->io_out) (void)hipStreamDestroy(c->io_out);
    if (c->comm2) ncclCommDestroy(c->comm2);
    if (c->comm) ncclCommDestroy(c->comm);
    for (int i = 0; i < 2; i++) if (c->cs[i]) (void)hipStreamDestroy(c->cs[i]);
    for (GeomCache& g : c->gc) { g.refs.release(); g.rslot.release(); g.tb.release(); g.scan_wgs.release(); }
    DevBuf* bufs[] = {&c->est, &c->g_num2, &c->g_den2, &c->n2, &c->e_basic, &c->refmap, &c->scores, &c->tables, &c->self_idx, &c->self_cnt, &c->best,
                      &c->t_noisy, &c->t_basic, &c->t_tnum, &c->t_tden, &c->und_num, &c->und_den, &c->sub_flags, &c->sub_cnt, &c->shape, &c->filt, &c->wgt, &c->aggpos, &c->gpos, &c->gofs, &c->gok, &c->sa_list, &c->gshape, &c->counters, &c->small, &c->t_num, &c->t_den, &c->d_mask, &c->g_num, &c->g_den, &c->w_noisy,
                      &c->w_basic, &c->w_num, &c->w_den, &c->h2d_noisy, &c->h2d_basic, &c->h2d_out, &c->d_own, &c->gscratch, &c->scan_lcol};
    for (DevBuf* b : bufs) b->release();
    for (hipEvent_t e : c->ev_pool) (void)hipEventDestroy(e);
    if (c->h_small) (void)hipHostFree(c->h_small);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

const char* lfbm5d_last_error(const lfbm5d_ctx* c) { return c ? c->err.c_str() : g_create_error.c_str(); }
void lfbm5d_reset_stats(lfbm5d_ctx* c) { if (c) std::memset(&c->stats, 0, sizeof(c->stats)); }
void lfbm5d_get_stats(const lfbm5d_ctx* c, lfbm5d_stats* out) { if (c && out) *out = c->stats; }
void* lfbm5d_stream(lfbm5d_ctx* c) { return c ? (void*)c->stream : nullptr; }

void lfbm5d_shard_rows(unsigned n_rows, int rank, int world, unsigned* begin, unsigned* end) {
    if (world < 1) world = 1;
    if (rank < 0) rank = 0;
    *begin = (unsigned)(((unsigned long long)n_rows * (unsigned)rank) / (unsigned)world);
    *end = (unsigned)(((unsigned long long)n_rows * (unsigned)(rank + 1)) / (unsigned)world);
}

int lfbm5d_comm_unique_id(void* id_out) {
    ncclUniqueId id;
    static_assert(sizeof(ncclUniqueId) <= LFBM5D_UNIQUE_ID_BYTES, "unique id size");
    if (ncclGetUniqueId(&id) != ncclSuccess) return 1;
    std::memset(id_out, 0, LFBM5D_UNIQUE_ID_BYTES);
    std::memcpy(id_out, &id, sizeof(id));
    return 0;
}

int lfbm5d_comm_init(lfbm5d_ctx* c, const void* idb, int rank, int world) {
    if (!c || world < 1 || rank < 0 || rank >= world) return 1;
    (void)hipSetDevice(c->device);
    c->rank = rank; c->world = world;
    if (world == 1) return 0;
    ncclUniqueId id;
    std::memcpy(&id, idb, sizeof(id));
    if (ncclCommInitRank(&c->comm, world, id, rank) != ncclSuccess) return fail(c, "ncclCommInitRank failed");
    /* second channel of the window-graph exchange: same ranks, independent progress.  Optional (one channel is only slower) */
    if (ncclCommSplit(c->comm, 0, rank, &c->comm2, nullptr) != ncclSuccess) c->comm2 = nullptr;
    return 0;
}

int lfbm5d_comm_init_ipc(lfbm5d_ctx* c, int rank, int world, const char* rendezvous_dir, double timeout_s) {
    if (!c || world < 1 || rank < 0 || rank >= world || !rendezvous_dir) return 1;
    (void)hipSetDevice(c->device);
    c->rank = rank; c->world = world;
    if (world == 1) return 0;
    c->ipc = true; c->ipc_dir = rendezvous_dir; c->ipc_timeout_s = timeout_s > 0 ? timeout_s : 30.0; c->ipc_epoch = 0;
    HIPCK(c, c->ipc_flags.reserve((2 * kIpcMaxMsgs + 16) * sizeof(unsigned)));
    HIPCK(c, hipMemset(c->ipc_flags.p, 0, (2 * kIpcMaxMsgs + 16) * sizeof(unsigned)));
    c->ipc_peers.assign((size_t)world, lfbm5d_ctx::IpcPeer());
    for (lfbm5d_ctx::IpcPeer& P : c->ipc_peers) std::memset(&P, 0, sizeof(P));
    std::vector<int> all;   /* every rank is there (and has zeroed its gating words) before anybody's first job */
    return ipc_allgather(c, "init", 1, all);
}

int lfbm5d_plan_windows(unsigned awidth, unsigned aheight, unsigned an, unsigned ang_major, const unsigned* mask,
                        unsigned* out_sai, unsigned cap) {
    if (!mask || !awidth || !aheight || 2 * an + 1 > awidth || 2 * an + 1 > aheight) return -1;
    if (ang_major != LFBM5D_ROWMAJOR && ang_major != LFBM5D_COLMAJOR) return -1;
    std::vector<unsigned> plan;
    plan_windows(mask, awidth, aheight, an, ang_major, plan);
    for (size_t i = 0; i < plan.size() && i < cap && out_sai; i++) out_sai[i] = plan[i];
    return (int)plan.size();
}

int lfbm5d_plan_graph(unsigned awidth, unsigned aheight, unsigned an, unsigned ang_major, const unsigned* mask, int world, int lanes,
                      unsigned* out_rank, unsigned* out_lane, unsigned* out_start, unsigned cap) {
    if (!mask || !awidth || !aheight || 2 * an + 1 > awidth || 2 * an + 1 > aheight || world < 1 || lanes < 1) return -1;
    if (ang_major != LFBM5D_ROWMAJOR && ang_major != LFBM5D_COLMAJOR) return -1;
    plan::Graph G;
    const plan::StepDesc sd = {an, LFBM5D_SADCT, 1u};
    plan::build(mask, awidth, aheight, ang_major, &sd, 1, world, lanes, 0, G);
    for (size_t i = 0; i < G.nodes.size() && i < cap; i++) {
        if (out_rank) out_rank[i] = (unsigned)G.nodes[i].rank;
        if (out_lane) out_lane[i] = (unsigned)G.nodes[i].lane;
        if (out_start) out_start[i] = G.nodes[i].start;
    }
    return (int)G.nodes.size();
}

int lfbm5d_plan_messages(unsigned awidth, unsigned aheight, unsigned an, unsigned ang_major, const unsigned* mask, int world,
                         unsigned* out, unsigned cap) {
    if (!mask || !awidth || !aheight || 2 * an + 1 > awidth || 2 * an + 1 > aheight || world < 1) return -1;
    if (ang_major != LFBM5D_ROWMAJOR && ang_major != LFBM5D_COLMAJOR) return -1;
    plan::Graph G;
    const plan::StepDesc sd = {an, LFBM5D_SADCT, 1u};
    plan::build(mask, awidth, aheight, ang_major, &sd, 1, world, 1, 0, G);
    for (size_t i = 0; i < G.xfers.size() && i < cap && out; i++) {
        out[4 * i] = G.xfers[i].from; out[4 * i + 1] = (unsigned)G.xfers[i].to_node; out[4 * i + 2] = G.xfers[i].sai; out[4 * i + 3] = (unsigned)G.xfers[i].channel;
    }
    return (int)G.xfers.size();
}

int lfbm5d_plan_job(unsigned awidth, unsigned aheight, unsigned ang_major, const unsigned* mask, int n_steps, const unsigned* an,
                    const unsigned* cost, int world, int lanes, unsigned* out_nodes, unsigned node_cap, unsigned* out_msgs, unsigned msg_cap,
                    unsigned* out_counts) {
    if (!mask || !awidth || !aheight || !an || world < 1 || lanes < 1 || n_steps < 1 || n_steps > 2) return -1;
    if (ang_major != LFBM5D_ROWMAJOR && ang_major != LFBM5D_COLMAJOR) return -1;
    plan::StepDesc sd[2];
    for (int i = 0; i < n_steps; i++) {
        if (2 * an[i] + 1 > awidth || 2 * an[i] + 1 > aheight) return -1;
        sd[i] = {an[i], LFBM5D_SADCT, cost ? cost[i] : (n_steps == 2 ? (i == 0 ? 10u : 9u) : 1u)};
    }
    plan::Graph G;
    plan::build(mask, awidth, aheight, ang_major, sd, n_steps, world, lanes, 0, G);
    std::vector<unsigned> pos(G.nodes.size(), 0);
    for (size_t i = 0; i < G.order.size(); i++) pos[G.order[i]] = (unsigned)i;
    for (size_t i = 0; i < G.nodes.size() && i < node_cap && out_nodes; i++) {
        const plan::Node& nd = G.nodes[i];
        unsigned* o = out_nodes + 8 * i;
        o[0] = (unsigned)nd.s; o[1] = nd.w; o[2] = nd.pst; o[3] = (unsigned)nd.rank; o[4] = (unsigned)nd.lane; o[5] = nd.start; o[6] = pos[i]; o[7] = nd.chain;
    }
    for (size_t i = 0; i < G.xfers.size() && i < msg_cap && out_msgs; i++) {
        const plan::Xfer& X = G.xfers[i];
        unsigned* o = out_msgs + 6 * i;
        o[0] = (unsigned)X.kind; o[1] = X.from; o[2] = X.to_node < 0 ? 0xffffffffu : (unsigned)X.to_node; o[3] = (unsigned)X.to_rank; o[4] = X.sai; o[5] = (unsigned)X.channel;
    }
    if (out_counts) { out_counts[0] = (unsigned)G.nodes.size(); out_counts[1] = (unsigned)G.xfers.size(); out_counts[2] = G.makespan; out_counts[3] = G.centre_ok ? 1u : 0u; }
    return (int)G.nodes.size();
}

int lfbm5d_last_windows(const lfbm5d_ctx* c, unsigned* out_sai, unsigned cap) {
    if (!c) return -1;
    for (size_t i = 0; i < c->last_windows.size() && i < cap && out_sai; i++) out_sai[i] = c->last_windows[i];
    return (int)c->last_windows.size();
}

int lfbm5d_comm_ranks(const lfbm5d_ctx* c) {
    if (!c || !c->comm) return 0;
    int n = 0;
    return ncclCommCount(c->comm, &n) == ncclSuccess ? n : -1;
}

int lfbm5d_comm_selftest(lfbm5d_ctx* c, unsigned n) {
    if (!c || !n) return 1;
    (void)hipSetDevice(c->device);
    ncclComm_t comm = c->comm;
    bool own = false;
    if (!comm) {   /* no communicator yet: a one-rank one exercises the same RCCL code path */
        ncclUniqueId id;
        if (ncclGetUniqueId(&id) != ncclSuccess) return fail(c, "ncclGetUniqueId failed");
        if (ncclCommInitRank(&comm, 1, id, 0) != ncclSuccess) return fail(c, "ncclCommInitRank failed");
        own = true;
    }
    const int world = own ? 1 : c->world;
    float* d = nullptr;
    std::vector<float> h(n);
    for (unsigned i = 0; i < n; i++) h[i] = (float)(i % 251) + 0.5f;
    int rc = 0;
    if (hipMalloc(&d, n * sizeof(float)) != hipSuccess) rc = fail(c, "hipMalloc failed");
    if (!rc && hipMemcpyAsync(d, h.data(), n * sizeof(float), hipMemcpyHostToDevice, c->stream) != hipSuccess) rc = fail(c, "copy failed");
    if (!rc && ncclAllReduce(d, d, n, ncclFloat, ncclSum, comm, c->stream) != ncclSuccess) rc = fail(c, "ncclAllReduce failed");
    std::vector<float> r(n);
    if (!rc && hipMemcpyAsync(r.data(), d, n * sizeof(float), hipMemcpyDeviceToHost, c->stream) != hipSuccess) rc = fail(c, "copy failed");
    if (!rc && hipStreamSynchronize(c->stream) != hipSuccess) rc = fail(c, "stream failed");
    for (unsigned i = 0; i < n && !rc; i++)
        if (r[i] != h[i] * (float)world) rc = fail(c, "all-reduce returned a wrong sum");
    /* the graph form's primitives: a split communicator, grouped send / recv (to this rank itself: the only peer a
     * one-GPU box has) and a grouped broadcast, on the exchange path's kind of stream */
    if (!rc && n >= 2) {
        ncclComm_t comm2 = nullptr;
        float* d2 = nullptr;
        const unsigned half = n / 2;
        int me = 0;
        if (ncclCommUserRank(comm, &me) != ncclSuccess) rc = fail(c, "ncclCommUserRank failed");
        if (!rc && ncclCommSplit(comm, 0, me, &comm2, nullptr) != ncclSuccess) rc = fail(c, "ncclCommSplit failed");
        if (!rc && hipMalloc(&d2, half * sizeof(float)) != hipSuccess) rc = fail(c, "hipMalloc failed");
        if (!rc && hipMemsetAsync(d2, 0, half * sizeof(float), c->stream) != hipSuccess) rc = fail(c, "memset failed");
        if (!rc) {
            bool ok = ncclGroupStart() == ncclSuccess;
            ok = ok && ncclSend(d, half, ncclFloat, me, comm2, c->stream) == ncclSuccess;
            ok = ok && ncclRecv(d2, half, ncclFloat, me, comm2, c->stream) == ncclSuccess;
            ok = ncclGroupEnd() == ncclSuccess && ok;
            ok = ok && ncclGroupStart() == ncclSuccess;
            ok = ok && ncclBroadcast(d + half, d + half, n - half, ncclFloat, 0, comm, c->stream) == ncclSuccess;
            ok = ncclGroupEnd() == ncclSuccess && ok;
            if (!ok) rc = fail(c, "RCCL send / recv / broadcast failed");
        }
        std::vector<float> r2(half);
        if (!rc && hipMemcpyAsync(r2.data(), d2, half * sizeof(float), hipMemcpyDeviceToHost, c->stream) != hipSuccess) rc = fail(c, "copy failed");
        if (!rc && hipStreamSynchronize(c->stream) != hipSuccess) rc = fail(c, "stream failed");
        for (unsigned i = 0; i < half && !rc; i++)
            if (r2[i] != r[i]) rc = fail(c, "send / recv returned wrong data");
        if (d2) (void)hipFree(d2);
        if (comm2) ncclCommDestroy(comm2);
    }
    /* several ranks: the exchange pattern of the window graph between real peers -- both channels at once, each on its
     * exchange stream, channel 0 passing a block to the next rank while channel 1 passes one to the previous rank */
    if (!rc && !own && world > 1 && c->comm2 && n >= 4) {
        int me = c->rank;
        const int nxt = (me + 1) % world, prv = (me + world - 1) % world;
        const unsigned q = n / 4;
        float* e = nullptr;
        hipStream_t xs[2] = {nullptr, nullptr};
        if (hipMalloc(&e, (size_t)4 * q * sizeof(float)) != hipSuccess) rc = fail(c, "hipMalloc failed");
        for (int ch = 0; ch < 2 && !rc; ch++)
            if (hipStreamCreateWithFlags(&xs[ch], hipStreamNonBlocking) != hipSuccess) rc = fail(c, "hipStreamCreate failed");
        std::vector<float> hs(2 * q);
        for (unsigned i = 0; i < 2 * q; i++) hs[i] = (float)(me * 1000) + (float)(i % 97);     /* [0,q): for next, [q,2q): for prev */
        if (!rc && hipMemcpyAsync(e, hs.data(), 2 * q * sizeof(float), hipMemcpyHostToDevice, c->stream) != hipSuccess) rc = fail(c, "copy failed");
        if (!rc && hipStreamSynchronize(c->stream) != hipSuccess) rc = fail(c, "stream failed");
        if (!rc) {
            bool ok = ncclGroupStart() == ncclSuccess;
            ok = ok && ncclSend(e, q, ncclFloat, nxt, c->comm, xs[0]) == ncclSuccess;
            ok = ok && ncclRecv(e + 2 * q, q, ncclFloat, prv, c->comm, xs[0]) == ncclSuccess;
            ok = ncclGroupEnd() == ncclSuccess && ok;
            ok = ok && ncclGroupStart() == ncclSuccess;
            ok = ok && ncclSend(e + q, q, ncclFloat, prv, c->comm2, xs[1]) == ncclSuccess;
            ok = ok && ncclRecv(e + 3 * q, q, ncclFloat, nxt, c->comm2, xs[1]) == ncclSuccess;
            ok = ncclGroupEnd() == ncclSuccess && ok;
            if (!ok) rc = fail(c, "RCCL two-channel exchange failed");
        }
        for (int ch = 0; ch < 2 && !rc; ch++)
            if (hipStreamSynchronize(xs[ch]) != hipSuccess) rc = fail(c, "exchange stream failed");
        std::vector<float> hr(2 * q);
        if (!rc && hipMemcpy(hr.data(), e + 2 * q, 2 * q * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) rc = fail(c, "copy failed");
        for (unsigned i = 0; i < q && !rc; i++)
            if (hr[i] != (float)(prv * 1000) + (float)(i % 97) || hr[q + i] != (float)(nxt * 1000) + (float)((q + i) % 97))
                rc = fail(c, "two-channel exchange returned wrong data");
        for (int ch = 0; ch < 2; ch++) if (xs[ch]) (void)hipStreamDestroy(xs[ch]);
        if (e) (void)hipFree(e);
    }
    if (d) (void)hipFree(d);
    if (own) ncclCommDestroy(comm);
    return rc;
}

int lfbm5d_set_tiles(lfbm5d_ctx* c, int nb_tiles) {
    if (!c || nb_tiles < 0) return 1;
    int n = 1;
    while (n * 2 <= nb_tiles) n *= 2;     /* main.cpp:101-102 floors nb_threads to a power of two */
    c->tiles = nb_tiles <= 1 ? 1 : n;
    return 0;
}

int lfbm5d_set_shard(lfbm5d_ctx* c, int rank, int world) {
    if (!c || world < 1 || rank < 0 || rank >= world) return 1;
    c->rank = rank; c->world = world;
    return 0;
}

int lfbm5d_pass_device(lfbm5d_ctx* c, int step, const lfbm5d_params* P, unsigned aw, unsigned ah,
                       unsigned Wb, unsigned Hb, unsigned C, const float* d_noisy, const float* d_basic,
                       float* d_num, float* d_den, const unsigned* h_mask, const unsigned* h_procSAI,
                       unsigned cst, unsigned pst) {
    if (!c || !P) return 1;
    (void)hipSetDevice(c->device);
    c->pass_rank = c->rank; c->pass_world = c->world; c->pass_reduce = c->comm != nullptr;
    const int rc = pass_impl(c, step, P, aw, ah, Wb, Hb, C, d_noisy, d_basic, d_num, d_den, h_mask, h_procSAI, cst, pst);
    c->pass_rank = 0; c->pass_world = 1; c->pass_reduce = false;
    if (rc) return 1;
    HIPCK(c, hipStreamSynchronize(c->stream));
    drain_events(c);
    return fold_counters(c, P, aw * ah, C, step);
}

int lfbm5d_step1_device(lfbm5d_ctx* c, const lfbm5d_params* P, float* d_noisy, const unsigned* h_mask,
                        float* d_basic, unsigned ang_major, unsigned awidth, unsigned aheight, unsigned an,
                        unsigned W, unsigned H, unsigned C) {
    if (!c || !P) return 1;
    (void)hipSetDevice(c->device);
    return run_step(c, 1, P, d_noisy, h_mask, nullptr, d_basic, ang_major, awidth, aheight, an, W, H, C);
}

int lfbm5d_step2_device(lfbm5d_ctx* c, const lfbm5d_params* P, float* d_noisy, const unsigned* h_mask,
                        float* d_basic, float* d_denoised, unsigned ang_major, unsigned awidth,
                        unsigned aheight, unsigned an, unsigned W, unsigned H, unsigned C) {
    if (!c || !P) return 1;
    (void)hipSetDevice(c->device);
    return run_step(c, 2, P, d_noisy, h_mask, d_basic, d_denoised, ang_major, awidth, aheight, an, W, H, C);
}

int lfbm5d_denoise_device(lfbm5d_ctx* c, const lfbm5d_params* P1, const lfbm5d_params* P2, float* d_noisy, const unsigned* h_mask,
                          float* d_basic, float* d_denoised, unsigned ang_major, unsigned awidth, unsigned aheight, unsigned an1,
                          unsigned an2, unsigned W, unsigned H, unsigned C) {
    if (!c || !P1 || !P2 || !h_mask) return 1;
    (void)hipSetDevice(c->device);
    return run_denoise(c, P1, P2, d_noisy, h_mask, d_basic, d_denoised, ang_major, awidth, aheight, an1, an2, W, H, C);
}

/* ---- host seam.  The *_host_sai entry points take the caller's light fields as ONE HOST POINTER PER SAI (what a
 * vector<vector<float>> is); the flat forms below are the same with pointers into one buffer.  On one rank with the window graph
 * the SAIs travel as the graph needs and finishes them; otherwise the whole light fields go up first and come down last. ---- */
namespace {
int host_job(lfbm5d_ctx* c, int kind /* 1 | 2: the step, 3: both */, const lfbm5d_params* P1, const lfbm5d_params* P2, float* const* h_noisy,
             const unsigned* h_mask, float* const* h_basic, float* const* h_out, unsigned ang_major, unsigned awidth, unsigned aheight,
             unsigned an1, unsigned an2, unsigned W, unsigned H, unsigned C) {
    (void)hipSetDevice(c->device);
    const unsigned asize = awidth * aheight;
    for (unsigned st = 0; st < asize; st++)
        if (h_mask[st] && (!h_noisy[st] || !h_basic[st] || (kind != 1 && !h_out[st]))) return fail(c, "host seam: NULL pointer for a non-empty SAI");
    const size_t bytes = (size_t)asize * C * W * H * sizeof(float);
    HIPCK(c, c->h2d_noisy.reserve(bytes));
    HIPCK(c, c->h2d_basic.reserve(bytes));
    if (kind != 1) HIPCK(c, c->h2d_out.reserve(bytes));
    HostIO io; io.noisy = h_noisy; io.basic = h_basic; io.out = h_out;
    float* const dn = c->h2d_noisy.as<float>(); float* const db = c->h2d_basic.as<float>(); float* const dd = c->h2d_out.as<float>();
    if (kind == 1) return run_step(c, 1, P1, dn, h_mask, nullptr, db, ang_major, awidth, aheight, an1, W, H, C, &io);
    if (kind == 2) return run_step(c, 2, P2, dn, h_mask, db, dd, ang_major, awidth, aheight, an2, W, H, C, &io);
    return run_denoise(c, P1, P2, dn, h_mask, db, dd, ang_major, awidth, aheight, an1, an2, W, H, C, &io);
}
/* pointers into flat [asize][C*H*W] buffers; the outputs of empty SAIs read as zeros (what the flat form has always returned) */
void flat_ptrs(std::vector<float*>& v, float* base, const unsigned* h_mask, unsigned asize, size_t img, bool zero_empty) {
    v.resize(asize);
    for (unsigned st = 0; st < asize; st++) {
        v[st] = base ? base + (size_t)st * img : nullptr;
        if (base && zero_empty && !h_mask[st]) std::memset(v[st], 0, img * sizeof(float));
    }
}
} /* namespace */

int lfbm5d_denoise_host_sai(lfbm5d_ctx* c, const lfbm5d_params* P1, const lfbm5d_params* P2, float* const* h_noisy, const unsigned* h_mask,
                            float* const* h_basic, float* const* h_denoised, unsigned ang_major, unsigned awidth, unsigned aheight,
                            unsigned an1, unsigned an2, unsigned W, unsigned H, unsigned C) {
    if (!c || !P1 || !P2 || !h_mask || !h_noisy || !h_basic || !h_denoised) return 1;
    return host_job(c, 3, P1, P2, h_noisy, h_mask, h_basic, h_denoised, ang_major, awidth, aheight, an1, an2, W, H, C);
}
int lfbm5d_step1_host_sai(lfbm5d_ctx* c, const lfbm5d_params* P, float* const* h_noisy, const unsigned* h_mask, float* const* h_basic,
                          unsigned ang_major, unsigned awidth, unsigned aheight, unsigned an, unsigned W, unsigned H, unsigned C) {
    if (!c || !P || !h_mask || !h_noisy || !h_basic) return 1;
    return host_job(c, 1, P, nullptr, h_noisy, h_mask, h_basic, nullptr, ang_major, awidth, aheight, an, an, W, H, C);
}
int lfbm5d_step2_host_sai(lfbm5d_ctx* c, const lfbm5d_params* P, float* const* h_noisy, const unsigned* h_mask, float* const* h_basic,
                          float* const* h_denoised, unsigned ang_major, unsigned awidth, unsigned aheight, unsigned an, unsigned W, unsigned H,
                          unsigned C) {
    if (!c || !P || !h_mask || !h_noisy || !h_basic || !h_denoised) return 1;
    return host_job(c, 2, nullptr, P, h_noisy, h_mask, h_basic, h_denoised, ang_major, awidth, aheight, an, an, W, H, C);
}

int lfbm5d_denoise_host(lfbm5d_ctx* c, const lfbm5d_params* P1, const lfbm5d_params* P2, float* h_noisy, const unsigned* h_mask,
                        float* h_basic, float* h_denoised, unsigned ang_major, unsigned awidth, unsigned aheight, unsigned an1,
                        unsigned an2, unsigned W, unsigned H, unsigned C) {
    if (!c || !P1 || !P2 || !h_mask || !h_noisy || !h_basic || !h_denoised) return 1;
    const size_t img = (size_t)C * W * H;
    std::vector<float*> pn, pb, pd;
    flat_ptrs(pn, h_noisy, h_mask, awidth * aheight, img, false);
    flat_ptrs(pb, h_basic, h_mask, awidth * aheight, img, true);
    flat_ptrs(pd, h_denoised, h_mask, awidth * aheight, img, true);
    return host_job(c, 3, P1, P2, pn.data(), h_mask, pb.data(), pd.data(), ang_major, awidth, aheight, an1, an2, W, H, C);
}

int lfbm5d_step1_host(lfbm5d_ctx* c, const lfbm5d_params* P, float* h_noisy, const unsigned* h_mask,
                      float* h_basic, unsigned ang_major, unsigned awidth, unsigned aheight, unsigned an,
                      unsigned W, unsigned H, unsigned C) {
    if (!c || !P || !h_mask || !h_noisy || !h_basic) return 1;
    const size_t img = (size_t)C * W * H;
    std::vector<float*> pn, pb;
    flat_ptrs(pn, h_noisy, h_mask, awidth * aheight, img, false);
    flat_ptrs(pb, h_basic, h_mask, awidth * aheight, img, true);
    return host_job(c, 1, P, nullptr, pn.data(), h_mask, pb.data(), nullptr, ang_major, awidth, aheight, an, an, W, H, C);
}

int lfbm5d_step2_host(lfbm5d_ctx* c, const lfbm5d_params* P, float* h_noisy, const unsigned* h_mask,
                      float* h_basic, float* h_denoised, unsigned ang_major, unsigned awidth,
                      unsigned aheight, unsigned an, unsigned W, unsigned H, unsigned C) {
    if (!c || !P || !h_mask || !h_noisy || !h_basic || !h_denoised) return 1;
    const size_t img = (size_t)C * W * H;
    std::vector<float*> pn, pb, pd;
    flat_ptrs(pn, h_noisy, h_mask, awidth * aheight, img, false);
    flat_ptrs(pb, h_basic, h_mask, awidth * aheight, img, false);
    flat_ptrs(pd, h_denoised, h_mask, awidth * aheight, img, true);
    return host_job(c, 2, nullptr, P, pn.data(), h_mask, pb.data(), pd.data(), ang_major, awidth, aheight, an, an, W, H, C);
}

/* ---- per-SAI BM3D (LFBM3Ddenoising) ---- */
namespace {
/* bm3d_1st_step / bm3d_2nd_step (bm3d.cpp:315-690) on a mirror-padded image already in HBM: the core pass with a
 * one-image "window" (A = 1, no disparity search, no angular transform, Hadamard along the stack), then
 * d_out = numerator / denominator over the whole padded image (pixels no patch reached keep the input). */
int bm3d_step(lfbm5d_ctx* c, int step, const lfbm5d_bm3d_params* B, unsigned Wb, unsigned Hb, unsigned C,
              const float* d_noisy, const float* d_basic, float* d_out) {
    lfbm5d_params P;
    std::memset(&P, 0, sizeof(P));
    P.sigma = B->sigma; P.lambda = B->lambda3D; P.N = B->N; P.nSim = B->nHW; P.nDisp = 0; P.k = B->k; P.p = B->p;
    P.useSD = B->useSD; P.tau_2D = B->tau_2D; P.tau_4D = LFBM5D_ID; P.tau_5D = LFBM5D_HADAMARD; P.color_space = B->color_space;
    const size_t n = (size_t)C * Wb * Hb;
    HIPCK(c, c->w_num.reserve(n * sizeof(float)));
    HIPCK(c, c->w_den.reserve(n * sizeof(float)));
    HIPCK(c, hipMemsetAsync(c->w_num.p, 0, n * sizeof(float), c->stream));
    HIPCK(c, hipMemsetAsync(c->w_den.p, 0, n * sizeof(float), c->stream));
    const unsigned one = 1, zero = 0;
    c->pass_rank = 0; c->pass_world = 1; c->pass_reduce = false;
    if (pass_impl(c, step, &P, 1, 1, Wb, Hb, C, d_noisy, d_basic, c->w_num.as<float>(), c->w_den.as<float>(), &one, &zero, 0, 0, true))
        return 1;
    HIPCK(c, launch_estimate(c->stream, c->w_num.as<float>(), c->w_den.as<float>(), step == 1 ? d_noisy : d_basic, d_out, n));
    return 0;
}
int bm3d_fold(lfbm5d_ctx* c, const lfbm5d_bm3d_params* B, unsigned C, int step) {
    lfbm5d_params P; std::memset(&P, 0, sizeof(P)); P.k = B->k;
    HIPCK(c, hipStreamSynchronize(c->stream));
    drain_events(c);
    return fold_counters(c, &P, 1, C, step);
}
/* run_bm3d_LF (bm3d_LF.cpp:75-125) -> run_bm3d (bm3d.cpp:86-300, nb_threads == 1) on device-resident buffers */
int run_bm3d_lf(lfbm5d_ctx* c, const lfbm5d_bm3d_params* Hd, const lfbm5d_bm3d_params* Wn, float* d_noisy, const unsigned* h_mask,
                float* d_basic, float* d_denoised, unsigned asize, unsigned W, unsigned H, unsigned C) {
    /* the reference pads both steps by nHard, searches the second within nWien and crops it at offset nWien of the nHard-padded
     * image (bm3d.cpp:126-189): a shifted picture unless the two are equal -- reproduced as is for nWien <= nHard.  With
     * nWien > nHard the crop reaches rows and columns the second step never aggregates into, which the reference returns
     * as 0 / 0: no result to reproduce */
    if (Wn->nHW > Hd->nHW) return fail(c, "unsupported: BM3D with nWien > nHard (the reference's own result is undefined there: 0 / 0 in the cropped border)");
    if (Hd->color_space != Wn->color_space || Hd->sigma != Wn->sigma) return fail(c, "BM3D: both steps share sigma and colour space");
    const unsigned nP = Hd->nHW, Wb = W + 2 * nP, Hb = H + 2 * nP;
    const size_t img = (size_t)C * W * H, imgb = (size_t)C * Wb * Hb;
    /* The SAIs are independent images (bm3d_LF.cpp:106-121 is a plain loop): they are dealt to lanes -- contexts with a stream and
     * work buffers of their own -- so that the kernels of several SAIs are in flight together.  One SAI's launches are small (a
     * 512 x 512 image: 99 workgroups of the table kernel, whose duration is one table walk however few they are).
     * LFBM5D_BM3D_LANES (default 3: 64 -> 105 SAI-MP/s on 512 x 512 SAIs; 1: the sequential form); results do not depend on it. */
    unsigned n_sai = 0;
    for (unsigned st = 0; st < asize; st++) n_sai += h_mask[st] ? 1u : 0u;
    const unsigned n_l = std::max(1u, std::min({8u, (unsigned)std::max(1, c->opt->bm3d_lanes), std::max(1u, n_sai)}));
    while (c->lanes.size() + 1 < n_l) {
        std::string e;
        lfbm5d_ctx* x = new_ctx(c->device, e);
        if (!x) return fail(c, "lane context: " + e);
        x->opt = c->opt;
        c->lanes.push_back(x);
    }
    HIPCK(c, hipStreamSynchronize(c->stream));   /* the caller's stream has produced d_noisy */
    /* an error return must not leave other lanes' kernels running on the caller's buffers */
    struct DrainOnError { bool armed; ~DrainOnError() { if (armed) (void)hipDeviceSynchronize(); } } drain_guard{true};
    unsigned turn = 0;
    for (unsigned st = 0; st < asize; st++) {
        if (!h_mask[st]) continue;
        lfbm5d_ctx* const x = turn % n_l == 0 ? c : c->lanes[turn % n_l - 1];
        turn++;
        hipStream_t s = x->stream;
        HIPCK(c, x->w_noisy.reserve(imgb * sizeof(float)));
        HIPCK(c, x->w_basic.reserve(imgb * sizeof(float)));
        HIPCK(c, x->t_num.reserve(imgb * sizeof(float)));
        float* const wn = x->w_noisy.as<float>(); float* const wb = x->w_basic.as<float>(); float* const wo = x->t_num.as<float>();
        float* noisy = d_noisy + st * img; float* basic = d_basic + st * img; float* deno = d_denoised + st * img;
        if (C == 3) HIPCK(c, launch_color(s, noisy, Hd->color_space, W * H, 1));
        HIPCK(c, launch_symetrize(s, noisy, wn, W, H, C, nP));
        if (bm3d_step(x, 1, Hd, Wb, Hb, C, wn, nullptr, wo)) { if (x != c) c->err = x->err; return 1; }
        HIPCK(c, launch_unsymetrize(s, basic, wo, W, H, C, nP));
        HIPCK(c, launch_symetrize(s, basic, wb, W, H, C, nP));
        if (bm3d_step(x, 2, Wn, Wb, Hb, C, wn, wb, wo)) { if (x != c) c->err = x->err; return 1; }
        HIPCK(c, launch_crop(s, deno, wo, W, H, C, nP, Wn->nHW));
        if (C == 3) {
            HIPCK(c, launch_color(s, deno, Hd->color_space, W * H, 0));
            HIPCK(c, launch_color(s, noisy, Hd->color_space, W * H, 0));
            HIPCK(c, launch_color(s, basic, Hd->color_space, W * H, 0));
        }
        if (x->pending.size() >= 64) {   /* bound the event pool on large light fields */
            if (bm3d_fold(x, Hd, C, 1)) { if (x != c) c->err = x->err; return 1; }
        }
    }
    /* drain the lanes, fold their counters and event times into this context */
    for (unsigned l = 1; l < n_l; l++) {
        lfbm5d_ctx* const x = c->lanes[l - 1];
        if (bm3d_fold(x, Wn, C, 2)) { c->err = x->err; return 1; }
        c->stats.passes += x->stats.passes; c->stats.groups += x->stats.groups; c->stats.stack_patches += x->stats.stack_patches;
        c->stats.sadct_groups += x->stats.sadct_groups; c->stats.algorithmic_bytes += x->stats.algorithmic_bytes;
        c->stats.ms_bm += x->stats.ms_bm; c->stats.ms_group += x->stats.ms_group; c->stats.ms_aggregate += x->stats.ms_aggregate;
        c->stats.launches_group += x->stats.launches_group; c->stats.launches_aggregate += x->stats.launches_aggregate;
        std::memset(&x->stats, 0, sizeof(x->stats));
    }
    drain_guard.armed = false;   /* every lane has been synchronised above */
    return bm3d_fold(c, Wn, C, 2);
}
} /* namespace */

int lfbm5d_bm3d_step_device(lfbm5d_ctx* c, int step, const lfbm5d_bm3d_params* P, unsigned Wb, unsigned Hb, unsigned C,
                            const float* d_noisy, const float* d_basic, float* d_out) {
    if (!c || !P || (step != 1 && step != 2)) return 1;
    (void)hipSetDevice(c->device);
    if (step == 2 && !d_basic) return fail(c, "step 2 needs the basic estimate");
    if (bm3d_step(c, step, P, Wb, Hb, C, d_noisy, d_basic, d_out)) return 1;
    return bm3d_fold(c, P, C, step);
}

int lfbm5d_bm3d_lf_device(lfbm5d_ctx* c, const lfbm5d_bm3d_params* hard, const lfbm5d_bm3d_params* wien, float* d_noisy,
                          const unsigned* h_mask, float* d_basic, float* d_denoised, unsigned asize, unsigned W, unsigned H,
                          unsigned C) {
    if (!c || !hard || !wien || !h_mask) return 1;
    (void)hipSetDevice(c->device);
    return run_bm3d_lf(c, hard, wien, d_noisy, h_mask, d_basic, d_denoised, asize, W, H, C);
}

int lfbm5d_bm3d_lf_host(lfbm5d_ctx* c, const lfbm5d_bm3d_params* hard, const lfbm5d_bm3d_params* wien, float* h_noisy,
                        const unsigned* h_mask, float* h_basic, float* h_denoised, unsigned asize, unsigned W, unsigned H,
                        unsigned C) {
    if (!c || !hard || !wien || !h_mask) return 1;
    (void)hipSetDevice(c->device);
    const size_t bytes = (size_t)asize * C * W * H * sizeof(float);
    HIPCK(c, c->h2d_noisy.reserve(bytes));
    HIPCK(c, c->h2d_basic.reserve(bytes));
    HIPCK(c, c->h2d_out.reserve(bytes));
    HIPCK(c, hipMemcpy(c->h2d_noisy.p, h_noisy, bytes, hipMemcpyHostToDevice));
    HIPCK(c, hipMemsetAsync(c->h2d_basic.p, 0, bytes, c->stream));
    HIPCK(c, hipMemsetAsync(c->h2d_out.p, 0, bytes, c->stream));
    if (run_bm3d_lf(c, hard, wien, c->h2d_noisy.as<float>(), h_mask, c->h2d_basic.as<float>(), c->h2d_out.as<float>(), asize, W, H, C)) return 1;
    HIPCK(c, hipMemcpy(h_noisy, c->h2d_noisy.p, bytes, hipMemcpyDeviceToHost));
    HIPCK(c, hipMemcpy(h_basic, c->h2d_basic.p, bytes, hipMemcpyDeviceToHost));
    HIPCK(c, hipMemcpy(h_denoised, c->h2d_out.p, bytes, hipMemcpyDeviceToHost));
    return 0;
}

int lfbm5d_last_bm(lfbm5d_ctx* c, unsigned* n_refs, unsigned* h_refs, unsigned* h_self_idx,
                   unsigned* h_self_cnt, unsigned* h_best, unsigned char* h_shape) {
    if (!c) return 1;
    (void)hipSetDevice(c->device);
    HIPCK(c, hipStreamSynchronize(c->stream));
    const unsigned R = c->last_n_refs;
    if (n_refs) *n_refs = R;
    if (!R) return 0;
    if (h_refs) std::memcpy(h_refs, c->gc[c->last_gslot].last_refs_host.data(), R * sizeof(unsigned));
    if (h_self_idx) HIPCK(c, hipMemcpy(h_self_idx, c->self_idx.p, (size_t)R * c->last_N * sizeof(unsigned), hipMemcpyDeviceToHost));
    if (h_self_cnt) HIPCK(c, hipMemcpy(h_self_cnt, c->self_cnt.p, (size_t)R * sizeof(unsigned), hipMemcpyDeviceToHost));
    if (h_best) HIPCK(c, hipMemcpy(h_best, c->best.p, c->last_A * c->last_plane * sizeof(unsigned), hipMemcpyDeviceToHost));
    if (h_shape) HIPCK(c, hipMemcpy(h_shape, c->shape.p, c->last_A * c->last_plane, hipMemcpyDeviceToHost));
    return 0;
}

size_t lfbm5d_last_tables(lfbm5d_ctx* c, float* h_tables, size_t n_floats) {
    if (!c) return 0;
    (void)hipSetDevice(c->device);
    if (hipStreamSynchronize(c->stream) != hipSuccess) return 0;
    const size_t have = c->tables.cap / sizeof(float);
    if (!h_tables) return have;
    const size_t n = std::min(have, n_floats);
    if (n && hipMemcpy(h_tables, c->tables.p, n * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) return 0;
    return n;
}

size_t lfbm5d_last_weights(lfbm5d_ctx* c, float* h_w, size_t n_floats) {
    if (!c) return 0;
    (void)hipSetDevice(c->device);
    if (hipStreamSynchronize(c->stream) != hipSuccess) return 0;
    const size_t have = c->wgt.cap / sizeof(float);
    if (!h_w) return have;
    const size_t n = std::min(have, n_floats);
    if (n && hipMemcpy(h_w, c->wgt.p, n * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) return 0;
    return n;
}

size_t lfbm5d_last_scores(lfbm5d_ctx* c, float* h_scores, size_t n_floats) {
    if (!c) return 0;
    (void)hipSetDevice(c->device);
    if (hipStreamSynchronize(c->stream) != hipSuccess) return 0;
    const size_t have = c->scores.cap / sizeof(float);
    if (!h_scores) return have;
    const size_t n = std::min(have, n_floats);
    if (n && hipMemcpy(h_scores, c->scores.p, n * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) return 0;
    return n;
}

int lfbm5d_last_scan_version(const lfbm5d_ctx* c) { return c ? c->last_scan_version : 0; }

int lfbm5d_malloc(void** dptr, size_t bytes) { return hipMalloc(dptr, bytes) == hipSuccess ? 0 : 1; }
int lfbm5d_free(void* dptr) { return hipFree(dptr) == hipSuccess ? 0 : 1; }
int lfbm5d_memcpy_h2d(void* dst, const void* src, size_t bytes) { return hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice) == hipSuccess ? 0 : 1; }
int lfbm5d_memcpy_d2h(void* dst, const void* src, size_t bytes) { return hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost) == hipSuccess ? 0 : 1; }

} /* extern "C" */
