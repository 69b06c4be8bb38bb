/*
 * lfbm5d_ctx.h -- what the host side of the library shares between its translation units (round 6: lfbm5d_api.hip, 2 500 lines,
 * was split along its seams):
 *   lfbm5d_pass.hip    one core pass (bm5d_1st_step / bm5d_2nd_step) as a sequence of kernels: geometry, tables, block matching,
 *                      group stage, aggregation; counters and event times
 *   lfbm5d_graph.hip   a job's windows as a dependency graph on lanes and ranks: run_graph, the RCCL and IPC transports, the
 *                      streamed host seam
 *   lfbm5d_steps.hip   run_bm5d_1st_step / run_bm5d_2nd_step / both as one job on device buffers: schedule forms, tile mode
 *   lfbm5d_api.hip     the C-ABI of include/lfbm5d.h: context, options, communicators, host entry points, BM3D, inspection
 * Internal: nothing here is part of the ABI.
 */
#ifndef LFBM5D_CTX_H
#define LFBM5D_CTX_H

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


namespace lfbm5d_host {

using namespace lfbm5d;



constexpr double kSqrt2 = 1.414213562373095;     /* core:33 */
constexpr double kSqrt2Inv = 0.7071067811865475; /* core:34 */
constexpr double kPi = 3.14159265358979323846;

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

} /* namespace lfbm5d_host */

using lfbm5d_host::DevBuf; using lfbm5d_host::PassEvents; using lfbm5d_host::GeomCache; using lfbm5d::Options;


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
    std::string ipc_tag;   /* prefix of the rendezvous names: "" for the whole world, "b<band>." while a band's team runs its job (spatial bands) */
    DevBuf ipc_flags, ipc_out;
    struct IpcPeer { unsigned char handle[7][64]; void* ptr[7]; };   /* flags, g_num[0..1], g_den[0..1], basic, out -- as this process maps them */
    std::vector<IpcPeer> ipc_peers;
    DevBuf pristine, pristine_b;
    /* spatial bands (option spatial_bands, lfbm5d_steps.hip): a band's crop of the three light fields, the input as it arrived (emulated
     * ranks), the all-gather's buffers; the communicators of this rank's team (split from comm once per band count) */
    DevBuf band_noisy, band_basic, band_out, band_src, band_pack, band_gather;
    ncclComm_t team_comm = nullptr, team_comm2 = nullptr; int team_S = 0;
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

namespace lfbm5d_host {


#define HIPCK(ctx, call)                                                                         \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess) {                                                                  \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                      \
            return 1;                                                                            \
        }                                                                                        \
    } while (0)

inline int fail(lfbm5d_ctx* c, const std::string& m) { c->err = m; return 1; }

/* lfbm5d_pass.hip */
lfbm5d_ctx* new_ctx(int device, std::string& err);
hipEvent_t get_event(lfbm5d_ctx* c);
void drain_events(lfbm5d_ctx* c);
int sigma_table(float sigma, unsigned C, unsigned cs, float* out);
void ind_init(std::vector<unsigned>& v, unsigned max_size, unsigned N, unsigned step);
void build_tables(GroupTables& t, unsigned k, unsigned aw, unsigned ah);
bool is_pow2(unsigned n);
int validate(lfbm5d_ctx* c, int step, const lfbm5d_params* P, unsigned aw, unsigned ah, unsigned C, bool bm3d = false);
int fold_counters(lfbm5d_ctx* c, const lfbm5d_params* P, unsigned A, unsigned C, int step, int slot = 0);
/* lfbm5d_pass.hip */
int pass_impl(lfbm5d_ctx* c, int step, const lfbm5d_params* P, unsigned aw, unsigned ah, unsigned Wb,
              unsigned Hb, unsigned C, const float* d_noisy, const float* d_basic, float* d_num,
              float* d_den, const unsigned* h_mask, const unsigned* h_proc, unsigned cst, unsigned pst,
              bool bm3d = false);
/* lfbm5d_graph.hip */
int io_upload_all(lfbm5d_ctx* c, const HostIO* io, const unsigned* h_mask, unsigned asize, size_t img, float* d_noisy, float* d_basic_in);
int io_download_all(lfbm5d_ctx* c, const HostIO* io, const unsigned* h_mask, unsigned asize, size_t img, const float* d_noisy,
                    const float* d_basic, const float* d_out);
constexpr unsigned kIpcMaxMsgs = 4096;   /* gating words: ready[kIpcMaxMsgs], taken[kIpcMaxMsgs], error */
int auto_bands(unsigned awidth, unsigned aheight, unsigned height, unsigned halo, int world);   /* lfbm5d_steps.hip: lfbm5d_auto_bands */
int ipc_allgather(lfbm5d_ctx* c, const char* tag, int mine, std::vector<int>& all);
bool ipc_put(const std::string& dir, const std::string& name, const void* data, size_t bytes);
bool ipc_get(const std::string& dir, const std::string& name, void* data, size_t bytes, double timeout_s);
/* lfbm5d_steps.hip */
int run_step(lfbm5d_ctx* c, int step, const lfbm5d_params* P, float* d_noisy, const unsigned* h_mask,
             float* d_basic, float* d_out, unsigned ang_major, unsigned awidth, unsigned aheight,
             unsigned an, unsigned W, unsigned H, unsigned C, const HostIO* io = nullptr);
int run_denoise(lfbm5d_ctx* c, const lfbm5d_params* P1, const lfbm5d_params* P2, float* d_noisy, const unsigned* h_mask, float* d_basic,
                float* d_out, unsigned ang_major, unsigned awidth, unsigned aheight, unsigned an1, unsigned an2, unsigned W, unsigned H,
                unsigned C, const HostIO* io = nullptr);

} /* namespace lfbm5d_host */
#endif
