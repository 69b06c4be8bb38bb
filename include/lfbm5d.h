/*
 * lfbm5d.h -- C-ABI of the MI355X-native LFBM5D denoising core (liblfbm5d_hip.so).
 *
 * The reference (V-Sense/LFBM5D) has no FFI layer; its seam for this path is two pairs of C++ free
 * functions with std::vector arguments:
 *   outer seam  run_bm5d_1st_step  src/bm5d.h:11-35   (called from src/main.cpp:195)
 *               run_bm5d_2nd_step  src/bm5d.h:38-62   (called from src/main.cpp:242)
 *   inner seam  bm5d_1st_step      src/bm5d_core_processing.h:6-42  (called from bm5d.cpp:351)
 *               bm5d_2nd_step      src/bm5d_core_processing.h:44-80 (called from bm5d.cpp:1050)
 * Every entry point below names the reference function it replaces.  Plain pointers, sizes and POD
 * structs only; return value 0 = success, 1 = failure (EXIT_SUCCESS / EXIT_FAILURE like the
 * reference), message via lfbm5d_last_error().  No C++ types, no exceptions cross this boundary.
 * One host thread per context.  All compute runs on the GPU: there is no CPU fallback, creation
 * fails when no HIP device is present.
 *
 * Stream contract: every context launches on a stream of its own (lfbm5d_stream) and is unaware of the
 * caller's streams.  Device buffers handed to an entry point must be READY on entry (whatever the caller
 * queued on them -- fills, copies, kernels -- has completed or the caller's stream has been synchronised);
 * every entry point returns with its results COMPLETE (it synchronises its stream before returning).
 * Several contexts may share a GPU and run concurrently from different host threads.
 *
 * Light-field layout (same as the reference, utilities_LF.cpp:140-146): asize = awidth*aheight
 * sub-aperture images (SAIs), each C planes of H*W float32 (planar, values nominally 0..255),
 * SAI index st = s*awidth + t (ang_major = LFBM5D_ROWMAJOR) or s + t*aheight (LFBM5D_COLMAJOR);
 * buffers are [asize][C*H*W] contiguous.
 */
#ifndef LFBM5D_H
#define LFBM5D_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* enum ints of the reference (src/bm5d.cpp:36-48) */
#define LFBM5D_YUV       0
#define LFBM5D_YCBCR     1
#define LFBM5D_OPP       2
#define LFBM5D_RGB       3
#define LFBM5D_ID        4
#define LFBM5D_DCT       5
#define LFBM5D_SADCT     6
#define LFBM5D_BIOR      7
#define LFBM5D_HADAMARD  8
#define LFBM5D_HAAR      9
#define LFBM5D_ROWMAJOR  11
#define LFBM5D_COLMAJOR  12

typedef struct lfbm5d_ctx lfbm5d_ctx;

/* The parameter tail of run_bm5d_1st_step / run_bm5d_2nd_step (bm5d.h:11-62). */
typedef struct {
    float    sigma;        /* sigma                                   */
    float    lambda;       /* lambdaHard5D (ignored by step 2)        */
    unsigned N;            /* NHard / NWien                           */
    unsigned nSim;         /* nSim                                    */
    unsigned nDisp;        /* nDisp                                   */
    unsigned k;            /* kHard / kWien                           */
    unsigned p;            /* pHard / pWien                           */
    unsigned useSD;        /* useSD                                   */
    unsigned tau_2D;       /* LFBM5D_ID | _DCT | _BIOR                */
    unsigned tau_4D;       /* LFBM5D_ID | _DCT | _SADCT               */
    unsigned tau_5D;       /* LFBM5D_HADAMARD | _HAAR | _DCT          */
    unsigned color_space;  /* LFBM5D_YUV | _YCBCR | _OPP | _RGB       */
} lfbm5d_params;

/* Counters and HIP-event timings accumulated since the last lfbm5d_reset_stats(). */
typedef struct {
    unsigned long long windows;        /* angular search windows visited                      */
    unsigned long long passes;         /* core passes (bm5d_*_step calls)                     */
    unsigned long long groups;         /* 5-D groups processed on this rank                   */
    unsigned long long stack_patches;  /* sum of nSx_r over those groups                      */
    unsigned long long sadct_groups;   /* groups that used the shape-adaptive 4-D transform   */
    double algorithmic_bytes;          /* sum_groups (4*S + 16) B * nSx*A*k^2*C  */
    double ms_bm;                      /* block-matching kernels, HIP-event time              */
    double ms_group;                   /* 5-D transform + shrink kernel                       */
    double ms_aggregate;               /* aggregation kernel                                  */
    double ms_other;                   /* padding / estimate / reductions / colour            */
    double ms_comm;                    /* RCCL all-reduce                                     */
    unsigned long long launches_group; /* launches of the transform kernel                    */
    unsigned long long launches_aggregate;
    unsigned long long lane_windows;   /* windows that ran on a lane other than the first (pipelined steps) */
    unsigned long long messages;       /* SAI messages (num + den of one SAI) exchanged between ranks, graph form   */
} lfbm5d_stats;

/* ---- context ---- */
/* Binds HIP device `device` (index within the visible devices), creates the stream the whole path
 * runs on.  Fails (returns 1, *out = NULL) when no HIP device is available. */
int  lfbm5d_create(lfbm5d_ctx** out, int device);
void lfbm5d_destroy(lfbm5d_ctx* ctx);
const char* lfbm5d_last_error(const lfbm5d_ctx* ctx); /* ctx may be NULL: creation errors */
void lfbm5d_reset_stats(lfbm5d_ctx* ctx);
void lfbm5d_get_stats(const lfbm5d_ctx* ctx, lfbm5d_stats* out);
/* The HIP stream (hipStream_t) the context launches on, for callers that time with events. */
void* lfbm5d_stream(lfbm5d_ctx* ctx);

/* ---- run-time options (round 6; lfbm5d_amd/csrc/lfbm5d_options.h has the list) ----
 * Every knob of the library is a per-context option.  lfbm5d_create fills them ONCE from the environment (LFBM5D_LANES=3 exported
 * before the context exists still means three window lanes); afterwards they change only through lfbm5d_set_option -- no entry
 * point reads the environment, so two contexts of one process can run different settings.  Keys: "lanes" (window lanes of the
 * graph form, 1..8, default 2), "fused" (0: lfbm5d_denoise_* runs the two calls), "step_sharding" ("rows" / "blocks" / 0),
 * "max_windows", "emulate_world", "data_driven_schedule", "host_blocking", "band_mb", "bm3d_lanes", "spatial_bands" / "band_halo"
 * (several ranks, lfbm5d_denoise_*: teams of ranks denoise horizontal bands of every SAI, see the multi-GPU notes below); test hooks that select
 * between implementations of the same arithmetic ("scan_v1", "scan_any", "scan_full_tables", "dct8w_v2", "group_generic",
 * "no_sa_kernels", "no_slab_kernel", "wide_nosplit", "agg_64bit", "agg_scalar_scan", "subset_list_host", "subset_scan_v1", "filt_group_major",
 * "scan_lds_cap", "force_redo").  The old variable names ("LFBM5D_LANES" ...) are accepted as keys.  Values are spelled as the
 * environment spelled them (integers; "rows" / "blocks"; flags: anything but "0" is on); value NULL or "" restores the default.
 * Unknown key: returns 1.  lfbm5d_get_option writes the current value as text (size: bytes of `value`). */
int lfbm5d_set_option(lfbm5d_ctx* ctx, const char* key, const char* value);
int lfbm5d_get_option(lfbm5d_ctx* ctx, const char* key, char* value, unsigned long long size);

/* ---- multi-GPU: one process per GPU, every rank holds the whole (read-only) light field.
 * Whole steps (lfbm5d_step*) and the two-step job (lfbm5d_denoise_*): the windows of the reference's schedule
 * (bm5d.cpp:165-407; a pure function of the SAI mask, see lfbm5d_plan_windows) form a dependency graph -- a window has to
 * wait exactly for the previous window of its step that touched each of its SAIs, because windows interact only through
 * num / den of shared SAIs (the running estimate block matching reads, the sums aggregation adds to); in the two-step job a
 * window of the second step also waits for the last first-step window on each of its SAIs, behind which that SAI's basic
 * estimate is final.  Every window has an owner rank, chosen along a simulated execution (a window follows the rank of the
 * windows of its row of SAIs while that rank is free; lfbm5d_plan_graph / lfbm5d_plan_job return the assignment -- a pure
 * function of the mask, the steps and the rank count); what a window needs from a window of another rank travels as one
 * RCCL send / recv per SAI (num and den of that SAI, or its basic estimate; xGMI point-to-point); at the end every SAI's
 * outputs are formed on the rank that holds them and broadcast.  Every window sees exactly the sums the single-GPU order shows
 * it: the result is BIT-IDENTICAL to one GPU for any rank count (lfbm5d_plan_job exposes owners, issue order and the
 * message list).  The reference's backward raster leaves a wavefront -- a row of windows may run two windows behind the row
 * before it -- so the speed-up is bounded by the graph's critical path: 22 of 64 window slots for ONE step of a 17x17 light
 * field (2.9x however many ranks), 27 of 128 window times for the two-step job (4.75x at eight ranks): use lfbm5d_denoise_*.
 * Environment LFBM5D_STEP_SHARDING selects the alternatives: "rows" = single-GPU window order with every core pass
 * row-sharded as below (exact, two all-reduces per pass; also what greyscale light fields need); "blocks" = round 1's
 * contiguous blocks of windows per rank + ONE all-reduce per step, which scales with the rank count but is NOT the
 * reference's result (a rank's block matching only sees its own earlier windows: -0.01 / -0.03 / -0.07 dB at 2 / 4 / 8 ranks).
 * Option "spatial_bands" = S > 1 (lfbm5d_denoise_* only; "0" = S from lfbm5d_auto_bands) adds a level above the graph for rank counts beyond what the graph keeps
 * busy: S teams of world / S ranks (rank = band * team size + team rank), team b runs the whole two-step job on rows
 * [b H / S, (b + 1) H / S) of every SAI plus "band_halo" rows on either side (default nSim + nDisp + k of the wider step) as a
 * light field of its own -- the same window graph, on a communicator split off for the team -- and ONE all-gather stitches the
 * three light fields.  NOT bit-identical to one GPU (a band's distance tables start their float recurrences at its first row, near
 * ties fall differently); PSNR within 1e-3 dB on 512 x 512 SAIs, a tenth of the +-0.01 dB the reference's own tile mode is held to.
 * Single core passes (lfbm5d_pass_device): the reference patches are sharded by rows over the ranks and
 * the window's num/den all-reduced.
 * Replaces the reference's only parallelism, the OpenMP tile loop + undivide_LF merge
 * (bm5d.cpp:411-708, utilities_LF.cpp:438-515), without its tile-border quality loss. ---- */
#define LFBM5D_UNIQUE_ID_BYTES 128
int lfbm5d_comm_unique_id(void* id_out /* LFBM5D_UNIQUE_ID_BYTES */);
/* (whole steps / jobs on several ranks run window lanes and two exchange streams at once: set GPU_MAX_HW_QUEUES >= 8 in the
 * environment BEFORE the HIP runtime initialises -- ROCm maps streams onto 4 hardware queues by default, and a send that
 * waits for its peer must not sit in front of a compute stream on the same queue; bench.py does it for itself) */
int lfbm5d_comm_init(lfbm5d_ctx* ctx, const void* id, int rank, int world);
/* TESTS ONLY -- the exchange of the window graph between PROCESSES THAT SHARE ONE GPU.  RCCL refuses several ranks on one device, so
 * the multi-rank form could never run between real processes on a one-GPU box; this second transport plays every message of the
 * graph with the same issue order, event gating, channels and abort path as the RCCL form (lfbm5d_api.hip, run_graph), only the
 * bytes move differently: the receiver's exchange stream waits for a word the sender publishes behind its window (device memory
 * both processes map through hipIpcMemHandle), copies the SAI out of the sender's buffers and publishes "taken".  Rendezvous
 * (handles, the completion vote, the closing barriers) goes through small files in `rendezvous_dir`, which must be empty and
 * visible to all ranks.  A peer that dies ends the job with return value 1 after `timeout_s` (<= 0: 30 s), never with a hung GPU.
 * After this call lfbm5d_step*_device / lfbm5d_denoise_device run as rank `rank` of `world`; results are bit-identical to one rank. */
int lfbm5d_comm_init_ipc(lfbm5d_ctx* ctx, int rank, int world, const char* rendezvous_dir, double timeout_s);
/* Diagnostics: all-reduce n floats on the context's stream through the context's communicator (or a
 * one-rank communicator created for the call) and verify the sums.  Returns 0 when RCCL works here. */
int lfbm5d_comm_selftest(lfbm5d_ctx* ctx, unsigned n);
/* Ranks of the context's RCCL communicator as RCCL itself counts them (ncclCommCount): 0 without a communicator, -1 on
 * error.  bench.py prints it so that a multi-GPU record shows the ranks that took part. */
int lfbm5d_comm_ranks(const lfbm5d_ctx* ctx);
/* Shard without a communicator (tests): this rank only processes its rows; no reduction. */
int lfbm5d_set_shard(lfbm5d_ctx* ctx, int rank, int world);
/* The reference's OpenMP tile mode for whole steps (bm5d.cpp:411-708, run_bm5d_* with nb_threads > 1): every window pass
 * runs tile by tile (sub_divide, utilities.cpp:312-395) and keeps the tiles' interiors only (undivide_LF,
 * utilities_LF.cpp:438-515).  nb_tiles is floored to a power of two like main.cpp:101-102 does with nbThreads; 0 / 1 =
 * off (the default: the untiled result, which is what nb_threads == 1 gives and about 0.5 dB better).  A compatibility
 * mode for reproducing a tiled reference run; one GPU, one window after the other.  Returns 0. */
int lfbm5d_set_tiles(lfbm5d_ctx* ctx, int nb_tiles);
/* Row range [begin,end) of n_rows reference-patch rows owned by `rank` of `world`. */
void lfbm5d_shard_rows(unsigned n_rows, int rank, int world, unsigned* begin, unsigned* end);
/* Spatial bands S for a two-step job on `world` ranks (option "spatial_bands"; host only): the window graph of an a x a light field
 * (a = the smaller angular side) keeps about 0.8 ceil(a / 3) ranks busy, so the graph takes the largest power of two of ranks within
 * that and bands take the rest -- while S divides `world` and a band (height / S rows) stays twice as tall as `halo` (0: 40, the
 * README parameters' nSim + nDisp + k).  1 = the graph alone.  Option value "0" applies this rule inside lfbm5d_denoise_*. */
int lfbm5d_auto_bands(unsigned awidth, unsigned aheight, unsigned height, unsigned halo, int world);
/* The window schedule of a step as the processed SAI (index in `ang_major` order) of each window, in
 * order: the centre SAI first, then always the last SAI not covered yet (bm5d.cpp:187-213 -- all
 * candidates tie on the zero-weight count because a window always finishes all of its SAIs).  Host
 * only, needs no GPU.  Returns the number of windows (-1 on bad arguments); writes min(n, cap) entries. */
int lfbm5d_plan_windows(unsigned awidth, unsigned aheight, unsigned an, unsigned ang_major, const unsigned* mask,
                        unsigned* out_sai, unsigned cap);
/* The graph form of a step for `world` ranks with `lanes` lanes each (host only, needs no GPU): owner rank, lane and
 * unit-time start slot of every window of lfbm5d_plan_windows' sequence in the simulated execution whose start order
 * (ties: the earlier window) is the ISSUE ORDER every rank enqueues in.  Returns the number of windows (-1 on bad
 * arguments); writes min(n, cap) entries to each non-NULL array. */
int lfbm5d_plan_graph(unsigned awidth, unsigned aheight, unsigned an, unsigned ang_major, const unsigned* mask, int world, int lanes,
                      unsigned* out_rank, unsigned* out_lane, unsigned* out_start, unsigned cap);
/* The messages of that graph in the order every rank issues them (by the producer's place in the issue order, then SAI
 * slot): out[4 i] = {producer window, consumer window, SAI, channel}; the producer's rank sends num and den of the SAI to the
 * consumer's rank once the producer window is done.
 * Returns the number of messages (-1 on bad arguments); writes min(n, cap) quadruples. */
int lfbm5d_plan_messages(unsigned awidth, unsigned aheight, unsigned an, unsigned ang_major, const unsigned* mask, int world,
                         unsigned* out, unsigned cap);
/* The graph of a JOB -- one step (n_steps = 1) or run_bm5d_1st_step + run_bm5d_2nd_step back to back (n_steps = 2, what
 * lfbm5d_denoise_* executes) -- for `world` graph ranks with `lanes` lanes each (host only, needs no GPU).  an[n_steps] =
 * half size of every step's angular search window; cost[n_steps] = relative cost of a window pass of every step in the
 * scheduling model (NULL: 10 / 9 for two steps, 1 for one).  In a two-step job a window of the second step waits, per SAI, for
 * the LAST window of the first step touching that SAI (the SAI's basic estimate is final then), so the second step's wavefront
 * follows the first's: 128 windows with a critical path of about 28 on a 17x17 light field instead of 2 x 22.
 *   out_nodes  8 unsigned per window, step 0's windows in plan order, then step 1's:
 *              {step slot, index in the step's sequence, processed SAI, graph rank, lane, start time (cost units) of the simulated
 *               execution, position in the ISSUE ORDER every rank walks, chain}
 *   out_msgs   6 unsigned per message, in issue order: {kind, producer node, consumer node (kind 0) or 0xffffffff, receiving graph
 *              rank, SAI, channel}; kind 0 = num and den of the SAI from the producer's rank to its next toucher's, kind 1 = the
 *              basic estimate of the SAI, finalised behind the producer node, to a rank whose second-step windows read it
 *   out_counts {windows, messages, makespan of the simulated execution in cost units, 1 if every window centre is non-empty}
 * Returns the number of windows (-1 on bad arguments); writes min(n, cap) entries to each non-NULL array. */
int lfbm5d_plan_job(unsigned awidth, unsigned aheight, unsigned ang_major, const unsigned* mask, int n_steps, const unsigned* an,
                    const unsigned* cost, int world, int lanes, unsigned* out_nodes, unsigned node_cap, unsigned* out_msgs,
                    unsigned msg_cap, unsigned* out_counts);
/* The windows the last lfbm5d_step* / lfbm5d_denoise_* call on this context actually ran (same encoding; a two-step job: the
 * first step's windows, then the second's). */
int lfbm5d_last_windows(const lfbm5d_ctx* ctx, unsigned* out_sai, unsigned cap);

/* ---- outer seam, device-resident: LF buffers already in HBM ----
 * lfbm5d_step1_device == run_bm5d_1st_step (bm5d.h:11-35, nb_threads == 1 semantics):
 *   d_noisy  [asize][C*H*W]  in/out: colour-transformed at entry and back at exit, exactly like the
 *                            reference mutates LF_noisy (bm5d.cpp:133, :713)
 *   h_mask   [asize] host    LF_SAI_mask (0 = empty SAI)
 *   d_basic  [asize][C*H*W]  out: basic estimate (RGB)
 * lfbm5d_step2_device == run_bm5d_2nd_step (bm5d.h:38-62): d_basic is in/out (bm5d.cpp:829,:1416),
 *   d_denoised is the output. */
int lfbm5d_step1_device(lfbm5d_ctx* ctx, const lfbm5d_params* P, float* d_noisy,
                        const unsigned* h_mask, float* d_basic, unsigned ang_major,
                        unsigned awidth, unsigned aheight, unsigned an, unsigned W, unsigned H,
                        unsigned C);
int lfbm5d_step2_device(lfbm5d_ctx* ctx, const lfbm5d_params* P, float* d_noisy,
                        const unsigned* h_mask, float* d_basic, float* d_denoised,
                        unsigned ang_major, unsigned awidth, unsigned aheight, unsigned an,
                        unsigned W, unsigned H, unsigned C);

/* Both steps as ONE job == run_bm5d_1st_step followed by run_bm5d_2nd_step (main.cpp:195, :242), bit-identical to
 * lfbm5d_step1_device + lfbm5d_step2_device: the windows of both steps form one dependency graph (lfbm5d_plan_job) and what the
 * reference does between the two calls (final estimate, inverse and forward colour transform: bm5d.cpp:405, :711-714, :827-830)
 * happens SAI by SAI as soon as a SAI's basic estimate is final.  On one GPU that closes the gap between the steps; on several
 * it is what lets all ranks work (the second step's wavefront follows the first's).  P1 / an1 = the hard-thresholding step's
 * parameters, P2 / an2 = the Wiener step's.  d_noisy in/out, d_basic and d_denoised out, exactly as the two calls leave them.
 * Light fields outside the graph form (greyscale, an empty SAI at a window centre, tile mode, LFBM5D_STEP_SHARDING,
 * LFBM5D_DATA_DRIVEN_SCHEDULE, LFBM5D_FUSED=0) run the two calls one after the other. */
int lfbm5d_denoise_device(lfbm5d_ctx* ctx, const lfbm5d_params* P1, const lfbm5d_params* P2, float* d_noisy,
                          const unsigned* h_mask, float* d_basic, float* d_denoised, unsigned ang_major,
                          unsigned awidth, unsigned aheight, unsigned an1, unsigned an2, unsigned W, unsigned H, unsigned C);

/* ---- outer seam, host buffers (what the run_bm5d_* wrappers of the drop-in call): same
 * semantics, the library stages through HBM (PCIe-inclusive).
 * ERRORS: the host forms write their in / out buffers (h_noisy, h_basic, h_denoised) while the job runs -- the streamed form
 * delivers a SAI's outputs as soon as the last window on it has run.  After a NON-ZERO return the contents of every in / out
 * buffer of the call are UNDEFINED (partly inputs, partly colour-round-tripped outputs): a caller that wants to retry keeps its
 * own copy of the inputs.  (The reference has the same contract: run_bm5d_* transform LF_noisy in place before anything can
 * fail, bm5d.cpp:133.) ---- */
int lfbm5d_step1_host(lfbm5d_ctx* ctx, const lfbm5d_params* P, float* h_noisy,
                      const unsigned* h_mask, float* h_basic, unsigned ang_major, unsigned awidth,
                      unsigned aheight, unsigned an, unsigned W, unsigned H, unsigned C);
int lfbm5d_step2_host(lfbm5d_ctx* ctx, const lfbm5d_params* P, float* h_noisy,
                      const unsigned* h_mask, float* h_basic, float* h_denoised,
                      unsigned ang_major, unsigned awidth, unsigned aheight, unsigned an,
                      unsigned W, unsigned H, unsigned C);

int lfbm5d_denoise_host(lfbm5d_ctx* ctx, const lfbm5d_params* P1, const lfbm5d_params* P2, float* h_noisy,
                        const unsigned* h_mask, float* h_basic, float* h_denoised, unsigned ang_major,
                        unsigned awidth, unsigned aheight, unsigned an1, unsigned an2, unsigned W, unsigned H, unsigned C);

/* The same with the caller's light fields as ONE HOST POINTER PER SAI -- h_x[st] = &LF_x[st][0], C*H*W floats each, exactly the
 * reference's vector<vector<float>> (src/bm5d.h:11-62) without a flat copy; entries of empty SAIs are ignored (may be NULL).
 * The flat forms above are these with pointers into one buffer.  On one rank with the window graph (colour light fields) the
 * SAIs are STREAMED: a SAI goes up when the first window that needs it is enqueued (forward colour transform behind the copy),
 * its outputs come down as soon as the last window on it has run, while the other windows compute -- the interval the reference
 * times (main.cpp:189-201, :241-247) then costs a few per cent more than device-resident buffers instead of the 20 % four
 * blocking copies of the light field cost.  Pageable memory is fine.  Everything else about the call (in / out arguments, result
 * bit-identical to the device form) is unchanged; LFBM5D_HOST_BLOCKING=1 selects the upload-all / download-all form. */
int lfbm5d_step1_host_sai(lfbm5d_ctx* ctx, const lfbm5d_params* P, float* const* h_noisy, const unsigned* h_mask,
                          float* const* h_basic, unsigned ang_major, unsigned awidth, unsigned aheight, unsigned an,
                          unsigned W, unsigned H, unsigned C);
int lfbm5d_step2_host_sai(lfbm5d_ctx* ctx, const lfbm5d_params* P, float* const* h_noisy, const unsigned* h_mask,
                          float* const* h_basic, float* const* h_denoised, unsigned ang_major, unsigned awidth,
                          unsigned aheight, unsigned an, unsigned W, unsigned H, unsigned C);
int lfbm5d_denoise_host_sai(lfbm5d_ctx* ctx, const lfbm5d_params* P1, const lfbm5d_params* P2, float* const* h_noisy,
                            const unsigned* h_mask, float* const* h_basic, float* const* h_denoised, unsigned ang_major,
                            unsigned awidth, unsigned aheight, unsigned an1, unsigned an2, unsigned W, unsigned H, unsigned C);

/* ---- inner seam: one core pass on a mirror-padded angular window, device pointers ----
 * == bm5d_1st_step (step = 1) / bm5d_2nd_step (step = 2) (bm5d_core_processing.h:6-80).
 * Buffers are [A][C*Wb*Hb], A = aw*ah; d_basic may be NULL for step 1; d_num / d_den are
 * accumulated into.  h_mask / h_procSAI are host arrays of A entries (LF_SAI_mask, procSAI). */
int lfbm5d_pass_device(lfbm5d_ctx* ctx, int step, const lfbm5d_params* P, unsigned aw,
                       unsigned ah, unsigned Wb, unsigned Hb, unsigned C, const float* d_noisy,
                       const float* d_basic, float* d_num, float* d_den, const unsigned* h_mask,
                       const unsigned* h_procSAI, unsigned cst, unsigned pst);

/* ---- per-SAI BM3D: the reference's comparison tool LFBM3Ddenoising, on the same kernels ----
 * One step's parameters of run_bm3d (src/bm3d.h:11-34). */
typedef struct {
    float    sigma;        /* sigma                                   */
    float    lambda3D;     /* lambdaHard3D (ignored by step 2)        */
    unsigned N;            /* NHard / NWien: power of two, 2..32      */
    unsigned nHW;          /* nHard / nWien: half search window       */
    unsigned k;            /* kHard / kWien                           */
    unsigned p;            /* pHard / pWien                           */
    unsigned useSD;        /* useSD_h / useSD_w                       */
    unsigned tau_2D;       /* LFBM5D_DCT | LFBM5D_BIOR                */
    unsigned color_space;  /* LFBM5D_YUV | _YCBCR | _OPP | _RGB       */
} lfbm5d_bm3d_params;
/* == bm3d_1st_step (step = 1, src/bm3d.h:37-55) / bm3d_2nd_step (step = 2, src/bm3d.h:58-76) on a mirror-padded,
 * colour-transformed image [C][Hb][Wb] in HBM; d_basic may be NULL for step 1.  d_out [C][Hb][Wb] receives
 * numerator / denominator (pixels no patch reached keep the step's input image). */
int lfbm5d_bm3d_step_device(lfbm5d_ctx* ctx, int step, const lfbm5d_bm3d_params* P, unsigned Wb, unsigned Hb,
                            unsigned C, const float* d_noisy, const float* d_basic, float* d_out);
/* == run_bm3d_LF (src/bm3d_LF.h:10-35; run_bm3d src/bm3d.h:11-34 with nb_threads == 1 for every SAI of the mask).
 * Buffers [asize][C*H*W]; d_noisy is colour-transformed at entry and back at exit like the reference mutates
 * LF_noisy (bm3d.cpp:115, :290); d_basic and d_denoised are outputs (RGB).  nHard != nWien reproduces the reference's
 * crop of the second step at offset nWien of the nHard-padded image (bm3d.cpp:181-189: a shifted picture); nWien <= nHard
 * (beyond that the reference itself returns 0 / 0 in the border). */
int lfbm5d_bm3d_lf_device(lfbm5d_ctx* ctx, const lfbm5d_bm3d_params* hard, const lfbm5d_bm3d_params* wien,
                          float* d_noisy, const unsigned* h_mask, float* d_basic, float* d_denoised,
                          unsigned asize, unsigned W, unsigned H, unsigned C);
int lfbm5d_bm3d_lf_host(lfbm5d_ctx* ctx, const lfbm5d_bm3d_params* hard, const lfbm5d_bm3d_params* wien,
                        float* h_noisy, const unsigned* h_mask, float* h_basic, float* h_denoised,
                        unsigned asize, unsigned W, unsigned H, unsigned C);

/* ---- inspection of the last pass's block matching (parity tests) ----
 * n_refs reference patches in raster order; h_refs[n_refs] flat index i*Wb+j;
 * h_self_idx[n_refs*N], h_self_cnt[n_refs] (precompute_BM, core:3301);
 * h_best[A*Wb*Hb] / h_shape[A*Wb*Hb] (precompute_BM_stereo, core:3479; entry st == pst unused).
 * Any output pointer may be NULL.  Returns the number of reference patches via n_refs. */
int lfbm5d_last_bm(lfbm5d_ctx* ctx, unsigned* n_refs, unsigned* h_refs, unsigned* h_self_idx,
                   unsigned* h_self_cnt, unsigned* h_best, unsigned char* h_shape);

/* Raw disparity distance tables of the last pass (the scratch precompute_BM_stereo's sum_table plays in the
 * reference, core:3513-3574), in the layout of the kernel generation that ran (lfbm5d_last_scan_version below): copies min(n_floats, size) floats and
 * returns the count; h_tables == NULL returns the buffer's size in floats.  For the bit-reproducibility tests. */
size_t lfbm5d_last_tables(lfbm5d_ctx* ctx, float* h_tables, size_t n_floats);
/* Candidate scores of the self-similarity search of the last pass, [reference patch][(2 nSim+1)^2] in the scan order of
 * core:3407-3420 (entries no table covers keep 2 * threshold): same calling convention. */
size_t lfbm5d_last_scores(lfbm5d_ctx* ctx, float* h_scores, size_t n_floats);
/* Aggregation weights of the groups of the last pass, [reference patch][channel] (core:413-421: 1 / (sigma_c^2 * retained
 * coefficients) in the hard-threshold step): same calling convention.  Lets a test compare survivor counts group by group. */
size_t lfbm5d_last_weights(lfbm5d_ctx* ctx, float* h_w, size_t n_floats);
/* Which generation of the table kernel the last pass used, i.e. what lfbm5d_last_tables returns:
 *   3 = ring-sharing workgroups, combined form (the default): the disparity tables never reach memory; the buffer holds
 *       [slot][workgroup of the slot][strip][chunk of 8 steps][lane * 8 + step] pairs of (smallest value of the workgroup's tables,
 *       its place in the reference's scan order), 8 bytes each, then per table an edge array [rows][strips] (table column 0 and the
 *       row-0 entry of every strip's first column) -- lfbm5d_kernels.h, stereo_part_stride / stereo_edge_stride;
 *   2 = ring-sharing workgroups with full tables (LFBM5D_SCAN_FULL_TABLES=1), [slot][(2 nDisp+1)^2]{[strip][Q / 4][lane][Q % 4],
 *       column 0}, lfbm5d_kernels.h stereo_table_stride2;
 *   1 = one wave per table (12x12 patches, irregular reference lists, estimates of 2 GiB and more, LFBM5D_SCAN_V1=1), skewed
 *       layout [slot][(2 nDisp+1)^2][strip][table row + lane][64], stereo_table_stride. */
int lfbm5d_last_scan_version(const lfbm5d_ctx* ctx);

/* ---- device memory helpers so hosts without a HIP binding (ctypes, cgo, JNI) can stage data ---- */
int lfbm5d_malloc(void** dptr, size_t bytes);
int lfbm5d_free(void* dptr);
int lfbm5d_memcpy_h2d(void* dst, const void* src, size_t bytes);
int lfbm5d_memcpy_d2h(void* dst, const void* src, size_t bytes);
int lfbm5d_device_count(void);

#ifdef __cplusplus
}
#endif
#endif
