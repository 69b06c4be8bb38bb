/*
 * lfbm5d_api.hip -- the C-ABI of include/lfbm5d.h: context and options, communicators, the device and host entry points of
 * the outer seam (lfbm5d_step* / lfbm5d_denoise_* -> lfbm5d_steps.hip), the inner seam (lfbm5d_pass_device -> lfbm5d_pass.hip),
 * per-SAI BM3D, inspection of the last pass.  The host side's other translation units: lfbm5d_ctx.h.
 */
#include "lfbm5d_graph.h"

using namespace lfbm5d_host;
using lfbm5d::plan::search_window;
using lfbm5d::plan::plan_windows;

namespace { std::string g_create_error; }

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
    if (c->team_comm2) ncclCommDestroy(c->team_comm2);
    if (c->team_comm) ncclCommDestroy(c->team_comm);
    if (c->comm2) ncclCommDestroy(c->comm2);
    if (c->comm) ncclCommDestroy(c->comm);
    for (int i = 0; i < 2; i++) if (c->cs[i]) (void)hipStreamDestroy(c->cs[i]);
    for (GeomCache& g : c->gc) { g.refs.release(); g.rslot.release(); g.tb.release(); g.scan_wgs.release(); }
    DevBuf* bufs[] = {&c->est, &c->g_num2, &c->g_den2, &c->n2, &c->e_basic, &c->refmap, &c->scores, &c->tables, &c->self_idx, &c->self_cnt, &c->best,
                      &c->t_noisy, &c->t_basic, &c->t_tnum, &c->t_tden, &c->und_num, &c->und_den, &c->sub_flags, &c->sub_cnt, &c->shape, &c->filt, &c->wgt, &c->aggpos, &c->gpos, &c->gofs, &c->gok, &c->sa_list, &c->gshape, &c->band_noisy, &c->band_basic, &c->band_out, &c->band_src, &c->band_pack, &c->band_gather, &c->counters, &c->small, &c->t_num, &c->t_den, &c->d_mask, &c->g_num, &c->g_den, &c->w_noisy,
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

int lfbm5d_auto_bands(unsigned awidth, unsigned aheight, unsigned height, unsigned halo, int world) {
    return auto_bands(awidth, aheight, height, halo ? halo : 40u, world);
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
    /* a second call replaces the communicators (after a job that tore them down, or to change the ranks) */
    if (c->team_comm2) { (void)ncclCommDestroy(c->team_comm2); c->team_comm2 = nullptr; }
    if (c->team_comm) { (void)ncclCommDestroy(c->team_comm); c->team_comm = nullptr; }
    c->team_S = 0;
    if (c->comm2) { (void)ncclCommDestroy(c->comm2); c->comm2 = nullptr; }
    if (c->comm) { (void)ncclCommDestroy(c->comm); c->comm = nullptr; }
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
