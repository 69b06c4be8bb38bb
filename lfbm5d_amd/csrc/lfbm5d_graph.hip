/*
 * lfbm5d_graph.hip -- a job's windows as a dependency graph (lfbm5d_plan.h) executed on lanes and ranks: run_graph with its RCCL
 * and IPC transports and the streamed host seam.  Split from lfbm5d_api.hip in round 6 (lfbm5d_ctx.h).
 */
#include "lfbm5d_graph.h"

namespace lfbm5d_host {

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
    const std::string base = c->ipc_tag + tag + "." + std::to_string(c->ipc_epoch) + ".";
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
    const std::string base = c->ipc_tag + "handles." + std::to_string(c->ipc_epoch) + ".";
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

} /* namespace lfbm5d_host */
