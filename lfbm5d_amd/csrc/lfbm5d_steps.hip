/*
 * lfbm5d_steps.hip -- run_bm5d_1st_step / run_bm5d_2nd_step (bm5d.cpp:88-747, :782-1452) and both as one job on device-resident
 * light fields: the forms of the window schedule (graph, planned sequence, data-driven, row / block sharding, tile mode).
 * Split from lfbm5d_api.hip in round 6 (lfbm5d_ctx.h).
 */
#include "lfbm5d_graph.h"

namespace lfbm5d_host {

using plan::search_window;
using plan::plan_windows;

/* bm5d.cpp:165-407 (step 1) / :861-1106 (step 2) on device-resident buffers */
int run_step(lfbm5d_ctx* c, int step, const lfbm5d_params* P, float* d_noisy, const unsigned* h_mask,
             float* d_basic, float* d_out, unsigned ang_major, unsigned awidth, unsigned aheight,
             unsigned an, unsigned W, unsigned H, unsigned C, const HostIO* io) {
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
static int run_denoise_whole(lfbm5d_ctx* c, const lfbm5d_params* P1, const lfbm5d_params* P2, float* d_noisy, const unsigned* h_mask, float* d_basic,
                             float* d_out, unsigned ang_major, unsigned awidth, unsigned aheight, unsigned an1, unsigned an2, unsigned W, unsigned H,
                             unsigned C, const HostIO* io) {
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

/* ------------------------------------------------------------------------------------------
 * Spatial bands (round 6, option spatial_bands = S): the ranks form S teams, team b denoises the horizontal BAND b of every SAI --
 * its H / S rows plus a halo (option band_halo; default nSim + nDisp + k of the wider step) -- as a complete two-step job of its
 * own on the window graph of its world / S ranks, and the bands' interiors are stitched.  The reference's tile mode
 * (bm5d.cpp:411-708, undivide_LF) is the same idea with OpenMP tiles; here a band is a whole-width strip, there are few of them,
 * and inside a band everything is the untiled algorithm.
 *
 * Why: the window graph alone stops at the light field's dependency chains (8 ranks: 4.2x on 17x17 SAIs, 3.2x on 15x15, 2.0x on
 * 9x9: tools/scale_model.py) and a window pass cannot be shared exactly -- the table kernel's recurrence is serial in rows
 * (DESIGN.md section 7).  A band's tables start their recurrence at the band's first row, so its distances differ from the
 * whole image's in the last bits, a fraction of a per cent of the matches differ, and pixels decorrelate exactly as they do
 * between this library and the CPU oracle (profiles/r06_d_near_tie_tail.txt): the result is NOT bit-identical to one GPU.  What
 * was measured instead (tools/band_accuracy.py, profiles/r06_i_band_accuracy.txt): the stitched light field's PSNR is within
 * 1e-3 dB of the whole-field job's for halos of 40 ... 96 rows -- a tenth of BASELINE.json's tolerance.  Off by default.
 * ------------------------------------------------------------------------------------------ */
struct BandRows { unsigned y0, y1, c0, c1; };   /* the band's own rows [y0, y1), the rows it is computed on [c0, c1) */
static BandRows band_rows(unsigned H, int S, int b, unsigned halo) {
    BandRows r;
    r.y0 = (unsigned)((unsigned long long)H * (unsigned)b / (unsigned)S); r.y1 = (unsigned)((unsigned long long)H * (unsigned)(b + 1) / (unsigned)S);
    r.c0 = r.y0 > halo ? r.y0 - halo : 0u; r.c1 = std::min(H, r.y1 + halo);
    return r;
}

static int run_denoise_banded(lfbm5d_ctx* c, const lfbm5d_params* P1, const lfbm5d_params* P2, float* d_noisy, const unsigned* h_mask, float* d_basic,
                              float* d_out, unsigned ang_major, unsigned awidth, unsigned aheight, unsigned an1, unsigned an2, unsigned W, unsigned H,
                              unsigned C, int S) {
    const unsigned asize = awidth * aheight, planes = asize * C;
    const int emu = c->opt->emulate_world, nranks = emu > 1 ? emu : c->world, T = nranks / S;
    const unsigned halo = c->opt->band_halo > 0 ? (unsigned)c->opt->band_halo
                                                : std::max(P1->nSim + P1->nDisp + P1->k, P2->nSim + P2->nDisp + P2->k);
    hipStream_t s = c->stream;
    unsigned Hc_max = 0;
    for (int b = 0; b < S; b++) { const BandRows r = band_rows(H, S, b, halo); Hc_max = std::max(Hc_max, r.c1 - r.c0); }
    const size_t crop_bytes = (size_t)planes * Hc_max * W * sizeof(float);
    HIPCK(c, c->band_noisy.reserve(crop_bytes)); HIPCK(c, c->band_basic.reserve(crop_bytes)); HIPCK(c, c->band_out.reserve(crop_bytes));
    float* const bn = c->band_noisy.as<float>(); float* const bb = c->band_basic.as<float>(); float* const bo = c->band_out.as<float>();
    auto job_on_band = [&](const float* src_noisy, const BandRows& r) -> int {
        const unsigned Hc = r.c1 - r.c0;
        HIPCK(c, launch_copy_rows(s, src_noisy, H, r.c0, bn, Hc, 0, Hc, W, planes));
        return run_denoise_whole(c, P1, P2, bn, h_mask, bb, bo, ang_major, awidth, aheight, an1, an2, W, Hc, C, nullptr);
    };
    auto interior_back = [&](const BandRows& r) -> int {   /* the band's own rows of the three outputs into the caller's light fields */
        const unsigned Hc = r.c1 - r.c0, n = r.y1 - r.y0, o = r.y0 - r.c0;
        HIPCK(c, launch_copy_rows(s, bn, Hc, o, d_noisy, H, r.y0, n, W, planes));
        HIPCK(c, launch_copy_rows(s, bb, Hc, o, d_basic, H, r.y0, n, W, planes));
        HIPCK(c, launch_copy_rows(s, bo, Hc, o, d_out, H, r.y0, n, W, planes));
        return 0;
    };
    /* the stitch: every member of a team holds its band's whole result, so member t contributes the t-th share of the band's rows --
     * equal chunks of chunk_rows rows (the last of a band may be shorter), every byte sent once */
    const unsigned rows_band_max = (H + (unsigned)S - 1) / (unsigned)S, chunk_rows = (rows_band_max + (unsigned)T - 1) / (unsigned)T;
    const size_t chunk_floats = (size_t)3 * planes * chunk_rows * W;
    auto chunk_of = [&](int rk, unsigned& ya, unsigned& n) {
        const BandRows q = band_rows(H, S, rk / T, halo);
        ya = std::min(q.y1, q.y0 + (unsigned)(rk % T) * chunk_rows);
        n = std::min(q.y1, ya + chunk_rows) - ya;
    };
    float* const lf[3] = {d_noisy, d_basic, d_out};
    auto pack_chunk = [&](int rk, float* dst) -> int {     /* rank rk's share of the three light fields -> dst[3][planes][chunk_rows][W] */
        unsigned ya, n; chunk_of(rk, ya, n);
        for (int i = 0; i < 3; i++) HIPCK(c, launch_copy_rows(s, lf[i], H, ya, dst + (size_t)i * planes * chunk_rows * W, chunk_rows, 0, n, W, planes));
        return 0;
    };
    auto unpack_chunk = [&](int rk, const float* src) -> int {
        unsigned ya, n; chunk_of(rk, ya, n);
        for (int i = 0; i < 3; i++) HIPCK(c, launch_copy_rows(s, src + (size_t)i * planes * chunk_rows * W, chunk_rows, 0, lf[i], H, ya, n, W, planes));
        return 0;
    };
    if (emu > 1) {
        /* every rank played on this GPU: band after band, each as a job of T emulated ranks; the bands are cut from the light field
         * as it arrived (a band's halo lies in its neighbour's rows, which the neighbour's outputs overwrite).  The stitch runs as
         * between real ranks -- every rank's chunk packed into the gather buffer, the outputs cleared, every chunk unpacked -- so
         * that the emulation covers the chunk arithmetic, not only the band jobs. */
        HIPCK(c, c->band_src.reserve((size_t)planes * H * W * sizeof(float)));
        HIPCK(c, hipMemcpyAsync(c->band_src.p, d_noisy, (size_t)planes * H * W * sizeof(float), hipMemcpyDeviceToDevice, s));
        HIPCK(c, c->band_gather.reserve(chunk_floats * sizeof(float) * (size_t)nranks));
        struct EmuScope { Options* o; int saved; ~EmuScope() { o->emulate_world = saved; } } scope{c->opt, c->opt->emulate_world};
        c->opt->emulate_world = T > 1 ? T : 0;
        for (int b = 0; b < S; b++) {
            const BandRows r = band_rows(H, S, b, halo);
            if (job_on_band(c->band_src.as<float>(), r)) return 1;
            if (interior_back(r)) return 1;
            for (int t = 0; t < T; t++)
                if (pack_chunk(b * T + t, c->band_gather.as<float>() + (size_t)(b * T + t) * chunk_floats)) return 1;
        }
        for (int i = 0; i < 3; i++) HIPCK(c, hipMemsetAsync(lf[i], 0, (size_t)planes * H * W * sizeof(float), s));
        for (int rk = 0; rk < nranks; rk++)
            if (unpack_chunk(rk, c->band_gather.as<float>() + (size_t)rk * chunk_floats)) return 1;
        HIPCK(c, hipStreamSynchronize(s));
        return 0;
    }
    /* real ranks: rank = band * T + team rank.  The team gets communicators of its own (split once, kept), the job runs on them,
     * then ONE all-gather over all ranks stitches the light fields: every member of a team holds its band's whole result, so
     * member t contributes the t-th share of the band's rows -- equal chunks, every byte sent once. */
    const int band = c->rank / T, trank = c->rank % T;
    if (c->ipc) {
        /* processes on one GPU (the IPC test transport, lfbm5d_comm_init_ipc): the same teams, the same stitch.  A team's job runs
         * under the team's numbering with rendezvous names of its own ("b<band>."); for the stitch every rank publishes the IPC
         * handle of its packed chunk, reads the chunks of the other teams straight out of their owners' buffers, and all ranks
         * leave together (nobody packs the next job's chunk over one that is still being read). */
        const BandRows r = band_rows(H, S, band, halo);
        c->ipc_epoch += 1;   /* names of this job's stitch; a team of one rank runs no exchange of its own that would advance it */
        const unsigned stitch_epoch = c->ipc_epoch;
        int rc;
        {
            struct IpcTeamScope {
                lfbm5d_ctx* c; int rank, world; std::vector<lfbm5d_ctx::IpcPeer> peers;
                ~IpcTeamScope() {
                    for (lfbm5d_ctx::IpcPeer& P : c->ipc_peers) for (void*& q : P.ptr) if (q) { (void)hipIpcCloseMemHandle(q); q = nullptr; }
                    c->ipc_peers = std::move(peers); c->rank = rank; c->world = world; c->ipc_tag.clear();
                }
            } scope{c, c->rank, c->world, std::move(c->ipc_peers)};
            c->ipc_peers.clear();
            c->rank = trank; c->world = T; c->ipc_tag = "b" + std::to_string(band) + ".";
            rc = job_on_band(d_noisy, r);
        }
        if (rc) return 1;   /* (run_graph has closed the transport: the other teams end in their watchdogs) */
        if (interior_back(r)) return 1;
        HIPCK(c, c->band_pack.reserve(chunk_floats * sizeof(float)));
        if (pack_chunk(c->rank, c->band_pack.as<float>())) return 1;
        HIPCK(c, hipStreamSynchronize(s));
        hipIpcMemHandle_t mine;
        HIPCK(c, hipIpcGetMemHandle(&mine, c->band_pack.p));
        const std::string base = "bandpack." + std::to_string(stitch_epoch) + ".";
        if (!ipc_put(c->ipc_dir, base + std::to_string(c->rank), &mine, sizeof(mine))) return fail(c, "ipc transport: cannot write to the rendezvous directory");
        std::vector<void*> opened;
        int bad = 0;
        for (int rk = 0; rk < nranks && !bad; rk++) {
            if (rk / T == band) continue;   /* this team's rows are in place */
            hipIpcMemHandle_t h;
            if (!ipc_get(c->ipc_dir, base + std::to_string(rk), &h, sizeof(h), c->ipc_timeout_s)) { bad = 1; break; }
            void* q = nullptr;
            if (hipIpcOpenMemHandle(&q, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess) { bad = 2; break; }
            opened.push_back(q);
            if (unpack_chunk(rk, reinterpret_cast<const float*>(q))) { bad = 3; break; }
        }
        (void)hipStreamSynchronize(s);
        for (void* q : opened) (void)hipIpcCloseMemHandle(q);
        if (bad == 1) return fail(c, "ipc transport: a rank did not publish its band within the watchdog (peer gone?)");
        if (bad == 2) return fail(c, "ipc transport: hipIpcOpenMemHandle of a peer's band failed");
        if (bad) return 1;
        std::vector<int> all;
        const unsigned keep = c->ipc_epoch; c->ipc_epoch = stitch_epoch;          /* (the teams' jobs may have advanced their epochs differently) */
        const int rs = ipc_allgather(c, "stitched", 1, all);
        (void)keep; c->ipc_epoch = stitch_epoch + 8;   /* the same on every rank whatever its team's job used (a redo takes more) */
        return rs;
    }
    if (!c->comm) return fail(c, "whole steps on several ranks need lfbm5d_comm_init");
    if (c->team_S != S) {
        if (c->team_comm2) { (void)ncclCommDestroy(c->team_comm2); c->team_comm2 = nullptr; }
        if (c->team_comm) { (void)ncclCommDestroy(c->team_comm); c->team_comm = nullptr; }
        if (ncclCommSplit(c->comm, band, trank, &c->team_comm, nullptr) != ncclSuccess) return fail(c, "ncclCommSplit (band team) failed");
        if (T > 1 && ncclCommSplit(c->team_comm, 0, trank, &c->team_comm2, nullptr) != ncclSuccess) c->team_comm2 = nullptr;
        c->team_S = S;
    }
    const BandRows r = band_rows(H, S, band, halo);
    int rc;
    {
        struct TeamScope {   /* the team's communicators and numbering stand in for the context's while its job runs */
            lfbm5d_ctx* c; ncclComm_t g1, g2; int rank, world;
            ~TeamScope() {
                /* run_graph aborts the communicators it ran on when it fails: those were the team's */
                if (!c->comm) { c->team_comm = nullptr; c->team_comm2 = nullptr; c->team_S = 0; } else { c->team_comm = c->comm; c->team_comm2 = c->comm2; }
                c->comm = g1; c->comm2 = g2; c->rank = rank; c->world = world;
            }
        } scope{c, c->comm, c->comm2, c->rank, c->world};
        c->comm = c->team_comm; c->comm2 = c->team_comm2; c->rank = trank; c->world = T;
        rc = job_on_band(d_noisy, r);
    }
    if (rc) {   /* the other teams wait in the all-gather below: take the whole job down, like run_graph does for its exchange */
        if (c->comm2) { (void)ncclCommAbort(c->comm2); c->comm2 = nullptr; }
        if (c->comm) { (void)ncclCommAbort(c->comm); c->comm = nullptr; }
        c->team_S = 0;
        c->err += " (banded job aborted: the RCCL communicators of this context were torn down, call lfbm5d_comm_init again)";
        return 1;
    }
    if (interior_back(r)) return 1;
    HIPCK(c, c->band_pack.reserve(chunk_floats * sizeof(float)));
    HIPCK(c, c->band_gather.reserve(chunk_floats * sizeof(float) * (size_t)nranks));
    if (pack_chunk(c->rank, c->band_pack.as<float>())) return 1;
    hipEvent_t e0 = get_event(c), e1 = get_event(c);
    HIPCK(c, hipEventRecord(e0, s));
    if (ncclAllGather(c->band_pack.p, c->band_gather.p, chunk_floats, ncclFloat, c->comm, s) != ncclSuccess) return fail(c, "ncclAllGather of the bands failed");
    HIPCK(c, hipEventRecord(e1, s));
    for (int rk = 0; rk < nranks; rk++) {
        if (rk / T == band) continue;   /* this team's rows are in place */
        if (unpack_chunk(rk, c->band_gather.as<float>() + (size_t)rk * chunk_floats)) return 1;
    }
    HIPCK(c, hipStreamSynchronize(s));
    float ms = 0.0f;
    if (hipEventElapsedTime(&ms, e0, e1) == hipSuccess) c->stats.ms_comm += ms;
    return 0;
}

/* the band count the scale model picks (tools/scale_model.py; include/lfbm5d.h lfbm5d_auto_bands) */
int auto_bands(unsigned awidth, unsigned aheight, unsigned height, unsigned halo, int world) {
    if (world < 1) return 1;
    const unsigned a = std::min(awidth, aheight);
    int t_max = 1;
    while (t_max * 2 <= 0.8 * (double)((a + 2) / 3)) t_max *= 2;
    int s = std::max(1, world / std::min(world, t_max));
    while (s > 1 && (world % s || height / (unsigned)s < 2 * halo)) s--;
    return s;
}

/* lfbm5d_denoise_*: the whole light field as one job (the default), or band by band (option spatial_bands) */
int run_denoise(lfbm5d_ctx* c, const lfbm5d_params* P1, const lfbm5d_params* P2, float* d_noisy, const unsigned* h_mask, float* d_basic,
                float* d_out, unsigned ang_major, unsigned awidth, unsigned aheight, unsigned an1, unsigned an2, unsigned W, unsigned H,
                unsigned C, const HostIO* io) {
    const int emu = c->opt->emulate_world, nranks = emu > 1 ? emu : c->world;
    const unsigned halo0 = c->opt->band_halo > 0 ? (unsigned)c->opt->band_halo : std::max(P1->nSim + P1->nDisp + P1->k, P2->nSim + P2->nDisp + P2->k);
    const int S = c->opt->spatial_bands == 0 ? auto_bands(awidth, aheight, H, halo0, nranks) : c->opt->spatial_bands;   /* 0: the rule */
    if (S <= 1 || nranks <= 1) return run_denoise_whole(c, P1, P2, d_noisy, h_mask, d_basic, d_out, ang_major, awidth, aheight, an1, an2, W, H, C, io);
    if (nranks % S) return fail(c, "spatial_bands must divide the number of ranks");
    if (H / (unsigned)S < std::max(halo0, 2 * std::max(P1->k, P2->k))) return fail(c, "spatial_bands: the bands would be narrower than their halo");
    if (c->tiles > 1) return fail(c, "spatial bands and the tile mode exclude each other");
    const unsigned asize = awidth * aheight;
    const size_t img = (size_t)C * W * H;
    if (io && io_upload_all(c, io, h_mask, asize, img, d_noisy, nullptr)) return 1;
    if (run_denoise_banded(c, P1, P2, d_noisy, h_mask, d_basic, d_out, ang_major, awidth, aheight, an1, an2, W, H, C, S)) return 1;
    return io ? io_download_all(c, io, h_mask, asize, img, d_noisy, d_basic, d_out) : 0;
}

} /* namespace lfbm5d_host */
