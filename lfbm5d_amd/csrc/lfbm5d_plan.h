/*
 * lfbm5d_plan.h -- the window schedule of run_bm5d_1st_step / run_bm5d_2nd_step (bm5d.cpp:165-407, :861-1106) as a
 * dependency graph, and its partition over GPUs.  Host code only (no HIP): what lfbm5d_plan_windows, lfbm5d_plan_graph,
 * lfbm5d_plan_messages and lfbm5d_plan_job expose, and what the step engine of lfbm5d_api.hip executes.
 *
 * A JOB is one step or both steps of a denoise.  Its nodes are the angular windows of the steps' planned sequences.
 * Two windows of a step interact only through num / den of the SAIs they share (the running estimate block matching
 * reads, the sums aggregation adds to): a window waits exactly for the previous window of its step that touched each of
 * its SAIs.  In a two-step job a window of the second step additionally waits, per SAI, for the LAST window of the
 * first step that touches that SAI -- the SAI's basic estimate is final then (bm5d.cpp:405) -- so the second step's
 * wavefront follows the first's instead of waiting for the whole first step.
 *
 * Partition: every window has an owner rank (chosen along a simulated execution: a window follows the rank of its chain --
 * the windows of its row of SAIs -- while that rank is free).  Whatever a window needs from a window of another rank
 * travels as messages: num and den of a shared SAI between consecutive
 * touchers, the basic estimate of a SAI from the rank that finalised it to the ranks whose second-step windows read it.
 * Every rank walks the nodes in the same ISSUE ORDER -- the start order of a simulated execution, a topological order of
 * the whole graph -- and enqueues its windows, sends and receives in that order on FIFO streams, which is what makes the
 * exchange deadlock-free (tests/test_dist_cpu.py replays it).
 */
#ifndef LFBM5D_PLAN_H
#define LFBM5D_PLAN_H

#include <algorithm>
#include <vector>

namespace lfbm5d {
namespace plan {

constexpr unsigned kRowMajor = 11;   /* LFBM5D_ROWMAJOR */

/* utilities_LF.cpp:881-901 */
inline void search_window(int aidx, unsigned asize, unsigned an, int& cc, int& mn, int& mx) {
    mn = aidx - (int)an; mx = aidx + (int)an;
    int shift = mn < 0 ? -mn : 0;
    mn += shift; mx += shift; cc = (int)an - shift;
    shift = mx >= (int)asize ? ((int)asize - mx - 1) : 0;
    mn += shift; mx += shift; cc -= shift;
}

/* The sequence of windows a step processes, as the processed SAI of each: first the centre SAI if it is not empty, then
 * always the last unprocessed SAI; every SAI of a window is processed when the window is done (bm5d.cpp:165-402 -- all
 * candidates of the reference's arg-max tie, see lfbm5d_plan_windows in include/lfbm5d.h). */
inline void plan_windows(const unsigned* h_mask, unsigned awidth, unsigned aheight, unsigned an, unsigned ang_major,
                         std::vector<unsigned>& out) {
    const unsigned asize = awidth * aheight, asw = 2 * an + 1;
    const unsigned cs = aheight / 2, ct = awidth / 2;
    const unsigned cst = ang_major == kRowMajor ? cs * awidth + ct : cs + ct * aheight;
    std::vector<unsigned> proc(asize);
    for (unsigned st = 0; st < asize; st++) proc[st] = !h_mask[st];
    unsigned remaining = (unsigned)std::count(proc.begin(), proc.end(), 0u);
    const unsigned total = remaining;
    out.clear();
    while (remaining) {
        unsigned pst = 0;
        if (remaining == total && h_mask[cst]) pst = cst;
        else for (unsigned st = 0; st < asize; st++) if (!proc[st]) pst = st;
        const unsigned ps = ang_major == kRowMajor ? pst / awidth : pst % aheight;
        const unsigned pt = ang_major == kRowMajor ? pst % awidth : pst / aheight;
        int cs_w, mins, maxs, ct_w, mint, maxt;
        search_window((int)ps, aheight, an, cs_w, mins, maxs);
        search_window((int)pt, awidth, an, ct_w, mint, maxt);
        for (unsigned si = 0; si < asw; si++)
            for (unsigned ti = 0; ti < asw; ti++) {
                const unsigned S = si + mins, T = ti + mint;
                const unsigned st = ang_major == kRowMajor ? S * awidth + T : S + T * aheight;
                if (h_mask[st]) proc[st] = 1;
            }
        out.push_back(pst);
        remaining = (unsigned)std::count(proc.begin(), proc.end(), 0u);
    }
}

struct StepDesc {
    unsigned an;        /* half size of the angular search window */
    unsigned tau4;      /* tau_4D at entry (switches from DCT to SADCT for good at the first window with an empty SAI, bm5d.cpp:276-280) */
    unsigned cost;      /* relative cost of one window pass of this step (scheduling model only) */
};

struct Node {
    int s = 0;                       /* step slot of the job: 0 | 1 */
    unsigned w = 0;                  /* index in the step's planned sequence */
    unsigned pst = 0, ps = 0, pt = 0, tau4 = 0, chain = 0, cost = 1;
    std::vector<unsigned> sai;       /* non-empty SAIs (light-field indices) of the window */
    std::vector<int> prev, next;     /* per SAI: previous / next node of the same step touching it (-1: none) */
    std::vector<unsigned> fin;       /* two-step jobs, first step: SAIs whose sums are final once this window is done */
    std::vector<int> deps;           /* every node this one waits for (prev + the finalising nodes of its SAIs) */
    int rank = 0, lane = 0;          /* owner (graph rank), lane within the owner */
    unsigned start = 0;              /* start time of the simulated execution, in `cost` units */
};

/* kind 0: num and den of `sai` (of step slot nodes[from].s) from the rank of node `from` to the rank of node `to_node`, its
 * next toucher; kind 1: the basic estimate of `sai`, finalised behind node `from`, to graph rank `to_rank`. */
struct Xfer { int kind; unsigned from; int to_node; int to_rank; unsigned sai; int channel; };

struct Graph {
    std::vector<Node> nodes;              /* step slot 0's windows in plan order, then step slot 1's */
    unsigned n_first = 0;                 /* nodes of step slot 0 */
    std::vector<unsigned> order;          /* issue order */
    std::vector<Xfer> xfers;              /* in issue order: position of the producer node, then SAI slot, then rank */
    std::vector<int> last_touch[2];       /* per step slot and SAI of the light field: last node touching it (-1: none) */
    bool centre_ok = true;                /* every window's centre SAI is non-empty */
    unsigned makespan = 0;                /* of the simulated execution (lanes as parallel servers) */
};

inline void build(const unsigned* h_mask, unsigned awidth, unsigned aheight, unsigned ang_major, const StepDesc* steps, int n_steps,
                  int world, int n_lanes, int max_windows, Graph& G) {
    const unsigned asize = awidth * aheight;
    const int Gn = std::max(1, world);
    n_lanes = std::max(1, n_lanes);
    G.nodes.clear(); G.order.clear(); G.xfers.clear(); G.centre_ok = true; G.n_first = 0; G.makespan = 0;
    unsigned chain = 0;
    for (int s = 0; s < n_steps; s++) {
        const unsigned an = steps[s].an, asw = 2 * an + 1, Aw = asw * asw;
        std::vector<unsigned> pl;
        plan_windows(h_mask, awidth, aheight, an, ang_major, pl);
        if (max_windows > 0 && pl.size() > (size_t)max_windows) pl.resize((size_t)max_windows);
        G.last_touch[s].assign(asize, -1);
        unsigned t4 = steps[s].tau4;
        const size_t base = G.nodes.size();
        for (size_t w = 0; w < pl.size(); w++) {
            Node nd;
            nd.s = s; nd.w = (unsigned)w; nd.pst = pl[w]; nd.cost = std::max(1u, steps[s].cost);
            nd.ps = ang_major == kRowMajor ? pl[w] / awidth : pl[w] % aheight;
            nd.pt = ang_major == kRowMajor ? pl[w] % awidth : pl[w] / aheight;
            int cs_w, mins, maxs, ct_w, mint, maxt;
            search_window((int)nd.ps, aheight, an, cs_w, mins, maxs);
            search_window((int)nd.pt, awidth, an, ct_w, mint, maxt);
            for (unsigned si = 0; si < asw; si++)
                for (unsigned ti = 0; ti < asw; ti++) {
                    const unsigned st = ang_major == kRowMajor ? (si + mins) * awidth + (ti + mint) : (si + mins) + (ti + mint) * aheight;
                    if (h_mask[st]) nd.sai.push_back(st);
                }
            if (nd.sai.size() != Aw && t4 == 5u /* LFBM5D_DCT */) t4 = 6u /* LFBM5D_SADCT */;
            nd.tau4 = t4;
            const unsigned cst_lf = ang_major == kRowMajor ? (unsigned)(mins + cs_w) * awidth + (unsigned)(mint + ct_w)
                                                           : (unsigned)(mins + cs_w) + (unsigned)(mint + ct_w) * aheight;
            if (!h_mask[cst_lf]) G.centre_ok = false;
            if (w > 0 && nd.ps != G.nodes.back().ps) chain++;
            else if (w == 0 && base > 0) chain++;
            nd.chain = chain;
            nd.prev.assign(nd.sai.size(), -1);
            nd.next.assign(nd.sai.size(), -1);
            const int me = (int)G.nodes.size();
            for (size_t i = 0; i < nd.sai.size(); i++) {
                const unsigned st = nd.sai[i];
                const int p = G.last_touch[s][st];
                nd.prev[i] = p;
                if (p >= 0) {
                    Node& pn = G.nodes[(size_t)p];
                    const size_t j = (size_t)(std::find(pn.sai.begin(), pn.sai.end(), st) - pn.sai.begin());
                    pn.next[j] = me;
                    if (std::find(nd.deps.begin(), nd.deps.end(), p) == nd.deps.end()) nd.deps.push_back(p);
                }
                G.last_touch[s][st] = me;
                if (s == 1) {   /* the SAI's basic estimate: final behind the first step's last window on it */
                    const int f = G.last_touch[0][st];
                    if (f >= 0 && std::find(nd.deps.begin(), nd.deps.end(), f) == nd.deps.end()) nd.deps.push_back(f);
                }
            }
            G.nodes.push_back(nd);
        }
        if (s == 0) G.n_first = (unsigned)G.nodes.size();
    }
    if (n_steps < 2) G.last_touch[1].assign(asize, -1);
    const size_t NN = G.nodes.size();
    if (n_steps == 2)
        for (unsigned st = 0; st < asize; st++) {
            const int f = G.last_touch[0][st];
            if (f >= 0) G.nodes[(size_t)f].fin.push_back(st);
        }

    /* Owner of every window, decided along a simulated execution with one server per rank: again and again the window that can
     * start first (ties: the lower index) goes to the rank on which it starts first; among equally early ranks to the one that
     * owns most of the window's predecessors -- one point per SAI whose sums, or basic estimate, would not have to travel -- then
     * to the lowest rank.  A window therefore follows its chain (the run of windows in its row of SAIs, each of which shares a
     * column of SAIs with the one before) while that rank is free and moves when it is not; round 4 measured this against
     * whole chains per rank (rounds 2-3): 4.76 against 4.19 of the serial time on a 17x17 light field at eight ranks, the
     * critical path of the two-step graph, with fewer cross-rank edges (lfbm5d_plan_job, tools/scale_model.py).  A pure function
     * of the mask, the steps and the rank count: every rank computes the same. */
    {
        std::vector<unsigned> fin(NN, 0), rank_free((size_t)Gn, 0);
        std::vector<char> done(NN, 0);
        for (size_t it = 0; it < NN; it++) {
            size_t pick = NN; unsigned pick_t = ~0u; int pick_r = 0;
            for (size_t n = 0; n < NN; n++) {
                if (done[n]) continue;
                const Node& nd = G.nodes[n];
                unsigned ready = 0; bool ok = true;
                for (int p : nd.deps) { if (!done[(size_t)p]) { ok = false; break; } ready = std::max(ready, fin[(size_t)p]); }
                if (!ok) continue;
                int r = 0; unsigned t = ~0u; int best_aff = -1;
                for (int q = 0; q < Gn; q++) {
                    const unsigned tq = std::max(ready, rank_free[(size_t)q]);
                    if (tq > t) continue;
                    int aff = 0;
                    for (int p : nd.prev) if (p >= 0 && G.nodes[(size_t)p].rank == q) aff++;
                    if (nd.s == 1) for (unsigned st : nd.sai) { const int f = G.last_touch[0][st]; if (f >= 0 && G.nodes[(size_t)f].rank == q) aff++; }
                    if (tq < t || aff > best_aff) { t = tq; best_aff = aff; r = q; }
                }
                if (t < pick_t) { pick = n; pick_t = t; pick_r = r; }
            }
            Node& nd = G.nodes[pick];
            nd.rank = pick_r; fin[pick] = pick_t + nd.cost; rank_free[(size_t)pick_r] = fin[pick]; done[pick] = 1;
        }
    }
    /* Simulated execution, the lanes of a rank as parallel servers: again and again the node that can start first (its
     * dependencies done, a lane of its owner free), ties to the lower index; among equally early lanes the lane of its
     * latest dependency (no event needed).  The sequence of starts is the issue order. */
    {
        std::vector<unsigned> finish(NN, 0), lane_free((size_t)Gn * (size_t)n_lanes, 0);
        std::vector<char> done(NN, 0);
        for (size_t it = 0; it < NN; it++) {
            size_t pick = NN; unsigned pick_t = ~0u; int pick_l = 0;
            for (size_t n = 0; n < NN; n++) {
                if (done[n]) continue;
                const Node& nd = G.nodes[n];
                unsigned ready = 0; int pref = -1; bool ok = true;
                for (int p : nd.deps) {
                    if (!done[(size_t)p]) { ok = false; break; }
                    if (finish[(size_t)p] >= ready) { ready = finish[(size_t)p]; pref = G.nodes[(size_t)p].rank == nd.rank ? G.nodes[(size_t)p].lane : -1; }
                }
                if (!ok) continue;
                int best_l = 0; unsigned best_t = ~0u;
                for (int l = 0; l < n_lanes; l++) {
                    const unsigned t = std::max(ready, lane_free[(size_t)nd.rank * n_lanes + l]);
                    if (t < best_t || (t == best_t && l == pref)) { best_t = t; best_l = l; }
                }
                if (best_t < pick_t) { pick = n; pick_t = best_t; pick_l = best_l; }
            }
            Node& nd = G.nodes[pick];
            nd.lane = pick_l; nd.start = pick_t; finish[pick] = pick_t + nd.cost; done[pick] = 1;
            lane_free[(size_t)nd.rank * n_lanes + pick_l] = finish[pick];
            G.makespan = std::max(G.makespan, finish[pick]);
            G.order.push_back((unsigned)pick);
        }
    }
    /* Messages, in the order every rank issues them: by the producer's place in the issue order, then SAI slot (sums), then the
     * SAIs it finalises and their readers in rank order (basic estimates).  Two channels (communicator + stream) alternate with
     * the producer's chain: a rank receives its inputs from the chain before its own on one channel and sends its outputs on the
     * other, so a send that is ready early never queues behind a receive that completes late. */
    for (unsigned n : G.order) {
        const Node& nd = G.nodes[n];
        const int ch = (int)(nd.chain & 1u);
        for (size_t i = 0; i < nd.sai.size(); i++) {
            const int nx = nd.next[i];
            if (nx >= 0 && G.nodes[(size_t)nx].rank != nd.rank) G.xfers.push_back({0, n, nx, G.nodes[(size_t)nx].rank, nd.sai[i], ch});
        }
        for (unsigned st : nd.fin) {
            std::vector<int> readers;
            for (size_t m = G.n_first; m < NN; m++) {
                const Node& rd = G.nodes[m];
                if (rd.rank != nd.rank && std::find(rd.sai.begin(), rd.sai.end(), st) != rd.sai.end() &&
                    std::find(readers.begin(), readers.end(), rd.rank) == readers.end()) readers.push_back(rd.rank);
            }
            std::sort(readers.begin(), readers.end());
            for (int r : readers) G.xfers.push_back({1, n, -1, r, st, ch});
        }
    }
}

} /* namespace plan */
} /* namespace lfbm5d */
#endif
