/*
 * lfbm5d_graph.h -- a JOB of the window graph (run_graph, lfbm5d_graph.hip) as the step functions hand it over.  Internal.
 */
#ifndef LFBM5D_GRAPH_H
#define LFBM5D_GRAPH_H

#include "lfbm5d_ctx.h"

namespace lfbm5d_host {

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
              unsigned ang_major, unsigned W, unsigned H, unsigned C, int nranks, bool emulate, int* complete_out);

} /* namespace lfbm5d_host */
#endif
