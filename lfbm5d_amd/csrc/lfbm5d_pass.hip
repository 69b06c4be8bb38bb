/*
 * lfbm5d_pass.hip -- one core pass (bm5d_1st_step / bm5d_2nd_step, core:90-822 / :859-1659) as a sequence of HIP kernels on the
 * context's stream: reference grid and transform tables of the geometry (cached), matching estimate, distance tables, selection,
 * group stage, aggregation; the context's events and counters.  Split from lfbm5d_api.hip in round 6 (lfbm5d_ctx.h).
 */
#include "lfbm5d_ctx.h"

namespace lfbm5d_host {

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

int validate(lfbm5d_ctx* c, int step, const lfbm5d_params* P, unsigned aw, unsigned ah, unsigned C, bool bm3d) {
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
              bool bm3d) {
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
    /* wide windows: filt SAI-major (kernels.h filt_patch) -- per SAI the launch's groups, [g][n][c][k2] */
    const bool sai_major = A >= kSaiMajorMinA && !(c->opt->kernels & kOptFiltGroupMajor);
    const size_t per_group_bias = sai_major ? per_group / A : per_group;   /* what one group takes in front of the patch the kernels address */
    if (n_groups <= band_groups) {   /* the whole pass (or this rank's rows) at once */
        ga.filt_sai_stride = aa.filt_sai_stride = sai_major ? (unsigned long long)n_groups * (per_group / A) : 0ull;
        ga.filt = c->filt.as<float>() - (size_t)ref_begin * per_group_bias;   /* (the group kernels index filt by absolute group number) */
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
            ga.filt_sai_stride = aa.filt_sai_stride = sai_major ? (unsigned long long)nb * (per_group / A) : 0ull;
            ga.ref_begin = b0; ga.n_groups = nb; ga.filt = c->filt.as<float>() - (size_t)b0 * per_group_bias;
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
int fold_counters(lfbm5d_ctx* c, const lfbm5d_params* P, unsigned A, unsigned C, int step, int slot) {
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

} /* namespace lfbm5d_host */
