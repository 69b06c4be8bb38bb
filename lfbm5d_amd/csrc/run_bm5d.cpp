/*
 * run_bm5d.cpp -- the reference's outer seam (src/bm5d.h:11-62) on top of the C-ABI.
 * Host side only: hands the library one pointer per SAI -- the vectors' own storage, no flat copy -- through
 * lfbm5d_step{1,2}_host_sai / lfbm5d_denoise_host_sai, which stream the SAIs through HBM as the window graph needs and
 * finishes them and run every kernel on the GPU (round 5; rounds 1-4 flattened, copied four light fields and unflattened).
 * Error behaviour follows the reference: message on stdout, EXIT_FAILURE.
 */
#include "run_bm5d.h"

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <memory>
#include <thread>

#include "../../include/lfbm5d.h"

namespace {

lfbm5d_ctx* context() {
    static lfbm5d_ctx* ctx = nullptr;
    if (!ctx) {
        const char* dev = std::getenv("LFBM5D_DEVICE");
        if (lfbm5d_create(&ctx, dev ? std::atoi(dev) : 0) != 0) {
            std::cout << "LFBM5D GPU backend: " << lfbm5d_last_error(nullptr) << std::endl;
            ctx = nullptr;
        }
    }
    return ctx;
}

/* the SAIs of the reference's vector-of-vectors light field <-> one flat buffer, by a few threads: at 17 x 17 x 512 x 512 x 3 a light
 * field is 0.9 GB and a step moves three to five of them -- one thread's memcpy of that is longer than the step on the GPU */
template <class F> void for_sais(size_t n, F fn) {
    const unsigned nt = (unsigned)std::min<size_t>(std::min(8u, std::max(1u, std::thread::hardware_concurrency())), std::max<size_t>(1, n));
    std::atomic<size_t> next(0);
    auto work = [&]() { for (size_t i = next++; i < n; i = next++) fn(i); };
    std::vector<std::thread> th;
    for (unsigned i = 1; i < nt; i++) th.emplace_back(work);
    work();
    for (auto& x : th) x.join();
}
/* flat buffers: uninitialised storage (every byte is written by flatten or by the library's device-to-host copy) */
typedef std::unique_ptr<float[]> Flat;
Flat flat_alloc(size_t n) { return Flat(new float[n]); }
void flatten(const std::vector<std::vector<float> >& LF, const std::vector<unsigned>& mask, size_t img, Flat& flat) {
    flat = flat_alloc(LF.size() * img);
    for_sais(LF.size(), [&](size_t st) {
        if (mask[st] && LF[st].size() == img) std::memcpy(&flat[st * img], LF[st].data(), img * sizeof(float));
        else std::memset(&flat[st * img], 0, img * sizeof(float));
    });
}
void unflatten(std::vector<std::vector<float> >& LF, const std::vector<unsigned>& mask, size_t img, const Flat& flat) {
    for_sais(LF.size(), [&](size_t st) {
        if (!mask[st]) return;
        if (LF[st].size() != img) LF[st].resize(img);
        std::memcpy(LF[st].data(), &flat[st * img], img * sizeof(float));
    });
}

/* one pointer per SAI into the caller's vectors; output vectors of non-empty SAIs get their size first (the reference resizes
 * them where it forms the estimates, utilities_LF.cpp:936-941), empty SAIs are left alone */
std::vector<float*> sai_ptrs(std::vector<std::vector<float> >& LF, const std::vector<unsigned>& mask, size_t img, bool output) {
    std::vector<float*> p(LF.size(), nullptr);
    if (output) for_sais(LF.size(), [&](size_t st) { if (mask[st] && LF[st].size() != img) LF[st].resize(img); });
    for (size_t st = 0; st < LF.size(); st++) if (mask[st] && LF[st].size() == img) p[st] = LF[st].data();
    return p;
}
bool inputs_ok(const std::vector<float*>& p, const std::vector<unsigned>& mask, const char* who) {
    for (size_t st = 0; st < p.size(); st++)
        if (mask[st] && !p[st]) { std::cout << who << ": a non-empty SAI does not hold width*height*chnls values" << std::endl; return false; }
    return true;
}

lfbm5d_params make(float sigma, float lambda, unsigned N, unsigned nSim, unsigned nDisp, unsigned k, unsigned p,
                   bool useSD, unsigned t2, unsigned t4, unsigned t5, unsigned cs) {
    lfbm5d_params P;
    P.sigma = sigma; P.lambda = lambda; P.N = N; P.nSim = nSim; P.nDisp = nDisp; P.k = k; P.p = p;
    P.useSD = useSD ? 1u : 0u; P.tau_2D = t2; P.tau_4D = t4; P.tau_5D = t5; P.color_space = cs;
    return P;
}

} // namespace

/* nb_threads selects the reference's tile mode (bm5d.cpp:411-708), whose result depends on the caller's core count and is
 * about 0.5 dB below the untiled one.  The GPU path therefore ignores it -- nb_threads == 1 semantics -- unless
 * LFBM5D_TILED is set in the environment: then nb_threads tiles (floored to a power of two) are reproduced. */
static int tiles_for(unsigned nb_threads) {
    const char* e = std::getenv("LFBM5D_TILED");
    return (e && *e && *e != '0' && nb_threads > 1) ? (int)nb_threads : 1;
}

int run_bm5d_1st_step(const float sigma, const float lambdaHard5D, std::vector<std::vector<float> >& LF_noisy,
                      std::vector<unsigned>& LF_SAI_mask, std::vector<std::vector<float> >& LF_basic,
                      const unsigned ang_major, const unsigned awidth, const unsigned aheight, const unsigned anHard,
                      const unsigned width, const unsigned height, const unsigned chnls, const unsigned NHard,
                      const unsigned nSim, const unsigned nDisp, const unsigned kHard, const unsigned pHard,
                      const bool useSD, const unsigned tau_2D, unsigned tau_4D, const unsigned tau_5D,
                      const unsigned color_space, const unsigned nb_threads) {
    const unsigned asize = awidth * aheight;
    if (LF_noisy.size() != asize || LF_SAI_mask.size() != asize) {
        std::cout << "run_bm5d_1st_step: light field and mask must hold awidth*aheight SAIs" << std::endl;
        return EXIT_FAILURE;
    }
    lfbm5d_ctx* ctx = context();
    if (!ctx) return EXIT_FAILURE;
    lfbm5d_set_tiles(ctx, tiles_for(nb_threads));
    if (LF_basic.size() != asize) LF_basic.resize(asize); /* bm5d.cpp:129-130 */
    const size_t img = (size_t)width * height * chnls;
    const std::vector<float*> noisy = sai_ptrs(LF_noisy, LF_SAI_mask, img, false), basic = sai_ptrs(LF_basic, LF_SAI_mask, img, true);
    if (!inputs_ok(noisy, LF_SAI_mask, "run_bm5d_1st_step")) return EXIT_FAILURE;
    const lfbm5d_params P = make(sigma, lambdaHard5D, NHard, nSim, nDisp, kHard, pHard, useSD, tau_2D, tau_4D, tau_5D, color_space);
    if (lfbm5d_step1_host_sai(ctx, &P, noisy.data(), LF_SAI_mask.data(), basic.data(), ang_major, awidth, aheight, anHard,
                              width, height, chnls) != 0) {
        std::cout << "LFBM5D GPU backend: " << lfbm5d_last_error(ctx) << std::endl;
        return EXIT_FAILURE;
    }
    return EXIT_SUCCESS;
}

int run_bm5d_2nd_step(const float sigma, std::vector<std::vector<float> >& LF_noisy, std::vector<unsigned>& LF_SAI_mask,
                      std::vector<std::vector<float> >& LF_basic, std::vector<std::vector<float> >& LF_denoised,
                      const unsigned ang_major, const unsigned awidth, const unsigned aheight, const unsigned anWien,
                      const unsigned width, const unsigned height, const unsigned chnls, const unsigned NWien,
                      const unsigned nSim, const unsigned nDisp, const unsigned kWien, const unsigned pWien,
                      const bool useSD, const unsigned tau_2D, unsigned tau_4D, const unsigned tau_5D,
                      const unsigned color_space, const unsigned nb_threads) {
    const unsigned asize = awidth * aheight;
    if (LF_noisy.size() != asize || LF_basic.size() != asize || LF_SAI_mask.size() != asize) {
        std::cout << "run_bm5d_2nd_step: light fields and mask must hold awidth*aheight SAIs" << std::endl;
        return EXIT_FAILURE;
    }
    lfbm5d_ctx* ctx = context();
    if (!ctx) return EXIT_FAILURE;
    lfbm5d_set_tiles(ctx, tiles_for(nb_threads));
    if (LF_denoised.size() != asize) LF_denoised.resize(asize); /* bm5d.cpp:823-824 */
    const size_t img = (size_t)width * height * chnls;
    const std::vector<float*> noisy = sai_ptrs(LF_noisy, LF_SAI_mask, img, false), basic = sai_ptrs(LF_basic, LF_SAI_mask, img, false),
                              den = sai_ptrs(LF_denoised, LF_SAI_mask, img, true);
    if (!inputs_ok(noisy, LF_SAI_mask, "run_bm5d_2nd_step") || !inputs_ok(basic, LF_SAI_mask, "run_bm5d_2nd_step")) return EXIT_FAILURE;
    const lfbm5d_params P = make(sigma, 0.0f, NWien, nSim, nDisp, kWien, pWien, useSD, tau_2D, tau_4D, tau_5D, color_space);
    if (lfbm5d_step2_host_sai(ctx, &P, noisy.data(), LF_SAI_mask.data(), basic.data(), den.data(), ang_major, awidth, aheight,
                              anWien, width, height, chnls) != 0) {
        std::cout << "LFBM5D GPU backend: " << lfbm5d_last_error(ctx) << std::endl;
        return EXIT_FAILURE;
    }
    return EXIT_SUCCESS;
}

/* main.cpp:195 + :242 as one job (lfbm5d_denoise_host): same buffers on return as the two calls.  The tile mode (LFBM5D_TILED) has no
 * graph form: the library then runs the two steps one after the other behind the same entry point. */
int run_bm5d(const float sigma, const float lambdaHard5D, std::vector<std::vector<float> >& LF_noisy, std::vector<unsigned>& LF_SAI_mask,
             std::vector<std::vector<float> >& LF_basic, std::vector<std::vector<float> >& LF_denoised, const unsigned ang_major,
             const unsigned awidth, const unsigned aheight, const unsigned anHard, const unsigned anWien, const unsigned width,
             const unsigned height, const unsigned chnls, const unsigned NHard, const unsigned nSimHard, const unsigned nDispHard,
             const unsigned kHard, const unsigned pHard, const bool useSDHard, const unsigned tau_2D_hard, unsigned tau_4D_hard,
             const unsigned tau_5D_hard, const unsigned NWien, const unsigned nSimWien, const unsigned nDispWien, const unsigned kWien,
             const unsigned pWien, const bool useSDWien, const unsigned tau_2D_wien, unsigned tau_4D_wien, const unsigned tau_5D_wien,
             const unsigned color_space, const unsigned nb_threads) {
    const unsigned asize = awidth * aheight;
    if (LF_noisy.size() != asize || LF_SAI_mask.size() != asize) {
        std::cout << "run_bm5d: light field and mask must hold awidth*aheight SAIs" << std::endl;
        return EXIT_FAILURE;
    }
    lfbm5d_ctx* ctx = context();
    if (!ctx) return EXIT_FAILURE;
    lfbm5d_set_tiles(ctx, tiles_for(nb_threads));
    if (LF_basic.size() != asize) LF_basic.resize(asize);
    if (LF_denoised.size() != asize) LF_denoised.resize(asize);
    const size_t img = (size_t)width * height * chnls;
    const std::vector<float*> noisy = sai_ptrs(LF_noisy, LF_SAI_mask, img, false), basic = sai_ptrs(LF_basic, LF_SAI_mask, img, true),
                              den = sai_ptrs(LF_denoised, LF_SAI_mask, img, true);
    if (!inputs_ok(noisy, LF_SAI_mask, "run_bm5d")) return EXIT_FAILURE;
    const lfbm5d_params P1 = make(sigma, lambdaHard5D, NHard, nSimHard, nDispHard, kHard, pHard, useSDHard, tau_2D_hard, tau_4D_hard, tau_5D_hard, color_space);
    const lfbm5d_params P2 = make(sigma, 0.0f, NWien, nSimWien, nDispWien, kWien, pWien, useSDWien, tau_2D_wien, tau_4D_wien, tau_5D_wien, color_space);
    if (lfbm5d_denoise_host_sai(ctx, &P1, &P2, noisy.data(), LF_SAI_mask.data(), basic.data(), den.data(), ang_major, awidth, aheight,
                                anHard, anWien, width, height, chnls) != 0) {
        std::cout << "LFBM5D GPU backend: " << lfbm5d_last_error(ctx) << std::endl;
        return EXIT_FAILURE;
    }
    return EXIT_SUCCESS;
}

/* run_bm3d_LF (src/bm3d_LF.h:10-35, bm3d_LF.cpp:75-125): BM3D on every SAI of the mask */
#include "run_bm3d_lf.h"
int run_bm3d_LF(const float sigma, std::vector<std::vector<float> >& LF_noisy, std::vector<unsigned>& LF_SAI_mask,
                std::vector<std::vector<float> >& LF_basic, std::vector<std::vector<float> >& LF_denoised,
                const unsigned width, const unsigned height, const unsigned chnls, const unsigned nHard, const unsigned nWien,
                const unsigned kHard, const unsigned kWien, const unsigned NHard, const unsigned NWien, const unsigned pHard,
                const unsigned pWien, const bool useSD_h, const bool useSD_w, const unsigned tau_2D_hard,
                const unsigned tau_2D_wien, const float lambdaHard3D, const unsigned color_space, unsigned /*nb_threads*/,
                char* sub_img_name) {
    const size_t asize = LF_noisy.size();
    if (LF_SAI_mask.size() != asize) {
        std::cout << "run_bm3d_LF: light field and mask must hold the same number of SAIs" << std::endl;
        return EXIT_FAILURE;
    }
    lfbm5d_ctx* ctx = context();
    if (!ctx) return EXIT_FAILURE;
    if (LF_basic.size() != asize) LF_basic.resize(asize);       /* bm3d_LF.cpp:99-102 */
    if (LF_denoised.size() != asize) LF_denoised.resize(asize);
    const size_t img = (size_t)width * height * chnls;
    Flat noisy, basic = flat_alloc(asize * img), den = flat_alloc(asize * img);
    flatten(LF_noisy, LF_SAI_mask, img, noisy);
    lfbm5d_bm3d_params Hd, Wn;
    Hd.sigma = sigma; Hd.lambda3D = lambdaHard3D; Hd.N = NHard; Hd.nHW = nHard; Hd.k = kHard; Hd.p = pHard;
    Hd.useSD = useSD_h ? 1u : 0u; Hd.tau_2D = tau_2D_hard; Hd.color_space = color_space;
    Wn = Hd; Wn.N = NWien; Wn.nHW = nWien; Wn.k = kWien; Wn.p = pWien; Wn.useSD = useSD_w ? 1u : 0u; Wn.tau_2D = tau_2D_wien;
    std::cout << " - > Running BM3D filter on every " << (sub_img_name ? sub_img_name : "SAI") << " (GPU)" << std::endl;
    if (lfbm5d_bm3d_lf_host(ctx, &Hd, &Wn, noisy.get(), LF_SAI_mask.data(), basic.get(), den.get(), (unsigned)asize, width,
                            height, chnls) != 0) {
        std::cout << "LFBM5D GPU backend: " << lfbm5d_last_error(ctx) << std::endl;
        return EXIT_FAILURE;
    }
    unflatten(LF_noisy, LF_SAI_mask, img, noisy);
    unflatten(LF_basic, LF_SAI_mask, img, basic);
    unflatten(LF_denoised, LF_SAI_mask, img, den);
    return EXIT_SUCCESS;
}

/* Measurement hook (bench.py `seam.dropin_vectors`, tests): the interval the reference times -- main.cpp:189-201 around
 * run_bm5d_1st_step plus :241-247 around run_bm5d_2nd_step -- through THIS file's functions on vector<vector<float>> light fields
 * built from a flat copy, output vectors sized beforehand like main.cpp:158-165 does.  mode 0: the two calls, 1: run_bm5d (one job).
 * hard / wien = {N, nSim, nDisp, k, p, useSD, tau_2D, tau_4D, tau_5D}.  ms_out[2 * rep + {0, 1}] = the two intervals (mode 1: the job,
 * 0).  The last repetition's light fields are copied to the non-NULL flat outputs. */
#include <chrono>
extern "C" int lfbm5d_dropin_probe(int mode, const float* noisy_flat, const unsigned* mask, float* noisy_out, float* basic_out,
                                   float* denoised_out, unsigned ang_major, unsigned awidth, unsigned aheight, unsigned anHard,
                                   unsigned anWien, unsigned width, unsigned height, unsigned chnls, float sigma, float lambda,
                                   const unsigned* hard, const unsigned* wien, unsigned color_space, int reps, double* ms_out) {
    const size_t asize = (size_t)awidth * aheight, img = (size_t)width * height * chnls;
    std::vector<unsigned> m(mask, mask + asize);
    std::vector<std::vector<float> > LF_noisy(asize), LF_basic(asize), LF_denoised(asize);
    for_sais(asize, [&](size_t st) { LF_noisy[st].resize(img, 0.0f); LF_basic[st].resize(img, 0.0f); LF_denoised[st].resize(img, 0.0f); });
    typedef std::chrono::steady_clock clk;
    for (int r = 0; r < reps; r++) {
        for_sais(asize, [&](size_t st) { std::memcpy(LF_noisy[st].data(), noisy_flat + st * img, img * sizeof(float)); });
        const clk::time_point t0 = clk::now();
        clk::time_point t1 = t0, t2 = t0;
        if (mode == 1) {
            if (run_bm5d(sigma, lambda, LF_noisy, m, LF_basic, LF_denoised, ang_major, awidth, aheight, anHard, anWien, width, height, chnls,
                         hard[0], hard[1], hard[2], hard[3], hard[4], hard[5] != 0, hard[6], hard[7], hard[8],
                         wien[0], wien[1], wien[2], wien[3], wien[4], wien[5] != 0, wien[6], wien[7], wien[8], color_space, 1) != EXIT_SUCCESS) return 1;
            t1 = t2 = clk::now();
        } else {
            if (run_bm5d_1st_step(sigma, lambda, LF_noisy, m, LF_basic, ang_major, awidth, aheight, anHard, width, height, chnls, hard[0], hard[1],
                                  hard[2], hard[3], hard[4], hard[5] != 0, hard[6], hard[7], hard[8], color_space, 1) != EXIT_SUCCESS) return 1;
            t1 = clk::now();
            if (run_bm5d_2nd_step(sigma, LF_noisy, m, LF_basic, LF_denoised, ang_major, awidth, aheight, anWien, width, height, chnls, wien[0],
                                  wien[1], wien[2], wien[3], wien[4], wien[5] != 0, wien[6], wien[7], wien[8], color_space, 1) != EXIT_SUCCESS) return 1;
            t2 = clk::now();
        }
        if (ms_out) {
            ms_out[2 * r] = std::chrono::duration<double, std::milli>(t1 - t0).count();
            ms_out[2 * r + 1] = std::chrono::duration<double, std::milli>(t2 - t1).count();
        }
    }
    for_sais(asize, [&](size_t st) {
        if (!m[st]) return;
        if (noisy_out) std::memcpy(noisy_out + st * img, LF_noisy[st].data(), img * sizeof(float));
        if (basic_out) std::memcpy(basic_out + st * img, LF_basic[st].data(), img * sizeof(float));
        if (denoised_out) std::memcpy(denoised_out + st * img, LF_denoised[st].data(), img * sizeof(float));
    });
    return 0;
}
