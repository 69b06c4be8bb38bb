/*
 * run_bm5d.cpp -- the reference's outer seam (src/bm5d.h:11-62) on top of the C-ABI.
 * Host side only: flattens the vector-of-vectors light field, calls lfbm5d_step{1,2}_host
 * (which stages through HBM and runs every kernel on the GPU), copies the results back.
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
    Flat noisy, basic = flat_alloc(asize * img);
    flatten(LF_noisy, LF_SAI_mask, img, noisy);
    const lfbm5d_params P = make(sigma, lambdaHard5D, NHard, nSim, nDisp, kHard, pHard, useSD, tau_2D, tau_4D, tau_5D, color_space);
    if (lfbm5d_step1_host(ctx, &P, noisy.get(), LF_SAI_mask.data(), basic.get(), ang_major, awidth, aheight, anHard,
                          width, height, chnls) != 0) {
        std::cout << "LFBM5D GPU backend: " << lfbm5d_last_error(ctx) << std::endl;
        return EXIT_FAILURE;
    }
    unflatten(LF_noisy, LF_SAI_mask, img, noisy);
    unflatten(LF_basic, LF_SAI_mask, img, basic);
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
    Flat noisy, basic, den = flat_alloc(asize * img);
    flatten(LF_noisy, LF_SAI_mask, img, noisy);
    flatten(LF_basic, LF_SAI_mask, img, basic);
    const lfbm5d_params P = make(sigma, 0.0f, NWien, nSim, nDisp, kWien, pWien, useSD, tau_2D, tau_4D, tau_5D, color_space);
    if (lfbm5d_step2_host(ctx, &P, noisy.get(), LF_SAI_mask.data(), basic.get(), den.get(), ang_major, awidth, aheight,
                          anWien, width, height, chnls) != 0) {
        std::cout << "LFBM5D GPU backend: " << lfbm5d_last_error(ctx) << std::endl;
        return EXIT_FAILURE;
    }
    unflatten(LF_noisy, LF_SAI_mask, img, noisy);
    unflatten(LF_basic, LF_SAI_mask, img, basic);
    unflatten(LF_denoised, LF_SAI_mask, img, den);
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
    Flat noisy, basic = flat_alloc(asize * img), den = flat_alloc(asize * img);
    flatten(LF_noisy, LF_SAI_mask, img, noisy);
    const lfbm5d_params P1 = make(sigma, lambdaHard5D, NHard, nSimHard, nDispHard, kHard, pHard, useSDHard, tau_2D_hard, tau_4D_hard, tau_5D_hard, color_space);
    const lfbm5d_params P2 = make(sigma, 0.0f, NWien, nSimWien, nDispWien, kWien, pWien, useSDWien, tau_2D_wien, tau_4D_wien, tau_5D_wien, color_space);
    if (lfbm5d_denoise_host(ctx, &P1, &P2, noisy.get(), LF_SAI_mask.data(), basic.get(), den.get(), ang_major, awidth, aheight,
                            anHard, anWien, width, height, chnls) != 0) {
        std::cout << "LFBM5D GPU backend: " << lfbm5d_last_error(ctx) << std::endl;
        return EXIT_FAILURE;
    }
    unflatten(LF_noisy, LF_SAI_mask, img, noisy);
    unflatten(LF_basic, LF_SAI_mask, img, basic);
    unflatten(LF_denoised, LF_SAI_mask, img, den);
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
