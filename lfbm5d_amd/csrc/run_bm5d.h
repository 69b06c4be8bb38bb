/*
 * run_bm5d.h -- drop-in declarations with the reference's exact signatures
 * (V-Sense/LFBM5D src/bm5d.h:11-35 and :38-62).  A program written against the reference's bm5d.h
 * links against liblfbm5d_dropin.so instead of the reference's bm5d.cpp / bm5d_core_processing.cpp
 * and runs the whole path on the GPU through the C-ABI of include/lfbm5d.h.
 */
#ifndef LFBM5D_RUN_BM5D_H
#define LFBM5D_RUN_BM5D_H

#include <vector>

//! Main function (hard-thresholding step) -- same argument list as the reference.  LF_noisy is
//! colour-transformed and transformed back in place (lossy for OPP, like the reference),
//! LF_basic is (re)sized and filled.  nb_threads is accepted for source compatibility; the GPU path
//! has the semantics of nb_threads == 1 (no tile-halo discard) unless LFBM5D_TILED is set in the
//! environment, in which case the reference's tile mode with nb_threads tiles is reproduced.
int run_bm5d_1st_step(
    const float sigma
,   const float lambdaHard5D
,   std::vector<std::vector<float> > &LF_noisy
,   std::vector<unsigned> &LF_SAI_mask
,   std::vector<std::vector<float> > &LF_basic
,   const unsigned ang_major
,   const unsigned awidth
,   const unsigned aheight
,   const unsigned anHard
,   const unsigned width
,   const unsigned height
,   const unsigned chnls
,   const unsigned NHard
,   const unsigned nSim
,   const unsigned nDisp
,   const unsigned kHard
,   const unsigned pHard
,   const bool     useSD
,   const unsigned tau_2D
,         unsigned tau_4D
,   const unsigned tau_5D
,   const unsigned color_space
,   const unsigned nb_threads
);

//! Main function (Wiener step)
int run_bm5d_2nd_step(
    const float sigma
,   std::vector<std::vector<float> > &LF_noisy
,   std::vector<unsigned> &LF_SAI_mask
,   std::vector<std::vector<float> > &LF_basic
,   std::vector<std::vector<float> > &LF_denoised
,   const unsigned ang_major
,   const unsigned awidth
,   const unsigned aheight
,   const unsigned anWien
,   const unsigned width
,   const unsigned height
,   const unsigned chnls
,   const unsigned NWien
,   const unsigned nSim
,   const unsigned nDisp
,   const unsigned kWien
,   const unsigned pWien
,   const bool     useSD
,   const unsigned tau_2D
,         unsigned tau_4D
,   const unsigned tau_5D
,   const unsigned color_space
,   const unsigned nb_threads
);

//! Both steps as ONE job -- not in the reference (whose main.cpp:195, :242 calls the two functions above one after the other; BASELINE.json
//! names it "run_bm5d()"): == run_bm5d_1st_step(...) followed by run_bm5d_2nd_step(...) with the same arguments, bit for bit, but
//! the windows of both steps run as one dependency graph on the GPU(s) (lfbm5d_denoise_host, include/lfbm5d.h).
int run_bm5d(
    const float sigma
,   const float lambdaHard5D
,   std::vector<std::vector<float> > &LF_noisy
,   std::vector<unsigned> &LF_SAI_mask
,   std::vector<std::vector<float> > &LF_basic
,   std::vector<std::vector<float> > &LF_denoised
,   const unsigned ang_major
,   const unsigned awidth
,   const unsigned aheight
,   const unsigned anHard
,   const unsigned anWien
,   const unsigned width
,   const unsigned height
,   const unsigned chnls
,   const unsigned NHard, const unsigned nSimHard, const unsigned nDispHard, const unsigned kHard, const unsigned pHard
,   const bool useSDHard, const unsigned tau_2D_hard, unsigned tau_4D_hard, const unsigned tau_5D_hard
,   const unsigned NWien, const unsigned nSimWien, const unsigned nDispWien, const unsigned kWien, const unsigned pWien
,   const bool useSDWien, const unsigned tau_2D_wien, unsigned tau_4D_wien, const unsigned tau_5D_wien
,   const unsigned color_space
,   const unsigned nb_threads
);

#endif
