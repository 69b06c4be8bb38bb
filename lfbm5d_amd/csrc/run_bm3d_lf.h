/*
 * run_bm3d_lf.h -- drop-in declaration with the reference's exact signature (V-Sense/LFBM5D src/bm3d_LF.h:10-35):
 * BM3D on every SAI of the light field independently, on the GPU through the C-ABI of include/lfbm5d.h.
 */
#ifndef LFBM5D_RUN_BM3D_LF_H
#define LFBM5D_RUN_BM3D_LF_H

#include <vector>

//! LF_noisy is colour-transformed and transformed back in place (lossy for OPP, like the reference); LF_basic and
//! LF_denoised are (re)sized and filled for the SAIs of the mask.  nb_threads is accepted for source compatibility:
//! the GPU path always has the semantics of nb_threads == 1 (no sub-image division).
int run_bm3d_LF(
    const float sigma
,   std::vector<std::vector<float> > &LF_noisy
,   std::vector<unsigned> &LF_SAI_mask
,   std::vector<std::vector<float> > &LF_basic
,   std::vector<std::vector<float> > &LF_denoised
,   const unsigned width
,   const unsigned height
,   const unsigned chnls
,   const unsigned nHard
,   const unsigned nWien
,   const unsigned kHard
,   const unsigned kWien
,   const unsigned NHard
,   const unsigned NWien
,   const unsigned pHard
,   const unsigned pWien
,   const bool useSD_h
,   const bool useSD_w
,   const unsigned tau_2D_hard
,   const unsigned tau_2D_wien
,   const float    lambdaHard3D
,   const unsigned color_space
,   unsigned nb_threads
,   char *sub_img_name
);

#endif
