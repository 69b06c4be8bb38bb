/*
 * ref_wrap.cpp -- C-callable entry points over the REFERENCE's own leaf routines.
 * TEST INFRASTRUCTURE ONLY.  Compiled together with the unmodified reference sources where they
 * lie (/root/reference/src/lib_transforms.cpp, mt19937ar.c) into oracle/_ref/libref_leaf.so by
 * oracle/Makefile.  These are the only reference translation units that build without FFTW3 /
 * libpng headers (absent from this image), so they are the only compiled-reference pins.
 */
#include <vector>
#include "lib_transforms.h" /* from /root/reference/src via -I */
extern "C" {
#include "mt19937ar.h"
}

extern "C" {

void ref_haar_forward(float* v, unsigned n) {
    std::vector<float> a(v, v + n), tmp(n);
    haar_forward(a, tmp, n, 0);
    for (unsigned i = 0; i < n; i++) v[i] = a[i];
}
void ref_haar_inverse(float* v, unsigned n) {
    std::vector<float> a(v, v + n), tmp(n);
    haar_inverse(a, tmp, 1, n, 0);
    for (unsigned i = 0; i < n; i++) v[i] = a[i];
}
void ref_hadamard(float* v, unsigned n) {
    std::vector<float> a(v, v + n), tmp(n);
    hadamard_transform(a, tmp, n, 0);
    for (unsigned i = 0; i < n; i++) v[i] = a[i];
}
void ref_bior_forward(const float* in, unsigned in_stride, unsigned in_size, float* out, unsigned n) {
    std::vector<float> a(in, in + in_size), o(n * n), lpd, hpd, lpr, hpr;
    bior15_coef(lpd, hpd, lpr, hpr);
    bior_2d_forward(a, o, n, 0, in_stride, 0, lpd, hpd);
    for (unsigned i = 0; i < n * n; i++) out[i] = o[i];
}
void ref_bior_inverse(float* patch, unsigned n) {
    std::vector<float> a(patch, patch + n * n), lpd, hpd, lpr, hpr;
    bior15_coef(lpd, hpd, lpr, hpr);
    bior_2d_inverse(a, n, 0, lpr, hpr);
    for (unsigned i = 0; i < n * n; i++) patch[i] = a[i];
}
void ref_mt_seed(unsigned long s) { mt_init_genrand(s); }
double ref_mt_res53(void) { return mt_genrand_res53(); }

}
