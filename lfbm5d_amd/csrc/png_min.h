/*
 * png_min.h -- minimal PNG reader/writer over zlib for the LFBM5Ddenoising CLI.
 * The reference uses IPOL's io_png.c on libpng (src/io_png.c); libpng headers are not part of this
 * image, zlib's are.  Reads 8-bit greyscale / grey+alpha / RGB / RGBA, non-interlaced (what
 * testing/sourceLF and the reference's own outputs are); writes 8-bit greyscale or RGB.
 * Pixel layout at this interface is the reference's planar float: img[c*W*H + y*W + x].
 */
#ifndef LFBM5D_PNG_MIN_H
#define LFBM5D_PNG_MIN_H
#include <cstddef>
#include <string>
#include <vector>

/* returns false on any error; channels = 1..4 as stored in the file */
bool png_read_planar_f32(const std::string& path, std::vector<float>& img, size_t& w, size_t& h, size_t& c);
/* values are rounded to nearest and must already be clipped to [0,255]; c = 1 or 3 */
bool png_write_planar_f32(const std::string& path, const float* img, size_t w, size_t h, size_t c);
#endif
