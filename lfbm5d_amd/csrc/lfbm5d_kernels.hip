/*
 * lfbm5d_kernels.hip -- HIP kernels of the LFBM5D core for gfx950 (MI355X), except block matching
 * (lfbm5d_bm.hip).
 *
 *   k_group_pos / k_group_shape   geometry pre-pass of a core pass: where every patch of every 5-D group
 *                is gathered from and aggregated at, SADCT bookkeeping per group (core:286-323, :503).
 *   k_group_id_haar / _any        hard-thresholding step with tau_2D = id: one thread per pixel holds the
 *                pixel of all nSx * A patches in registers (packed fp32), no LDS stack.
 *   k_group_dct8w                  Wiener step with 8x8 patches and a 2-D DCT: noisy and pilot stacks as the
 *                two halves of float2 values, LDS stack [coefficient][patch], packed fp32.
 *   k_group_dct8, k_group          the other configurations: one workgroup per (group, channel), the
 *                nSx * A * k^2 stack(s) in LDS, 2-D transform (id / DCT / bior1.5), 4-D angular
 *                transform (DCT or shape-adaptive DCT), 5th-dimension Haar / Hadamard / DCT + hard
 *                threshold or Wiener shrinkage, inverses, filtered patches and group weight out.
 *                Restate core:277-481 / :1054-1282.  No MFMA: the largest transform is 16 points.
 *   k_aggregate  per tile of 64 output pixels and SAI: gathers every filtered patch that overlaps the
 *                tile, in the reference's own order (reference patches in raster order, then match
 *                index), so num/den are reproducible run to run and need no float atomics.
 *                Restates core:484-528.
 *   small elementwise / reduction kernels for the window schedule of bm5d.cpp:165-407, batched over the
 *                SAIs of a window / of the light field.
 */
#include "lfbm5d_kernels.h"

#include <algorithm>

#include <type_traits>

namespace lfbm5d {

namespace {

/* transform tables are constant during a kernel: constant address space, so uniform reads become scalar loads */
typedef const __attribute__((address_space(4))) GroupTables* TbPtr;
typedef const __attribute__((address_space(4))) float* TbFloats;

/* ============================== elementwise helpers ======================================= */

/* colour transform of one pixel: utilities.cpp:482-599, same expressions; contraction off to keep their rounding */
__device__ __forceinline__ void color_px(unsigned cs, int fwd, const float R, const float G, const float B, float& x, float& y, float& z) {
#pragma clang fp contract(off)
    if (cs == 0) { /* YUV */
        if (fwd) { x = 0.299f * R + 0.587f * G + 0.114f * B; y = -0.14713f * R - 0.28886f * G + 0.436f * B; z = 0.615f * R - 0.51498f * G - 0.10001f * B; }
        else     { x = R + 1.13983f * B; y = R - 0.39465f * G - 0.5806f * B; z = R + 2.03211f * G; }
    } else if (cs == 1) { /* YCbCr */
        if (fwd) { x = 0.299f * R + 0.587f * G + 0.114f * B; y = -0.169f * R - 0.331f * G + 0.500f * B; z = 0.500f * R - 0.419f * G - 0.081f * B; }
        else     { x = 1.000f * R + 0.000f * G + 1.402f * B; y = 1.000f * R - 0.344f * G - 0.714f * B; z = 1.000f * R + 1.772f * G + 0.000f * B; }
    } else {       /* OPP */
        if (fwd) { x = 0.333f * R + 0.333f * G + 0.333f * B; y = 0.500f * R + 0.000f * G - 0.500f * B; z = 0.250f * R - 0.500f * G + 0.250f * B; }
        else     { x = 1.0f * R + 1.0f * G + 0.666f * B; y = 1.0f * R + 0.0f * G - 1.333f * B; z = 1.0f * R - 1.0f * G + 0.666f * B; }
    }
}

/* blockIdx.y = SAI of a light field laid out [SAI][3][n]; SAIs whose mask entry is 0 are left alone */
__global__ void k_color(float* __restrict__ img, size_t sai_stride, const unsigned* __restrict__ mask, unsigned cs, unsigned n, int fwd) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || (mask && !mask[blockIdx.y])) return;
    img += blockIdx.y * sai_stride;
    float x, y, z;
    color_px(cs, fwd, img[i], img[i + n], img[i + 2 * n], x, y, z);
    img[i] = x; img[i + n] = y; img[i + 2 * n] = z;
}

/* out = fwd(inv(in)) per pixel: what a light field looks like to the second step after the first step's closing inverse
 * transform and the second's opening forward transform (bm5d.cpp:711-714, :827-830; the reference's matrices are not
 * inverses of each other).  in == out is allowed. */
__global__ void k_color_roundtrip(const float* in, float* out, size_t sai_stride, const unsigned* __restrict__ mask,
                                  unsigned cs, unsigned n) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || (mask && !mask[blockIdx.y])) return;
    in += blockIdx.y * sai_stride; out += blockIdx.y * sai_stride;
    float x, y, z, u, v, w;
    color_px(cs, 0, in[i], in[i + n], in[i + 2 * n], x, y, z);
    color_px(cs, 1, x, y, z, u, v, w);
    out[i] = u; out[i + n] = v; out[i + 2 * n] = w;
}

__device__ __forceinline__ int mirror(int x, int n) { return x < 0 ? -x - 1 : (x >= n ? 2 * n - x - 1 : x); }

/* utilities.cpp:215-263 in closed form: padded (i,j) reads source (mirror(i-N), mirror(j-N)) */
__global__ void k_symetrize(const float* __restrict__ src, float* __restrict__ dst, int W, int H, int C, int N) {
    const int w = W + 2 * N, h = H + 2 * N;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)w * h * C) return;
    const int c = (int)(i / ((size_t)w * h));
    const int r = (int)(i % ((size_t)w * h));
    const int y = mirror(r / w - N, H), x = mirror(r % w - N, W);
    dst[i] = src[(size_t)c * W * H + (size_t)y * W + x];
}

__global__ void k_unsymetrize(float* __restrict__ dst, const float* __restrict__ src, int W, int H, int C, int N, int off) {
    /* N: padding of src; off: offset of the crop (== N except for the reference's BM3D second-step crop, bm3d.cpp:181-189) */
    const int w = W + 2 * N, h = H + 2 * N;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)W * H * C) return;
    const int c = (int)(i / ((size_t)W * H));
    const int r = (int)(i % ((size_t)W * H));
    dst[i] = src[(size_t)c * w * h + (size_t)(r / W + off) * w + r % W + off];
}

/* the same for all SAIs of an angular window in one launch: blockIdx.y = window slot, L.st[slot] = SAI
 * index in the light-field buffer (0xffffffff: empty slot) */
__global__ void k_symetrize_multi(const float* __restrict__ src, size_t src_stride, float* __restrict__ dst, size_t dst_stride,
                                  SaiList L, int W, int H, int C, int N) {
    const unsigned st = L.st[blockIdx.y];
    if (st == 0xffffffffu) return;
    const int w = W + 2 * N, h = H + 2 * N;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)w * h * C) return;
    const int c = (int)(i / ((size_t)w * h));
    const int r = (int)(i % ((size_t)w * h));
    const int y = mirror(r / w - N, H), x = mirror(r % w - N, W);
    dst[blockIdx.y * dst_stride + i] = src[st * src_stride + (size_t)c * W * H + (size_t)y * W + x];
}
__global__ void k_unsymetrize_multi(float* __restrict__ dst, size_t dst_stride, const float* __restrict__ src, size_t src_stride,
                                    SaiList L, int W, int H, int C, int N) {
    const unsigned st = L.st[blockIdx.y];
    if (st == 0xffffffffu) return;
    const int w = W + 2 * N, h = H + 2 * N;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)W * H * C) return;
    const int c = (int)(i / ((size_t)W * H));
    const int r = (int)(i % ((size_t)W * H));
    dst[st * dst_stride + i] = src[blockIdx.y * src_stride + (size_t)c * w * h + (size_t)(r / W + N) * w + r % W + N];
}

/* compute_LF_estimate utilities_LF.cpp:944-950 (IEEE division) */
__global__ void k_estimate(const float* __restrict__ num, const float* __restrict__ den,
                           const float* __restrict__ sub, float* __restrict__ est, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float d = den[i];
    est[i] = d ? __fdiv_rn(num[i], d) : sub[i];
}

/* matching estimate (channel 0) of every non-empty SAI of a window: blockIdx.y = SAI */
__global__ void k_estimate_multi(const float* __restrict__ num, const float* __restrict__ den, const float* __restrict__ sub,
                                 float* __restrict__ est, size_t plane, unsigned C, SaiMask mask_bits) {
    const unsigned st = blockIdx.y;
    if (!mask_bits.test(st)) return;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= plane) return;
    const size_t o = (size_t)st * C * plane + i;
    const float d = den[o];
    est[st * plane + i] = d ? __fdiv_rn(num[o], d) : sub[o];
}

/* final estimate of a whole light field [SAI][seg]: SAIs whose mask entry is 0 are left alone */
__global__ void k_estimate_lf(const float* __restrict__ num, const float* __restrict__ den, const float* __restrict__ sub,
                              float* __restrict__ est, size_t seg, const unsigned* __restrict__ mask) {
    if (mask && !mask[blockIdx.y]) return;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= seg) return;
    const size_t o = blockIdx.y * seg + i;
    const float d = den[o];
    est[o] = d ? __fdiv_rn(num[o], d) : sub[o];
}

/* Two-step jobs: the basic estimate of the SAIs whose first-step sums have just become final, as the second step reads it:
 * compute_LF_estimate (bm5d.cpp:405), inverse colour transform (bm5d.cpp:711), forward colour transform (bm5d.cpp:829) --
 * the operations run_bm5d_1st_step ends with and run_bm5d_2nd_step begins with, pixel by pixel.  blockIdx.y = entry of L. */
__global__ void k_finalize_multi(const float* __restrict__ num, const float* __restrict__ den, const float* __restrict__ sub,
                                 float* __restrict__ basic, size_t sai_stride, SaiList L, unsigned cs, unsigned n, int colour) {
    const unsigned st = L.st[blockIdx.y];
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const size_t o = (size_t)st * sai_stride + i;
    float e[3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float d = den[o + (size_t)c * n];
        e[c] = d ? __fdiv_rn(num[o + (size_t)c * n], d) : sub[o + (size_t)c * n];
    }
    if (colour) {
        float x, y, z;
        color_px(cs, 0, e[0], e[1], e[2], x, y, z);
        color_px(cs, 1, x, y, z, e[0], e[1], e[2]);
    }
    basic[o] = e[0]; basic[o + n] = e[1]; basic[o + 2 * (size_t)n] = e[2];
}

/* Streamed host seam (lfbm5d_*_host with the window graph): everything a step -- or the two-step job -- leaves in the caller's
 * light fields for the SAIs whose sums have just become final, pixel by pixel what the tail of run_bm5d_* does for the whole
 * light field: the estimate (bm5d.cpp:405 / :1106) and the closing inverse colour transforms of the result, of the basic estimate
 * (second step) and of LF_noisy (bm5d.cpp:711-714, :1414-1418).  In-place operands are read before anything is written.
 * blockIdx.y = entry of L. */
__global__ void k_output_multi(const float* __restrict__ num, const float* __restrict__ den, const float* sub, float* out,
                               float* basic, const float* noisy_src, float* noisy_dst, size_t sai_stride, SaiList L, unsigned cs,
                               unsigned n, int colour) {
    const unsigned st = L.st[blockIdx.y];
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const size_t o = (size_t)st * sai_stride + i;
    float e[3], b[3] = {0.f, 0.f, 0.f}, v[3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float d = den[o + (size_t)c * n];
        e[c] = d ? __fdiv_rn(num[o + (size_t)c * n], d) : sub[o + (size_t)c * n];
        if (basic) b[c] = basic[o + (size_t)c * n];
        v[c] = noisy_src[o + (size_t)c * n];
    }
    if (colour) {
        float x, y, z;
        color_px(cs, 0, e[0], e[1], e[2], x, y, z); e[0] = x; e[1] = y; e[2] = z;
        color_px(cs, 0, b[0], b[1], b[2], x, y, z); b[0] = x; b[1] = y; b[2] = z;
        color_px(cs, 0, v[0], v[1], v[2], x, y, z); v[0] = x; v[1] = y; v[2] = z;
    }
#pragma unroll
    for (int c = 0; c < 3; c++) {
        out[o + (size_t)c * n] = e[c];
        if (basic) basic[o + (size_t)c * n] = b[c];
        noisy_dst[o + (size_t)c * n] = v[c];
    }
}

/* Two processes on one GPU (the exchange's second transport, tests only: RCCL refuses two ranks on one device): a message is gated
 * by words in device memory both processes map (hipIpcMemHandle).  set: publish `v` behind everything queued on the stream; wait:
 * hold the stream until the word has reached `want` -- or until timeout_ticks of the 100 MHz clock have passed (a dead peer must end
 * in an error return, never in a hung GPU): then *err is set, and every later wait of the job returns at once. */
__global__ void k_ipc_set(unsigned* p, unsigned v) {
    __threadfence_system();
    __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void k_ipc_wait(const unsigned* p, unsigned want, unsigned* err, unsigned long long timeout_ticks) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while ((int)(__hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - want) < 0) {
        if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
        if (__builtin_amdgcn_s_memrealtime() - t0 > timeout_ticks) { __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return; }
        __builtin_amdgcn_s_sleep(64);
    }
    __threadfence_system();
}

__global__ void k_copy_rect(float* __restrict__ dst, size_t dst_stride, int dW, int dH, int dx0, int dy0,
                            const float* __restrict__ src, size_t src_stride, int sW, int sH, int sx0, int sy0,
                            int w, int h, int C, SaiMask mask_bits) {
    if (!mask_bits.test(blockIdx.y)) return;      /* blockIdx.y = window slot */
    const size_t total = (size_t)w * h * C;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i / ((size_t)w * h)), r = (int)(i % ((size_t)w * h));
    const int y = r / w, x = r % w;
    dst[blockIdx.y * dst_stride + ((size_t)c * dH + dy0 + y) * dW + dx0 + x] =
        src[blockIdx.y * src_stride + ((size_t)c * sH + sy0 + y) * sW + sx0 + x];
}
__global__ void k_fill_f32(float* p, float v, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}
__global__ void k_add(float* __restrict__ dst, const float* __restrict__ src, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] += src[i];
}
__global__ void k_fill_i32(int* p, int v, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

__device__ __forceinline__ unsigned block_sum_u32(unsigned v, unsigned* red) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    unsigned t = 0;
    if (threadIdx.x == 0) for (unsigned w = 0; w < blockDim.x / 64; w++) t += red[w];
    return t; /* valid in thread 0 */
}

__global__ void k_count_zeros(const float* __restrict__ den, size_t seg, unsigned* __restrict__ counts) {
    __shared__ unsigned red[4];
    const float* p = den + (size_t)blockIdx.y * seg;
    unsigned c = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < seg; i += (size_t)gridDim.x * blockDim.x)
        c += p[i] == 0.0f ? 1u : 0u;
    const unsigned t = block_sum_u32(c, red);
    if (threadIdx.x == 0 && t) atomicAdd(&counts[blockIdx.y], t);
}

__global__ void k_count_denoised(const float* __restrict__ den, size_t sai_stride, SaiMask mask_bits, int W, int H, int C, int N, int k,
                                 unsigned* __restrict__ count) {
    __shared__ unsigned red[4];
    if (!mask_bits.test(blockIdx.y)) return;      /* blockIdx.y = window slot */
    den += blockIdx.y * sai_stride;
    const int w = W + 2 * N, h = H + 2 * N;
    const int sw = W - k + 1, sh = H - k + 1;
    const size_t total = (size_t)sw * sh * C;
    unsigned c = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int ch = (int)(i / ((size_t)sw * sh));
        const int r = (int)(i % ((size_t)sw * sh));
        c += den[(size_t)ch * w * h + (size_t)(N + r / sw) * w + N + r % sw] > 0.0f ? 1u : 0u;
    }
    const unsigned t = block_sum_u32(c, red);
    if (threadIdx.x == 0 && t) atomicAdd(count, t);
}

/* ---- the two ends of a window of the graph form (run_graph, lfbm5d_api.hip), one launch each ----
 * begin: mirror-pad noisy (+ basic) + num + den of every SAI of the window (symetrize, utilities.cpp:215-263, bm5d.cpp:252-265)
 * and form the matching estimate of channel 0 (compute_LF_estimate, core:167-170) from the padded sums -- what
 * k_symetrize_multi x 3..4 + k_estimate_multi do, with one index computation per pixel instead of a 64-bit division per
 * element.  Thread = one padded pixel of one SAI, all channels; blockIdx.z = window slot. */
struct WinBeginArgs {
    const float* noisy; const float* basic; const float* num; const float* den;   /* light field [SAI][C][H][W] (basic: NULL in step 1) */
    float* w_noisy; float* w_basic; float* w_num; float* w_den;                   /* window [slot][C][Hb][Wb] */
    float* est;                                                                   /* [slot][Hb][Wb] */
    size_t lf_stride, w_stride;
    SaiList L;
    int W, H, C, N;
    unsigned* zero;                                                               /* the window's partial coverage counters (kWinEndCounters of them), cleared here */
};
__global__ __launch_bounds__(256) void k_window_begin(WinBeginArgs a) {
    if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.y == 0 && threadIdx.x < kWinCounters && a.zero) a.zero[threadIdx.x] = 0u;
    const unsigned st = a.L.st[blockIdx.z];
    if (st == 0xffffffffu) return;
    const int w = a.W + 2 * a.N, h = a.H + 2 * a.N;
    const int xx = blockIdx.x * 64 + threadIdx.x, yy = blockIdx.y * 4 + threadIdx.y;
    if (xx >= w || yy >= h) return;
    const size_t so = (size_t)st * a.lf_stride + (size_t)mirror(yy - a.N, a.H) * a.W + mirror(xx - a.N, a.W);
    const size_t d_o = (size_t)blockIdx.z * a.w_stride + (size_t)yy * w + xx;
    const size_t sp = (size_t)a.W * a.H, dp = (size_t)w * h;
    float n0 = 0.0f, u0 = 0.0f, d0 = 0.0f, b0 = 0.0f;
    for (int c = 0; c < a.C; c++) {
        const float nv = a.noisy[so + c * sp], uv = a.num[so + c * sp], dv = a.den[so + c * sp];
        a.w_noisy[d_o + c * dp] = nv; a.w_num[d_o + c * dp] = uv; a.w_den[d_o + c * dp] = dv;
        float bv = 0.0f;
        if (a.basic) { bv = a.basic[so + c * sp]; a.w_basic[d_o + c * dp] = bv; }
        if (c == 0) { n0 = nv; u0 = uv; d0 = dv; b0 = bv; }
    }
    a.est[(size_t)blockIdx.z * dp + (size_t)yy * w + xx] = d0 ? __fdiv_rn(u0, d0) : (a.basic ? b0 : n0);
}

/* end: the window's sums back into the light field (unsymetrize, utilities.cpp:265-298, bm5d.cpp:388-396) and the coverage
 * count of the pass (LF_denoised_percent, utilities_LF.cpp:985-992: (i, j, c) triples with den > 0 over the (H-k+1) x (W-k+1)
 * patch origins) -- k_unsymetrize_multi x 2 + k_count_denoised.  Thread = one pixel of one SAI, all channels. */
struct WinEndArgs {
    float* num; float* den; const float* w_num; const float* w_den;
    size_t lf_stride, w_stride;
    SaiList L;
    int W, H, C, N, k;
    unsigned* count;
};
constexpr int kWinEndRows = 32;     /* rows of a SAI per workgroup (8 rounds of 4) */
static_assert(kWinCounters == 32, "k_window_end");
constexpr int kWinEndCounters = 32; /* partial coverage counters: atomics to ONE address serialise (a hundred million a second) */
__global__ __launch_bounds__(256) void k_window_end(WinEndArgs a) {
    __shared__ unsigned red[4];
    const unsigned st = a.L.st[blockIdx.z];
    if (st == 0xffffffffu) return;      /* (uniform per workgroup) */
    const int w = a.W + 2 * a.N, h = a.H + 2 * a.N;
    const int x = blockIdx.x * 64 + threadIdx.x;
    const size_t dp = (size_t)a.W * a.H, sp = (size_t)w * h;
    unsigned cnt = 0;
    if (x < a.W)
        for (int r = 0; r < kWinEndRows; r += 4) {
            const int y = blockIdx.y * kWinEndRows + r + threadIdx.y;
            if (y >= a.H) break;
            const size_t d_o = (size_t)st * a.lf_stride + (size_t)y * a.W + x;
            const size_t so = (size_t)blockIdx.z * a.w_stride + (size_t)(y + a.N) * w + x + a.N;
            const bool counted = x < a.W - a.k + 1 && y < a.H - a.k + 1;
            for (int c = 0; c < a.C; c++) {
                const float dv = a.w_den[so + c * sp];
                a.num[d_o + c * dp] = a.w_num[so + c * sp];
                a.den[d_o + c * dp] = dv;
                cnt += (counted && dv > 0.0f) ? 1u : 0u;
            }
        }
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
    if (threadIdx.x == 0) red[threadIdx.y] = cnt;
    __syncthreads();
    if (threadIdx.x == 0 && threadIdx.y == 0) {
        const unsigned t = red[0] + red[1] + red[2] + red[3];
        if (t) atomicAdd(a.count + (blockIdx.x + blockIdx.y * 5 + blockIdx.z * 11) % kWinEndCounters, t);
    }
}

/* ================================== group kernel ========================================== */

constexpr int kThreads = 256;

template <int MA, int MW>
struct ShapeInfoT {          /* SADCT bookkeeping of one group (core:302-323, :2036-2049, :2102-2104) */
    int mask[MA], idx[MA], mask_col[MA], idx_col[MA], mask_dct[MA];   /* [s * aw + t], aw = window side */
    int row_n[MW], col_n[MW];
    int use_sadct;
};
typedef ShapeInfoT<kMaxA, kMaxAw> ShapeInfo;        /* windows of up to 7x7 SAIs */
typedef ShapeInfoT<kBigA, kBigAw> ShapeInfoBig;     /* larger windows (general forms only) */
static_assert(sizeof(ShapeInfo) == kShapeInfoBytes && sizeof(ShapeInfoBig) == kShapeInfoBigBytes, "GroupArgs::gshape stride");
/* the per-group ShapeInfo written by the pre-pass is constant during the group kernels: scalar loads */
typedef const __attribute__((address_space(4))) ShapeInfo& ShRef;
typedef const __attribute__((address_space(4))) ShapeInfoBig& ShRefBig;
/* side of a square window of A SAIs */
__device__ __forceinline__ int window_side(int A) { int w = 1; while (w * w < A) w++; return w; }

template <class SH>
__device__ void build_shape(SH& sh, const int* m, int aw) {
    const int A = aw * aw;
    int size = 0;
    for (int i = 0; i < A; i++) { sh.mask[i] = m[i]; sh.idx[i] = 0; sh.mask_col[i] = 0; sh.idx_col[i] = 0; sh.mask_dct[i] = 0; size += m[i]; }
    for (int s = 0; s < aw; s++) {
        int r = 0;
        for (int t = 0; t < aw; t++) if (m[s * aw + t]) sh.idx[s * aw + r++] = t;
        sh.row_n[s] = r;
        for (int t = 0; t < r; t++) sh.mask_col[s * aw + t] = 1;
    }
    for (int t = 0; t < aw; t++) {
        int r = 0;
        for (int s = 0; s < aw; s++) if (sh.mask_col[s * aw + t]) sh.idx_col[(r++) * aw + t] = s;
        sh.col_n[t] = r;
        for (int s = 0; s < r; s++) sh.mask_dct[s * aw + t] = 1;
    }
    sh.use_sadct = size != A;
}

/* orthonormalised 3x3 angular DCT as the reference applies it (core:1862-1954) */
__device__ __forceinline__ void dct9_fwd(float* x, TbPtr tb) {
    float t[9];
#pragma unroll
    for (int s = 0; s < 3; s++)
#pragma unroll
        for (int u = 0; u < 3; u++)
            t[s * 3 + u] = 2.0f * (x[s * 3] * tb->cos3[u * 3] + x[s * 3 + 1] * tb->cos3[u * 3 + 1] + x[s * 3 + 2] * tb->cos3[u * 3 + 2]);
#pragma unroll
    for (int v = 0; v < 3; v++)
#pragma unroll
        for (int u = 0; u < 3; u++)
            x[v * 3 + u] = 2.0f * (t[u] * tb->cos3[v * 3] + t[3 + u] * tb->cos3[v * 3 + 1] + t[6 + u] * tb->cos3[v * 3 + 2]) * tb->cn4[v * 3 + u];
}
__device__ __forceinline__ void dct9_inv(float* x, TbPtr tb) {
    float t[9];
#pragma unroll
    for (int i = 0; i < 9; i++) x[i] *= tb->cni4[i];
#pragma unroll
    for (int s = 0; s < 3; s++)
#pragma unroll
        for (int j = 0; j < 3; j++)
            t[s * 3 + j] = x[s * 3] + 2.0f * (x[s * 3 + 1] * tb->cos3[3 + j] + x[s * 3 + 2] * tb->cos3[6 + j]);
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++)
            x[i * 3 + j] = (t[j] + 2.0f * (t[3 + j] * tb->cos3[3 + i] + t[6 + j] * tb->cos3[6 + i])) * tb->coef4inv;
}

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));   /* two values per lane: v_pk_{add,mul,fma}_f32 */
struct __attribute__((packed, aligned(4))) f4u { float v[4]; };   /* 16-byte load at 4-byte alignment */
/* dct9_fwd / dct9_inv on a pair of fibres: the same operation sequence as the scalar versions (folding the
 * constant factors into the stage matrices saves a third of the multiplies but moves results by an ulp, enough
 * to flip the odd hard-threshold decision against the reference) */
__device__ __forceinline__ void dct9_fwd2(v2f* x, TbPtr tb) {
    v2f t[9];
#pragma unroll
    for (int s = 0; s < 3; s++)
#pragma unroll
        for (int u = 0; u < 3; u++)
            t[s * 3 + u] = 2.0f * (x[s * 3] * tb->cos3[u * 3] + x[s * 3 + 1] * tb->cos3[u * 3 + 1] + x[s * 3 + 2] * tb->cos3[u * 3 + 2]);
#pragma unroll
    for (int v = 0; v < 3; v++)
#pragma unroll
        for (int u = 0; u < 3; u++)
            x[v * 3 + u] = 2.0f * (t[u] * tb->cos3[v * 3] + t[3 + u] * tb->cos3[v * 3 + 1] + t[6 + u] * tb->cos3[v * 3 + 2]) * tb->cn4[v * 3 + u];
}
__device__ __forceinline__ void dct9_inv2(v2f* x, TbPtr tb) {
    v2f t[9];
#pragma unroll
    for (int i = 0; i < 9; i++) x[i] *= tb->cni4[i];
#pragma unroll
    for (int s = 0; s < 3; s++)
#pragma unroll
        for (int j = 0; j < 3; j++)
            t[s * 3 + j] = x[s * 3] + 2.0f * (x[s * 3 + 1] * tb->cos3[3 + j] + x[s * 3 + 2] * tb->cos3[6 + j]);
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++)
            x[i * 3 + j] = (t[j] + 2.0f * (t[3 + j] * tb->cos3[3 + i] + t[6 + j] * tb->cos3[6 + i])) * tb->coef4inv;
}
/* The same 3x3 transforms with the table's symmetries used (cos3 = {1, 1, 1; c, 0, -c; 1/2, -1, 1/2}: the table's
 * middle entry is cos(pi/2) in double, 6e-17, taken as 0): less than half the operations, results within an ulp or two of
 * dct9_fwd2 / dct9_inv2.  For the Wiener kernels only -- a hard-threshold decision can flip on an ulp. */
__device__ __forceinline__ void dct9_fwd2_fast(v2f* x, TbPtr tb) {
    const float c2 = 2.0f * tb->cos3[3];
    v2f t[9];
#pragma unroll
    for (int s = 0; s < 3; s++) {
        const v2f a = x[s * 3] + x[s * 3 + 2];
        t[s * 3] = 2.0f * (a + x[s * 3 + 1]);
        t[s * 3 + 1] = c2 * (x[s * 3] - x[s * 3 + 2]);
        t[s * 3 + 2] = a - 2.0f * x[s * 3 + 1];
    }
#pragma unroll
    for (int u = 0; u < 3; u++) {
        const v2f a = t[u] + t[6 + u];
        x[u] = (a + t[3 + u]) * (2.0f * tb->cn4[u]);
        x[3 + u] = (t[u] - t[6 + u]) * (c2 * tb->cn4[3 + u]);
        x[6 + u] = (a - 2.0f * t[3 + u]) * tb->cn4[6 + u];
    }
}
__device__ __forceinline__ void dct9_inv2_fast(v2f* x, TbPtr tb) {
    const float c2 = 2.0f * tb->cos3[3];
    v2f t[9];
#pragma unroll
    for (int s = 0; s < 3; s++) {
        const v2f X0 = x[s * 3] * tb->cni4[s * 3], X1 = x[s * 3 + 1] * (c2 * tb->cni4[s * 3 + 1]), X2 = x[s * 3 + 2] * tb->cni4[s * 3 + 2];
        const v2f p = X0 + X2;
        t[s * 3] = p + X1; t[s * 3 + 1] = X0 - 2.0f * X2; t[s * 3 + 2] = p - X1;
    }
#pragma unroll
    for (int j = 0; j < 3; j++) {
        const v2f T0 = t[j] * tb->coef4inv, T1 = t[3 + j] * (c2 * tb->coef4inv), T2 = t[6 + j] * tb->coef4inv;
        const v2f p = T0 + T2;
        x[j] = p + T1; x[3 + j] = T0 - 2.0f * T2; x[6 + j] = p - T1;
    }
}
/* Haar over the NS patches of a group held as pairs P[h] = {patch h, patch h + NS/2} (lib_transforms.cpp:403-471
 * / :290-321, same butterflies and scaling, evaluated two at a time).  Forward leaves the coefficients in
 * C[0..NS/2) (a permutation of the reference order -- the shrinkage treats all coefficients alike); the inverse
 * takes that layout back to P. */
template <int NS> __device__ __forceinline__ void haar_fwd_pairs(v2f* P) {
    const float s = 0.70710678118654752f;
    if (NS == 8) {
        const v2f S01 = (P[0] + P[1]) * s, D01 = (P[0] - P[1]) * s, S23 = (P[2] + P[3]) * s, D23 = (P[2] - P[3]) * s;
        const v2f SS = (S01 + S23) * s, DD = (S01 - S23) * s;
        P[0] = v2f{(SS.x + SS.y) * s, (SS.x - SS.y) * s}; P[1] = DD; P[2] = D01; P[3] = D23;
    } else if (NS == 4) {
        const v2f S = (P[0] + P[1]) * s, D = (P[0] - P[1]) * s;
        P[0] = v2f{(S.x + S.y) * s, (S.x - S.y) * s}; P[1] = D;
    } else if (NS == 2) {
        P[0] = v2f{(P[0].x + P[0].y) * s, (P[0].x - P[0].y) * s};
    }
}
template <int NS> __device__ __forceinline__ void haar_inv_pairs(v2f* P) {
    const float s = 0.70710678118654752f;
    if (NS == 8) {
        const v2f X = v2f{(P[0].x + P[0].y) * s, (P[0].x - P[0].y) * s};
        const v2f U = (X + P[1]) * s, V = (X - P[1]) * s, D01 = P[2], D23 = P[3];
        P[0] = (U + D01) * s; P[1] = (U - D01) * s; P[2] = (V + D23) * s; P[3] = (V - D23) * s;
    } else if (NS == 4) {
        const v2f X = v2f{(P[0].x + P[0].y) * s, (P[0].x - P[0].y) * s}, D = P[1];
        P[0] = (X + D) * s; P[1] = (X - D) * s;
    } else if (NS == 2) {
        P[0] = v2f{(P[0].x + P[0].y) * s, (P[0].x - P[0].y) * s};
    }
}
/* 1-D REDFT10 / REDFT01 of runtime length n <= 3 (SADCT rows / columns) */
__device__ void r10_small(const float* x, float* y, int n, TbPtr tb) {
    for (int u = 0; u < n; u++) {
        float a = 0.0f;
        for (int j = 0; j < n; j++) a += x[j] * tb->cos1[n][u * n + j];
        y[u] = 2.0f * a;
    }
}
__device__ void r01_small(const float* x, float* y, int n, TbPtr tb) {
    for (int j = 0; j < n; j++) {
        float a = 0.0f;
        for (int u = 1; u < n; u++) a += x[u] * tb->cos1[n][u * n + j];
        y[j] = x[0] + 2.0f * a;
    }
}
/* core:1969-2116 on one 3x3 vector */
__device__ __noinline__ void sadct9_fwd(float* v, ShRef sh, TbPtr tb) {
    float x[3], y[3];
    for (int s = 0; s < 3; s++) {
        const int n = sh.row_n[s];
        if (n == 1) v[s * 3] = v[s * 3 + sh.idx[s * 3]];
        else if (n > 1) {
            for (int t = 0; t < n; t++) x[t] = v[s * 3 + sh.idx[s * 3 + t]];
            r10_small(x, y, n, tb);
            for (int t = 0; t < n; t++) v[s * 3 + t] = y[t] * tb->cn1[n][t];
        }
    }
    for (int t = 0; t < 3; t++) {
        const int n = sh.col_n[t];
        if (n == 1) v[t] = v[sh.idx_col[t] * 3 + t];
        else if (n > 1) {
            for (int s = 0; s < n; s++) x[s] = v[sh.idx_col[s * 3 + t] * 3 + t];
            r10_small(x, y, n, tb);
            for (int s = 0; s < n; s++) v[s * 3 + t] = y[s] * tb->cn1[n][s];
        }
    }
    const float coef = 0.5f * 0.70710678118654752f;
    for (int i = 0; i < 9; i++) v[i] *= (float)sh.mask_dct[i] * coef;
}
/* core:2131-2264 */
__device__ __noinline__ void sadct9_inv(float* v, ShRef sh, TbPtr tb) {
    float x[3], y[3];
    const float coef = 2.0f * 1.41421356237309505f;
    for (int t = 0; t < 3; t++) {
        const int n = sh.col_n[t];
        if (n == 1) v[sh.idx_col[t] * 3 + t] = v[t] * coef;
        else if (n > 1) {
            for (int s = 0; s < n; s++) x[s] = v[s * 3 + t] * tb->cni1[n][s] * coef;
            r01_small(x, y, n, tb);
            for (int s = 0; s < n; s++) v[sh.idx_col[s * 3 + t] * 3 + t] = y[s] * tb->c1inv[n];
        }
    }
    for (int s = 0; s < 3; s++) {
        const int n = sh.row_n[s];
        if (n == 1) v[s * 3 + sh.idx[s * 3]] = v[s * 3];
        else if (n > 1) {
            for (int t = 0; t < n; t++) x[t] = v[s * 3 + t] * tb->cni1[n][t];
            r01_small(x, y, n, tb);
            for (int t = 0; t < n; t++) v[s * 3 + sh.idx[s * 3 + t]] = y[t] * tb->c1inv[n];
        }
    }
    for (int i = 0; i < 9; i++) v[i] *= (float)sh.mask[i];
}
/* The same two transforms for callers that cannot afford a call: inline, the vector in LDS (dynamic indices), the length-n
 * transforms unrolled to three with the terms past n left out -- the same products added in the same order.  (Around a call the
 * 72 live values of the 16x16 kernels' register stage have to sit in the sparse callee-saved registers: 124 VGPRs become 168 and
 * 80 spills for every group, shape-adaptive or not.) */
__device__ __forceinline__ void r10_small3(const float (&x)[3], float (&y)[3], int n, TbPtr tb) {
#pragma unroll
    for (int u = 0; u < 3; u++) {
        float a = 0.0f;
#pragma unroll
        for (int j = 0; j < 3; j++) if (j < n && u < n) a += x[j] * tb->cos1[n][u * n + j];
        y[u] = 2.0f * a;
    }
}
__device__ __forceinline__ void r01_small3(const float (&x)[3], float (&y)[3], int n, TbPtr tb) {
#pragma unroll
    for (int j = 0; j < 3; j++) {
        float a = 0.0f;
#pragma unroll
        for (int u = 1; u < 3; u++) if (u < n && j < n) a += x[u] * tb->cos1[n][u * n + j];
        y[j] = x[0] + 2.0f * a;
    }
}
/* ... and in REGISTERS: a group's shape is uniform, so every index of the transform is a scalar -- the gathers become two selects on
 * scalar conditions, the scatters three conditional moves per destination, the loops over the run lengths predicated code; no
 * dynamic indexing, no call, no scratch.  Same products in the same order as sadct9_fwd / sadct9_inv.  (A window with an empty SAI
 * makes every group shape-adaptive, bm5d.cpp:276-280: with the call form such a 560^2 pass took 13.6 instead of 0.94 ms in the
 * HT group kernel and 9.3 instead of 1.1 ms in the Wiener one.) */
__device__ __forceinline__ float pick3(float a, float b, float c, int i) { return i == 0 ? a : (i == 1 ? b : c); }
/* the length-N transforms with N a compile-time constant (the run length is a scalar: one uniform branch per row / column selects
 * the instance; a full row -- two of three in a window with one empty SAI -- needs no gather at all) */
template <int N> __device__ __forceinline__ void r10_n(const float (&x)[3], float (&y)[3], TbPtr tb) {
#pragma unroll
    for (int u = 0; u < N; u++) {
        float a = 0.0f;
#pragma unroll
        for (int j = 0; j < N; j++) a += x[j] * tb->cos1[N][u * N + j];
        y[u] = 2.0f * a;
    }
}
template <int N> __device__ __forceinline__ void r01_n(const float (&x)[3], float (&y)[3], TbPtr tb) {
#pragma unroll
    for (int j = 0; j < N; j++) {
        float a = 0.0f;
#pragma unroll
        for (int u = 1; u < N; u++) a += x[u] * tb->cos1[N][u * N + j];
        y[j] = x[0] + 2.0f * a;
    }
}
__device__ __forceinline__ void sadct9_fwd_sel(float (&v)[9], ShRef sh, TbPtr tb) {
    float x[3], y[3];
#pragma unroll
    for (int s = 0; s < 3; s++) {
        const int n = sh.row_n[s];
        if (n == 3) {   /* idx = 0, 1, 2 */
            x[0] = v[s * 3]; x[1] = v[s * 3 + 1]; x[2] = v[s * 3 + 2];
            r10_n<3>(x, y, tb);
#pragma unroll
            for (int t = 0; t < 3; t++) v[s * 3 + t] = y[t] * tb->cn1[3][t];
        } else if (n == 2) {
            x[0] = pick3(v[s * 3], v[s * 3 + 1], v[s * 3 + 2], sh.idx[s * 3]);
            x[1] = pick3(v[s * 3], v[s * 3 + 1], v[s * 3 + 2], sh.idx[s * 3 + 1]);
            r10_n<2>(x, y, tb);
            v[s * 3] = y[0] * tb->cn1[2][0]; v[s * 3 + 1] = y[1] * tb->cn1[2][1];
        } else if (n == 1) v[s * 3] = pick3(v[s * 3], v[s * 3 + 1], v[s * 3 + 2], sh.idx[s * 3]);
    }
#pragma unroll
    for (int t = 0; t < 3; t++) {
        const int n = sh.col_n[t];
        if (n == 3) {
            x[0] = v[t]; x[1] = v[3 + t]; x[2] = v[6 + t];
            r10_n<3>(x, y, tb);
#pragma unroll
            for (int s2 = 0; s2 < 3; s2++) v[s2 * 3 + t] = y[s2] * tb->cn1[3][s2];
        } else if (n == 2) {
            x[0] = pick3(v[t], v[3 + t], v[6 + t], sh.idx_col[t]);
            x[1] = pick3(v[t], v[3 + t], v[6 + t], sh.idx_col[3 + t]);
            r10_n<2>(x, y, tb);
            v[t] = y[0] * tb->cn1[2][0]; v[3 + t] = y[1] * tb->cn1[2][1];
        } else if (n == 1) v[t] = pick3(v[t], v[3 + t], v[6 + t], sh.idx_col[t]);
    }
    const float coef = 0.5f * 0.70710678118654752f;
#pragma unroll
    for (int i = 0; i < 9; i++) v[i] *= (float)sh.mask_dct[i] * coef;
}
__device__ __forceinline__ void sadct9_inv_sel(float (&v)[9], ShRef sh, TbPtr tb) {
    float x[3], y[3];
    const float coef = 2.0f * 1.41421356237309505f;
#pragma unroll
    for (int t = 0; t < 3; t++) {
        const int n = sh.col_n[t];
        if (n == 3) {   /* rows 0, 1, 2: in place */
#pragma unroll
            for (int s2 = 0; s2 < 3; s2++) x[s2] = v[s2 * 3 + t] * tb->cni1[3][s2] * coef;
            r01_n<3>(x, y, tb);
#pragma unroll
            for (int s2 = 0; s2 < 3; s2++) v[s2 * 3 + t] = y[s2] * tb->c1inv[3];
        } else if (n >= 1) {
            if (n == 1) { y[0] = v[t] * coef; y[1] = 0.0f; }
            else {
                x[0] = v[t] * tb->cni1[2][0] * coef; x[1] = v[3 + t] * tb->cni1[2][1] * coef;
                r01_n<2>(x, y, tb);
                y[0] *= tb->c1inv[2]; y[1] *= tb->c1inv[2];
            }
            /* v[idx_col[s2][t]][t] = y[s2] for s2 < n: the destinations are distinct rows */
#pragma unroll
            for (int r = 0; r < 3; r++) {
                float w = v[r * 3 + t];
                w = sh.idx_col[t] == r ? y[0] : w;
                w = (n == 2 && sh.idx_col[3 + t] == r) ? y[1] : w;
                v[r * 3 + t] = w;
            }
        }
    }
#pragma unroll
    for (int s = 0; s < 3; s++) {
        const int n = sh.row_n[s];
        if (n == 3) {
#pragma unroll
            for (int t = 0; t < 3; t++) x[t] = v[s * 3 + t] * tb->cni1[3][t];
            r01_n<3>(x, y, tb);
#pragma unroll
            for (int t = 0; t < 3; t++) v[s * 3 + t] = y[t] * tb->c1inv[3];
        } else if (n >= 1) {
            if (n == 1) { y[0] = v[s * 3]; y[1] = 0.0f; }
            else {
                x[0] = v[s * 3] * tb->cni1[2][0]; x[1] = v[s * 3 + 1] * tb->cni1[2][1];
                r01_n<2>(x, y, tb);
                y[0] *= tb->c1inv[2]; y[1] *= tb->c1inv[2];
            }
#pragma unroll
            for (int q = 0; q < 3; q++) {
                float w = v[s * 3 + q];
                w = sh.idx[s * 3] == q ? y[0] : w;
                w = (n == 2 && sh.idx[s * 3 + 1] == q) ? y[1] : w;
                v[s * 3 + q] = w;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 9; i++) v[i] *= (float)sh.mask[i];
}
__device__ __forceinline__ void sadct9_fwd_lds(float* v, ShRef sh, TbPtr tb) {
    float x[3], y[3];
#pragma unroll
    for (int s = 0; s < 3; s++) {
        const int n = sh.row_n[s];
        if (n == 1) v[s * 3] = v[s * 3 + sh.idx[s * 3]];
        else if (n > 1) {
#pragma unroll
            for (int t = 0; t < 3; t++) x[t] = t < n ? v[s * 3 + sh.idx[s * 3 + t]] : 0.0f;
            r10_small3(x, y, n, tb);
#pragma unroll
            for (int t = 0; t < 3; t++) if (t < n) v[s * 3 + t] = y[t] * tb->cn1[n][t];
        }
    }
#pragma unroll
    for (int t = 0; t < 3; t++) {
        const int n = sh.col_n[t];
        if (n == 1) v[t] = v[sh.idx_col[t] * 3 + t];
        else if (n > 1) {
#pragma unroll
            for (int s2 = 0; s2 < 3; s2++) x[s2] = s2 < n ? v[sh.idx_col[s2 * 3 + t] * 3 + t] : 0.0f;
            r10_small3(x, y, n, tb);
#pragma unroll
            for (int s2 = 0; s2 < 3; s2++) if (s2 < n) v[s2 * 3 + t] = y[s2] * tb->cn1[n][s2];
        }
    }
    const float coef = 0.5f * 0.70710678118654752f;
#pragma unroll
    for (int i = 0; i < 9; i++) v[i] *= (float)sh.mask_dct[i] * coef;
}
__device__ __forceinline__ void sadct9_inv_lds(float* v, ShRef sh, TbPtr tb) {
    float x[3], y[3];
    const float coef = 2.0f * 1.41421356237309505f;
#pragma unroll
    for (int t = 0; t < 3; t++) {
        const int n = sh.col_n[t];
        if (n == 1) v[sh.idx_col[t] * 3 + t] = v[t] * coef;
        else if (n > 1) {
#pragma unroll
            for (int s2 = 0; s2 < 3; s2++) x[s2] = s2 < n ? v[s2 * 3 + t] * tb->cni1[n][s2] * coef : 0.0f;
            r01_small3(x, y, n, tb);
#pragma unroll
            for (int s2 = 0; s2 < 3; s2++) if (s2 < n) v[sh.idx_col[s2 * 3 + t] * 3 + t] = y[s2] * tb->c1inv[n];
        }
    }
#pragma unroll
    for (int s = 0; s < 3; s++) {
        const int n = sh.row_n[s];
        if (n == 1) v[s * 3 + sh.idx[s * 3]] = v[s * 3];
        else if (n > 1) {
#pragma unroll
            for (int t = 0; t < 3; t++) x[t] = t < n ? v[s * 3 + t] * tb->cni1[n][t] : 0.0f;
            r01_small3(x, y, n, tb);
#pragma unroll
            for (int t = 0; t < 3; t++) if (t < n) v[s * 3 + sh.idx[s * 3 + t]] = y[t] * tb->c1inv[n];
        }
    }
#pragma unroll
    for (int i = 0; i < 9; i++) v[i] *= (float)sh.mask[i];
}

/* General aw x aw angular window (aswSize 2: 5x5): the same transforms with run-time sizes, generic kernel only.
 * dct_4d_process / dct_4d_inverse (core:1862-1954) and sadct_4d_process / _inverse (core:1969-2264) on one vector. */
/* The angular DCT of a 5x5 / 7x7 window (dct_4d_process / dct_4d_inverse, core:1862-1954) with the window side a compile-time
 * constant: the vector and the intermediate in registers, the loops unrolled.  (Rounds 2-3 had call forms that walked a scratch
 * vector with run-time indices: a 5x5 window's group kernel took 24-30 ms per 304^2 pass, a 7x7 window's 73-78 ms.) */
template <int AW>
__device__ __forceinline__ void dctw_fwd_t(float (&x)[AW * AW], TbPtr tb) {
    float t[AW * AW];
#pragma unroll
    for (int s = 0; s < AW; s++)
#pragma unroll
        for (int u = 0; u < AW; u++) {
            float acc = 0.0f;
#pragma unroll
            for (int j = 0; j < AW; j++) acc += x[s * AW + j] * tb->cosw[u * AW + j];
            t[s * AW + u] = 2.0f * acc;
        }
#pragma unroll
    for (int v = 0; v < AW; v++)
#pragma unroll
        for (int u = 0; u < AW; u++) {
            float acc = 0.0f;
#pragma unroll
            for (int j = 0; j < AW; j++) acc += t[j * AW + u] * tb->cosw[v * AW + j];
            x[v * AW + u] = 2.0f * acc * tb->cn4[v * AW + u];
        }
}
template <int AW>
__device__ __forceinline__ void dctw_inv_t(float (&x)[AW * AW], TbPtr tb) {
    float t[AW * AW];
#pragma unroll
    for (int i = 0; i < AW * AW; i++) x[i] *= tb->cni4[i];
#pragma unroll
    for (int s = 0; s < AW; s++)
#pragma unroll
        for (int j = 0; j < AW; j++) {
            float acc = 0.0f;
#pragma unroll
            for (int u = 1; u < AW; u++) acc += x[s * AW + u] * tb->cosw[u * AW + j];
            t[s * AW + j] = x[s * AW] + 2.0f * acc;
        }
#pragma unroll
    for (int i = 0; i < AW; i++)
#pragma unroll
        for (int j = 0; j < AW; j++) {
            float acc = 0.0f;
#pragma unroll
            for (int v = 1; v < AW; v++) acc += t[v * AW + j] * tb->cosw[v * AW + i];
            x[i * AW + j] = (t[j] + 2.0f * acc) * tb->coef4inv;
        }
}
/* ... and with a run-time window side, on a vector in scratch memory (windows larger than 7x7) */
__device__ __noinline__ void dctw_fwd_rt(float* x, float* t, int aw, TbPtr tb) {
    for (int s = 0; s < aw; s++)
        for (int u = 0; u < aw; u++) {
            float acc = 0.0f;
            for (int j = 0; j < aw; j++) acc += x[s * aw + j] * tb->cosw[u * aw + j];
            t[s * aw + u] = 2.0f * acc;
        }
    for (int v = 0; v < aw; v++)
        for (int u = 0; u < aw; u++) {
            float acc = 0.0f;
            for (int j = 0; j < aw; j++) acc += t[j * aw + u] * tb->cosw[v * aw + j];
            x[v * aw + u] = 2.0f * acc * tb->cn4[v * aw + u];
        }
}
__device__ __noinline__ void dctw_inv_rt(float* x, float* t, int aw, TbPtr tb) {
    for (int i = 0; i < aw * aw; i++) x[i] *= tb->cni4[i];
    for (int s = 0; s < aw; s++)
        for (int j = 0; j < aw; j++) {
            float acc = 0.0f;
            for (int u = 1; u < aw; u++) acc += x[s * aw + u] * tb->cosw[u * aw + j];
            t[s * aw + j] = x[s * aw] + 2.0f * acc;
        }
    for (int i = 0; i < aw; i++)
        for (int j = 0; j < aw; j++) {
            float acc = 0.0f;
            for (int v = 1; v < aw; v++) acc += t[v * aw + j] * tb->cosw[v * aw + i];
            x[i * aw + j] = (t[j] + 2.0f * acc) * tb->coef4inv;
        }
}
template <class SHR>
__device__ __noinline__ void sadctw_fwd(float* v, int aw, SHR sh, TbPtr tb) {
    float x[kBigAw], y[kBigAw];
    for (int s = 0; s < aw; s++) {
        const int n = sh.row_n[s];
        if (n == 1) v[s * aw] = v[s * aw + sh.idx[s * aw]];
        else if (n > 1) {
            for (int t = 0; t < n; t++) x[t] = v[s * aw + sh.idx[s * aw + t]];
            r10_small(x, y, n, tb);
            for (int t = 0; t < n; t++) v[s * aw + t] = y[t] * tb->cn1[n][t];
        }
    }
    for (int t = 0; t < aw; t++) {
        const int n = sh.col_n[t];
        if (n == 1) v[t] = v[sh.idx_col[t] * aw + t];
        else if (n > 1) {
            for (int s = 0; s < n; s++) x[s] = v[sh.idx_col[s * aw + t] * aw + t];
            r10_small(x, y, n, tb);
            for (int s = 0; s < n; s++) v[s * aw + t] = y[s] * tb->cn1[n][s];
        }
    }
    const float coef = 0.5f * 0.70710678118654752f;
    for (int i = 0; i < aw * aw; i++) v[i] *= (float)sh.mask_dct[i] * coef;
}
template <class SHR>
__device__ __noinline__ void sadctw_inv(float* v, int aw, SHR sh, TbPtr tb) {
    float x[kBigAw], y[kBigAw];
    const float coef = 2.0f * 1.41421356237309505f;
    for (int t = 0; t < aw; t++) {
        const int n = sh.col_n[t];
        if (n == 1) v[sh.idx_col[t] * aw + t] = v[t] * coef;
        else if (n > 1) {
            for (int s = 0; s < n; s++) x[s] = v[s * aw + t] * tb->cni1[n][s] * coef;
            r01_small(x, y, n, tb);
            for (int s = 0; s < n; s++) v[sh.idx_col[s * aw + t] * aw + t] = y[s] * tb->c1inv[n];
        }
    }
    for (int s = 0; s < aw; s++) {
        const int n = sh.row_n[s];
        if (n == 1) v[s * aw + sh.idx[s * aw]] = v[s * aw];
        else if (n > 1) {
            for (int t = 0; t < n; t++) x[t] = v[s * aw + t] * tb->cni1[n][t];
            r01_small(x, y, n, tb);
            for (int t = 0; t < n; t++) v[s * aw + sh.idx[s * aw + t]] = y[t] * tb->c1inv[n];
        }
    }
    for (int i = 0; i < aw * aw; i++) v[i] *= (float)sh.mask[i];
}

/* lib_transforms.cpp:403-471 / :290-321 on a register vector of compile-time length */
template <int NS> __device__ __forceinline__ void haar_fwd(float* v) {
    const float s = 0.70710678118654752f;
#pragma unroll
    for (int n = NS; n > 1; n /= 2) {
        float t[NS > 1 ? NS : 1];
#pragma unroll
        for (int i = 0; i < n / 2; i++) { t[i] = (v[2 * i] + v[2 * i + 1]) * s; t[n / 2 + i] = (v[2 * i] - v[2 * i + 1]) * s; }
#pragma unroll
        for (int i = 0; i < n; i++) v[i] = t[i];
    }
}
template <int NS> __device__ __forceinline__ void haar_inv(float* v) {
    const float s = 0.70710678118654752f;
#pragma unroll
    for (int h = 1; h < NS; h *= 2) {
        float t[NS > 1 ? NS : 1];
#pragma unroll
        for (int i = 0; i < h; i++) { t[2 * i] = (v[i] + v[h + i]) * s; t[2 * i + 1] = (v[i] - v[h + i]) * s; }
#pragma unroll
        for (int i = 0; i < 2 * h; i++) v[i] = t[i];
    }
}
template <int NS> __device__ __forceinline__ void hadamard(float* v) {
    /* sums to the first half, differences to the second, recurse on both: log2(NS) levels of
     * the same butterfly applied block-wise */
#pragma unroll
    for (int n = NS; n > 1; n /= 2) {
        float t[NS > 1 ? NS : 1];
#pragma unroll
        for (int b = 0; b < NS; b += n)
#pragma unroll
            for (int i = 0; i < n / 2; i++) { t[b + i] = v[b + 2 * i] + v[b + 2 * i + 1]; t[b + n / 2 + i] = v[b + 2 * i] - v[b + 2 * i + 1]; }
#pragma unroll
        for (int i = 0; i < NS; i++) v[i] = t[i];
    }
}

/* 5th-dimension filter of one (st, pq) fibre held in registers.
 * HT: core:2408-2505 / :2281-2391; Wiener: core:2826-2925 / :2706-2810. */
/* 5th-dimension DCT of a fibre (tau_5D = dct): REDFT10 * coef_norm / coef_norm_inv * REDFT01 * coef
 * (core:2546-2593, norms preProcess_5d core:3262-3276) */
template <int NS> __device__ __forceinline__ int log2c() { return NS == 1 ? 0 : NS == 2 ? 1 : NS == 4 ? 2 : NS == 8 ? 3 : NS == 16 ? 4 : 5; }
template <int NS> __device__ __forceinline__ void dct5_fwd(float* v, TbPtr tb) {
    TbFloats ct = NS == 32 ? tb->cos5x : tb->cos5[NS == 32 ? 0 : log2c<NS>()];
    float y[NS];
#pragma unroll
    for (int u = 0; u < NS; u++) {
        float a = 0.0f;
#pragma unroll
        for (int j = 0; j < NS; j++) a += v[j] * ct[u * NS + j];
        y[u] = 2.0f * a * (u == 0 ? tb->cn5_0[log2c<NS>()] : tb->cn5[log2c<NS>()]);
    }
#pragma unroll
    for (int u = 0; u < NS; u++) v[u] = y[u];
}
template <int NS> __device__ __forceinline__ void dct5_inv(float* v, TbPtr tb) {
    TbFloats ct = NS == 32 ? tb->cos5x : tb->cos5[NS == 32 ? 0 : log2c<NS>()];
    float y[NS];
    const float x0 = v[0] * 1.41421356237309505f;   /* coef_norm_inv[0] = sqrt2, others 1 */
#pragma unroll
    for (int j = 0; j < NS; j++) {
        float a = 0.0f;
#pragma unroll
        for (int u = 1; u < NS; u++) a += v[u] * ct[u * NS + j];
        y[j] = (x0 + 2.0f * a) * tb->c5inv[log2c<NS>()];
    }
#pragma unroll
    for (int j = 0; j < NS; j++) v[j] = y[j];
}

/* 5th-dimension transform + shrinkage + inverse of one (st, pq) fibre held in registers.
 * o: noisy fibre, e: pilot fibre (Wiener); the filtered fibre is returned in o (HT) / e (Wiener).
 * HT: core:2408-2505 / :2281-2391; Wiener: core:2826-2925 / :2706-2810. */
template <int NS, int STEP>
__device__ __forceinline__ void shrink_fibre(float* o, float* e, unsigned tau5, float T, float sig2,
                                             bool in_shape, float& wacc, TbPtr tb) {
    const bool haar = tau5 == 9, dct = tau5 == 5;
    if (dct) { dct5_fwd<NS>(o, tb); if (STEP == 2) dct5_fwd<NS>(e, tb); }
    else if (NS > 1) {
        if (haar) { haar_fwd<NS>(o); if (STEP == 2) haar_fwd<NS>(e); }
        else      { hadamard<NS>(o); if (STEP == 2) hadamard<NS>(e); }
    }
    if (in_shape) {
        if (STEP == 1) {
            /* T = lambda*sigma*sqrt2; Hadamard: * sqrt(nSx) (core:2306); DCT: * 2 (core:2567) */
            const float Th = haar ? T : (dct ? T * 2.0f : T * sqrtf((float)NS));
#pragma unroll
            for (int n = 0; n < NS; n++) {
                if (fabsf(o[n]) > Th) wacc += 1.0f; else o[n] = 0.0f;
            }
        } else {
            const float hc = 1.0f / (float)NS;
            const bool plain = haar || dct;
#pragma unroll
            for (int n = 0; n < NS; n++) {
                float value = plain ? e[n] * e[n] : e[n] * e[n] * hc;
                value = __fdiv_rn(value, value + sig2);
                e[n] = plain ? o[n] * value : o[n] * value * hc;
                wacc += value;
            }
        }
    }
    float* r = STEP == 1 ? o : e;
    if (dct) dct5_inv<NS>(r, tb);
    else if (NS > 1) {
        if (haar) haar_inv<NS>(r);
        else {
            hadamard<NS>(r);
            if (STEP == 1) {
                const float hc = 1.0f / (float)NS;
#pragma unroll
                for (int n = 0; n < NS; n++) r[n] *= hc;
            }
        }
    }
}

/* the same on a fibre stored in the LDS stack */
template <int NS, int STEP>
__device__ __forceinline__ void filter5(float* S0, float* S1, int base, int stride, unsigned tau5,
                                        float T, float sig2, bool in_shape, float& wacc, float& s1, float& s2,
                                        TbPtr tb) {
    float o[NS], e[NS];
#pragma unroll
    for (int n = 0; n < NS; n++) o[n] = S0[base + n * stride];
    if (STEP == 2) {
#pragma unroll
        for (int n = 0; n < NS; n++) e[n] = S1[base + n * stride];
    }
    shrink_fibre<NS, STEP>(o, e, tau5, T, sig2, in_shape, wacc, tb);
    float* r = STEP == 1 ? o : e;
    float* dst = STEP == 1 ? S0 : S1;
#pragma unroll
    for (int n = 0; n < NS; n++) { dst[base + n * stride] = r[n]; s1 += r[n]; s2 += r[n] * r[n]; }
}

__device__ __forceinline__ int per_ext(int j, int L, int N) { int m = (j - L) % N; return m < 0 ? m + N : m; }

/* ------------------------------------------------------------------------------------------
 * bior1.5 2-D transform of K x K patches (K = 8, 16; lib_transforms.cpp:46-120 forward, :135-204 inverse), fast
 * path: K threads per patch, thread = one row in the row passes and one column in the column passes, the row /
 * column held in registers, all periodic-extension indices resolved at compile time.  A patch is copied from
 * the stack into a work area laid out [patch][K][K+1] (the odd row stride makes both access directions free of
 * LDS bank conflicts), transformed there through all levels and copied back.  The K threads of a patch sit in
 * one wavefront, whose DS operations execute in order: no workgroup barrier inside.  Same taps in the same
 * order as the generic path -- identical results.
 * ------------------------------------------------------------------------------------------ */
constexpr int bior_ext(int j, int N) { return (((j - 4) % N) + N) % N; }   /* per_ext(j, 4, N) */

template <int K, int N1>
__device__ __forceinline__ void bior_fwd_level(float* Tp, int r, TbPtr tb) {
#pragma clang fp contract(off)   /* the reference's separate multiply and add: bit-identical coefficients */
    if constexpr (N1 > 1) {
        constexpr int N2 = N1 / 2, RS = K + 1;
        if (r < N1) {   /* rows: first N2 outputs low-pass, next N2 high-pass */
            float v[N1], o[N1];
#pragma unroll
            for (int c = 0; c < N1; c++) v[c] = Tp[r * RS + c];
#pragma unroll
            for (int j = 0; j < N1; j++) {
                const int jj = j < N2 ? j : j - N2;
                float acc = 0.0f;
                if (j < N2) {
#pragma unroll
                    for (int t = 0; t < 10; t++) acc += v[bior_ext(t + 2 * jj, N1)] * tb->lpd[t];
                } else {   /* the high-pass analysis filter has two taps (lib_transforms.cpp:215-277); the eight products with
                            * its zero taps only ever add a zero */
                    acc = v[bior_ext(4 + 2 * jj, N1)] * tb->hpd[4];
                    acc += v[bior_ext(5 + 2 * jj, N1)] * tb->hpd[5];
                }
                o[j] = acc;
            }
#pragma unroll
            for (int c = 0; c < N1; c++) Tp[r * RS + c] = o[c];
        }
        __builtin_amdgcn_wave_barrier();
        if (r < N1) {   /* columns (thread = column r) */
            float v[N1], o[N1];
#pragma unroll
            for (int i = 0; i < N1; i++) v[i] = Tp[i * RS + r];
#pragma unroll
            for (int i = 0; i < N1; i++) {
                const int ii = i < N2 ? i : i - N2;
                float acc = 0.0f;
                if (i < N2) {
#pragma unroll
                    for (int t = 0; t < 10; t++) acc += v[bior_ext(t + 2 * ii, N1)] * tb->lpd[t];
                } else {
                    acc = v[bior_ext(4 + 2 * ii, N1)] * tb->hpd[4];
                    acc += v[bior_ext(5 + 2 * ii, N1)] * tb->hpd[5];
                }
                o[i] = acc;
            }
#pragma unroll
            for (int i = 0; i < N1; i++) Tp[i * RS + r] = o[i];
        }
        __builtin_amdgcn_wave_barrier();
        bior_fwd_level<K, N1 / 2>(Tp, r, tb);
    }
}
template <int K, int N1>
__device__ __forceinline__ void bior_inv_level(float* Tp, int r, TbPtr tb) {
#pragma clang fp contract(off)
    if constexpr (N1 <= K) {
        constexpr int N2 = N1 / 2, RS = K + 1;
        if (r < N1) {   /* columns: out[2m] from the high-pass taps, out[2m+1] from the low-pass taps */
            float v[N1], o[N1];
#pragma unroll
            for (int i = 0; i < N1; i++) v[i] = Tp[i * RS + r];
#pragma unroll
            for (int i = 0; i < N1; i++) {
                const int m = i / 2;
                float acc = 0.0f;
                if (i & 1) {   /* the low-pass synthesis filter has two taps */
                    acc = tb->lpr[4] * v[(4 * N2 + m) % N1];
                    acc += tb->lpr[5] * v[(5 * N2 + m) % N1];
                } else {
#pragma unroll
                    for (int t = 0; t < 10; t++) acc += tb->hpr[t] * v[(t * N2 + m) % N1];
                }
                o[i] = acc;
            }
#pragma unroll
            for (int i = 0; i < N1; i++) Tp[i * RS + r] = o[i];
        }
        __builtin_amdgcn_wave_barrier();
        if (r < N1) {   /* rows */
            float v[N1], o[N1];
#pragma unroll
            for (int c = 0; c < N1; c++) v[c] = Tp[r * RS + c];
#pragma unroll
            for (int j = 0; j < N1; j++) {
                const int m = j / 2;
                float acc = 0.0f;
                if (j & 1) {
                    acc = tb->lpr[4] * v[(4 * N2 + m) % N1];
                    acc += tb->lpr[5] * v[(5 * N2 + m) % N1];
                } else {
#pragma unroll
                    for (int t = 0; t < 10; t++) acc += tb->hpr[t] * v[(t * N2 + m) % N1];
                }
                o[j] = acc;
            }
#pragma unroll
            for (int c = 0; c < N1; c++) Tp[r * RS + c] = o[c];
        }
        __builtin_amdgcn_wave_barrier();
        bior_inv_level<K, N1 * 2>(Tp, r, tb);
    }
}
/* Floats per patch of the 16x16 kernels' work area [patch][16][17]: 16 * 17 = 272 = 16 (mod 32) put every other patch of a wave on the
 * same banks -- two-way conflicts in every pass of the 16x16 level, 62 % of the LDS-active cycles in conflicts
 * (profiles/r04_a_sq_counters.txt).  280 = 8 (mod 32): the four patches of a half-wave (eight lanes each) start 0 / 8 / 16 / 24 banks
 * apart, and rows 17 r + c as well as columns r + 17 c of eight lanes then fall on 32 distinct banks; two workgroups of 72 patches
 * still fit a CU (2 x 80 640 B). */
constexpr int kT16Patch = 16 * 17 + 8;
/* The same levels for the 16x16 kernel below, two rows (then two columns) per thread as packed pairs: a level of
 * side N1 takes N1/2 threads per patch, rows r and r + N1/2 travel as one v2f (one ds_read2 / ds_write2 per
 * element: the partner sits N1/2 rows, or N1/2 floats, away), every tap is one packed multiply or add.  Same
 * taps, order and unfused arithmetic as bior_fwd_level / bior_inv_level -> identical coefficients. */
template <int N1, bool FWD>
__device__ __forceinline__ void bior_taps2(const v2f* v, v2f* o, TbPtr tb) {
#pragma clang fp contract(off)
    constexpr int N2 = N1 / 2;
    if (FWD) {
#pragma unroll
        for (int j = 0; j < N2; j++) {
            v2f acc = v[bior_ext(2 * j, N1)] * tb->lpd[0];
#pragma unroll
            for (int t = 1; t < 10; t++) acc += v[bior_ext(t + 2 * j, N1)] * tb->lpd[t];
            o[j] = acc;
            v2f hi = v[bior_ext(4 + 2 * j, N1)] * tb->hpd[4];
            hi += v[bior_ext(5 + 2 * j, N1)] * tb->hpd[5];
            o[N2 + j] = hi;
        }
    } else {
#pragma unroll
        for (int m = 0; m < N2; m++) {
            v2f acc = v[m % N1] * tb->hpr[0];
#pragma unroll
            for (int t = 1; t < 10; t++) acc += v[(t * N2 + m) % N1] * tb->hpr[t];
            o[2 * m] = acc;
            v2f lo = v[(4 * N2 + m) % N1] * tb->lpr[4];
            lo += v[(5 * N2 + m) % N1] * tb->lpr[5];
            o[2 * m + 1] = lo;
        }
    }
}
template <int N1, bool FWD, bool ROWS>
__device__ __forceinline__ void bior16_pass2(float* Tp, int r, TbPtr tb) {
    constexpr int N2 = N1 / 2, RS = 17;
    v2f v[N1], o[N1];
#pragma unroll
    for (int c = 0; c < N1; c++)
        v[c] = ROWS ? v2f{Tp[r * RS + c], Tp[(r + N2) * RS + c]} : v2f{Tp[c * RS + r], Tp[c * RS + r + N2]};
    bior_taps2<N1, FWD>(v, o, tb);
#pragma unroll
    for (int c = 0; c < N1; c++) {
        if (ROWS) { Tp[r * RS + c] = o[c].x; Tp[(r + N2) * RS + c] = o[c].y; }
        else      { Tp[c * RS + r] = o[c].x; Tp[c * RS + r + N2] = o[c].y; }
    }
}
/* one level of all NP patches of the work area [patch][16][17]; all 256 threads call it */
template <int N1, bool FWD>
__device__ __forceinline__ void bior16_level_all(float* work, int NP, int tid, TbPtr tb) {
    constexpr int TPP = N1 / 2, PPI = kThreads / TPP, PSZ = kT16Patch;   /* threads per patch (one wavefront holds them all) */
    const int slot = tid / TPP, r = tid % TPP;
    for (int p0 = 0; p0 < NP; p0 += PPI) {
        const int patch = p0 + slot;
        if (patch < NP) {
            float* Tp = work + patch * PSZ;
            bior16_pass2<N1, FWD, FWD>(Tp, r, tb);         /* forward: rows first; inverse: columns first */
            __builtin_amdgcn_wave_barrier();
            bior16_pass2<N1, FWD, !FWD>(Tp, r, tb);
        }
    }
    __syncthreads();   /* the next level deals the patches to other threads */
}
/* work area: (kThreads / K) patches of K x (K+1) floats */
template <int K> constexpr int bior_tmp_floats() { return (kThreads / K) * K * (K + 1); }
template <int K, bool FWD>
__device__ void bior2d_fast(float* S, float* tmp, int np, TbPtr tb) {
    constexpr int PPI = kThreads / K, RS = K + 1;
    const int tid = threadIdx.x, slot = tid / K, r = tid % K;
    float* Tp = tmp + slot * K * RS;
    for (int p0 = 0; p0 < np; p0 += PPI) {
        const int patch = p0 + slot;
        if (patch < np) {     /* uniform for the K threads of a patch */
            float* X = S + (size_t)patch * K * K + r * K;
            float x[K];
#pragma unroll
            for (int c4 = 0; c4 < K; c4 += 4) {
                const v4f q = *reinterpret_cast<const v4f*>(X + c4);
                x[c4] = q[0]; x[c4 + 1] = q[1]; x[c4 + 2] = q[2]; x[c4 + 3] = q[3];
            }
#pragma unroll
            for (int c = 0; c < K; c++) Tp[r * RS + c] = x[c];
            __builtin_amdgcn_wave_barrier();
            if (FWD) bior_fwd_level<K, K>(Tp, r, tb); else bior_inv_level<K, 2>(Tp, r, tb);
#pragma unroll
            for (int c = 0; c < K; c++) x[c] = Tp[r * RS + c];
#pragma unroll
            for (int c4 = 0; c4 < K; c4 += 4) *reinterpret_cast<v4f*>(X + c4) = v4f{x[c4], x[c4 + 1], x[c4 + 2], x[c4 + 3]};
            __builtin_amdgcn_wave_barrier();
        }
    }
    __syncthreads();
}

/* 2-D forward transform of `np` patches stored back to back at S (k*k floats each).  All threads
 * of the workgroup call this. */

/* barrier between the two passes of a patch: when a patch is exactly one wavefront (k = 8) the LDS
 * traffic of a patch stays inside that wave, whose DS operations execute in order -- no workgroup
 * barrier needed */
#define PATCH_SYNC() do { if (wave_local) __builtin_amdgcn_wave_barrier(); else __syncthreads(); } while (0)

/* 2-D DCT of all patches with the per-thread table entries held in registers (thread = coefficient
 * (i,j) of a patch; its cosine rows never change from patch to patch) */
template <int K>
__device__ void fwd2d_dct(float* S, float* tmp, int np, TbPtr tb) {
    constexpr int K2 = K * K;
    const int tid = threadIdx.x;
    const bool wave_local = K2 == 64;
    constexpr int ppi = kThreads / K2 > 0 ? kThreads / K2 : 1;
    const int slot = tid / K2, pq = tid % K2, i = pq / K, j = pq % K;
    float cj[K], ci[K];
#pragma unroll
    for (int t = 0; t < K; t++) { cj[t] = tb->cos2[j * K + t]; ci[t] = tb->cos2[i * K + t]; }
    const float cn = tb->cn2[pq];
    float* Tm = tmp + slot * K2;
    for (int p0 = 0; p0 < np; p0 += ppi) {
        const int patch = p0 + slot;
        const bool on = slot < ppi && patch < np;
        float* X = S + (size_t)patch * K2;
        if (on) { float a = 0.0f;
#pragma unroll
            for (int t = 0; t < K; t++) a += X[i * K + t] * cj[t];
            Tm[pq] = 2.0f * a; }
        PATCH_SYNC();
        if (on) { float a = 0.0f;
#pragma unroll
            for (int t = 0; t < K; t++) a += Tm[t * K + j] * ci[t];
            X[pq] = 2.0f * a * cn; }
        PATCH_SYNC();
    }
    __syncthreads();
}
template <int K>
__device__ void inv2d_dct(float* S, float* tmp, int np, TbPtr tb) {
    constexpr int K2 = K * K;
    const int tid = threadIdx.x;
    const bool wave_local = K2 == 64;
    constexpr int ppi = kThreads / K2 > 0 ? kThreads / K2 : 1;
    const int slot = tid / K2, pq = tid % K2, i = pq / K, j = pq % K;
    float cc[K], ni[K], cu[K];
#pragma unroll
    for (int t = 0; t < K; t++) { cc[t] = tb->cos2[t * K + j]; ni[t] = tb->cni2[i * K + t]; cu[t] = tb->cos2[t * K + i]; }
    const float c2 = tb->coef2inv;
    float* Tm = tmp + slot * K2;
    for (int p0 = 0; p0 < np; p0 += ppi) {
        const int patch = p0 + slot;
        const bool on = slot < ppi && patch < np;
        float* X = S + (size_t)patch * K2;
        if (on) { float a = 0.0f;
#pragma unroll
            for (int v = 1; v < K; v++) a += X[i * K + v] * ni[v] * cc[v];
            Tm[pq] = X[i * K] * ni[0] + 2.0f * a; }
        PATCH_SYNC();
        if (on) { float a = 0.0f;
#pragma unroll
            for (int u = 1; u < K; u++) a += Tm[u * K + j] * cu[u];
            X[pq] = c2 * (Tm[j] + 2.0f * a); }
        PATCH_SYNC();
    }
    __syncthreads();
}

/* patches of more coefficients than the workgroup has threads (k > 16): one patch at a time, the threads stride over its coefficients;
 * the same sums in the same order as the general form below */
__device__ void fwd2d_big(float* S, float* Tm, int np, int k, unsigned tau2, TbPtr tb) {
    const int k2 = k * k, tid = threadIdx.x;
    for (int patch = 0; patch < np; patch++) {
        float* X = S + (size_t)patch * k2;
        if (tau2 == 5) {
            for (int pq = tid; pq < k2; pq += kThreads) { const int i = pq / k, j = pq % k; float a = 0.0f; for (int t = 0; t < k; t++) a += X[i * k + t] * tb->cos2[j * k + t]; Tm[pq] = 2.0f * a; }
            __syncthreads();
            for (int pq = tid; pq < k2; pq += kThreads) { const int i = pq / k, j = pq % k; float a = 0.0f; for (int t = 0; t < k; t++) a += Tm[t * k + j] * tb->cos2[i * k + t]; X[pq] = 2.0f * a * tb->cn2[pq]; }
            __syncthreads();
        } else {
            for (int N1 = k; N1 > 1; N1 /= 2) {
                const int N2 = N1 / 2;
                for (int pq = tid; pq < k2; pq += kThreads) {
                    const int i = pq / k, j = pq % k;
                    if (i < N1 && j < N1) {
                        const bool lo = j < N2; const int jj = lo ? j : j - N2;
                        TbFloats f = lo ? tb->lpd : tb->hpd;
                        float a = 0.0f;
                        for (int t = 0; t < 10; t++) a += X[i * k + per_ext(t + 2 * jj, 4, N1)] * f[t];
                        Tm[pq] = a;
                    }
                }
                __syncthreads();
                for (int pq = tid; pq < k2; pq += kThreads) {
                    const int i = pq / k, j = pq % k;
                    if (i < N1 && j < N1) {
                        const bool lo = i < N2; const int ii = lo ? i : i - N2;
                        TbFloats f = lo ? tb->lpd : tb->hpd;
                        float a = 0.0f;
                        for (int t = 0; t < 10; t++) a += Tm[per_ext(t + 2 * ii, 4, N1) * k + j] * f[t];
                        X[pq] = a;
                    }
                }
                __syncthreads();
            }
        }
    }
    __syncthreads();
}
__device__ void inv2d_big(float* S, float* Tm, int np, int k, unsigned tau2, TbPtr tb) {
    const int k2 = k * k, tid = threadIdx.x;
    for (int patch = 0; patch < np; patch++) {
        float* X = S + (size_t)patch * k2;
        if (tau2 == 5) {
            for (int pq = tid; pq < k2; pq += kThreads) {
                const int i = pq / k, j = pq % k;
                float a = 0.0f;
                for (int v = 1; v < k; v++) a += X[i * k + v] * tb->cni2[i * k + v] * tb->cos2[v * k + j];
                Tm[pq] = X[i * k] * tb->cni2[i * k] + 2.0f * a;
            }
            __syncthreads();
            for (int pq = tid; pq < k2; pq += kThreads) {
                const int i = pq / k, j = pq % k;
                float a = 0.0f;
                for (int u = 1; u < k; u++) a += Tm[u * k + j] * tb->cos2[u * k + i];
                X[pq] = tb->coef2inv * (Tm[j] + 2.0f * a);
            }
            __syncthreads();
        } else {
            for (int N1 = 2; N1 <= k; N1 *= 2) {
                const int N2 = N1 / 2;
                for (int pq = tid; pq < k2; pq += kThreads) {
                    const int i = pq / k, j = pq % k;
                    if (i < N1 && j < N1) {
                        const int m = i / 2; TbFloats f = (i & 1) ? tb->lpr : tb->hpr;
                        float a = 0.0f;
                        for (int t = 0; t < 10; t++) a += f[t] * X[((t * N2 + m) % N1) * k + j];
                        Tm[pq] = a;
                    }
                }
                __syncthreads();
                for (int pq = tid; pq < k2; pq += kThreads) {
                    const int i = pq / k, j = pq % k;
                    if (i < N1 && j < N1) {
                        const int m = j / 2; TbFloats f = (j & 1) ? tb->lpr : tb->hpr;
                        float a = 0.0f;
                        for (int t = 0; t < 10; t++) a += f[t] * Tm[i * k + (t * N2 + m) % N1];
                        X[pq] = a;
                    }
                }
                __syncthreads();
            }
        }
    }
    __syncthreads();
}

__device__ void fwd2d(float* S, float* tmp, int np, int k, unsigned tau2, TbPtr tb) {
    if (k * k > kThreads) return fwd2d_big(S, tmp, np, k, tau2, tb);
    if (tau2 == 5) {
        if (k == 8) return fwd2d_dct<8>(S, tmp, np, tb);
        if (k == 12) return fwd2d_dct<12>(S, tmp, np, tb);
        if (k == 16) return fwd2d_dct<16>(S, tmp, np, tb);
    }   /* (other sizes: the run-time form below) */
    if (tau2 == 7 && k == 16) return bior2d_fast<16, true>(S, tmp, np, tb);
    if (tau2 == 7 && k == 8) return bior2d_fast<8, true>(S, tmp, np, tb);
    const int k2 = k * k, tid = threadIdx.x;
    const bool wave_local = k2 == 64;
    const int ppi = kThreads / k2 > 0 ? kThreads / k2 : 1; /* patches per iteration */
    for (int p0 = 0; p0 < np; p0 += ppi) {
        const int slot = tid / k2, pq = tid % k2, patch = p0 + slot;
        const bool on = slot < ppi && patch < np;
        float* X = S + (size_t)patch * k2;
        float* Tm = tmp + slot * k2;
        const int i = pq / k, j = pq % k;
        if (tau2 == 5) { /* DCT: REDFT10 rows, REDFT10 columns, * coef_norm (bm3d.cpp:745-757) */
            if (on) { float a = 0.0f; for (int t = 0; t < k; t++) a += X[i * k + t] * tb->cos2[j * k + t]; Tm[pq] = 2.0f * a; }
            PATCH_SYNC();
            if (on) { float a = 0.0f; for (int t = 0; t < k; t++) a += Tm[t * k + j] * tb->cos2[i * k + t]; X[pq] = 2.0f * a * tb->cn2[pq]; }
            PATCH_SYNC();
        } else {         /* bior1.5 (lib_transforms.cpp:46-120) */
            for (int N1 = k; N1 > 1; N1 /= 2) {
                const int N2 = N1 / 2;
                if (on && i < N1 && j < N1) {
                    const bool lo = j < N2; const int jj = lo ? j : j - N2;
                    TbFloats f = lo ? tb->lpd : tb->hpd;
                    float a = 0.0f;
                    for (int t = 0; t < 10; t++) a += X[i * k + per_ext(t + 2 * jj, 4, N1)] * f[t];
                    Tm[pq] = a;
                }
                PATCH_SYNC();
                if (on && i < N1 && j < N1) {
                    const bool lo = i < N2; const int ii = lo ? i : i - N2;
                    TbFloats f = lo ? tb->lpd : tb->hpd;
                    float a = 0.0f;
                    for (int t = 0; t < 10; t++) a += Tm[per_ext(t + 2 * ii, 4, N1) * k + j] * f[t];
                    X[pq] = a;
                }
                PATCH_SYNC();
            }
        }
    }
    __syncthreads();
}
__device__ void inv2d(float* S, float* tmp, int np, int k, unsigned tau2, TbPtr tb) {
    if (k * k > kThreads) return inv2d_big(S, tmp, np, k, tau2, tb);
    if (tau2 == 5) {
        if (k == 8) return inv2d_dct<8>(S, tmp, np, tb);
        if (k == 12) return inv2d_dct<12>(S, tmp, np, tb);
        if (k == 16) return inv2d_dct<16>(S, tmp, np, tb);
    }
    if (tau2 == 7 && k == 16) return bior2d_fast<16, false>(S, tmp, np, tb);
    if (tau2 == 7 && k == 8) return bior2d_fast<8, false>(S, tmp, np, tb);
    const int k2 = k * k, tid = threadIdx.x;
    const bool wave_local = k2 == 64;
    const int ppi = kThreads / k2 > 0 ? kThreads / k2 : 1;
    for (int p0 = 0; p0 < np; p0 += ppi) {
        const int slot = tid / k2, pq = tid % k2, patch = p0 + slot;
        const bool on = slot < ppi && patch < np;
        float* X = S + (size_t)patch * k2;
        float* Tm = tmp + slot * k2;
        const int i = pq / k, j = pq % k;
        if (tau2 == 5) { /* bm3d.cpp:1039-1071 */
            if (on) {
                float a = 0.0f;
                for (int v = 1; v < k; v++) a += X[i * k + v] * tb->cni2[i * k + v] * tb->cos2[v * k + j];
                Tm[pq] = X[i * k] * tb->cni2[i * k] + 2.0f * a;
            }
            PATCH_SYNC();
            if (on) {
                float a = 0.0f;
                for (int u = 1; u < k; u++) a += Tm[u * k + j] * tb->cos2[u * k + i];
                X[pq] = tb->coef2inv * (Tm[j] + 2.0f * a);
            }
            PATCH_SYNC();
        } else { /* lib_transforms.cpp:135-204 */
            for (int N1 = 2; N1 <= k; N1 *= 2) {
                const int N2 = N1 / 2;
                if (on && i < N1 && j < N1) { /* columns: out[2m] = high, out[2m+1] = low */
                    const int m = i / 2; TbFloats f = (i & 1) ? tb->lpr : tb->hpr;
                    float a = 0.0f;
                    for (int t = 0; t < 10; t++) a += f[t] * X[((t * N2 + m) % N1) * k + j];
                    Tm[pq] = a;
                }
                PATCH_SYNC();
                if (on && i < N1 && j < N1) { /* rows */
                    const int m = j / 2; TbFloats f = (j & 1) ? tb->lpr : tb->hpr;
                    float a = 0.0f;
                    for (int t = 0; t < 10; t++) a += f[t] * Tm[i * k + (t * N2 + m) % N1];
                    X[pq] = a;
                }
                PATCH_SYNC();
            }
        }
    }
    __syncthreads();
}

/* ------------------------------------------------------------------------------------------
 * Group geometry pre-pass (core:286-323, :503): for every group of the launch, the window position
 * of each of its N x A patches (0xffffffff: no patch), the positions the aggregation kernel adds
 * them at, and the 9-bit angular shape.  Doing this once, fully parallel, takes the
 * self_idx -> best -> patch dependent-load chain out of every group workgroup.
 * ------------------------------------------------------------------------------------------ */
__global__ __launch_bounds__(256) void k_group_pos(GroupArgs a) {
    /* one thread per (group, match): the A disparity look-ups of a match are independent loads */
    const int A = a.A, N = a.N, NA = N * A;
    const size_t plane = (size_t)a.Wb * a.Hb;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)a.n_groups * N) return;
    const unsigned g = a.ref_begin + (unsigned)(idx / N);
    const int n = (int)(idx % N);
    const int nSx = (int)a.self_cnt[g];
    const unsigned k_r = a.refs[g];
    const unsigned ind_pst = n < nSx ? a.self_idx[(size_t)g * N + n] : 0u;
    /* byte offsets and presence bits for the scalar loads of the register-resident HT kernel (launch_group) */
    const bool want_ofs = A == 9 && a.tau2 == 4 && a.step == 1 && (size_t)A * a.C * plane * 4 < 0x7fffffffull;
    unsigned bits = 0;
    for (int st = 0; st < A; st++) {
        const bool masked = a.mask_bits.test((unsigned)st);
        unsigned p = 0xffffffffu;
        if (n < nSx && masked) p = (st == (int)a.pst) ? ind_pst : a.best[(size_t)st * plane + ind_pst];
        /* gather position: patches whose column equals Wb-k read the reference's never-filled table
         * column, i.e. zeros (core:1697, bm3d.cpp:737) on the centre path -- they are still aggregated */
        const bool zero_patch = a.fill_quirk && p != 0xffffffffu && (p % a.Wb) >= a.Wb - a.k;
        const int i = n * A + st;
        a.gpos[(size_t)g * NA + i] = zero_patch ? 0xffffffffu : p;
        if (want_ofs) {
            const bool there = !zero_patch && p != 0xffffffffu;
            a.gofs[(size_t)g * NA + i] = there ? (unsigned)(((size_t)st * a.C * plane + p) * 4) : 0u;
            bits |= there ? 1u << st : 0u;
        }
        const bool in_shape = st == (int)a.pst || (masked && a.shape[(size_t)st * plane + k_r]);
        /* positions the aggregation kernel will add this group's patches at; 0xffffffff = none
         * (match slot unused, empty SAI, or SAI outside the SADCT shape, core:503) */
        /* stored as (row << 16) | column: the aggregation kernel tests each position against many tiles */
        a.aggpos[((size_t)st * a.n_refs_total + g) * N + n] = ((a.tau4 != 6 || in_shape) && p != 0xffffffffu) ? ((p / a.Wb) << 16) | (p % a.Wb) : 0xffffffffu;
    }
    if (want_ofs) a.gok[(size_t)g * N + n] = bits;
}
template <bool BIG>   /* BIG: windows of more than 7x7 SAIs (records of kShapeInfoBigBytes, built in place in global memory) */
__global__ __launch_bounds__(256) void k_group_shape(GroupArgs a) {
    typedef typename std::conditional<BIG, ShapeInfoBig, ShapeInfo>::type SH;
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n_groups) return;
    const unsigned g = a.ref_begin + i;
    const size_t plane = (size_t)a.Wb * a.Hb;
    const unsigned k_r = a.refs[g];
    /* only what the group kernels read is written: the first A entries of each array (the record is sized for a 7x7
     * window, a 3x3 one uses a fifth of it), and nothing but the flag when the angular transform is not shape-adaptive */
    SH* out = reinterpret_cast<SH*>(a.gshape) + g;
    if (a.tau4 != 6) { out->use_sadct = 0; return; }
    const int A = (int)a.A, aw = window_side(A);
    if (BIG) {
        int* m = out->mask_dct;   /* (scratch until build_shape overwrites it: the shape goes in through a second array) */
        int full = 0;
        for (int st = 0; st < A; st++) {
            const bool masked = a.mask_bits.test((unsigned)st);
            m[st] = (st == (int)a.pst || (masked && a.shape[(size_t)st * plane + k_r])) ? 1 : 0;
            full += m[st];
        }
        if (full == A) { out->use_sadct = 0; return; }
        for (int st = 0; st < A; st++) out->mask[st] = m[st];
        build_shape(*out, out->mask, aw);
        return;
    }
    SH sh;
    int m[BIG ? 1 : kMaxA], full = 0;
    for (int st = 0; st < A; st++) {
        const bool masked = a.mask_bits.test((unsigned)st);
        m[st] = (st == (int)a.pst || (masked && a.shape[(size_t)st * plane + k_r])) ? 1 : 0;
        full += m[st];
    }
    /* the usual case, every SAI in the shape: the plain angular DCT, and the group kernels read nothing but the flag */
    if (full == A) { out->use_sadct = 0; return; }
    build_shape(sh, m, aw);
    for (int q = 0; q < A; q++) {
        out->mask[q] = sh.mask[q]; out->idx[q] = sh.idx[q]; out->mask_col[q] = sh.mask_col[q];
        out->idx_col[q] = sh.idx_col[q]; out->mask_dct[q] = sh.mask_dct[q];
    }
    for (int q = 0; q < aw; q++) { out->row_n[q] = sh.row_n[q]; out->col_n[q] = sh.col_n[q]; }
    out->use_sadct = sh.use_sadct;
}
/* SADCT bookkeeping of group g (pre-pass output) */
__device__ __forceinline__ ShRef group_shape(const GroupArgs& a, unsigned g) {
    return *reinterpret_cast<const __attribute__((address_space(4))) ShapeInfo*>(
        (const __attribute__((address_space(4))) char*)a.gshape + (size_t)g * sizeof(ShapeInfo));
}
__device__ __forceinline__ ShRefBig group_shape_big(const GroupArgs& a, unsigned g) {
    return *reinterpret_cast<const __attribute__((address_space(4))) ShapeInfoBig*>(
        (const __attribute__((address_space(4))) char*)a.gshape + (size_t)g * sizeof(ShapeInfoBig));
}

/* One (group, channel) of the generic path.  S0 / S1: the group's stack(s) [n][st][pq] -- in LDS (k_group) or, when the
 * stacks do not fit the 160 KiB, in a per-workgroup slice of an HBM scratch buffer (k_group_big); tmp: the 2-D stage's
 * LDS work area.  Any patch size, any transform combination, 3x3 and 5x5 angular windows. */
/* XCD-aware group numbering: hardware deals consecutive workgroup ids round-robin to the 8 XCDs (each with an L2 of its
 * own), so workgroup b takes group (b % 8) * per_xcd + b / 8: an XCD works its way through a contiguous band of reference
 * patches and its L2 fetches that band's window rows once instead of every XCD fetching every row.  The launch rounds
 * grid.x up to 8 * per_xcd; indices past the last group return. */
__device__ __forceinline__ unsigned xcd_group_index(const GroupArgs& a) {
    const unsigned per_xcd = (a.n_groups + 7) / 8;
    return (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
}

template <int STEP, bool BIG = false>
__device__ __forceinline__ void group_generic(const GroupArgs& a, const unsigned g, const int c, float* S0, float* S1, float* tmp,
                                              unsigned* pos, float (*red)[kThreads / 64]) {
    const int tid = threadIdx.x;
    const int k = a.k, k2 = k * k, A = a.A, N = a.N;
    const int aw = window_side(A);
    const int nSx = (int)a.self_cnt[g];
    const size_t plane = (size_t)a.Wb * a.Hb;
    const int stack = nSx * A * k2;
    const TbPtr tb = (TbPtr)a.tb;

    /* patch positions (core:286-299) and the SADCT shape of this group (core:302-323): from the pre-pass */
    for (int i = tid; i < nSx * A; i += kThreads) pos[i] = a.gpos[(size_t)g * N * A + i];   /* up to 32 x 9 > 256 */
    typename std::conditional<BIG, ShRefBig, ShRef>::type sh = [&]() -> typename std::conditional<BIG, ShRefBig, ShRef>::type {
        if constexpr (BIG) return group_shape_big(a, g); else return group_shape(a, g); }();
    __syncthreads();
    const bool use_sadct = a.tau4 == 6 && sh.use_sadct;

    /* gather (core:286-299).  Patches whose column equals Wb-k read the reference's never-filled
     * table column, i.e. zeros (core:1697, bm3d.cpp:737) -- reproduce. */
    {
        constexpr int G = 12; /* loads in flight per thread and stack */
        for (int e0 = tid; e0 < stack; e0 += kThreads * G) {
            float v0[G], v1[G];
#pragma unroll
            for (int u = 0; u < G; u++) {
                const int e = e0 + u * kThreads;
                v0[u] = 0.0f; v1[u] = 0.0f;
                if (e < stack) {
                    const int pq = e % k2, ns = e / k2;
                    const int st = ns % A;
                    const unsigned p = pos[ns];
                    if (p != 0xffffffffu) {   /* the pre-pass folds the never-filled table column in */
                        const size_t off = ((size_t)st * a.C + c) * plane + p + (size_t)(pq / k) * a.Wb + pq % k;
                        v0[u] = a.noisy[off];
                        if (STEP == 2) v1[u] = a.basic[off];
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < G; u++) {
                const int e = e0 + u * kThreads;
                if (e < stack) { S0[e] = v0[u]; if (STEP == 2) S1[e] = v1[u]; }
            }
        }
    }
    __syncthreads();

    if (a.tau2 != 4) {
        fwd2d(S0, tmp, nSx * A, k, a.tau2, tb);
        if (STEP == 2) fwd2d(S1, tmp, nSx * A, k, a.tau2, tb);
    }

    /* 4-D forward (core:353-360): one (n, pq) fibre of A values per thread */
    const bool do_dct4 = a.tau4 == 5 || (a.tau4 == 6 && !use_sadct);
    const bool do_sa4 = !do_dct4 && a.tau4 == 6;
    if (do_dct4 || do_sa4) {
        for (int f = tid; f < nSx * k2; f += kThreads) {
            const int n = f / k2, pq = f % k2;
            for (int s = 0; s < (STEP == 2 ? 2 : 1); s++) {
                float* S = s ? S1 : S0;
                if constexpr (BIG) {   /* more than 7x7 SAIs: run-time sizes, the vector in scratch memory */
                    float x[kBigA], t[kBigA];
                    for (int st = 0; st < A; st++) x[st] = S[(size_t)(n * A + st) * k2 + pq];
                    if (do_dct4) dctw_fwd_rt(x, t, aw, tb); else sadctw_fwd<ShRefBig>(x, aw, sh, tb);
                    for (int st = 0; st < A; st++) S[(size_t)(n * A + st) * k2 + pq] = x[st];
                } else
                if (A == 9) {
                    float x[9];
#pragma unroll
                    for (int st = 0; st < 9; st++) x[st] = S[(n * A + st) * k2 + pq];
                    if (do_dct4) dct9_fwd(x, tb); else sadct9_fwd(x, sh, tb);
#pragma unroll
                    for (int st = 0; st < 9; st++) S[(n * A + st) * k2 + pq] = x[st];
                } else if (do_dct4 && A == 25) {   /* 5x5 / 7x7 window, plain DCT: in registers */
                    float x[25];
#pragma unroll
                    for (int st = 0; st < 25; st++) x[st] = S[(n * 25 + st) * k2 + pq];
                    dctw_fwd_t<5>(x, tb);
#pragma unroll
                    for (int st = 0; st < 25; st++) S[(n * 25 + st) * k2 + pq] = x[st];
                } else if (do_dct4) {
                    float x[49];
#pragma unroll
                    for (int st = 0; st < 49; st++) x[st] = S[(n * 49 + st) * k2 + pq];
                    dctw_fwd_t<7>(x, tb);
#pragma unroll
                    for (int st = 0; st < 49; st++) S[(n * 49 + st) * k2 + pq] = x[st];
                } else {   /* shape-adaptive: the call form */
                    float x[kMaxA];
                    for (int st = 0; st < A; st++) x[st] = S[(n * A + st) * k2 + pq];
                    if constexpr (!BIG) sadctw_fwd<ShRef>(x, aw, sh, tb);
                    for (int st = 0; st < A; st++) S[(n * A + st) * k2 + pq] = x[st];
                }
            }
        }
        __syncthreads();
    }

    /* 5th dimension + shrinkage (core:371-410): one (st, pq) fibre of nSx values per thread */
    float wacc = 0.0f, s1 = 0.0f, s2 = 0.0f;
    {
        const float sig = a.sigma[c];
        const float T = a.bm3d ? a.lambda * sig : a.lambda * sig * 1.41421356237309505f; /* core:2431; bm3d.cpp:941 */
        const float sig2 = sig * sig;
        for (int f = tid; f < A * k2; f += kThreads) {
            const int st = f / k2, pq = f % k2;
            const bool in_shape = !use_sadct || sh.mask_dct[st];
            const int base = st * k2 + pq, stride = A * k2;
            switch (nSx) {
                case 1:  filter5<1, STEP>(S0, S1, base, stride, a.tau5, T, sig2, in_shape, wacc, s1, s2, tb); break;
                case 2:  filter5<2, STEP>(S0, S1, base, stride, a.tau5, T, sig2, in_shape, wacc, s1, s2, tb); break;
                case 4:  filter5<4, STEP>(S0, S1, base, stride, a.tau5, T, sig2, in_shape, wacc, s1, s2, tb); break;
                case 8:  filter5<8, STEP>(S0, S1, base, stride, a.tau5, T, sig2, in_shape, wacc, s1, s2, tb); break;
                case 16: filter5<16, STEP>(S0, S1, base, stride, a.tau5, T, sig2, in_shape, wacc, s1, s2, tb); break;
                default: filter5<32, STEP>(S0, S1, base, stride, a.tau5, T, sig2, in_shape, wacc, s1, s2, tb); break;   
            }
        }
    }
    /* group weight (core:412-421, sd_weighting_5d core:3140-3173) */
    for (int o = 32; o > 0; o >>= 1) { wacc += __shfl_xor(wacc, o); s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
    if ((tid & 63) == 0) { red[0][tid >> 6] = wacc; red[1][tid >> 6] = s1; red[2][tid >> 6] = s2; }
    __syncthreads();
    if (tid == 0) {
        float w = 0.0f, m = 0.0f, q = 0.0f;
        for (int i = 0; i < kThreads / 64; i++) { w += red[0][i]; m += red[1][i]; q += red[2][i]; }
        float wx;
        if (a.useSD) {
            const float Nn = a.bm3d ? (float)(nSx * k2) : (float)(nSx * A);
            const float res = (q - m * m / Nn) / (Nn - 1.0f);
            wx = res > 0.0f ? 1.0f / sqrtf(res) : 0.0f;
        } else {
            const float sig = a.sigma[c];
            wx = w > 0.0f ? (sig > 0.0f ? 1.0f / (sig * sig * w) : 1.0f / w) : 1.0f;
        }
        a.wgt[(size_t)g * a.C + c] = wx;
        if (c == 0) {
            atomicAdd(&a.counters[0], (unsigned long long)nSx);
            if (use_sadct) atomicAdd(&a.counters[1], 1ull);
        }
    }
    float* F = STEP == 2 ? S1 : S0;

    /* 4-D inverse (core:431-451) */
    if (do_dct4 || do_sa4) {
        for (int f = tid; f < nSx * k2; f += kThreads) {
            const int n = f / k2, pq = f % k2;
            if constexpr (BIG) {
                float x[kBigA], t[kBigA];
                for (int st = 0; st < A; st++) x[st] = F[(size_t)(n * A + st) * k2 + pq];
                if (do_dct4) dctw_inv_rt(x, t, aw, tb); else sadctw_inv<ShRefBig>(x, aw, sh, tb);
                for (int st = 0; st < A; st++) F[(size_t)(n * A + st) * k2 + pq] = x[st];
            } else
            if (A == 9) {
                float x[9];
#pragma unroll
                for (int st = 0; st < 9; st++) x[st] = F[(n * A + st) * k2 + pq];
                if (do_dct4) dct9_inv(x, tb); else sadct9_inv(x, sh, tb);
#pragma unroll
                for (int st = 0; st < 9; st++) F[(n * A + st) * k2 + pq] = x[st];
            } else if (do_dct4 && A == 25) {
                float x[25];
#pragma unroll
                for (int st = 0; st < 25; st++) x[st] = F[(n * 25 + st) * k2 + pq];
                dctw_inv_t<5>(x, tb);
#pragma unroll
                for (int st = 0; st < 25; st++) F[(n * 25 + st) * k2 + pq] = x[st];
            } else if (do_dct4) {
                float x[49];
#pragma unroll
                for (int st = 0; st < 49; st++) x[st] = F[(n * 49 + st) * k2 + pq];
                dctw_inv_t<7>(x, tb);
#pragma unroll
                for (int st = 0; st < 49; st++) F[(n * 49 + st) * k2 + pq] = x[st];
            } else {
                float x[kMaxA];
                for (int st = 0; st < A; st++) x[st] = F[(n * A + st) * k2 + pq];
                if constexpr (!BIG) sadctw_inv<ShRef>(x, aw, sh, tb);
                for (int st = 0; st < A; st++) F[(n * A + st) * k2 + pq] = x[st];
            }
        }
    }
    __syncthreads();
    if (a.tau2 != 4) inv2d(F, tmp, nSx * A, k, a.tau2, tb);

    /* filtered patches out: [g][n][st][c][k2] */
    float* out = a.filt + (size_t)g * N * A * a.C * k2;
    for (int e = tid; e < stack; e += kThreads) {
        const int pq = e % k2, ns = e / k2;
        out[((size_t)ns * a.C + c) * k2 + pq] = F[e];
    }
}

template <int STEP>
__global__ __launch_bounds__(kThreads) void k_group(GroupArgs a) {
    extern __shared__ float lds[];
    __shared__ unsigned pos[kMaxN3 * kMaxA];
    __shared__ float red[3][kThreads / 64];
    const unsigned g = a.ref_begin + blockIdx.x;
    const int stack = (int)a.self_cnt[g] * (int)a.A * (int)(a.k * a.k);
    group_generic<STEP>(a, g, (int)blockIdx.y, lds, STEP == 2 ? lds + stack : nullptr, lds + (STEP == 2 ? 2 : 1) * stack, pos, red);
}

/* Stacks beyond the LDS (a 5x5 window with the README's 16x16 patches: 8 x 25 x 256 floats = 200 KiB per stack; N = 32
 * with 16x16 patches): a persistent launch, every workgroup owns a slice of an HBM scratch buffer for its stack(s) and
 * walks over (group, channel) items.  Global memory written by a workgroup is visible to it after a barrier (one CU, one
 * vector L1), so the phases are the LDS kernel's, only slower; the 2-D stage's work area stays in LDS. */
template <int STEP, bool BIG>
__global__ __launch_bounds__(kThreads) void k_group_big(GroupArgs a, float* scratch, unsigned long long slice_floats, unsigned tmp_floats) {
    extern __shared__ float lds[];
    __shared__ unsigned pos_small[BIG ? 1 : kMaxN3 * kMaxA];
    __shared__ float red[3][kThreads / 64];
    unsigned* pos = BIG ? reinterpret_cast<unsigned*>(lds + tmp_floats) : pos_small;   /* BIG: N x A positions behind the 2-D work area */
    float* S0 = scratch + (size_t)blockIdx.x * slice_floats;
    const unsigned items = a.n_groups * a.C;
    for (unsigned it = blockIdx.x; it < items; it += gridDim.x) {
        const unsigned g = a.ref_begin + it / a.C;
        const int stack = (int)a.self_cnt[g] * (int)a.A * (int)(a.k * a.k);
        group_generic<STEP, BIG>(a, g, (int)(it % a.C), S0, STEP == 2 ? S0 + stack : nullptr, lds, pos, red);
        __syncthreads();   /* pos / red / the scratch slice are reused by the next item */
    }
}

/* ------------------------------------------------------------------------------------------
 * Register-resident variant for tau_2D = id (the README hard-thresholding configuration): with no
 * 2-D transform, pixel pq of the group never mixes with other pixels, so one thread owns pixel pq
 * of all nSx * A patches (72 registers for N = 8), loads them straight from the window images and
 * runs the 3x3 angular DCT / SADCT, the Haar/Hadamard fibre transforms, the shrinkage and the
 * inverses without touching LDS (only the group weight is reduced through it).  No LDS stack means
 * occupancy is set by registers, not by the 72 KiB stack of k_group.
 * ------------------------------------------------------------------------------------------ */
/* One pixel of all NS * 9 patches of the group (hard-thresholding step).  Loads and stores go
 * through buffer resources with the per-patch part of the address in a scalar register (the patch
 * positions are uniform), so none of the 2 * NS * 9 memory operations needs address VGPRs; the
 * 3x3 angular DCTs run on pairs of patches (n, n + 1) with packed fp32 arithmetic. */
/* the angular transform, the 5th-dimension transform with the hard threshold and their inverses on one pixel's NS * 9 values
 * V[h][st] = {patch h, patch h + NS/2} (the register stage shared by the tau_2D = id kernel and the 16x16 kernels) */
/* SA_MODE: how the (rare) shape-adaptive transform is reached -- 0: calls (scratch vector; keeps its code out of the caller's
 * register allocation), 1: inline on sa_lds, nine floats of LDS of this thread's, 2: inline in registers (the *_sa kernels, which
 * the host launches for windows with an empty SAI, where EVERY group is shape-adaptive) */
template <int NS, bool HAAR, int SA_MODE = 0>
__device__ __forceinline__ void group_id_compute(const GroupArgs& a, int c, ShRef sh, bool use_sadct, v2f (&V)[NS > 1 ? NS / 2 : 1][9],
                                                 float& wacc, float& s1, float& s2, float* sa_lds = nullptr) {
    constexpr int NH = NS > 1 ? NS / 2 : 1;
    const TbPtr tb = (TbPtr)a.tb;
    const bool do_dct4 = a.tau4 == 5 || (a.tau4 == 6 && !use_sadct);
    const bool do_sa4 = !do_dct4 && a.tau4 == 6;
    auto sadct_pairs = [&](bool fwd) {   /* rare: shape-adaptive transform on the scalar path, staged through t9 */
#pragma unroll
        for (int h = 0; h < NH; h++)
#pragma unroll
            for (int half = 0; half < (NS > 1 ? 2 : 1); half++) {
                if (SA_MODE == 1) {
#pragma unroll
                    for (int i = 0; i < 9; i++) sa_lds[i] = half ? V[h][i].y : V[h][i].x;
                    if (fwd) sadct9_fwd_lds(sa_lds, sh, tb); else sadct9_inv_lds(sa_lds, sh, tb);
#pragma unroll
                    for (int i = 0; i < 9; i++) { if (half) V[h][i].y = sa_lds[i]; else V[h][i].x = sa_lds[i]; }
                } else {
                    float t9[9];
#pragma unroll
                    for (int i = 0; i < 9; i++) t9[i] = half ? V[h][i].y : V[h][i].x;
                    if (SA_MODE == 2) { if (fwd) sadct9_fwd_sel(t9, sh, tb); else sadct9_inv_sel(t9, sh, tb); }
                    else if (fwd) sadct9_fwd(t9, sh, tb); else sadct9_inv(t9, sh, tb);
#pragma unroll
                    for (int i = 0; i < 9; i++) { if (half) V[h][i].y = t9[i]; else V[h][i].x = t9[i]; }
                }
            }
    };
    if (do_dct4) {
#pragma unroll
        for (int h = 0; h < NH; h++) dct9_fwd2(V[h], tb);
    } else if (do_sa4) sadct_pairs(true);
    const float sig = a.sigma[c];
    const float T = a.lambda * sig * 1.41421356237309505f;
#pragma unroll
    for (int st = 0; st < 9; st++) {
        const bool in_shape = !use_sadct || sh.mask_dct[st];
        if (HAAR) {
            v2f P[NH];
#pragma unroll
            for (int h = 0; h < NH; h++) P[h] = V[h][st];
            haar_fwd_pairs<NS>(P);
            if (in_shape) {
#pragma unroll
                for (int h = 0; h < NH; h++) {
                    const bool kx = fabsf(P[h].x) > T, ky = (NS > 1) && fabsf(P[h].y) > T;
                    wacc += (kx ? 1.0f : 0.0f) + (ky ? 1.0f : 0.0f);
                    P[h].x = kx ? P[h].x : 0.0f;
                    P[h].y = ky ? P[h].y : 0.0f;
                }
            }
            haar_inv_pairs<NS>(P);
#pragma unroll
            for (int h = 0; h < NH; h++) V[h][st] = P[h];
        } else {
            float o[NS], e[1] = {0.0f};
#pragma unroll
            for (int n = 0; n < NS; n++) o[n] = n < NH ? V[n][st].x : V[n - NH][st].y;
            shrink_fibre<NS, 1>(o, e, a.tau5, T, sig * sig, in_shape, wacc, tb);
#pragma unroll
            for (int n = 0; n < NS; n++) { if (n < NH) V[n][st].x = o[n]; else V[n - NH][st].y = o[n]; }
        }
    }
    if (a.useSD) {
#pragma unroll
        for (int h = 0; h < NH; h++)
#pragma unroll
            for (int st = 0; st < 9; st++) {
                s1 += V[h][st].x; s2 += V[h][st].x * V[h][st].x;
                if (NS > 1) { s1 += V[h][st].y; s2 += V[h][st].y * V[h][st].y; }
            }
    }
    if (do_dct4) {
#pragma unroll
        for (int h = 0; h < NH; h++) dct9_inv2(V[h], tb);
    } else if (do_sa4) sadct_pairs(false);
}

template <int NS, bool HAAR, bool LDSW = false, int SA_MODE = 0>   /* LDSW: values come from / go back to an LDS work area [patch][k][k+1] (2-D transformed patches) */
__device__ __forceinline__ void group_id_body(const GroupArgs& a, unsigned g, int c, int pq, const __attribute__((address_space(4))) unsigned* pos,
                                              ShRef sh, bool use_sadct, float& wacc, float& s1, float& s2, float* work = nullptr) {
    const int k = a.k, k2 = k * k, A = 9;
    const unsigned plane = a.Wb * a.Hb;
    const unsigned kRsrcFlags = 0x00020000u;
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)a.noisy, 0, (int)((size_t)a.A * a.C * plane * 4), kRsrcFlags);
    float* const out = a.filt + (size_t)g * a.N * A * a.C * k2;
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc((void*)out, 0, (int)((size_t)a.N * A * a.C * k2 * 4), kRsrcFlags);
    /* the pixel's NS * 9 values as pairs of patches: V[h][st] = {patch h, patch h + NS/2} (NS = 1: .y unused) */
    constexpr int NH = NS > 1 ? NS / 2 : 1;
    v2f V[NH][9];
    const int voff = (int)(((unsigned)(pq / k) * a.Wb + pq % k) * 4u);
    const int woff = (pq / k) * (k + 1) + pq % k;      /* this pixel inside a work-area patch */
    unsigned okbits[NS];
    typedef const __attribute__((address_space(4))) unsigned* cuptr_;
    const cuptr_ ofs = (cuptr_)(a.gofs + (size_t)g * a.N * A), ok = (cuptr_)(a.gok + (size_t)g * a.N);
    const unsigned cbase = (unsigned)c * plane * 4u;
    if (LDSW) {
#pragma unroll
        for (int n = 0; n < NS; n++)
#pragma unroll
            for (int st = 0; st < 9; st++) {
                const float x = work[(n * A + st) * kT16Patch + woff];
                if (n < NH) V[n][st].x = x; else V[n - NH][st].y = x;
                okbits[n] = 0x1ffu;
            }
    } else
#pragma unroll
    for (int n = 0; n < NS; n++) {
        okbits[n] = ok[n];                            /* uniform: scalar loads (pre-pass: k_group_pos) */
#pragma unroll
        for (int st = 0; st < 9; st++) {
            const unsigned so = ofs[n * A + st] + cbase;   /* absent patches read offset 0 and are zeroed below */
            const float x = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_in, voff, (int)so, 0));
            if (n < NH) V[n][st].x = x; else V[n - NH][st].y = x;
        }
    }
    if (NS == 1) {
#pragma unroll
        for (int st = 0; st < 9; st++) V[0][st].y = 0.0f;
    }
#pragma unroll
    for (int n = 0; n < NS; n++)
        if (okbits[n] != 0x1ffu) {   /* uniform, rare: patches of empty SAIs / never-filled table column read as zeros */
#pragma unroll
            for (int st = 0; st < 9; st++) {
                if (n < NH) V[n][st].x = ((okbits[n] >> st) & 1) ? V[n][st].x : 0.0f;
                else V[n - NH][st].y = ((okbits[n] >> st) & 1) ? V[n - NH][st].y : 0.0f;
            }
        }
    group_id_compute<NS, HAAR, SA_MODE>(a, c, sh, use_sadct, V, wacc, s1, s2);
    const int vout = pq * 4;
#pragma unroll
    for (int n = 0; n < NS; n++)
#pragma unroll
        for (int st = 0; st < 9; st++) {
            const float r = n < NH ? V[n][st].x : V[n - NH][st].y;
            if (LDSW) work[(n * A + st) * kT16Patch + woff] = r;
            else __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, r), rs_out, vout, (int)((((unsigned)(n * A + st) * a.C + c) * k2) * 4u), 0);
        }
}

template <bool HAAR, bool SA = false>
__device__ __forceinline__ void group_id_kernel(const GroupArgs& a) {
    __shared__ float red[3][4];
    const int tid = threadIdx.x;
    const unsigned gi = xcd_group_index(a);
    if (gi >= a.n_groups) return;
    const unsigned g = a.ref_begin + gi;
    const int c = blockIdx.y;
    const int A = 9, N = a.N;
    const int nSx = (int)a.self_cnt[g];
    /* positions are uniform per workgroup and constant during this kernel: constant address space -> scalar loads */
    typedef const __attribute__((address_space(4))) unsigned* cuptr;
    const cuptr pos = (cuptr)(a.gpos + (size_t)g * N * A);
    ShRef sh = group_shape(a, g);
    const bool use_sadct = a.tau4 == 6 && sh.use_sadct;
    float wacc = 0.0f, s1 = 0.0f, s2 = 0.0f;
    if (tid < (int)(a.k * a.k)) {
        switch (nSx) {
            case 1:  group_id_body<1, HAAR, false, SA ? 2 : 0>(a, g, c, tid, pos, sh, use_sadct, wacc, s1, s2); break;
            case 2:  group_id_body<2, HAAR, false, SA ? 2 : 0>(a, g, c, tid, pos, sh, use_sadct, wacc, s1, s2); break;
            case 4:  group_id_body<4, HAAR, false, SA ? 2 : 0>(a, g, c, tid, pos, sh, use_sadct, wacc, s1, s2); break;
            default: group_id_body<8, HAAR, false, SA ? 2 : 0>(a, g, c, tid, pos, sh, use_sadct, wacc, s1, s2); break;
        }
    }
    for (int o = 32; o > 0; o >>= 1) { wacc += __shfl_xor(wacc, o); s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
    if ((tid & 63) == 0) { red[0][tid >> 6] = wacc; red[1][tid >> 6] = s1; red[2][tid >> 6] = s2; }
    __syncthreads();
    if (tid == 0) {
        float w = 0.0f, m = 0.0f, q = 0.0f;
        for (unsigned i = 0; i < (blockDim.x + 63) / 64; i++) { w += red[0][i]; m += red[1][i]; q += red[2][i]; }
        float wx;
        if (a.useSD) {
            const float Nn = (float)(nSx * A);
            const float res = (q - m * m / Nn) / (Nn - 1.0f);
            wx = res > 0.0f ? 1.0f / sqrtf(res) : 0.0f;
        } else {
            const float sig = a.sigma[c];
            wx = w > 0.0f ? (sig > 0.0f ? 1.0f / (sig * sig * w) : 1.0f / w) : 1.0f;
        }
        a.wgt[(size_t)g * a.C + c] = wx;
        if (c == 0) {
            atomicAdd(&a.counters[0], (unsigned long long)nSx);
            if (use_sadct) atomicAdd(&a.counters[1], 1ull);
        }
    }
}

/* Haar configuration (README): capped at 168 VGPRs so that three waves fit a SIMD (a dozen spilled values buy 20 %);
 * the Hadamard / DCT fibre transforms need more registers and keep two */
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_group_id_haar(GroupArgs a) { group_id_kernel<true>(a); }
__global__ __launch_bounds__(256) void k_group_id_any(GroupArgs a) { group_id_kernel<false>(a); }
/* the same kernels for windows with an empty SAI (every group shape-adaptive): the transform inline, in registers */
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_group_id_haar_sa(GroupArgs a) { group_id_kernel<true, true>(a); }
__global__ __launch_bounds__(256) void k_group_id_any_sa(GroupArgs a) { group_id_kernel<false, true>(a); }

__device__ __forceinline__ void dct8_fwd(float* x) {
    const float a0 = 0.35355339059327376f;   /* 1/sqrt(8) */
    const float h = 0.5f;
    const float c1 = 0.98078528040323044f, c2 = 0.92387953251128674f, c3 = 0.83146961230254524f, c4 = 0.70710678118654752f,
                c5 = 0.55557023301960222f, c6 = 0.38268343236508977f, c7 = 0.19509032201612827f;
    const float s0 = x[0] + x[7], s1 = x[1] + x[6], s2 = x[2] + x[5], s3 = x[3] + x[4];
    const float d0 = x[0] - x[7], d1 = x[1] - x[6], d2 = x[2] - x[5], d3 = x[3] - x[4];
    const float p0 = s0 + s3, p1 = s1 + s2, m0 = s0 - s3, m1 = s1 - s2;
    x[0] = a0 * (p0 + p1);
    x[4] = (h * c4) * (p0 - p1);
    x[2] = h * (c2 * m0 + c6 * m1);
    x[6] = h * (c6 * m0 - c2 * m1);
    x[1] = h * (c1 * d0 + c3 * d1 + c5 * d2 + c7 * d3);
    x[3] = h * (c3 * d0 - c7 * d1 - c1 * d2 - c5 * d3);
    x[5] = h * (c5 * d0 - c1 * d1 + c7 * d2 + c3 * d3);
    x[7] = h * (c7 * d0 - c5 * d1 + c3 * d2 - c1 * d3);
}
__device__ __forceinline__ void dct8_inv(float* X) {
    const float a0 = 0.35355339059327376f;
    const float h = 0.5f;
    const float c1 = 0.98078528040323044f, c2 = 0.92387953251128674f, c3 = 0.83146961230254524f, c4 = 0.70710678118654752f,
                c5 = 0.55557023301960222f, c6 = 0.38268343236508977f, c7 = 0.19509032201612827f;
    const float e0 = a0 * X[0] + (h * c4) * X[4], e1 = a0 * X[0] - (h * c4) * X[4];
    const float f0 = h * (c2 * X[2] + c6 * X[6]), f1 = h * (c6 * X[2] - c2 * X[6]);
    const float E0 = e0 + f0, E1 = e1 + f1, E2 = e1 - f1, E3 = e0 - f0;
    const float O0 = h * (c1 * X[1] + c3 * X[3] + c5 * X[5] + c7 * X[7]);
    const float O1 = h * (c3 * X[1] - c7 * X[3] - c1 * X[5] - c5 * X[7]);
    const float O2 = h * (c5 * X[1] - c1 * X[3] + c7 * X[5] + c3 * X[7]);
    const float O3 = h * (c7 * X[1] - c5 * X[3] + c3 * X[5] - c1 * X[7]);
    X[0] = E0 + O0; X[7] = E0 - O0;
    X[1] = E1 + O1; X[6] = E1 - O1;
    X[2] = E2 + O2; X[5] = E2 - O2;
    X[3] = E3 + O3; X[4] = E3 - O3;
}

/* orthonormal 16-point DCT-II / its inverse in registers: even outputs = the 8-point transform of the folded sums
 * (scaled by 1/sqrt2), odd outputs = an 8x8 product of the folded differences with cos((2n+1)(2k+1) pi/32) */
constexpr float kCos32[16] = {1.0f, 0.99518472667219689f, 0.98078528040323044f, 0.95694033573220887f, 0.92387953251128674f,
                              0.88192126434835503f, 0.83146961230254524f, 0.77301045336273696f, 0.70710678118654752f,
                              0.63439328416364549f, 0.55557023301960222f, 0.47139673682599764f, 0.38268343236508977f,
                              0.29028467725446236f, 0.19509032201612827f, 0.09801714032956060f};
constexpr float cos32(int m) {   /* cos(m pi / 32), m odd */
    m &= 63;
    return m < 16 ? kCos32[m] : m < 32 ? -kCos32[32 - m] : m < 48 ? -kCos32[m - 32] : kCos32[64 - m];
}
__device__ __forceinline__ void dct16_fwd(float* x) {
    const float r2 = 0.70710678118654752f, h = 0.35355339059327376f;   /* sqrt(2/16) */
    float e[8], o[8];
#pragma unroll
    for (int n = 0; n < 8; n++) { e[n] = (x[n] + x[15 - n]) * r2; o[n] = (x[n] - x[15 - n]) * h; }
    dct8_fwd(e);
#pragma unroll
    for (int u = 0; u < 8; u++) {
        float acc = 0.0f;
#pragma unroll
        for (int n = 0; n < 8; n++) acc += o[n] * cos32((2 * n + 1) * (2 * u + 1));
        x[2 * u] = e[u]; x[2 * u + 1] = acc;
    }
}
__device__ __forceinline__ void dct16_inv(float* X) {
    const float r2 = 0.70710678118654752f, h = 0.35355339059327376f;
    float e[8], o[8];
#pragma unroll
    for (int u = 0; u < 8; u++) { e[u] = X[2 * u] * r2; o[u] = X[2 * u + 1] * h; }
    dct8_inv(e);
#pragma unroll
    for (int n = 0; n < 8; n++) {
        float acc = 0.0f;
#pragma unroll
        for (int u = 0; u < 8; u++) acc += o[u] * cos32((2 * n + 1) * (2 * u + 1));
        X[n] = e[n] + acc; X[15 - n] = e[n] - acc;
    }
}
/* ------------------------------------------------------------------------------------------
 * Hard-thresholding step with tau_2D = bior1.5 or dct and 16x16 patches (BASELINE configurations 2, 4 and 5):
 * the register-resident kernel above with a 2-D stage in front and behind it.  All nSx * A patches of the group
 * go through an LDS work area [patch][16][17]: the first pass of the forward transform reads whole patch rows from
 * the window images, the per-pixel threads run the angular and 5th-dimension stages on the work area
 * (group_id_body), the last pass of the inverse transform stores the filtered rows.  bior1.5: two rows / columns
 * per thread as packed pairs (bior_taps2), levels 8, 4, 2 with the patches re-dealt to fewer threads each; DCT: 16
 * threads per patch (dct16_fwd / dct16_inv).
 * ------------------------------------------------------------------------------------------ */
/* MULTI (N = 1, BASELINE configuration 5): a group is nine patches, so a workgroup takes kT16Groups consecutive groups
 * through the 2-D stages together (their patches are contiguous in gpos and filt) and runs the per-pixel stage once
 * per group.  Measured at 560^2: 0.90 ms with one group per workgroup, 0.34 / 0.33 / 0.36 / 0.38 / 0.46 / 0.47 / 0.71 ms
 * with 2 / 3 / 4 / 5 / 7 / 8 / 14 -- three groups fill one round of the 16x16 level (216 of 256 threads) and leave
 * room for five workgroups per CU. */
#ifndef LFBM5D_T16_GROUPS
#define LFBM5D_T16_GROUPS 3
#endif
constexpr int kT16Groups = LFBM5D_T16_GROUPS;
#ifndef LFBM5D_T16_ROUND
#define LFBM5D_T16_ROUND 40
#endif
#ifndef LFBM5D_T16_WAVES
#define LFBM5D_T16_WAVES 3
#endif
constexpr int kT16Half = LFBM5D_T16_ROUND;   /* patches per round of 2-D transforms: the work area of the Haar kernels (40: two rounds for a full group of 72, three workgroups per CU) */
template <bool HAAR, bool BIOR, bool MULTI, bool SPLIT = false, bool SA = false>   /* SA: windows with an empty SAI -- the shape-adaptive transform inline, in registers (N = 1 form) */
__device__ __forceinline__ void group_t16_kernel(const GroupArgs& a) {
    extern __shared__ float lds[];
    __shared__ float red[MULTI ? kT16Groups : 1][3][4];
    constexpr int K = 16, RS = K + 1, PSZ = kT16Patch, A = 9;
    const int tid = threadIdx.x;
    const unsigned g = a.ref_begin + blockIdx.x * (MULTI ? kT16Groups : 1);     /* first group of the workgroup */
    const int ngr = MULTI ? (int)min((unsigned)kT16Groups, a.ref_begin + a.n_groups - g) : 1;
    const int c = blockIdx.y;
    const int N = a.N;
    const int nSx = MULTI ? 1 : (int)a.self_cnt[g], NP = MULTI ? ngr * A : nSx * A;
    typedef const __attribute__((address_space(4))) unsigned* cuptr;
    const cuptr pos = (cuptr)(a.gpos + (size_t)g * N * A);
    ShRef sh = group_shape(a, g);
    const bool use_sadct = a.tau4 == 6 && sh.use_sadct;
    const TbPtr tb = (TbPtr)a.tb;
    float* work = lds;
#ifdef LFBM5D_PHASE_TIMING
    long long tq[6]; int tqi = 0;
#define T16_MARK() do { if (tid == 0) tq[tqi] = (long long)__builtin_readcyclecounter(); tqi++; } while (0)
    T16_MARK();
#else
#define T16_MARK() do {} while (0)
#endif
    /* forward 2-D transform; its first pass (the rows of the 16x16 level) takes the patches straight from the window
     * images: a thread loads whole 64-byte patch rows (four 16-byte loads at 4-byte alignment), transforms them and
     * parks the result in the work area -- no separate gather, and 8 (bior) / 4 (DCT) loads per thread and round where
     * a thread-per-pixel gather issues one 4-byte load per patch */
    const size_t plane = (size_t)a.Wb * a.Hb;
    auto patch_src = [&](int patch, bool& ok) -> const float* {
        const unsigned p = a.gpos[(size_t)g * N * A + patch];
        ok = p != 0xffffffffu;            /* empty SAI / never-filled table column: zeros */
        return a.noisy + ((size_t)(patch % A) * a.C + c) * plane + (ok ? p : 0u);
    };
    /* patches base .. base + np - 1 of the group -> work area slots 0 .. np - 1 */
    auto fwd2d = [&](const int base, const int np) {
        if (BIOR) {
            constexpr int TPP = 8, PPI = kThreads / TPP;   /* rows r and r + 8 per thread */
            const int slot = tid / TPP, r = tid % TPP;
            for (int p0 = 0; p0 < np; p0 += PPI) {
                const int patch = p0 + slot;
                if (patch < np) {
                    bool ok;
                    const float* src = patch_src(base + patch, ok);
                    v2f v[K], o[K];
    #pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const f4u lo = *reinterpret_cast<const f4u*>(src + (size_t)r * a.Wb + 4 * q);
                        const f4u hi = *reinterpret_cast<const f4u*>(src + (size_t)(r + 8) * a.Wb + 4 * q);
    #pragma unroll
                        for (int e = 0; e < 4; e++) v[4 * q + e] = ok ? v2f{lo.v[e], hi.v[e]} : v2f{0.0f, 0.0f};
                    }
                    bior_taps2<K, true>(v, o, tb);
                    float* Tp = work + patch * PSZ;
    #pragma unroll
                    for (int cc = 0; cc < K; cc++) { Tp[r * RS + cc] = o[cc].x; Tp[(r + 8) * RS + cc] = o[cc].y; }
                    __builtin_amdgcn_wave_barrier();
                    bior16_pass2<K, true, false>(Tp, r, tb);
                }
            }
            __syncthreads();
            bior16_level_all<8, true>(work, np, tid, tb);
            bior16_level_all<4, true>(work, np, tid, tb);
            bior16_level_all<2, true>(work, np, tid, tb);
        } else {
            const int slot = tid / K, r = tid % K;         /* DCT: 16 threads per patch, thread = row, then column */
            for (int p0 = 0; p0 < np; p0 += kThreads / K) {
                const int patch = p0 + slot;
                if (patch < np) {
                    bool ok;
                    const float* src = patch_src(base + patch, ok);
                    float x[K];
    #pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const f4u t4 = *reinterpret_cast<const f4u*>(src + (size_t)r * a.Wb + 4 * q);
    #pragma unroll
                        for (int e = 0; e < 4; e++) x[4 * q + e] = ok ? t4.v[e] : 0.0f;
                    }
                    dct16_fwd(x);
                    float* Tp = work + patch * PSZ;
    #pragma unroll
                    for (int cc = 0; cc < K; cc++) Tp[r * RS + cc] = x[cc];
                    __builtin_amdgcn_wave_barrier();
    #pragma unroll
                    for (int i = 0; i < K; i++) x[i] = Tp[i * RS + r];
                    dct16_fwd(x);
    #pragma unroll
                    for (int i = 0; i < K; i++) Tp[i * RS + r] = x[i];
                }
            }
            __syncthreads();
        }
    };
    /* inverse 2-D transform; its last pass (the rows of the 16x16 level) stores the filtered patches: filt[g][n][st][c][256] */
    float* const out = a.filt + (size_t)g * N * A * a.C * K * K;
    auto inv2d = [&](const int base, const int np) {
        if (BIOR) {
            bior16_level_all<2, false>(work, np, tid, tb);
            bior16_level_all<4, false>(work, np, tid, tb);
            bior16_level_all<8, false>(work, np, tid, tb);
            constexpr int TPP = 8, PPI = kThreads / TPP;
            const int slot = tid / TPP, r = tid % TPP;
            for (int p0 = 0; p0 < np; p0 += PPI) {
                const int patch = p0 + slot;
                if (patch < np) {
                    float* Tp = work + patch * PSZ;
                    bior16_pass2<K, false, false>(Tp, r, tb);
                    __builtin_amdgcn_wave_barrier();
                    v2f v[K], o[K];
    #pragma unroll
                    for (int cc = 0; cc < K; cc++) v[cc] = v2f{Tp[r * RS + cc], Tp[(r + 8) * RS + cc]};
                    bior_taps2<K, false>(v, o, tb);
                    float4* dst = reinterpret_cast<float4*>(out + ((size_t)(base + patch) * a.C + c) * K * K);
    #pragma unroll
                    for (int q = 0; q < 4; q++) {
                        dst[r * 4 + q] = make_float4(o[4 * q].x, o[4 * q + 1].x, o[4 * q + 2].x, o[4 * q + 3].x);
                        dst[(r + 8) * 4 + q] = make_float4(o[4 * q].y, o[4 * q + 1].y, o[4 * q + 2].y, o[4 * q + 3].y);
                    }
                }
            }
        } else {
            const int slot = tid / K, r = tid % K;
            for (int p0 = 0; p0 < np; p0 += kThreads / K) {
                const int patch = p0 + slot;
                if (patch < np) {
                    float* Tp = work + patch * PSZ;
                    float x[K];
    #pragma unroll
                    for (int i = 0; i < K; i++) x[i] = Tp[i * RS + r];
                    dct16_inv(x);
    #pragma unroll
                    for (int i = 0; i < K; i++) Tp[i * RS + r] = x[i];
                    __builtin_amdgcn_wave_barrier();
    #pragma unroll
                    for (int cc = 0; cc < K; cc++) x[cc] = Tp[r * RS + cc];
                    dct16_inv(x);
                    float4* dst = reinterpret_cast<float4*>(out + ((size_t)(base + patch) * a.C + c) * K * K);
    #pragma unroll
                    for (int q = 0; q < 4; q++) dst[r * 4 + q] = make_float4(x[4 * q], x[4 * q + 1], x[4 * q + 2], x[4 * q + 3]);
                }
            }
        }
    };
    float wacc[MULTI ? kT16Groups : 1], s1[MULTI ? kT16Groups : 1], s2[MULTI ? kT16Groups : 1];
    const bool split = SPLIT && !MULTI;
    if (!split) { fwd2d(0, NP); T16_MARK(); }
    if (MULTI) {
#pragma unroll
        for (int gi = 0; gi < kT16Groups; gi++) {
            wacc[gi] = 0.0f; s1[gi] = 0.0f; s2[gi] = 0.0f;
            if (gi < ngr) {
                ShRef shg = group_shape(a, g + gi);
                group_id_body<1, HAAR, true, SA ? 2 : 0>(a, g + gi, c, tid, pos + gi * A, shg, a.tau4 == 6 && shg.use_sadct, wacc[gi], s1[gi], s2[gi],
                                                         work + gi * A * PSZ);
            }
        }
    } else if (SPLIT) {
        /* Round 4: a work area of kT16Half patches instead of the whole group's (three workgroups per CU instead of two).  A thread
         * collects its pixel's coefficients in registers from the forward transforms -- one round of them for groups of up to four
         * matches, two for the full group --, runs the register stage of the tau_2D = id kernel on them, and the results go back
         * through the same area for the inverse transforms.  Between the two the area is free: the shape-adaptive transform of the
         * rare groups that need it runs inline on nine floats of it per thread (as a call it costs every group 44 VGPRs). */
        wacc[0] = 0.0f; s1[0] = 0.0f; s2[0] = 0.0f;
        const int woff = (tid / K) * RS + tid % K;
        auto rounds = [&](auto ns_tag) {
            constexpr int NS = decltype(ns_tag)::value, NH = NS > 1 ? NS / 2 : 1, NPc = NS * A, PR = kT16Half, NR = (NPc + PR - 1) / PR, F0 = NPc - (NR - 1) * PR;
            v2f V[NH][9];
            auto put = [&](const int pch, const float x) { if (pch / 9 < NH) V[pch / 9][pch % 9].x = x; else V[pch / 9 - NH][pch % 9].y = x; };
            auto get = [&](const int pch) { return pch / 9 < NH ? V[pch / 9][pch % 9].x : V[pch / 9 - NH][pch % 9].y; };
            if (NS == 1) {
#pragma unroll
                for (int st = 0; st < 9; st++) V[0][st].y = 0.0f;
            }
#pragma unroll
            for (int r = 0; r < NR; r++) {   /* the first round is the short one: fewest coefficients in registers while the transforms of the others run */
                const int b0 = r == 0 ? 0 : F0 + (r - 1) * PR, cnt = r == 0 ? F0 : PR;
                fwd2d(b0, cnt);
#pragma unroll
                for (int q = 0; q < PR; q++) if (q < cnt) put(b0 + q, work[q * PSZ + woff]);
                __syncthreads();
            }
            T16_MARK();
            /* the shape-adaptive transform inline: in registers in the wavelet kernel (4.2 -> 3.0 ms per pass of shape-adaptive groups), on LDS
             * scratch in the DCT kernel, whose register allocation the register form upsets (69 spills, +20 % on ordinary groups) */
            group_id_compute<NS, HAAR, BIOR ? 2 : 1>(a, c, sh, use_sadct, V, wacc[0], s1[0], s2[0], work + tid * 9);
            __syncthreads();
            T16_MARK();
#pragma unroll
            for (int r = NR - 1; r >= 0; r--) {
                const int b0 = r == 0 ? 0 : F0 + (r - 1) * PR, cnt = r == 0 ? F0 : PR;
#pragma unroll
                for (int q = 0; q < PR; q++) if (q < cnt) work[q * PSZ + woff] = get(b0 + q);
                __syncthreads();
                inv2d(b0, cnt);
                if (r > 0) __syncthreads();
            }
        };
        switch (nSx) {
            case 1:  rounds(std::integral_constant<int, 1>{}); break;
            case 2:  rounds(std::integral_constant<int, 2>{}); break;
            case 4:  rounds(std::integral_constant<int, 4>{}); break;
            default: rounds(std::integral_constant<int, 8>{}); break;
        }
    } else {
        wacc[0] = 0.0f; s1[0] = 0.0f; s2[0] = 0.0f;
        switch (nSx) {
            case 1:  group_id_body<1, HAAR, true>(a, g, c, tid, pos, sh, use_sadct, wacc[0], s1[0], s2[0], work); break;
            case 2:  group_id_body<2, HAAR, true>(a, g, c, tid, pos, sh, use_sadct, wacc[0], s1[0], s2[0], work); break;
            case 4:  group_id_body<4, HAAR, true>(a, g, c, tid, pos, sh, use_sadct, wacc[0], s1[0], s2[0], work); break;
            default: group_id_body<8, HAAR, true>(a, g, c, tid, pos, sh, use_sadct, wacc[0], s1[0], s2[0], work); break;
        }
    }
    if (!split) { __syncthreads(); T16_MARK(); inv2d(0, NP); }
    T16_MARK();
#ifdef LFBM5D_PHASE_TIMING
    T16_MARK();
    if (tid == 0) { for (int i = 0; i < 5; i++) atomicAdd(&a.counters[4 + i], (unsigned long long)(tq[i + 1] - tq[i])); atomicAdd(&a.counters[9], 1ull); }
#endif
#pragma unroll
    for (int gi = 0; gi < (MULTI ? kT16Groups : 1); gi++) {
        for (int o = 32; o > 0; o >>= 1) { wacc[gi] += __shfl_xor(wacc[gi], o); s1[gi] += __shfl_xor(s1[gi], o); s2[gi] += __shfl_xor(s2[gi], o); }
        if ((tid & 63) == 0) { red[gi][0][tid >> 6] = wacc[gi]; red[gi][1][tid >> 6] = s1[gi]; red[gi][2][tid >> 6] = s2[gi]; }
    }
    __syncthreads();
    if (tid < ngr) {   /* group weights (core:412-421, sd_weighting_5d core:3140-3173) */
        const int gi = tid;
        float w = 0.0f, m = 0.0f, q = 0.0f;
        for (int i = 0; i < 4; i++) { w += red[gi][0][i]; m += red[gi][1][i]; q += red[gi][2][i]; }
        float wx;
        if (a.useSD) {
            const float Nn = (float)(nSx * A);
            const float res = (q - m * m / Nn) / (Nn - 1.0f);
            wx = res > 0.0f ? 1.0f / sqrtf(res) : 0.0f;
        } else {
            const float sig = a.sigma[c];
            wx = w > 0.0f ? (sig > 0.0f ? 1.0f / (sig * sig * w) : 1.0f / w) : 1.0f;
        }
        a.wgt[(size_t)(g + gi) * a.C + c] = wx;
        if (c == 0) {
            atomicAdd(&a.counters[0], (unsigned long long)nSx);
            if (a.tau4 == 6 && group_shape(a, g + gi).use_sadct) atomicAdd(&a.counters[1], 1ull);
        }
    }
}
#ifndef LFBM5D_T16_NOSPLIT
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(LFBM5D_T16_WAVES, LFBM5D_T16_WAVES))) void k_group_bior16_haar(GroupArgs a) { group_t16_kernel<true, true, false, true>(a); }
#else
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_group_bior16_haar(GroupArgs a) { group_t16_kernel<true, true, false>(a); }
#endif
__global__ __launch_bounds__(256) void k_group_bior16_any(GroupArgs a) { group_t16_kernel<false, true, false>(a); }
#ifndef LFBM5D_T16_NOSPLIT
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(LFBM5D_T16_WAVES, LFBM5D_T16_WAVES))) void k_group_dct16_haar(GroupArgs a) { group_t16_kernel<true, false, false, true>(a); }
#else
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_group_dct16_haar(GroupArgs a) { group_t16_kernel<true, false, false>(a); }
#endif
__global__ __launch_bounds__(256) void k_group_dct16_any(GroupArgs a) { group_t16_kernel<false, false, false>(a); }
/* N = 1: kT16Groups groups per workgroup (the 5th-dimension transform is the identity, HAAR or not) */
__global__ __launch_bounds__(256) void k_group_bior16_n1(GroupArgs a) { group_t16_kernel<true, true, true>(a); }
__global__ __launch_bounds__(256) void k_group_dct16_n1(GroupArgs a) { group_t16_kernel<true, false, true>(a); }
__global__ __launch_bounds__(256) void k_group_bior16_n1_sa(GroupArgs a) { group_t16_kernel<true, true, true, false, true>(a); }
__global__ __launch_bounds__(256) void k_group_dct16_n1_sa(GroupArgs a) { group_t16_kernel<true, false, true, false, true>(a); }

/* ------------------------------------------------------------------------------------------
 * 8x8 2-D DCT variant (the README Wiener configuration: k = 8, tau_2D = dct).  The 2-D transform
 * is done with ONE THREAD PER PATCH entirely in registers: the thread loads its 8x8 patch from the
 * window image, runs 8 row + 8 column 8-point DCTs (even/odd factorisation, orthonormal scaling
 * = REDFT10 x REDFT10 x coef_norm of bm3d.cpp:745-757,1148-1168) and scatters the 64 coefficients
 * into an LDS stack laid out [coefficient pq][patch] (+1 padding, conflict-free for every phase).
 * That replaces the gather + per-coefficient LDS matrix products of k_group (about 20 LDS
 * operations per stacked pixel) by one LDS write per pixel.  4-D / 5th-dimension phases work on
 * fibres of that stack as in k_group; the inverse 2-D DCT is again one thread per patch, reading
 * its 64 coefficients and storing the 64 pixels of the filtered patch as four-float vectors.
 * ------------------------------------------------------------------------------------------ */
constexpr int kDct8Threads = 320;

template <int STEP>
__global__ __launch_bounds__(kDct8Threads) void k_group_dct8(GroupArgs a) {
    extern __shared__ float lds[];
    __shared__ unsigned pos[kMaxN * kA3];
    __shared__ float red[3][kDct8Threads / 64];
    const int tid = threadIdx.x;
    const unsigned g = a.ref_begin + blockIdx.x;
    const int c = blockIdx.y;
    constexpr int A = 9, K2 = 64;
    const int N = a.N;
    const int nSx = (int)a.self_cnt[g];
    const size_t plane = (size_t)a.Wb * a.Hb;
    const int NP = nSx * A;               /* patches per stack */
    const int NPp = (N * A) | 1;          /* row stride of the [pq][patch] stack, odd */
    float* S0 = lds;
    float* S1 = lds + K2 * NPp;
    const TbPtr tb = (TbPtr)a.tb;
    constexpr int S = STEP == 2 ? 2 : 1;

    for (int i = tid; i < NP; i += kDct8Threads) pos[i] = a.gpos[(size_t)g * N * A + i];
    ShRef sh = group_shape(a, g);
    __syncthreads();
    const bool use_sadct = a.tau4 == 6 && sh.use_sadct;

    /* gather + forward 2-D DCT, one thread per patch */
    for (int task = tid; task < S * NP; task += kDct8Threads) {
        const int s = task / NP, patch = task % NP, st = patch % A;
        const unsigned p = pos[patch];
        const bool ok = p != 0xffffffffu;
        const float* img = (s ? a.basic : a.noisy) + ((size_t)st * a.C + c) * plane + (ok ? p : 0u);
        float x[8][8];
#pragma unroll
        for (int i = 0; i < 8; i++)
#pragma unroll
            for (int j = 0; j < 8; j++) { const float v = img[(size_t)i * a.Wb + j]; x[i][j] = ok ? v : 0.0f; }
#pragma unroll
        for (int i = 0; i < 8; i++) dct8_fwd(x[i]);
#pragma unroll
        for (int j = 0; j < 8; j++) {
            float col[8];
#pragma unroll
            for (int i = 0; i < 8; i++) col[i] = x[i][j];
            dct8_fwd(col);
#pragma unroll
            for (int i = 0; i < 8; i++) x[i][j] = col[i];
        }
        float* dst = (s ? S1 : S0) + patch;
#pragma unroll
        for (int i = 0; i < 8; i++)
#pragma unroll
            for (int j = 0; j < 8; j++) dst[(i * 8 + j) * NPp] = x[i][j];
    }
    __syncthreads();

    /* 4-D forward: one (n, pq) fibre of 9 values per thread */
    const bool do_dct4 = a.tau4 == 5 || (a.tau4 == 6 && !use_sadct);
    const bool do_sa4 = !do_dct4 && a.tau4 == 6;
    if (do_dct4 || do_sa4) {
        for (int f = tid; f < S * nSx * K2; f += kDct8Threads) {
            const int s = f / (nSx * K2), r = f % (nSx * K2), n = r / K2, pq = r % K2;
            float* base = (s ? S1 : S0) + pq * NPp + n * A;
            float x[9];
#pragma unroll
            for (int st = 0; st < 9; st++) x[st] = base[st];
            if (do_dct4) dct9_fwd(x, tb); else sadct9_fwd(x, sh, tb);
#pragma unroll
            for (int st = 0; st < 9; st++) base[st] = x[st];
        }
        __syncthreads();
    }

    /* 5th dimension + shrinkage: one (st, pq) fibre of nSx values per thread */
    float wacc = 0.0f, s1 = 0.0f, s2 = 0.0f;
    {
        const float sig = a.sigma[c];
        const float T = a.lambda * sig * 1.41421356237309505f;
        const float sig2 = sig * sig;
        for (int f = tid; f < A * K2; f += kDct8Threads) {
            const int st = f / K2, pq = f % K2;
            const bool in_shape = !use_sadct || sh.mask_dct[st];
            const int base = pq * NPp + st;
            switch (nSx) {
                case 1:  filter5<1, STEP>(S0, S1, base, A, a.tau5, T, sig2, in_shape, wacc, s1, s2, tb); break;
                case 2:  filter5<2, STEP>(S0, S1, base, A, a.tau5, T, sig2, in_shape, wacc, s1, s2, tb); break;
                case 4:  filter5<4, STEP>(S0, S1, base, A, a.tau5, T, sig2, in_shape, wacc, s1, s2, tb); break;
                case 8:  filter5<8, STEP>(S0, S1, base, A, a.tau5, T, sig2, in_shape, wacc, s1, s2, tb); break;
                default: filter5<16, STEP>(S0, S1, base, A, a.tau5, T, sig2, in_shape, wacc, s1, s2, tb); break;
            }
        }
    }
    for (int o = 32; o > 0; o >>= 1) { wacc += __shfl_xor(wacc, o); s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
    if ((tid & 63) == 0) { red[0][tid >> 6] = wacc; red[1][tid >> 6] = s1; red[2][tid >> 6] = s2; }
    __syncthreads();
    if (tid == 0) {
        float w = 0.0f, m = 0.0f, q = 0.0f;
        for (int i = 0; i < kDct8Threads / 64; i++) { w += red[0][i]; m += red[1][i]; q += red[2][i]; }
        float wx;
        if (a.useSD) {
            const float Nn = (float)(nSx * A);
            const float res = (q - m * m / Nn) / (Nn - 1.0f);
            wx = res > 0.0f ? 1.0f / sqrtf(res) : 0.0f;
        } else {
            const float sig = a.sigma[c];
            wx = w > 0.0f ? (sig > 0.0f ? 1.0f / (sig * sig * w) : 1.0f / w) : 1.0f;
        }
        a.wgt[(size_t)g * a.C + c] = wx;
        if (c == 0) {
            atomicAdd(&a.counters[0], (unsigned long long)nSx);
            if (use_sadct) atomicAdd(&a.counters[1], 1ull);
        }
    }
    float* F = STEP == 2 ? S1 : S0;

    /* 4-D inverse */
    if (do_dct4 || do_sa4) {
        for (int f = tid; f < nSx * K2; f += kDct8Threads) {
            const int n = f / K2, pq = f % K2;
            float* base = F + pq * NPp + n * A;
            float x[9];
#pragma unroll
            for (int st = 0; st < 9; st++) x[st] = base[st];
            if (do_dct4) dct9_inv(x, tb); else sadct9_inv(x, sh, tb);
#pragma unroll
            for (int st = 0; st < 9; st++) base[st] = x[st];
        }
    }
    __syncthreads();

    /* inverse 2-D DCT + store, one thread per patch: filt[g][n][st][c][64] */
    for (int patch = tid; patch < NP; patch += kDct8Threads) {
        float x[8][8];
        const float* src = F + patch;
#pragma unroll
        for (int i = 0; i < 8; i++)
#pragma unroll
            for (int j = 0; j < 8; j++) x[i][j] = src[(i * 8 + j) * NPp];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            float col[8];
#pragma unroll
            for (int i = 0; i < 8; i++) col[i] = x[i][j];
            dct8_inv(col);
#pragma unroll
            for (int i = 0; i < 8; i++) x[i][j] = col[i];
        }
#pragma unroll
        for (int i = 0; i < 8; i++) dct8_inv(x[i]);
        float4* out = reinterpret_cast<float4*>(a.filt + ((size_t)g * N * A + patch) * a.C * K2 + (size_t)c * K2);
#pragma unroll
        for (int i = 0; i < 8; i++) {
            out[2 * i] = make_float4(x[i][0], x[i][1], x[i][2], x[i][3]);
            out[2 * i + 1] = make_float4(x[i][4], x[i][5], x[i][6], x[i][7]);
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * Wiener step of the 8x8 DCT configuration (step 2, k = 8, tau_2D = dct): packed-fp32 variant.
 * The noisy and the pilot (basic) stacks go through identical forward transforms, so they are kept
 * as the two halves of a float2 everywhere: one LDS stack of float2 laid out [coefficient pq][patch],
 * 64-bit LDS accesses, and v_pk_{add,mul,fma}_f32 arithmetic that transforms both stacks at once.
 * Where only the filtered stack remains (inverse transforms) two fibres / two patches are paired
 * instead.  The arithmetic per element is the same sequence as in k_group_dct8<2>.
 * Phases (256 threads, barriers between them):
 *   1  cooperative 16-byte gather of both images through LDS, then thread = patch: 16 packed 8-point DCTs, 64 LDS writes
 *   2  thread = (n, pq) fibre over the 9 SAIs: packed 3x3 DCT (shape-adaptive variant on the scalar path)
 *   3  thread = (st, pq) fibre over the nSx patches: packed Haar, Wiener shrinkage, inverse Haar
 *   4  thread = two (n, pq) fibres of the filtered stack: packed inverse 3x3 DCT
 *   5  thread = two patches: packed inverse 8x8 DCT, 16-byte stores of the filtered patches
 * ------------------------------------------------------------------------------------------ */

/* 8-point orthonormal DCT-II / DCT-III of the Wiener kernels (T = float or a packed pair).  The 1/2 of the orthonormal
 * scaling is folded into the cosines (round 1 multiplied by it separately: a seventh more instructions, no effect then
 * because the kernel was latency-bound; at 60 % VALU utilisation it counts).  No threshold follows these transforms, so
 * the last-bit differences against the unfolded form stay far inside the Wiener step's tolerance. */
template <class T> __device__ __forceinline__ void dct8_fwd_t(T* x) {
    const float a0 = 0.35355339059327376f;
    const float c1 = 0.5f * 0.98078528040323044f, c2 = 0.5f * 0.92387953251128674f, c3 = 0.5f * 0.83146961230254524f,
                c4 = 0.5f * 0.70710678118654752f, c5 = 0.5f * 0.55557023301960222f, c6 = 0.5f * 0.38268343236508977f,
                c7 = 0.5f * 0.19509032201612827f;
    const T s0 = x[0] + x[7], s1 = x[1] + x[6], s2 = x[2] + x[5], s3 = x[3] + x[4];
    const T d0 = x[0] - x[7], d1 = x[1] - x[6], d2 = x[2] - x[5], d3 = x[3] - x[4];
    const T p0 = s0 + s3, p1 = s1 + s2, m0 = s0 - s3, m1 = s1 - s2;
    x[0] = a0 * (p0 + p1);
    x[4] = c4 * (p0 - p1);
    x[2] = c2 * m0 + c6 * m1;
    x[6] = c6 * m0 - c2 * m1;
    x[1] = c1 * d0 + c3 * d1 + c5 * d2 + c7 * d3;
    x[3] = c3 * d0 - c7 * d1 - c1 * d2 - c5 * d3;
    x[5] = c5 * d0 - c1 * d1 + c7 * d2 + c3 * d3;
    x[7] = c7 * d0 - c5 * d1 + c3 * d2 - c1 * d3;
}
template <class T> __device__ __forceinline__ void dct8_inv_t(T* X) {
    const float a0 = 0.35355339059327376f;
    const float c1 = 0.5f * 0.98078528040323044f, c2 = 0.5f * 0.92387953251128674f, c3 = 0.5f * 0.83146961230254524f,
                c4 = 0.5f * 0.70710678118654752f, c5 = 0.5f * 0.55557023301960222f, c6 = 0.5f * 0.38268343236508977f,
                c7 = 0.5f * 0.19509032201612827f;
    const T e0 = a0 * X[0] + c4 * X[4], e1 = a0 * X[0] - c4 * X[4];
    const T f0 = c2 * X[2] + c6 * X[6], f1 = c6 * X[2] - c2 * X[6];
    const T E0 = e0 + f0, E1 = e1 + f1, E2 = e1 - f1, E3 = e0 - f0;
    const T O0 = c1 * X[1] + c3 * X[3] + c5 * X[5] + c7 * X[7];
    const T O1 = c3 * X[1] - c7 * X[3] - c1 * X[5] - c5 * X[7];
    const T O2 = c5 * X[1] - c1 * X[3] + c7 * X[5] + c3 * X[7];
    const T O3 = c7 * X[1] - c5 * X[3] + c3 * X[5] - c1 * X[7];
    X[0] = E0 + O0; X[7] = E0 - O0;
    X[1] = E1 + O1; X[6] = E1 - O1;
    X[2] = E2 + O2; X[5] = E2 - O2;
    X[3] = E3 + O3; X[4] = E3 - O3;
}
template <int NS> __device__ __forceinline__ void haar_fwd2(v2f* v) {
    const float s = 0.70710678118654752f;
#pragma unroll
    for (int n = NS; n > 1; n /= 2) {
        v2f t[NS > 1 ? NS : 1];
#pragma unroll
        for (int i = 0; i < n / 2; i++) { t[i] = (v[2 * i] + v[2 * i + 1]) * s; t[n / 2 + i] = (v[2 * i] - v[2 * i + 1]) * s; }
#pragma unroll
        for (int i = 0; i < n; i++) v[i] = t[i];
    }
}

template <int NS> __device__ __forceinline__ void haar_inv2(v2f* v) {
    const float s = 0.70710678118654752f;
#pragma unroll
    for (int h = 1; h < NS; h *= 2) {
        v2f t[NS > 1 ? NS : 1];
#pragma unroll
        for (int i = 0; i < h; i++) { t[2 * i] = (v[i] + v[h + i]) * s; t[2 * i + 1] = (v[i] - v[h + i]) * s; }
#pragma unroll
        for (int i = 0; i < 2 * h; i++) v[i] = t[i];
    }
}

/* a / b for the Wiener coefficient e^2 / (e^2 + sigma^2) (0 <= a < b, both normal or a = 0): reciprocal estimate and
 * one correction step instead of the IEEE division sequence -- within one ulp of the quotient, which is well inside
 * the float tolerance of this stage (the division was a tenth of the kernel's instructions) */
__device__ __forceinline__ float wiener_div(float a, float b) {
#ifdef LFBM5D_WIENER_DIV_REFINED
    const float r = __builtin_amdgcn_rcpf(b);
    const float q = a * r;
    return __builtin_fmaf(__builtin_fmaf(-b, q, a), r, q);
#else
    return a * __builtin_amdgcn_rcpf(b);   /* v_rcp_f32 is good to 1 ulp: the quotient to 2 ulp, a Wiener coefficient needs no more */
#endif
}

/* phase 3 of k_group_dct8w on one (st, pq) fibre of nSx = NS float2 entries (x: noisy, y: pilot) */
template <int NS, bool HAAR>
__device__ __forceinline__ void wiener_fibre2(v2f* stack, int base, int stride, unsigned tau5, float sig2, bool in_shape,
                                              bool useSD, float& wacc, float& s1, float& s2, TbPtr tb) {
    v2f f[NS];
#pragma unroll
    for (int n = 0; n < NS; n++) f[n] = stack[base + n * stride];
    float o[NS], e[NS];
    if (HAAR) {
        if (NS > 1) haar_fwd2<NS>(f);
#pragma unroll
        for (int n = 0; n < NS; n++) { o[n] = f[n].x; e[n] = f[n].y; }
        if (in_shape) {
#pragma unroll
            for (int n = 0; n < NS; n++) {
                float value = e[n] * e[n];
                value = wiener_div(value, value + sig2);
                e[n] = o[n] * value;
                wacc += value;
            }
        }
        if (NS > 1) haar_inv<NS>(e);
    } else {
#pragma unroll
        for (int n = 0; n < NS; n++) { o[n] = f[n].x; e[n] = f[n].y; }
        shrink_fibre<NS, 2>(o, e, tau5, 0.0f, sig2, in_shape, wacc, tb);
    }
    float* dst = reinterpret_cast<float*>(stack);
#pragma unroll
    for (int n = 0; n < NS; n++) dst[2 * (base + n * stride) + 1] = e[n];
    if (useSD) {
#pragma unroll
        for (int n = 0; n < NS; n++) { s1 += e[n]; s2 += e[n] * e[n]; }
    }
}

#ifndef LFBM5D_DCT8W_THREADS
#define LFBM5D_DCT8W_THREADS 256
#endif
constexpr int kDct8wThreads = LFBM5D_DCT8W_THREADS;

/* bior1.5 on an 8x8 patch held by ONE thread (rows x[i][0..8)), all three levels in registers; T = float or a packed pair.
 * Same taps, order and unfused arithmetic as bior_fwd_level / bior_inv_level (lib_transforms.cpp:46-204). */
template <int N1, class T> __device__ __forceinline__ void bior_fwd_vec(T* v, TbPtr tb) {
#pragma clang fp contract(off)
    constexpr int N2 = N1 / 2;
    T o[N1];
#pragma unroll
    for (int j = 0; j < N2; j++) {
        T acc = v[bior_ext(2 * j, N1)] * tb->lpd[0];
#pragma unroll
        for (int t = 1; t < 10; t++) acc += v[bior_ext(t + 2 * j, N1)] * tb->lpd[t];
        o[j] = acc;
        T hi = v[bior_ext(4 + 2 * j, N1)] * tb->hpd[4];
        hi += v[bior_ext(5 + 2 * j, N1)] * tb->hpd[5];
        o[N2 + j] = hi;
    }
#pragma unroll
    for (int j = 0; j < N1; j++) v[j] = o[j];
}
template <int N1, class T> __device__ __forceinline__ void bior_inv_vec(T* v, TbPtr tb) {
#pragma clang fp contract(off)
    constexpr int N2 = N1 / 2;
    T o[N1];
#pragma unroll
    for (int m = 0; m < N2; m++) {
        T acc = v[m % N1] * tb->hpr[0];
#pragma unroll
        for (int t = 1; t < 10; t++) acc += v[(t * N2 + m) % N1] * tb->hpr[t];
        o[2 * m] = acc;
        T lo = v[(4 * N2 + m) % N1] * tb->lpr[4];
        lo += v[(5 * N2 + m) % N1] * tb->lpr[5];
        o[2 * m + 1] = lo;
    }
#pragma unroll
    for (int j = 0; j < N1; j++) v[j] = o[j];
}
template <int N1, bool FWD, bool ROWS, class T> __device__ __forceinline__ void bior8_pass(T (*x)[8], TbPtr tb) {
#pragma unroll
    for (int a = 0; a < N1; a++) {
        T v[N1];
#pragma unroll
        for (int b = 0; b < N1; b++) v[b] = ROWS ? x[a][b] : x[b][a];
        if (FWD) bior_fwd_vec<N1>(v, tb); else bior_inv_vec<N1>(v, tb);
#pragma unroll
        for (int b = 0; b < N1; b++) { if (ROWS) x[a][b] = v[b]; else x[b][a] = v[b]; }
    }
}
template <class T> __device__ __forceinline__ void bior8_fwd_2d(T (*x)[8], TbPtr tb) {
    bior8_pass<8, true, true>(x, tb); bior8_pass<8, true, false>(x, tb);
    bior8_pass<4, true, true>(x, tb); bior8_pass<4, true, false>(x, tb);
    bior8_pass<2, true, true>(x, tb); bior8_pass<2, true, false>(x, tb);
}
template <class T> __device__ __forceinline__ void bior8_inv_2d(T (*x)[8], TbPtr tb) {
    bior8_pass<2, false, false>(x, tb); bior8_pass<2, false, true>(x, tb);
    bior8_pass<4, false, false>(x, tb); bior8_pass<4, false, true>(x, tb);
    bior8_pass<8, false, false>(x, tb); bior8_pass<8, false, true>(x, tb);
}

template <bool HAAR, bool BIOR = false>   /* BIOR: tau_2D = bior1.5 instead of the DCT */
__global__ __launch_bounds__(kDct8wThreads) void k_group_dct8w(GroupArgs a) {
    extern __shared__ float lds[];
    __shared__ float red[3][kDct8wThreads / 64];
    const int tid = threadIdx.x;
    const unsigned g = a.ref_begin + blockIdx.x;
    const int c = blockIdx.y;
    constexpr int A = 9, K2 = 64;
    const int N = a.N;
    const int nSx = (int)a.self_cnt[g];
    const size_t plane = (size_t)a.Wb * a.Hb;
    const int NP = nSx * A;               /* patches per stack */
    const int NPp = (N * A) | 1;          /* row stride of the [pq][patch] stack (float2 units), odd */
    v2f* stack = reinterpret_cast<v2f*>(lds);
    float* stackf = lds;
    const TbPtr tb = (TbPtr)a.tb;

    ShRef sh = group_shape(a, g);

#ifdef LFBM5D_PHASE_TIMING
    long long tc[6]; int tci = 0;
#define PHASE_MARK() do { if (tid == 0) tc[tci] = (long long)__builtin_readcyclecounter(); tci++; } while (0)
    PHASE_MARK();
#else
#define PHASE_MARK() do {} while (0)
#endif
    /* 1a: cooperative gather.  The unit is one 16-byte half of a patch row: 16 adjacent lanes fetch the 8 rows of
     * one patch (two lanes per 32-byte row, one cache-line request), instead of every lane walking its own
     * patch.  The pieces are parked in LDS, [image][patch][16 pieces], piece index XOR-ed with the patch
     * index so that phase 1b reads them without bank conflicts; the area is reused by the stack afterwards. */
    {
        __shared__ unsigned pos[kMaxN * kA3];
        for (int i = tid; i < NP; i += kDct8wThreads) pos[i] = a.gpos[(size_t)g * N * A + i];
        __syncthreads();
        constexpr int kItems = (kMaxN * kA3 * 16 + kDct8wThreads - 1) / kDct8wThreads;   /* 9 */
        f4u v0[kItems], v1[kItems];
#pragma unroll
        for (int j = 0; j < kItems; j++) {
            const int it = tid + j * kDct8wThreads, patch = it >> 4, piece = it & 15;
            if (patch < NP) {
                const unsigned p = pos[patch];
                const size_t off = ((size_t)(patch % A) * a.C + c) * plane + (p != 0xffffffffu ? p : 0u)
                                   + (size_t)(piece >> 1) * a.Wb + 4 * (piece & 1);
                v0[j] = *reinterpret_cast<const f4u*>(a.noisy + off);
                v1[j] = *reinterpret_cast<const f4u*>(a.basic + off);
            }
        }
        v4f* stage = reinterpret_cast<v4f*>(lds);
#pragma unroll
        for (int j = 0; j < kItems; j++) {
            const int it = tid + j * kDct8wThreads, patch = it >> 4, piece = it & 15;
            if (patch < NP) {
                const bool ok = pos[patch] != 0xffffffffu;     /* empty SAI / never-filled table column: zeros */
                const int slot = patch * 16 + (piece ^ (patch & 15));
                stage[slot] = ok ? v4f{v0[j].v[0], v0[j].v[1], v0[j].v[2], v0[j].v[3]} : v4f{0.f, 0.f, 0.f, 0.f};
                stage[NP * 16 + slot] = ok ? v4f{v1[j].v[0], v1[j].v[1], v1[j].v[2], v1[j].v[3]} : v4f{0.f, 0.f, 0.f, 0.f};
            }
        }
        __syncthreads();
    }
    /* 1b: forward 2-D DCT of both images, one thread per patch */
    {
        const int patch = tid;       /* NP <= 144 < 256 threads */
        v2f x[8][8];
        if (patch < NP) {
            const v4f* stage = reinterpret_cast<const v4f*>(lds);
#pragma unroll
            for (int piece = 0; piece < 16; piece++) {
                const v4f l = stage[patch * 16 + (piece ^ (patch & 15))], r = stage[NP * 16 + patch * 16 + (piece ^ (patch & 15))];
#pragma unroll
                for (int j = 0; j < 4; j++) x[piece >> 1][4 * (piece & 1) + j] = v2f{l[j], r[j]};
            }
        }
        __syncthreads();             /* every patch is in registers: the area becomes the stack */
        if (patch < NP) {
            if (BIOR) bior8_fwd_2d(x, tb);
            else {
#pragma unroll
                for (int i = 0; i < 8; i++) dct8_fwd_t(x[i]);
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    v2f col[8];
#pragma unroll
                    for (int i = 0; i < 8; i++) col[i] = x[i][j];
                    dct8_fwd_t(col);
#pragma unroll
                    for (int i = 0; i < 8; i++) x[i][j] = col[i];
                }
            }
            v2f* dst = stack + patch;
#pragma unroll
            for (int i = 0; i < 8; i++)
#pragma unroll
                for (int j = 0; j < 8; j++) dst[(i * 8 + j) * NPp] = x[i][j];
        }
    }
    __syncthreads();
    PHASE_MARK();

    /* 2: 4-D forward, one (n, pq) fibre of 9 float2 per thread */
    const bool use_sadct = a.tau4 == 6 && sh.use_sadct;
    const bool do_dct4 = a.tau4 == 5 || (a.tau4 == 6 && !use_sadct);
    const bool do_sa4 = !do_dct4 && a.tau4 == 6;
    if (do_dct4 || do_sa4) {
        for (int f = tid; f < nSx * K2; f += kDct8wThreads) {
            const int n = f / K2, pq = f % K2;
            v2f* base = stack + pq * NPp + n * A;
            v2f x[9];
#pragma unroll
            for (int st = 0; st < 9; st++) x[st] = base[st];
            if (do_dct4) dct9_fwd2(x, tb);
            else {   /* rare: shape-adaptive transform on the scalar path */
                float t9[9];
#pragma unroll
                for (int i = 0; i < 9; i++) t9[i] = x[i].x;
                sadct9_fwd(t9, sh, tb);
#pragma unroll
                for (int i = 0; i < 9; i++) { x[i].x = t9[i]; t9[i] = x[i].y; }
                sadct9_fwd(t9, sh, tb);
#pragma unroll
                for (int i = 0; i < 9; i++) x[i].y = t9[i];
            }
#pragma unroll
            for (int st = 0; st < 9; st++) base[st] = x[st];
        }
        __syncthreads();
    }

    PHASE_MARK();
    /* 3: 5th dimension + Wiener shrinkage, one (st, pq) fibre of nSx float2 per thread; result -> .y */
    float wacc = 0.0f, s1 = 0.0f, s2 = 0.0f;
    {
        const float sig = a.sigma[c];
        const float sig2 = sig * sig;
        const bool useSD = a.useSD != 0;
        for (int f = tid; f < A * K2; f += kDct8wThreads) {
            const int st = f / K2, pq = f % K2;
            const bool in_shape = !use_sadct || sh.mask_dct[st];
            const int base = pq * NPp + st;
            switch (nSx) {
                case 1:  wiener_fibre2<1, HAAR>(stack, base, A, a.tau5, sig2, in_shape, useSD, wacc, s1, s2, tb); break;
                case 2:  wiener_fibre2<2, HAAR>(stack, base, A, a.tau5, sig2, in_shape, useSD, wacc, s1, s2, tb); break;
                case 4:  wiener_fibre2<4, HAAR>(stack, base, A, a.tau5, sig2, in_shape, useSD, wacc, s1, s2, tb); break;
                case 8:  wiener_fibre2<8, HAAR>(stack, base, A, a.tau5, sig2, in_shape, useSD, wacc, s1, s2, tb); break;
                default: wiener_fibre2<16, HAAR>(stack, base, A, a.tau5, sig2, in_shape, useSD, wacc, s1, s2, tb); break;
            }
        }
    }
    for (int o = 32; o > 0; o >>= 1) { wacc += __shfl_xor(wacc, o); s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
    if ((tid & 63) == 0) { red[0][tid >> 6] = wacc; red[1][tid >> 6] = s1; red[2][tid >> 6] = s2; }
    __syncthreads();
    PHASE_MARK();
    /* 4: 4-D inverse of the filtered stack (.y), two (n, pq) fibres per thread: pq and pq + 32 */
    if (do_dct4 || do_sa4) {
        for (int f = tid; f < nSx * (K2 / 2); f += kDct8wThreads) {
            const int n = f / (K2 / 2), pq = f % (K2 / 2);
            float* b0 = stackf + 2 * (pq * NPp + n * A) + 1;
            float* b1 = stackf + 2 * ((pq + K2 / 2) * NPp + n * A) + 1;
            v2f x[9];
#pragma unroll
            for (int st = 0; st < 9; st++) x[st] = v2f{b0[2 * st], b1[2 * st]};
            if (do_dct4) dct9_inv2(x, tb);
            else {
                float t9[9];
#pragma unroll
                for (int i = 0; i < 9; i++) t9[i] = x[i].x;
                sadct9_inv(t9, sh, tb);
#pragma unroll
                for (int i = 0; i < 9; i++) { x[i].x = t9[i]; t9[i] = x[i].y; }
                sadct9_inv(t9, sh, tb);
#pragma unroll
                for (int i = 0; i < 9; i++) x[i].y = t9[i];
            }
#pragma unroll
            for (int st = 0; st < 9; st++) { b0[2 * st] = x[st].x; b1[2 * st] = x[st].y; }
        }
    }
    __syncthreads();

    if (tid == kDct8wThreads - 64) {   /* the last wave takes no part in phase 5 (NP / 2 <= 72 patch pairs): the weight costs nothing there */
        float w = 0.0f, m = 0.0f, q = 0.0f;
        for (int i = 0; i < kDct8wThreads / 64; i++) { w += red[0][i]; m += red[1][i]; q += red[2][i]; }
        float wx;
        if (a.useSD) {
            const float Nn = (float)(nSx * A);
            const float res = (q - m * m / Nn) / (Nn - 1.0f);
            wx = res > 0.0f ? 1.0f / sqrtf(res) : 0.0f;
        } else {
            const float sig = a.sigma[c];
            wx = w > 0.0f ? (sig > 0.0f ? 1.0f / (sig * sig * w) : 1.0f / w) : 1.0f;
        }
        a.wgt[(size_t)g * a.C + c] = wx;
        if (c == 0) {
            atomicAdd(&a.counters[0], (unsigned long long)nSx);
            if (use_sadct) atomicAdd(&a.counters[1], 1ull);
        }
    }

    PHASE_MARK();
    /* 5: inverse 2-D DCT + store, two patches per thread (patch, patch + NPh): filt[g][n][st][c][64] */
    const int NPh = (NP + 1) / 2;
    for (int pa = tid; pa < NPh; pa += kDct8wThreads) {
        const int pb = pa + NPh;
        const bool has_b = pb < NP;
        const float* sa = stackf + 2 * pa + 1;
        const float* sb = stackf + 2 * (has_b ? pb : pa) + 1;
        v2f x[8][8];
#pragma unroll
        for (int i = 0; i < 8; i++)
#pragma unroll
            for (int j = 0; j < 8; j++) x[i][j] = v2f{sa[2 * (i * 8 + j) * NPp], sb[2 * (i * 8 + j) * NPp]};
        if (BIOR) bior8_inv_2d(x, tb);
        else {
#pragma unroll
            for (int j = 0; j < 8; j++) {
                v2f col[8];
#pragma unroll
                for (int i = 0; i < 8; i++) col[i] = x[i][j];
                dct8_inv_t(col);
#pragma unroll
                for (int i = 0; i < 8; i++) x[i][j] = col[i];
            }
#pragma unroll
            for (int i = 0; i < 8; i++) dct8_inv_t(x[i]);
        }
        float4* oa = reinterpret_cast<float4*>(a.filt + ((size_t)g * N * A + pa) * a.C * K2 + (size_t)c * K2);
#pragma unroll
        for (int i = 0; i < 8; i++) {
            oa[2 * i] = make_float4(x[i][0].x, x[i][1].x, x[i][2].x, x[i][3].x);
            oa[2 * i + 1] = make_float4(x[i][4].x, x[i][5].x, x[i][6].x, x[i][7].x);
        }
        if (has_b) {
            float4* ob = reinterpret_cast<float4*>(a.filt + ((size_t)g * N * A + pb) * a.C * K2 + (size_t)c * K2);
#pragma unroll
            for (int i = 0; i < 8; i++) {
                ob[2 * i] = make_float4(x[i][0].y, x[i][1].y, x[i][2].y, x[i][3].y);
                ob[2 * i + 1] = make_float4(x[i][4].y, x[i][5].y, x[i][6].y, x[i][7].y);
            }
        }
    }
#ifdef LFBM5D_PHASE_TIMING
    PHASE_MARK();
    if (tid == 0) {
        for (int i = 0; i < 5; i++) atomicAdd(&a.counters[4 + i], (unsigned long long)(tc[i + 1] - tc[i]));
        atomicAdd(&a.counters[9], 1ull);
    }
#endif
}

/* ------------------------------------------------------------------------------------------
 * Wiener step, 8x8 DCT, second generation: the same arithmetic as k_group_dct8w with the 2-D stages dealt to ALL
 * threads.  k_group_dct8w runs the forward 2-D DCT with one thread per patch (144 of 256 threads busy, fed through an
 * LDS staging area) and the inverse with one thread per patch PAIR (72 of 256): those two phases were 63 % of its time
 * at a quarter to a half of the lanes.  Here a 2-D DCT is two passes over the LDS stack with an item = one 8-point
 * packed transform:
 *   1a  item = (row i, patch): the thread loads its 32-byte row of both images straight from the window (no staging),
 *       transforms the noisy / pilot pair, writes 8 float2 into the stack [coefficient][patch]
 *   1b  item = (column j, patch): 8-point transform down the column, in place
 *   2-4 as before (3x3 angular DCT per (n, pq) fibre; Haar + Wiener + inverse Haar per (st, pq) fibre; inverse 3x3)
 *   5a  item = (column j, patch pair): inverse transform of the filtered stack, two patches packed
 *   5b  item = (row i, patch pair): inverse transform along the row, two 32-byte stores
 * Workgroups are 512 threads (eight wavefronts, two workgroups per CU at the stack's 74 KiB: 16 waves per CU; 192 threads --
 * every phase divides evenly -- measured three times slower, odd wave counts leave SIMDs unevenly loaded).
 * Items are numbered patch-fastest so that the stack accesses of a wavefront are consecutive float2 (no bank conflicts);
 * the global accesses are 32-byte row segments either way.
 * ------------------------------------------------------------------------------------------ */
#ifndef LFBM5D_DCT8W2_THREADS
#define LFBM5D_DCT8W2_THREADS 512
#endif
constexpr int kDct8w2Threads = LFBM5D_DCT8W2_THREADS;

template <bool HAAR>
__global__ __launch_bounds__(kDct8w2Threads) void k_group_dct8w2(GroupArgs a) {
    extern __shared__ float lds[];
    __shared__ float red[3][kDct8w2Threads / 64];
    __shared__ unsigned pos[kMaxN * kA3];
    constexpr int TH = kDct8w2Threads;
    const int tid = threadIdx.x;
    const unsigned gi = xcd_group_index(a);
    if (gi >= a.n_groups) return;
    const unsigned g = a.ref_begin + gi;
    const int c = blockIdx.y;
    constexpr int A = 9, K2 = 64;
    const int N = a.N;
    const int nSx = (int)a.self_cnt[g];
    const size_t plane = (size_t)a.Wb * a.Hb;
    const int NP = nSx * A;               /* patches per stack */
    const int NPp = (N * A) | 1;          /* row stride of the [pq][patch] stack (float2 units), odd */
    v2f* stack = reinterpret_cast<v2f*>(lds);
    float* stackf = lds;
    const TbPtr tb = (TbPtr)a.tb;
    /* item index -> (row or column, patch): division by NP / NPh through a 32-bit reciprocal (exact for items < 2^16) */
    const unsigned rcpNP = 0xffffffffu / (unsigned)NP + 1u, rcpNPh = 0xffffffffu / (unsigned)((NP + 1) / 2) + 1u;

    ShRef sh = group_shape(a, g);
    for (int i = tid; i < NP; i += TH) pos[i] = a.gpos[(size_t)g * N * A + i];
    __syncthreads();
#ifdef LFBM5D_PHASE_TIMING
    long long tc[6]; int tci = 0;
    PHASE_MARK();
#endif

    /* 1a: rows.  All loads of a thread's items are issued before the first transform */
    {
        constexpr int kIt = (kMaxN * kA3 * 8 + TH - 1) / TH;   /* 6 */
        f4u n0[kIt], n1[kIt], b0[kIt], b1[kIt];
#pragma unroll
        for (int q = 0; q < kIt; q++) {
            const int it = tid + q * TH;
#ifndef LFBM5D_W2_LOAD_ROWFAST
            const int i = (int)__umulhi((unsigned)it, rcpNP), patch = it - i * NP;
#else
            const int patch = it >> 3, i = it & 7;     /* row-fastest (measured: the stack writes then conflict, 1.46 vs 1.44 ms) */
#endif
            if (it < NP * 8) {
                const unsigned p = pos[patch];
                const size_t off = ((size_t)(patch % A) * a.C + c) * plane + (p != 0xffffffffu ? p : 0u) + (size_t)i * a.Wb;
                n0[q] = *reinterpret_cast<const f4u*>(a.noisy + off); n1[q] = *reinterpret_cast<const f4u*>(a.noisy + off + 4);
                b0[q] = *reinterpret_cast<const f4u*>(a.basic + off); b1[q] = *reinterpret_cast<const f4u*>(a.basic + off + 4);
            }
        }
#pragma unroll
        for (int q = 0; q < kIt; q++) {
            const int it = tid + q * TH;
#ifndef LFBM5D_W2_LOAD_ROWFAST
            const int i = (int)__umulhi((unsigned)it, rcpNP), patch = it - i * NP;
#else
            const int patch = it >> 3, i = it & 7;
#endif
            if (it < NP * 8) {
                const bool ok = pos[patch] != 0xffffffffu;     /* empty SAI / never-filled table column: zeros */
                v2f x[8];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    x[j] = ok ? v2f{n0[q].v[j], b0[q].v[j]} : v2f{0.f, 0.f};
                    x[4 + j] = ok ? v2f{n1[q].v[j], b1[q].v[j]} : v2f{0.f, 0.f};
                }
                dct8_fwd_t(x);
                v2f* dst = stack + (i * 8) * NPp + patch;
#pragma unroll
                for (int j = 0; j < 8; j++) dst[j * NPp] = x[j];
            }
        }
    }
    __syncthreads();
    PHASE_MARK();
    /* 1b: columns, in place */
    for (int it = tid; it < NP * 8; it += TH) {
        const int j = (int)__umulhi((unsigned)it, rcpNP), patch = it - j * NP;
        v2f* col = stack + j * NPp + patch;
        v2f x[8];
#pragma unroll
        for (int i = 0; i < 8; i++) x[i] = col[(i * 8) * NPp];
        dct8_fwd_t(x);
#pragma unroll
        for (int i = 0; i < 8; i++) col[(i * 8) * NPp] = x[i];
    }
    __syncthreads();
    PHASE_MARK();

    /* 2: 4-D forward, one (n, pq) fibre of 9 float2 per thread */
    const bool use_sadct = a.tau4 == 6 && sh.use_sadct;
    const bool do_dct4 = a.tau4 == 5 || (a.tau4 == 6 && !use_sadct);
    const bool do_sa4 = !do_dct4 && a.tau4 == 6;
    if (do_dct4 || do_sa4) {
        for (int f = tid; f < nSx * K2; f += TH) {
            const int n = f / K2, pq = f % K2;
            v2f* base = stack + pq * NPp + n * A;
            v2f x[9];
#pragma unroll
            for (int st = 0; st < 9; st++) x[st] = base[st];
            if (do_dct4) dct9_fwd2(x, tb);
            else {   /* rare: shape-adaptive transform on the scalar path */
                float t9[9];
#pragma unroll
                for (int i = 0; i < 9; i++) t9[i] = x[i].x;
                sadct9_fwd(t9, sh, tb);
#pragma unroll
                for (int i = 0; i < 9; i++) { x[i].x = t9[i]; t9[i] = x[i].y; }
                sadct9_fwd(t9, sh, tb);
#pragma unroll
                for (int i = 0; i < 9; i++) x[i].y = t9[i];
            }
#pragma unroll
            for (int st = 0; st < 9; st++) base[st] = x[st];
        }
        __syncthreads();
    }

    /* 3: 5th dimension + Wiener shrinkage, one (st, pq) fibre of nSx float2 per thread; result -> .y */
    float wacc = 0.0f, s1 = 0.0f, s2 = 0.0f;
    {
        const float sig = a.sigma[c];
        const float sig2 = sig * sig;
        const bool useSD = a.useSD != 0;
        for (int f = tid; f < A * K2; f += TH) {
            const int st = f / K2, pq = f % K2;
            const bool in_shape = !use_sadct || sh.mask_dct[st];
            const int base = pq * NPp + st;
            switch (nSx) {
                case 1:  wiener_fibre2<1, HAAR>(stack, base, A, a.tau5, sig2, in_shape, useSD, wacc, s1, s2, tb); break;
                case 2:  wiener_fibre2<2, HAAR>(stack, base, A, a.tau5, sig2, in_shape, useSD, wacc, s1, s2, tb); break;
                case 4:  wiener_fibre2<4, HAAR>(stack, base, A, a.tau5, sig2, in_shape, useSD, wacc, s1, s2, tb); break;
                case 8:  wiener_fibre2<8, HAAR>(stack, base, A, a.tau5, sig2, in_shape, useSD, wacc, s1, s2, tb); break;
                default: wiener_fibre2<16, HAAR>(stack, base, A, a.tau5, sig2, in_shape, useSD, wacc, s1, s2, tb); break;
            }
        }
    }
    for (int o = 32; o > 0; o >>= 1) { wacc += __shfl_xor(wacc, o); s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
    if ((tid & 63) == 0) { red[0][tid >> 6] = wacc; red[1][tid >> 6] = s1; red[2][tid >> 6] = s2; }
    __syncthreads();
    /* 4: 4-D inverse of the filtered stack (.y), two (n, pq) fibres per thread: pq and pq + 32 */
    if (do_dct4 || do_sa4) {
        for (int f = tid; f < nSx * (K2 / 2); f += TH) {
            const int n = f / (K2 / 2), pq = f % (K2 / 2);
            float* b0 = stackf + 2 * (pq * NPp + n * A) + 1;
            float* b1 = stackf + 2 * ((pq + K2 / 2) * NPp + n * A) + 1;
            v2f x[9];
#pragma unroll
            for (int st = 0; st < 9; st++) x[st] = v2f{b0[2 * st], b1[2 * st]};
            if (do_dct4) dct9_inv2(x, tb);
            else {
                float t9[9];
#pragma unroll
                for (int i = 0; i < 9; i++) t9[i] = x[i].x;
                sadct9_inv(t9, sh, tb);
#pragma unroll
                for (int i = 0; i < 9; i++) { x[i].x = t9[i]; t9[i] = x[i].y; }
                sadct9_inv(t9, sh, tb);
#pragma unroll
                for (int i = 0; i < 9; i++) x[i].y = t9[i];
            }
#pragma unroll
            for (int st = 0; st < 9; st++) { b0[2 * st] = x[st].x; b1[2 * st] = x[st].y; }
        }
    }
    __syncthreads();

    if (tid == 0) {
        float w = 0.0f, m = 0.0f, q = 0.0f;
        for (int i = 0; i < TH / 64; i++) { w += red[0][i]; m += red[1][i]; q += red[2][i]; }
        float wx;
        if (a.useSD) {
            const float Nn = (float)(nSx * A);
            const float res = (q - m * m / Nn) / (Nn - 1.0f);
            wx = res > 0.0f ? 1.0f / sqrtf(res) : 0.0f;
        } else {
            const float sig = a.sigma[c];
            wx = w > 0.0f ? (sig > 0.0f ? 1.0f / (sig * sig * w) : 1.0f / w) : 1.0f;
        }
        a.wgt[(size_t)g * a.C + c] = wx;
        if (c == 0) {
            atomicAdd(&a.counters[0], (unsigned long long)nSx);
            if (use_sadct) atomicAdd(&a.counters[1], 1ull);
        }
    }

    PHASE_MARK();
    /* 5a: inverse transform down the columns of the filtered stack (.y), two patches (pa, pa + NPh) packed, in place */
    const int NPh = (NP + 1) / 2;
    for (int it = tid; it < NPh * 8; it += TH) {
        const int j = (int)__umulhi((unsigned)it, rcpNPh), pa = it - j * NPh;
        const int pb = pa + NPh < NP ? pa + NPh : pa;
        float* ca = stackf + 2 * (j * NPp + pa) + 1;
        float* cb = stackf + 2 * (j * NPp + pb) + 1;
        v2f x[8];
#pragma unroll
        for (int i = 0; i < 8; i++) x[i] = v2f{ca[2 * (i * 8) * NPp], cb[2 * (i * 8) * NPp]};
        dct8_inv_t(x);
#pragma unroll
        for (int i = 0; i < 8; i++) { ca[2 * (i * 8) * NPp] = x[i].x; if (pb != pa) cb[2 * (i * 8) * NPp] = x[i].y; }
    }
    __syncthreads();
    PHASE_MARK();
    /* 5b: inverse transform along the rows + store: filt[g][n][st][c][64].  Items are numbered row-fastest: the eight
     * lanes that hold the rows of one patch write its 256 contiguous bytes */
#ifdef LFBM5D_W2_ROWS_PAIRED
    for (int it = tid; it < NPh * 8; it += TH) {
        const int i = it / NPh, pa = it - i * NPh;
        const int pb = pa + NPh;
        const bool has_b = pb < NP;
        const float* ra = stackf + 2 * ((i * 8) * NPp + pa) + 1;
        const float* rb = stackf + 2 * ((i * 8) * NPp + (has_b ? pb : pa)) + 1;
        v2f x[8];
#pragma unroll
        for (int j = 0; j < 8; j++) x[j] = v2f{ra[2 * j * NPp], rb[2 * j * NPp]};
        dct8_inv_t(x);
        float4* oa = reinterpret_cast<float4*>(a.filt + ((size_t)g * N * A + pa) * a.C * K2 + (size_t)c * K2 + i * 8);
        oa[0] = make_float4(x[0].x, x[1].x, x[2].x, x[3].x);
        oa[1] = make_float4(x[4].x, x[5].x, x[6].x, x[7].x);
        if (has_b) {
            float4* ob = reinterpret_cast<float4*>(a.filt + ((size_t)g * N * A + pb) * a.C * K2 + (size_t)c * K2 + i * 8);
            ob[0] = make_float4(x[0].y, x[1].y, x[2].y, x[3].y);
            ob[1] = make_float4(x[4].y, x[5].y, x[6].y, x[7].y);
        }
    }
#else
    for (int it = tid; it < NP * 8; it += TH) {
        const int patch = it >> 3, i = it & 7;
        const float* ra = stackf + 2 * ((i * 8) * NPp + patch) + 1;
        float x[8];
#pragma unroll
        for (int j = 0; j < 8; j++) x[j] = ra[2 * j * NPp];
        dct8_inv_t(x);
        float4* oa = reinterpret_cast<float4*>(a.filt + ((size_t)g * N * A + patch) * a.C * K2 + (size_t)c * K2 + i * 8);
        oa[0] = make_float4(x[0], x[1], x[2], x[3]);
        oa[1] = make_float4(x[4], x[5], x[6], x[7]);
    }
#endif
#ifdef LFBM5D_PHASE_TIMING
    PHASE_MARK();
    if (tid == 0) {
        for (int i = 0; i < 5; i++) atomicAdd(&a.counters[4 + i], (unsigned long long)(tc[i + 1] - tc[i]));
        atomicAdd(&a.counters[9], 1ull);
    }
#endif
}

/* ------------------------------------------------------------------------------------------
 * Wiener step, 8x8 DCT, Haar along the matches, third generation: ONE image in LDS at a time.  k_group_dct8w2 holds the
 * noisy and the pilot stack side by side (float2, 74 KiB: two workgroups per CU) and the time of a pass follows the
 * number of INDEPENDENT workgroups on a CU, not the number of waves (one workgroup of 8 or 16 waves per CU: 2.2 / 2.1 ms;
 * two of 8 waves: 1.4 ms -- a workgroup spends its phases waiting on one kind of unit, a second one in another phase
 * fills it).  Here the pilot goes through the forward transforms first, its Wiener coefficients e^2 / (e^2 + sigma^2)
 * stay in REGISTERS of the thread that owns the (st, pq) fibre, the noisy image then takes the same LDS: a float stack
 * of 36.5 KiB, four workgroups per CU.  The arithmetic per value is that of k_group_dct8w2 (same transforms on packed
 * pairs -- here two PATCHES per lane where that kernel packs the two images -- same shrinkage).
 *   P1-P3  pilot:  rows from the window (item = row i of patches 2pp, 2pp + 1), columns in place, 3x3 angular
 *   P4     Haar of the pilot fibre -> coefficient (or, outside the SADCT shape, the pilot value: quirk 10) in registers
 *   P5-P7  noisy: as P1-P3
 *   P8     Haar of the noisy fibre, times the coefficient, inverse Haar, in place
 *   P9-P11 inverse 3x3 (fibres pq, pq + 32 packed), inverse columns (patch pairs), inverse rows + store
 * Stack [pq][patch] floats, row stride 146: pairs of patches are 8-byte aligned, and 146 = 18 mod 32 with the lane
 * numbering of the fibre phases (16 values of pq x 2 neighbours in n or st) touches 32 distinct banks.
 * ------------------------------------------------------------------------------------------ */
#ifndef LFBM5D_DCT8W3_THREADS
#define LFBM5D_DCT8W3_THREADS 256
#endif
constexpr int kDct8w3Threads = LFBM5D_DCT8W3_THREADS;
constexpr int kW3Stride = 146;
constexpr unsigned kW3Lds = 64 * kW3Stride * sizeof(float);
constexpr unsigned kW3Empty = 0xf0000000u;   /* byte offset of an absent patch: beyond any window this kernel is launched on */

template <int NS, int TH, bool SA>
__device__ __forceinline__ void w3_forward(__amdgpu_buffer_rsrc_t img, unsigned row_bytes, float* S, const unsigned* pos,
                                           int tid, ShRef sh, bool do_dct4, bool do_sa4, TbPtr tb) {
    constexpr int A = 9, NP = NS * A, NPh = (NP + 1) / 2, NPf = kW3Stride;
    constexpr int kIt = (NPh * 8 + TH - 1) / TH;
    {
        /* pos[] holds byte offsets into the window (out of range for an empty SAI / never-filled column: the buffer load returns zeros) */
        v4f ra0[kIt], ra1[kIt], rb0[kIt], rb1[kIt];
#pragma unroll
        for (int q = 0; q < kIt; q++) {
            const int it = tid + q * TH;
            if (it < NPh * 8) {
                const int i = it / NPh, pp = it - i * NPh;
                const int pA = 2 * pp, pB = pA + 1 < NP ? pA + 1 : pA;
                const int oa = (int)(pos[pA] + (unsigned)i * row_bytes), ob = (int)(pos[pB] + (unsigned)i * row_bytes);
                ra0[q] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(img, oa, 0, 0));
                ra1[q] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(img, oa + 16, 0, 0));
                rb0[q] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(img, ob, 0, 0));
                rb1[q] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(img, ob + 16, 0, 0));
            }
        }
#pragma unroll
        for (int q = 0; q < kIt; q++) {
            const int it = tid + q * TH;
            if (it < NPh * 8) {
                const int i = it / NPh, pp = it - i * NPh;
                v2f x[8];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    x[j] = v2f{ra0[q][j], rb0[q][j]};
                    x[4 + j] = v2f{ra1[q][j], rb1[q][j]};
                }
                dct8_fwd_t(x);
                float* dst = S + (i * 8) * NPf + 2 * pp;
#pragma unroll
                for (int j = 0; j < 8; j++) *reinterpret_cast<v2f*>(dst + j * NPf) = x[j];
            }
        }
    }
    __syncthreads();
    for (int it = tid; it < NPh * 8; it += TH) {
        const int j = it / NPh, pp = it - j * NPh;
        float* col = S + j * NPf + 2 * pp;
        v2f x[8];
#pragma unroll
        for (int i = 0; i < 8; i++) x[i] = *reinterpret_cast<const v2f*>(col + (i * 8) * NPf);
        dct8_fwd_t(x);
#pragma unroll
        for (int i = 0; i < 8; i++) *reinterpret_cast<v2f*>(col + (i * 8) * NPf) = x[i];
    }
    __syncthreads();
    if (do_dct4 || do_sa4) {
        for (int f = tid; f < NS * 32; f += TH) {
            int n, pq;
            if (NS > 1) { pq = (f & 15) | ((f >> 1) & 16); n = ((f >> 4) & 1) | ((f >> 5) & ~1); }
            else { n = 0; pq = f; }
            float* b0 = S + pq * NPf + n * A;
            float* b1 = b0 + 32 * NPf;
            v2f x[9];
#pragma unroll
            for (int st = 0; st < 9; st++) x[st] = v2f{b0[st], b1[st]};
            if (do_dct4) dct9_fwd2_fast(x, tb);
            else {   /* rare: shape-adaptive transform on the scalar path */
                float t9[9];
#pragma unroll
                for (int i = 0; i < 9; i++) t9[i] = x[i].x;
                if (SA) sadct9_fwd_sel(t9, sh, tb); else sadct9_fwd(t9, sh, tb);
#pragma unroll
                for (int i = 0; i < 9; i++) { x[i].x = t9[i]; t9[i] = x[i].y; }
                if (SA) sadct9_fwd_sel(t9, sh, tb); else sadct9_fwd(t9, sh, tb);
#pragma unroll
                for (int i = 0; i < 9; i++) x[i].y = t9[i];
            }
#pragma unroll
            for (int st = 0; st < 9; st++) { b0[st] = x[st].x; b1[st] = x[st].y; }
        }
        __syncthreads();
    }
}

template <int NS, int TH, bool SA>
__device__ __forceinline__ void w3_body(const GroupArgs& a, float* S, const unsigned* pos, float (*red)[TH / 64], int tid,
                                        unsigned g, int c) {
    constexpr int A = 9, K2 = 64, NP = NS * A, NPh = (NP + 1) / 2, NPf = kW3Stride;
    static_assert(TH == 256, "the fibre phases deal st 0..3 / 4..7 to 256 threads");
    const int N = a.N;
    const unsigned win_bytes = (unsigned)((size_t)A * a.C * a.Wb * a.Hb * 4);
    const TbPtr tb = (TbPtr)a.tb;
    ShRef sh = group_shape(a, g);
    const bool use_sadct = a.tau4 == 6 && sh.use_sadct;
    const bool do_dct4 = a.tau4 == 5 || (a.tau4 == 6 && !use_sadct);
    const bool do_sa4 = !do_dct4 && a.tau4 == 6;
    const float sig = a.sigma[c];
    const float sig2 = sig * sig;
    const bool useSD = a.useSD != 0;

    w3_forward<NS, TH, SA>(__builtin_amdgcn_make_buffer_rsrc((void*)a.basic, 0, (int)win_bytes, 0x00020000u), a.Wb * 4u, S, pos, tid, sh, do_dct4, do_sa4, tb);
    /* fibres (st, pq): every thread owns st and st + 4 (st < 4) as a packed pair, the first wave also st = 8 */
    const int fpq = (tid & 15) | ((tid >> 1) & 48), fst = ((tid >> 4) & 1) | ((tid >> 6) & 2);
    float* const fbase = S + fpq * NPf + fst;
    float* const f8base = S + (tid & 63) * NPf + 8;
    const bool own8 = tid < 64;
    v2f vv[NS];
    float v8[NS];
    {
        v2f e[NS];
#pragma unroll
        for (int n = 0; n < NS; n++) e[n] = v2f{fbase[n * A], fbase[n * A + 4]};
        if (NS > 1) haar_fwd2<NS>(e);
#pragma unroll
        for (int n = 0; n < NS; n++) {
            const v2f value = e[n] * e[n], den = value + sig2;
            vv[n] = v2f{wiener_div(value.x, den.x), wiener_div(value.y, den.y)};
        }
        if (use_sadct) {   /* outside the shape the pilot's own coefficient passes through (quirk 10) */
#pragma unroll
            for (int n = 0; n < NS; n++) {
                if (!sh.mask_dct[fst]) vv[n].x = e[n].x;
                if (!sh.mask_dct[fst + 4]) vv[n].y = e[n].y;
            }
        }
    }
    if (own8) {
        float e[NS];
#pragma unroll
        for (int n = 0; n < NS; n++) e[n] = f8base[n * A];
        if (NS > 1) haar_fwd<NS>(e);
        const bool in8 = !use_sadct || sh.mask_dct[8];
#pragma unroll
        for (int n = 0; n < NS; n++) {
            const float value = e[n] * e[n];
            v8[n] = in8 ? wiener_div(value, value + sig2) : e[n];
        }
    }
    __syncthreads();
    w3_forward<NS, TH, SA>(__builtin_amdgcn_make_buffer_rsrc((void*)a.noisy, 0, (int)win_bytes, 0x00020000u), a.Wb * 4u, S, pos, tid, sh, do_dct4, do_sa4, tb);
    float wacc = 0.0f, s1 = 0.0f, s2 = 0.0f;
    {
        v2f o[NS];
#pragma unroll
        for (int n = 0; n < NS; n++) o[n] = v2f{fbase[n * A], fbase[n * A + 4]};
        if (NS > 1) haar_fwd2<NS>(o);
        if (!use_sadct) {
            v2f w2 = v2f{0.f, 0.f};
#pragma unroll
            for (int n = 0; n < NS; n++) { o[n] = o[n] * vv[n]; w2 += vv[n]; }
            wacc = w2.x + w2.y;
        } else {
            const bool ina = sh.mask_dct[fst], inb = sh.mask_dct[fst + 4];
#pragma unroll
            for (int n = 0; n < NS; n++) {
                o[n].x = ina ? o[n].x * vv[n].x : vv[n].x;
                o[n].y = inb ? o[n].y * vv[n].y : vv[n].y;
                wacc += (ina ? vv[n].x : 0.f) + (inb ? vv[n].y : 0.f);
            }
        }
        if (NS > 1) haar_inv2<NS>(o);
#pragma unroll
        for (int n = 0; n < NS; n++) { fbase[n * A] = o[n].x; fbase[n * A + 4] = o[n].y; }
        if (useSD) {
#pragma unroll
            for (int n = 0; n < NS; n++) { s1 += o[n].x + o[n].y; s2 += o[n].x * o[n].x + o[n].y * o[n].y; }
        }
    }
    if (own8) {
        float o[NS];
#pragma unroll
        for (int n = 0; n < NS; n++) o[n] = f8base[n * A];
        if (NS > 1) haar_fwd<NS>(o);
        if (!use_sadct || sh.mask_dct[8]) {
#pragma unroll
            for (int n = 0; n < NS; n++) { o[n] = o[n] * v8[n]; wacc += v8[n]; }
        } else {
#pragma unroll
            for (int n = 0; n < NS; n++) o[n] = v8[n];
        }
        if (NS > 1) haar_inv<NS>(o);
#pragma unroll
        for (int n = 0; n < NS; n++) f8base[n * A] = o[n];
        if (useSD) {
#pragma unroll
            for (int n = 0; n < NS; n++) { s1 += o[n]; s2 += o[n] * o[n]; }
        }
    }
    for (int o = 32; o > 0; o >>= 1) { wacc += __shfl_xor(wacc, o); s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
    if ((tid & 63) == 0) { red[0][tid >> 6] = wacc; red[1][tid >> 6] = s1; red[2][tid >> 6] = s2; }
    __syncthreads();
    if (do_dct4 || do_sa4) {
        for (int f = tid; f < NS * 32; f += TH) {
            int n, pq;
            if (NS > 1) { pq = (f & 15) | ((f >> 1) & 16); n = ((f >> 4) & 1) | ((f >> 5) & ~1); }
            else { n = 0; pq = f; }
            float* b0 = S + pq * NPf + n * A;
            float* b1 = b0 + 32 * NPf;
            v2f x[9];
#pragma unroll
            for (int st = 0; st < 9; st++) x[st] = v2f{b0[st], b1[st]};
            if (do_dct4) dct9_inv2_fast(x, tb);
            else {
                float t9[9];
#pragma unroll
                for (int i = 0; i < 9; i++) t9[i] = x[i].x;
                if (SA) sadct9_inv_sel(t9, sh, tb); else sadct9_inv(t9, sh, tb);
#pragma unroll
                for (int i = 0; i < 9; i++) { x[i].x = t9[i]; t9[i] = x[i].y; }
                if (SA) sadct9_inv_sel(t9, sh, tb); else sadct9_inv(t9, sh, tb);
#pragma unroll
                for (int i = 0; i < 9; i++) x[i].y = t9[i];
            }
#pragma unroll
            for (int st = 0; st < 9; st++) { b0[st] = x[st].x; b1[st] = x[st].y; }
        }
    }
    __syncthreads();
    if (tid == 0) {
        float w = 0.0f, m = 0.0f, q = 0.0f;
        for (int i = 0; i < TH / 64; i++) { w += red[0][i]; m += red[1][i]; q += red[2][i]; }
        float wx;
        if (a.useSD) {
            const float Nn = (float)(NS * A);
            const float res = (q - m * m / Nn) / (Nn - 1.0f);
            wx = res > 0.0f ? 1.0f / sqrtf(res) : 0.0f;
        } else {
            wx = w > 0.0f ? (sig > 0.0f ? 1.0f / (sig * sig * w) : 1.0f / w) : 1.0f;
        }
        a.wgt[(size_t)g * a.C + c] = wx;
        if (c == 0) {
            atomicAdd(&a.counters[0], (unsigned long long)NS);
            if (use_sadct) atomicAdd(&a.counters[1], 1ull);
        }
    }
    for (int it = tid; it < NPh * 8; it += TH) {
        const int j = it / NPh, pp = it - j * NPh;
        float* col = S + j * NPf + 2 * pp;
        v2f x[8];
#pragma unroll
        for (int i = 0; i < 8; i++) x[i] = *reinterpret_cast<const v2f*>(col + (i * 8) * NPf);
        dct8_inv_t(x);
#pragma unroll
        for (int i = 0; i < 8; i++) *reinterpret_cast<v2f*>(col + (i * 8) * NPf) = x[i];
    }
    __syncthreads();
    /* rows + store: filt[g][n][st][c][64].  Patches (pa, pa + NPh) packed, row-fastest: the eight lanes with the rows of one
     * patch write its 256 bytes */
    for (int it = tid; it < NPh * 8; it += TH) {
        const int pa = it >> 3, i = it & 7;
        const bool has_b = pa + NPh < NP;
        const float* ra = S + (i * 8) * NPf + pa;
        v2f x[8];
#pragma unroll
        for (int j = 0; j < 8; j++) x[j] = v2f{ra[j * NPf], ra[j * NPf + (NP > NPh ? NPh : 0)]};
        dct8_inv_t(x);
        float4* oa = reinterpret_cast<float4*>(a.filt + ((size_t)g * N * A + pa) * a.C * K2 + (size_t)c * K2 + i * 8);
        oa[0] = make_float4(x[0].x, x[1].x, x[2].x, x[3].x);
        oa[1] = make_float4(x[4].x, x[5].x, x[6].x, x[7].x);
        if (has_b) {
            float4* ob = oa + (size_t)NPh * a.C * (K2 / 4);
            ob[0] = make_float4(x[0].y, x[1].y, x[2].y, x[3].y);
            ob[1] = make_float4(x[4].y, x[5].y, x[6].y, x[7].y);
        }
    }
}

#ifndef LFBM5D_W3SA_WAVES
#define LFBM5D_W3SA_WAVES 4   /* 128 VGPRs (a few spills): four workgroups per CU beat 129 without */
#endif
template <bool SA>   /* SA: for windows with an empty SAI (every group shape-adaptive): the transform inline, in registers */
__global__ __launch_bounds__(kDct8w3Threads) __attribute__((amdgpu_waves_per_eu(SA ? LFBM5D_W3SA_WAVES : 1))) void k_group_dct8w3(GroupArgs a) {
    extern __shared__ float lds[];
    __shared__ float red[3][kDct8w3Threads / 64];
    __shared__ unsigned pos[kMaxN * kA3];
    constexpr int TH = kDct8w3Threads;
    const int tid = threadIdx.x;
    const unsigned gi = xcd_group_index(a);
    if (gi >= a.n_groups) return;
    const unsigned g = a.ref_begin + gi;
    const int c = blockIdx.y;
    const int nSx = (int)a.self_cnt[g];
    const unsigned plane = a.Wb * a.Hb;
    for (int i = tid; i < nSx * 9; i += TH) {
        const unsigned p = a.gpos[(size_t)g * a.N * 9 + i];
        pos[i] = p != 0xffffffffu ? (((unsigned)(i % 9) * a.C + c) * plane + p) * 4u : kW3Empty;
    }
    __syncthreads();
#ifdef LFBM5D_W3_ONLY16
    w3_body<16, TH, SA>(a, lds, pos, red, tid, g, c); return;
#endif
    switch (nSx) {
        case 1:  w3_body<1, TH, SA>(a, lds, pos, red, tid, g, c); break;
        case 2:  w3_body<2, TH, SA>(a, lds, pos, red, tid, g, c); break;
        case 4:  w3_body<4, TH, SA>(a, lds, pos, red, tid, g, c); break;
        case 8:  w3_body<8, TH, SA>(a, lds, pos, red, tid, g, c); break;
        default: w3_body<16, TH, SA>(a, lds, pos, red, tid, g, c); break;
    }
}

/* ------------------------------------------------------------------------------------------
 * Per-SAI BM3D flavour, 8x8 patches (LFBM3Ddenoising's parameters: bm3d.cpp:315-690 with kHard = kWien = 8).
 * A group is nSx <= 32 patches of ONE image, so a whole group fits a WAVEFRONT: four groups per workgroup, no
 * workgroup barrier anywhere.  Lane = patch for the 2-D stages (the 8x8 patch and its transform in registers, the
 * noisy / pilot pair packed in the Wiener step), lane = coefficient for the Hadamard + shrinkage along the stack
 * (ht_filtering_hadamard :914-966, wiener_filtering_hadamard :980-1027), through an LDS stack [coefficient][patch]
 * of the wave's own.  The generic group kernel spends a 256-thread workgroup on such a group.
 * ------------------------------------------------------------------------------------------ */
constexpr int kBm3dWaves = 4;
template <int STEP> struct Bm3dT { typedef float type; };
template <> struct Bm3dT<2> { typedef v2f type; };
__device__ __forceinline__ float bm3d_first(float x) { return x; }
__device__ __forceinline__ float bm3d_first(v2f x) { return x.x; }

template <int STEP, bool BIOR>
__global__ __launch_bounds__(64 * kBm3dWaves) void k_group_bm3d8(GroupArgs a) {
    extern __shared__ float lds[];
    typedef typename Bm3dT<STEP>::type T;
    constexpr int K2 = 64, ST = kMaxN3 + 1;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const unsigned gi = blockIdx.x * kBm3dWaves + wave;
    if (gi >= a.n_groups) return;                      /* the whole wavefront leaves: nothing below synchronises waves */
    const unsigned g = a.ref_begin + gi;
    const int c = blockIdx.y;
    const int N = a.N, nSx = (int)a.self_cnt[g];
    const size_t plane = (size_t)a.Wb * a.Hb;
    const TbPtr tb = (TbPtr)a.tb;
    T* S = reinterpret_cast<T*>(lds) + (size_t)wave * K2 * ST;

    /* A: lane = patch: load, forward 2-D transform, scatter to the stack */
    if (lane < nSx) {
        const unsigned p = a.gpos[(size_t)g * N + lane];
        const bool ok = p != 0xffffffffu;              /* never-filled table column: zeros (bm3d.cpp:737, :857) */
        const size_t off = (size_t)c * plane + (ok ? p : 0u);
        T x[8][8];
#pragma unroll
        for (int i = 0; i < 8; i++)
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const f4u q0 = *reinterpret_cast<const f4u*>(a.noisy + off + (size_t)i * a.Wb + 4 * h);
                if (STEP == 2) {
                    const f4u q1 = *reinterpret_cast<const f4u*>(a.basic + off + (size_t)i * a.Wb + 4 * h);
#pragma unroll
                    for (int e = 0; e < 4; e++) reinterpret_cast<v2f&>(x[i][4 * h + e]) = ok ? v2f{q0.v[e], q1.v[e]} : v2f{0.0f, 0.0f};
                } else {
#pragma unroll
                    for (int e = 0; e < 4; e++) reinterpret_cast<float&>(x[i][4 * h + e]) = ok ? q0.v[e] : 0.0f;
                }
            }
        if (BIOR) bior8_fwd_2d(x, tb);
        else {
#pragma unroll
            for (int i = 0; i < 8; i++) dct8_fwd_t(x[i]);
#pragma unroll
            for (int j = 0; j < 8; j++) {
                T col[8];
#pragma unroll
                for (int i = 0; i < 8; i++) col[i] = x[i][j];
                dct8_fwd_t(col);
#pragma unroll
                for (int i = 0; i < 8; i++) x[i][j] = col[i];
            }
        }
#pragma unroll
        for (int i = 0; i < 8; i++)
#pragma unroll
            for (int j = 0; j < 8; j++) S[(i * 8 + j) * ST + lane] = x[i][j];
    }
    __builtin_amdgcn_wave_barrier();

    /* B: lane = coefficient: Hadamard along the stack, shrinkage, inverse; filtered value back in place (Wiener: into .y) */
    float wacc = 0.0f, s1 = 0.0f, s2 = 0.0f;
    {
        const float sig = a.sigma[c];
        const float Tthr = a.lambda * sig;             /* * sqrt(nSx) inside shrink_fibre (bm3d.cpp:941) */
        T* F = S + lane * ST;
        auto fibre = [&](auto ns_tag) {
            constexpr int NS = decltype(ns_tag)::value;
            float o[NS], e[NS];
#pragma unroll
            for (int n = 0; n < NS; n++) {
                if (STEP == 2) { const v2f t = reinterpret_cast<const v2f&>(F[n]); o[n] = t.x; e[n] = t.y; }
                else { o[n] = reinterpret_cast<const float&>(F[n]); e[n] = 0.0f; }
            }
            shrink_fibre<NS, STEP>(o, e, 8u, Tthr, sig * sig, true, wacc, tb);
#pragma unroll
            for (int n = 0; n < NS; n++) {
                const float r = STEP == 1 ? o[n] : e[n];
                s1 += r; s2 += r * r;
                if (STEP == 2) reinterpret_cast<v2f&>(F[n]).y = r; else reinterpret_cast<float&>(F[n]) = r;
            }
        };
        switch (nSx) {
            case 2:  fibre(std::integral_constant<int, 2>{}); break;
            case 4:  fibre(std::integral_constant<int, 4>{}); break;
            case 8:  fibre(std::integral_constant<int, 8>{}); break;
            case 16: fibre(std::integral_constant<int, 16>{}); break;
            default: fibre(std::integral_constant<int, 32>{}); break;
        }
    }
    for (int o = 32; o > 0; o >>= 1) { wacc += __shfl_xor(wacc, o); s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
    if (lane == 0) {
        float wx;
        if (a.useSD) {                                 /* sd_weighting, bm3d.cpp:1345-1373 */
            const float Nn = (float)(nSx * K2);
            const float res = (s2 - s1 * s1 / Nn) / (Nn - 1.0f);
            wx = res > 0.0f ? 1.0f / sqrtf(res) : 0.0f;
        } else {
            const float sig = a.sigma[c];
            wx = wacc > 0.0f ? (sig > 0.0f ? 1.0f / (sig * sig * wacc) : 1.0f / wacc) : 1.0f;
        }
        a.wgt[(size_t)g * a.C + c] = wx;
        if (c == 0) atomicAdd(&a.counters[0], (unsigned long long)nSx);
    }
    __builtin_amdgcn_wave_barrier();

    /* C: lane = patch: inverse 2-D transform of the filtered coefficients, 16-byte stores: filt[g][n][c][64] */
    if (lane < nSx) {
        float x[8][8];
#pragma unroll
        for (int i = 0; i < 8; i++)
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const T t = S[(i * 8 + j) * ST + lane];
                if (STEP == 2) x[i][j] = reinterpret_cast<const v2f&>(t).y; else x[i][j] = bm3d_first(t);
            }
        if (BIOR) bior8_inv_2d(x, tb);
        else {
#pragma unroll
            for (int j = 0; j < 8; j++) {
                float col[8];
#pragma unroll
                for (int i = 0; i < 8; i++) col[i] = x[i][j];
                dct8_inv_t(col);
#pragma unroll
                for (int i = 0; i < 8; i++) x[i][j] = col[i];
            }
#pragma unroll
            for (int i = 0; i < 8; i++) dct8_inv_t(x[i]);
        }
        float4* dst = reinterpret_cast<float4*>(a.filt + (((size_t)g * N + lane) * a.C + c) * K2);
#pragma unroll
        for (int i = 0; i < 8; i++) {
            dst[2 * i] = make_float4(x[i][0], x[i][1], x[i][2], x[i][3]);
            dst[2 * i + 1] = make_float4(x[i][4], x[i][5], x[i][6], x[i][7]);
        }
    }
}

/* ================================ aggregation kernel ====================================== */


/* Gather form of core:484-528.  One wavefront = one tile of 64 pixels of one SAI (thread = pixel).
 * Candidates are the patch instances (reference patch in raster order, then match index n) of the
 * reference patches whose search range can reach the tile; every pixel adds its contributions in
 * that order -- the reference's order for that pixel -- starting from the value already in num/den.
 * Everything is wave-synchronous and built to keep many loads in flight (the kernel is a chain of
 * dependent gathers, so latency, not bandwidth, is what has to be hidden):
 *   scan     kAggPF chunks of 64 candidates at a time: their aggregation positions are loaded
 *            together, then the group weights of the hits, then the hits are appended in candidate
 *            order (ballot prefix) to a hit list in LDS: position, patch offset in filt, weights;
 *   consume  once the list holds enough hits (or is full, or at the end) all lanes walk it in order, kAggU hits
 *            per round: the loads of a round are issued together, the adds stay in list order.
 * Workgroups are renumbered so that the tiles one XCD works on at a time are neighbours: the
 * filtered patches they share are then fetched into that XCD's L2 once. */
constexpr int kAggFlush = 64;   /* hits that make a consume phase worth starting */
/* WINDOWED: Kaiser window (k = 8, 12); any other size has an all-ones window (bm3d.cpp:1144-1146).
 * TW x TH: tile shape (64 pixels).  A filtered patch row is k floats, so wide flat tiles read longer
 * contiguous runs of it: 16x4 for k >= 12 (64-byte rows), 8x8 for k = 8 (a whole patch is two cache lines).
 * kAggPF chunks per scan round and kAggU hits per load round trade latency hiding against registers and LDS
 * (occupancy): the k = 8 pass has few hits per candidate and wants occupancy, the k = 16 pass deeper rounds.
 * BIG: filt is 4 GiB or more (windows far beyond 560 x 560): 64-bit gather addresses instead of a buffer resource. */
template <bool WINDOWED, int TW, int TH, int kAggPF, int kAggU, bool BIG, bool VEC4>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu((VEC4 && TW == 8 && !BIG) ? 8 : 1))) void k_aggregate(AggArgs a) {
    constexpr int kAggCap = kAggFlush + kAggPF * 64 + kAggU;   /* + padding of the last round */
    __shared__ uint4 hit_a[kAggCap];       /* (py << 16) | px, offset of the patch in filt, weights of channels 0 and 1 */
    __shared__ float hit_w2[kAggCap];      /* weight of channel 2 */
    __shared__ float kai[WINDOWED ? 256 : 1];   /* Kaiser windows exist for 8x8 and 12x12 patches only (bm3d.cpp:1101-1146) */
    const int lane = threadIdx.x;
    /* XCD-aware renumbering: hardware deals consecutive workgroup ids round-robin to the 8 XCDs */
    const unsigned gx = (a.Wb + TW - 1) / TW, gy = (a.Hb + TH - 1) / TH, total_wg = gx * gy * a.A;
    const unsigned per_xcd = gridDim.x / 8;          /* the launch is rounded up to a multiple of 8 workgroups */
    const unsigned lin2 = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
    if (lin2 >= total_wg) return;
    const int st = (int)(lin2 / (gx * gy));
    const int tile_y = (int)((lin2 / gx) % gy), tile_x = (int)(lin2 % gx);
    if (a.proc_bits.test((unsigned)st)) return;      /* procSAI[st] != 0: skipped (core:486) */
    if (!a.mask_bits.test((unsigned)st)) return;
    const int tx0 = tile_x * TW, ty0 = tile_y * TH;
    const int x = tx0 + lane % TW, y = ty0 + lane / TW;
    const bool inside = x < (int)a.Wb && y < (int)a.Hb;
    const int k = a.k, k2 = k * k, C = a.C, N = a.N, A = a.A;
    const int logN = 31 - __builtin_clz((unsigned)N);   /* N is a power of two (lfbm5d_api.hip validate) */
    const size_t plane = (size_t)a.Wb * a.Hb;
    const int reach = (st == (int)a.pst) ? (int)a.nSim : (int)a.nHW;
    if (WINDOWED) for (int i = lane; i < k2; i += 64) kai[i] = a.tb->kaiser[i];

    /* reference-grid index ranges that can reach this tile (grid = nHW + i*p, plus a forced last
     * index, utilities.cpp:697-712) */
    const int last_r = (int)a.Hb - k - (int)a.nHW, last_c = (int)a.Wb - k - (int)a.nHW;
    auto lo_idx = [&](int v) { int d = v - (int)a.nHW; return d <= 0 ? 0 : (d + (int)a.p - 1) / (int)a.p; };
    auto hi_idx = [&](int v, int n, int lastv) { /* largest index whose coordinate <= v */
        if (v >= lastv) return n - 1;
        int d = v - (int)a.nHW; if (d < 0) return -1;
        int i = d / (int)a.p; return i > n - 2 ? n - 2 : i;
    };
    int r_lo = lo_idx(ty0 - k + 1 - reach), r_hi = hi_idx(ty0 + TH - 1 + reach, (int)a.n_ref_rows, last_r);
    int c_lo = lo_idx(tx0 - k + 1 - reach), c_hi = hi_idx(tx0 + TW - 1 + reach, (int)a.n_ref_cols, last_c);
    if (r_lo > (int)a.n_ref_rows - 1) r_lo = (int)a.n_ref_rows - 1; /* the forced last index may sit closer than p */
    if (c_lo > (int)a.n_ref_cols - 1) c_lo = (int)a.n_ref_cols - 1;
    /* a launch covers the groups [ref_begin, ref_begin + n_groups): whole rows of the reference grid (a band of a pass processed band by
     * band, a rank's share of a row-sharded pass) -- only those rows' candidates are enumerated, and a tile none of them can reach
     * returns before it has touched num / den.  Bands launched in raster order add up in the reference's order (core:484-528): the sums
     * do not depend on how a pass is cut. */
    if (!a.irregular && a.n_ref_cols) {
        r_lo = max(r_lo, (int)(a.ref_begin / a.n_ref_cols));
        r_hi = min(r_hi, (int)((a.ref_begin + a.n_groups - 1) / a.n_ref_cols));
    }
    const int ncols_span = c_hi - c_lo + 1;
    /* (an irregular list -- subset passes -- is in raster order too: the launch's slice of it) */
    const int n_rr = a.irregular ? (int)a.n_groups : ((r_hi >= r_lo && c_hi >= c_lo) ? (r_hi - r_lo + 1) * ncols_span : 0);
    const int n_cand = n_rr << logN;
    if (n_cand == 0) return;

    float accn[3] = {0, 0, 0}, accd[3] = {0, 0, 0};
    const size_t pix = (size_t)st * C * plane + (size_t)y * a.Wb + x;
    if (inside) for (int c = 0; c < C; c++) { accn[c] = a.num[pix + c * plane]; accd[c] = a.den[pix + c * plane]; }

    /* rr / ncols_span by multiplication: exact while rr < 2^20 / ncols_span (rr is a few hundred) */
    const bool mul_div = !a.irregular && (long long)n_rr * ncols_span < (1 << 20);
    const unsigned div_m = ((1u << 20) + (unsigned)max(ncols_span, 1) - 1) / (unsigned)max(ncols_span, 1);
    const unsigned g_end = a.ref_begin + a.n_groups;
    /* every patch index (g N + n) A + st below 2^24: full-rate 24-bit multiplies for the per-candidate index arithmetic */
    const bool small24 = (((unsigned long long)a.n_refs_total << logN) + 1) * A < (1ull << 24);
    const unsigned* apos = a.aggpos + (size_t)st * a.n_refs_total * N;
    /* channel stride inside a filtered patch, bytes; greyscale: all three loads read channel 0 (its weights are 0) */
    const unsigned cstride = C > 1 ? (unsigned)k2 * 4u : 0u;
    const __amdgpu_buffer_rsrc_t rs_filt = __builtin_amdgcn_make_buffer_rsrc((void*)a.filt, 0, BIG ? 0 : (int)(unsigned)a.filt_bytes, 0x00020000u);
    unsigned nh = 0;   /* hits in the list (uniform) */

    auto consume = [&]() {
        /* pad the list to whole rounds with entries no pixel is covered by (position 0xffff, 0xffff; first patch; weight 0) */
        const unsigned nh_pad = (nh + kAggU - 1) / kAggU * kAggU;
        if (nh + lane < nh_pad) { hit_a[nh + lane] = make_uint4(0xffffffffu, 0u, 0u, 0u); hit_w2[nh + lane] = 0.0f; }
        __builtin_amdgcn_wave_barrier();
        for (unsigned h0 = 0; h0 < nh_pad; h0 += kAggU) {
            float val[kAggU][3], kw[kAggU][3];
#pragma unroll
            for (int u = 0; u < kAggU; u++) {
                const uint4 ha = hit_a[h0 + u];
                const float w2 = hit_w2[h0 + u];
                const int dy = y - (int)(ha.x >> 16), dx = x - (int)(ha.x & 0xffffu);
                const bool on = (unsigned)dy < (unsigned)k && (unsigned)dx < (unsigned)k;
                /* pixels the patch does not cover read the nearest pixel it does cover -- a pixel of this tile,
                 * so no extra cache line is touched ... */
                const unsigned o = __umul24((unsigned)min(max(dy, 0), k - 1), (unsigned)k) + (unsigned)min(max(dx, 0), k - 1);   /* v_mad_u32_u24: a 32-bit multiply is quarter rate */
                const float kz = WINDOWED ? kai[o] : 1.0f;
                if (BIG) {
                    const char* fp = reinterpret_cast<const char*>(a.filt) + ((size_t)ha.y + o) * 4;
                    val[u][0] = *reinterpret_cast<const float*>(fp);
                    val[u][1] = *reinterpret_cast<const float*>(fp + cstride);
                    val[u][2] = *reinterpret_cast<const float*>(fp + 2 * cstride);
                } else {
                    const int vo = (int)((ha.y + o) * 4u);
#if defined(LFBM5D_AGG_EXP) && LFBM5D_AGG_EXP == 1     /* timing experiment (results garbage): one gather per hit instead of three */
                    val[u][0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_filt, vo, 0, 0));
                    val[u][1] = val[u][0]; val[u][2] = val[u][0];
#elif defined(LFBM5D_AGG_EXP) && LFBM5D_AGG_EXP == 2   /* timing experiment: one 12-byte gather per hit (as if the channels were interleaved) */
                    { typedef float v3f __attribute__((ext_vector_type(3)));
                      const v3f t3 = __builtin_bit_cast(v3f, __builtin_amdgcn_raw_buffer_load_b96(rs_filt, (int)(ha.y * 4u + o * 12u), 0, 0));
                      val[u][0] = t3[0]; val[u][1] = t3[1]; val[u][2] = t3[2]; }
#else
                    val[u][0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_filt, vo, 0, 0));
                    val[u][1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_filt, vo, (int)cstride, 0));
                    val[u][2] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_filt, vo, (int)(2 * cstride), 0));
#endif
                }
                kw[u][0] = on ? kz * __uint_as_float(ha.z) : 0.0f;   /* ... and add it with weight zero */
                kw[u][1] = on ? kz * __uint_as_float(ha.w) : 0.0f;
                kw[u][2] = on ? kz * w2 : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < kAggU; u++)
#pragma unroll
                for (int c = 0; c < 3; c++) {
#pragma clang fp contract(off)
                    accn[c] += kw[u][c] * val[u][c];   /* core:516-520 */
                    accd[c] += kw[u][c];
                }
        }
        nh = 0;
        __builtin_amdgcn_wave_barrier();
    };

    __builtin_amdgcn_wave_barrier();
    if (VEC4) {
        /* N a multiple of four (round 4): a lane takes FOUR consecutive candidates -- matches n0 .. n0 + 3 of ONE reference patch -- with
         * one 16-byte load of their aggregation positions; the group index, the weights and the patch offset are per lane instead of per
         * candidate, the tile test is two unsigned range checks per candidate.  Hits are appended in candidate order (lane-major: an
         * exclusive prefix of the lanes' hit counts from three ballots) -- the list the consume phase walks is the same as before, entry
         * for entry.  A super-chunk of 256 candidates can hold more hits than the list has room for: lanes are then taken in
         * runs that fit, with a consume phase in between. */
        constexpr unsigned cap_hits = (unsigned)(kAggCap - kAggU);
        const unsigned lo_y = (unsigned)(ty0 - k + 1), lo_x = (unsigned)(tx0 - k + 1);          /* (wrap around for tiles at the border: the */
        const unsigned span_y = (unsigned)(TH + k - 1), span_x = (unsigned)(TW + k - 1);        /*  unsigned test below still means lo <= v < lo + span) */
        const unsigned pstep = (unsigned)(A * C * k2);                                          /* filt offset from match n to n + 1 */
        /* the candidates c0 + 4 lane .. + 3 of the lanes [l0, l1): test, and append the hits if the list has room (else: false, nothing
         * appended).  Nothing computed here is alive across a consume phase -- that is what keeps the kernel at its wave count. */
        auto try_append = [&](const int c0, const unsigned l0, const unsigned l1) -> bool {
            const int e = c0 + lane * 4;
            unsigned g = 0, n0 = 0;
            uint4 p4 = make_uint4(0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu);
            if (e < n_cand && (unsigned)lane >= l0 && (unsigned)lane < l1) {
                n0 = (unsigned)e & (unsigned)(N - 1);
                const unsigned rr = (unsigned)e >> logN;
                if (a.irregular) g = a.ref_begin + rr;
                else {
                    unsigned q = __umul24(rr, div_m) >> 20;
                    if (!mul_div) { asm volatile("" ::: "memory"); q = rr / (unsigned)ncols_span; }
                    g = __umul24((unsigned)r_lo + q, a.n_ref_cols) + (unsigned)c_lo + (rr - __umul24(q, (unsigned)ncols_span));
                }
                if (g >= a.ref_begin && g < g_end) p4 = *reinterpret_cast<const uint4*>(apos + ((size_t)g << logN) + n0);
            }
            const unsigned pj[4] = {p4.x, p4.y, p4.z, p4.w};
            unsigned m4 = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) {   /* an absent patch (0xffffffff) is at row / column 65535: outside every tile's range */
                const bool h = ((pj[j] >> 16) - lo_y) < span_y && ((pj[j] & 0xffffu) - lo_x) < span_x;
                m4 |= h ? (1u << j) : 0u;
            }
            const unsigned cnt = (unsigned)__popc(m4);
            /* exclusive prefix of cnt (0 .. 4) over the lanes below this one */
            const unsigned long long b0 = __ballot(cnt & 1u), b1 = __ballot(cnt & 2u), b2 = __ballot(cnt & 4u);
            auto below = [&](unsigned long long b) { return __builtin_amdgcn_mbcnt_hi((unsigned)(b >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)b, 0u)); };
            const unsigned total = (unsigned)__popcll(b0) + 2u * (unsigned)__popcll(b1) + 4u * (unsigned)__popcll(b2);
            if (nh + total > cap_hits) return false;
            if (m4) {
                unsigned slot = nh + below(b0) + 2u * below(b1) + 4u * below(b2);
                size_t wbase; unsigned off;
                const unsigned gl = g - a.ref_begin;   /* filt holds the launch's groups: [g - ref_begin][n][st][c][k2] */
                if (small24) { wbase = __umul24(g, (unsigned)C); off = __umul24(__umul24((gl << logN) + n0, (unsigned)A) + (unsigned)st, (unsigned)(C * k2)); }
                else { asm volatile("" ::: "memory"); wbase = (size_t)g * C; off = (((gl << logN) + n0) * A + st) * C * k2; }
                float w[3];
#pragma unroll
                for (int c = 0; c < 3; c++) w[c] = c < C ? a.wgt[wbase + (a.wchan0 ? 0 : c)] : 0.0f;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    if (m4 & (1u << j)) {
                        hit_a[slot] = make_uint4(pj[j], off, __float_as_uint(w[0]), __float_as_uint(w[1]));
                        hit_w2[slot] = w[2];
                        slot++;
                    }
                    off += pstep;
                }
            }
            nh += total;
            __builtin_amdgcn_wave_barrier();
            return true;
        };
        for (int c0 = 0; c0 < n_cand; c0 += 256) {
            if (!try_append(c0, 0u, 64u))                     /* more hits than the list has room for: sixteen lanes (<= 64 hits) at a time */
                for (unsigned l0 = 0; l0 < 64; l0 += 16) { consume(); (void)try_append(c0, l0, l0 + 16); }
            if (nh >= kAggFlush) consume();
        }
    } else
    for (int c0 = 0; c0 < n_cand; c0 += 64 * kAggPF) {
        unsigned g[kAggPF], p[kAggPF], nn[kAggPF];
#pragma unroll
        for (int u = 0; u < kAggPF; u++) {
            const int e = c0 + u * 64 + lane;
            p[u] = 0xffffffffu; g[u] = 0; nn[u] = 0;
            if (e < n_cand) {
                const unsigned n = (unsigned)e & (unsigned)(N - 1), rr = (unsigned)e >> logN;
                nn[u] = n;
                if (a.irregular) g[u] = a.ref_begin + rr;   /* the list is in raster order too (row lists, then columns) */
                else {
                    /* 24-bit multiplies (full rate): rr, q < 2^20 and div_m <= 2^20 under mul_div; grid rows / columns < 2^16 */
                    unsigned q = __umul24(rr, div_m) >> 20;
                    if (!mul_div) { asm volatile("" ::: "memory"); q = rr / (unsigned)ncols_span; }   /* kept a branch: the division's 32-bit multiplies are quarter rate */
                    g[u] = __umul24((unsigned)r_lo + q, a.n_ref_cols) + (unsigned)c_lo + (rr - __umul24(q, (unsigned)ncols_span));
                }
                if (g[u] >= a.ref_begin && g[u] < g_end) p[u] = apos[((size_t)g[u] << logN) + n];
            }
        }
        bool hit[kAggPF];
        float w[kAggPF][3];
        size_t wbase[kAggPF];
        if (small24) {
#pragma unroll
            for (int u = 0; u < kAggPF; u++) wbase[u] = __umul24(g[u], (unsigned)C);
        } else {
            asm volatile("" ::: "memory");
#pragma unroll
            for (int u = 0; u < kAggPF; u++) wbase[u] = (size_t)g[u] * C;
        }
#pragma unroll
        for (int u = 0; u < kAggPF; u++) {
            const int py = (int)(p[u] >> 16), px = (int)(p[u] & 0xffffu);
            hit[u] = p[u] != 0xffffffffu && py < ty0 + TH && py + k > ty0 && px < tx0 + TW && px + k > tx0;
#pragma unroll
            for (int c = 0; c < 3; c++) w[u][c] = (hit[u] && c < C) ? a.wgt[wbase[u] + (a.wchan0 ? 0 : c)] : 0.0f;
        }
        /* ordered append of the hits of these chunks */
#pragma unroll
        for (int u = 0; u < kAggPF; u++) {
            const unsigned long long bal = __ballot(hit[u]);
            if (hit[u]) {
                const unsigned slot = nh + __popcll(bal & ((1ull << lane) - 1ull));
                unsigned off;
                const unsigned gl = g[u] - a.ref_begin;
                if (small24) off = __umul24(__umul24((gl << logN) + nn[u], (unsigned)A) + (unsigned)st, (unsigned)(C * k2));
                else { asm volatile("" ::: "memory"); off = (((gl << logN) + nn[u]) * A + st) * C * k2; }
                hit_a[slot] = make_uint4(p[u], off, __float_as_uint(w[u][0]), __float_as_uint(w[u][1]));
                hit_w2[slot] = w[u][2];
            }
            nh += __popcll(bal);
        }
        __builtin_amdgcn_wave_barrier();
        if (nh >= kAggFlush) consume();
    }
    if (nh) consume();
    if (inside) for (int c = 0; c < C; c++) { a.num[pix + c * plane] = accn[c]; a.den[pix + c * plane] = accd[c]; }
}

} /* namespace */

/* ================================== launchers ============================================= */

static inline dim3 grid1d(size_t n, unsigned b = 256) { return dim3((unsigned)((n + b - 1) / b)); }

hipError_t launch_color(hipStream_t s, float* img, unsigned cs, unsigned n_px, int forward) {
    hipLaunchKernelGGL(k_color, grid1d(n_px), dim3(256), 0, s, img, (size_t)0, (const unsigned*)nullptr, cs, n_px, forward);
    return hipGetLastError();
}
hipError_t launch_color_lf(hipStream_t s, float* lf, size_t sai_stride, unsigned n_sai, const unsigned* d_mask, unsigned cs,
                           unsigned n_px, int forward) {
    hipLaunchKernelGGL(k_color, dim3(grid1d(n_px).x, n_sai), dim3(256), 0, s, lf, sai_stride, d_mask, cs, n_px, forward);
    return hipGetLastError();
}
hipError_t launch_color_roundtrip_lf(hipStream_t s, const float* in, float* out, size_t sai_stride, unsigned n_sai, const unsigned* d_mask,
                                     unsigned cs, unsigned n_px) {
    hipLaunchKernelGGL(k_color_roundtrip, dim3((n_px + 255) / 256, n_sai), dim3(256), 0, s, in, out, sai_stride, d_mask, cs, n_px);
    return hipGetLastError();
}
hipError_t launch_finalize_multi(hipStream_t s, const float* num, const float* den, const float* sub, float* basic, size_t sai_stride,
                                 const SaiList& L, unsigned cs, unsigned n_px, int colour) {
    if (!L.n) return hipSuccess;
    hipLaunchKernelGGL(k_finalize_multi, dim3((n_px + 255) / 256, L.n), dim3(256), 0, s, num, den, sub, basic, sai_stride, L, cs, n_px, colour);
    return hipGetLastError();
}
hipError_t launch_output_multi(hipStream_t s, const float* num, const float* den, const float* sub, float* out, float* basic,
                               const float* noisy_src, float* noisy_dst, size_t sai_stride, const SaiList& L, unsigned cs, unsigned n_px,
                               int colour) {
    if (!L.n) return hipSuccess;
    hipLaunchKernelGGL(k_output_multi, dim3((n_px + 255) / 256, L.n), dim3(256), 0, s, num, den, sub, out, basic, noisy_src, noisy_dst,
                       sai_stride, L, cs, n_px, colour);
    return hipGetLastError();
}
hipError_t launch_ipc_set(hipStream_t s, unsigned* p, unsigned v) {
    hipLaunchKernelGGL(k_ipc_set, dim3(1), dim3(1), 0, s, p, v);
    return hipGetLastError();
}
hipError_t launch_ipc_wait(hipStream_t s, const unsigned* p, unsigned want, unsigned* err, double timeout_s) {
    hipLaunchKernelGGL(k_ipc_wait, dim3(1), dim3(1), 0, s, p, want, err, (unsigned long long)(timeout_s * 1e8));
    return hipGetLastError();
}
hipError_t launch_estimate_lf(hipStream_t s, const float* num, const float* den, const float* sub, float* est, size_t seg,
                              unsigned n_sai, const unsigned* d_mask) {
    hipLaunchKernelGGL(k_estimate_lf, dim3(grid1d(seg).x, n_sai), dim3(256), 0, s, num, den, sub, est, seg, d_mask);
    return hipGetLastError();
}
hipError_t launch_copy_rect(hipStream_t s, float* dst, size_t dst_stride, unsigned dW, unsigned dH, unsigned dx0, unsigned dy0,
                            const float* src, size_t src_stride, unsigned sW, unsigned sH, unsigned sx0, unsigned sy0,
                            unsigned w, unsigned h, unsigned C, unsigned n_slots, const SaiMask& mask_bits) {
    hipLaunchKernelGGL(k_copy_rect, dim3(grid1d((size_t)w * h * C).x, n_slots), dim3(256), 0, s, dst, dst_stride, (int)dW, (int)dH, (int)dx0,
                       (int)dy0, src, src_stride, (int)sW, (int)sH, (int)sx0, (int)sy0, (int)w, (int)h, (int)C, mask_bits);
    return hipGetLastError();
}
hipError_t launch_symetrize(hipStream_t s, const float* src, float* dst, unsigned W, unsigned H, unsigned C, unsigned N) {
    hipLaunchKernelGGL(k_symetrize, grid1d((size_t)(W + 2 * N) * (H + 2 * N) * C), dim3(256), 0, s, src, dst, (int)W, (int)H, (int)C, (int)N);
    return hipGetLastError();
}
hipError_t launch_crop(hipStream_t s, float* dst, const float* src, unsigned W, unsigned H, unsigned C, unsigned N, unsigned off) {
    hipLaunchKernelGGL(k_unsymetrize, grid1d((size_t)W * H * C), dim3(256), 0, s, dst, src, (int)W, (int)H, (int)C, (int)N, (int)off);
    return hipGetLastError();
}
hipError_t launch_unsymetrize(hipStream_t s, float* dst, const float* src, unsigned W, unsigned H, unsigned C, unsigned N) {
    hipLaunchKernelGGL(k_unsymetrize, grid1d((size_t)W * H * C), dim3(256), 0, s, dst, src, (int)W, (int)H, (int)C, (int)N, (int)N);
    return hipGetLastError();
}
hipError_t launch_symetrize_multi(hipStream_t s, const float* src, size_t src_stride, float* dst, size_t dst_stride,
                                  const SaiList& L, unsigned W, unsigned H, unsigned C, unsigned N) {
    hipLaunchKernelGGL(k_symetrize_multi, dim3(grid1d((size_t)(W + 2 * N) * (H + 2 * N) * C).x, L.n), dim3(256), 0, s,
                       src, src_stride, dst, dst_stride, L, (int)W, (int)H, (int)C, (int)N);
    return hipGetLastError();
}
hipError_t launch_unsymetrize_multi(hipStream_t s, float* dst, size_t dst_stride, const float* src, size_t src_stride,
                                    const SaiList& L, unsigned W, unsigned H, unsigned C, unsigned N) {
    hipLaunchKernelGGL(k_unsymetrize_multi, dim3(grid1d((size_t)W * H * C).x, L.n), dim3(256), 0, s,
                       dst, dst_stride, src, src_stride, L, (int)W, (int)H, (int)C, (int)N);
    return hipGetLastError();
}
hipError_t launch_window_begin(hipStream_t s, const float* noisy, const float* basic, const float* num, const float* den, size_t lf_stride,
                               float* w_noisy, float* w_basic, float* w_num, float* w_den, float* est, size_t w_stride, const SaiList& L,
                               unsigned W, unsigned H, unsigned C, unsigned N, unsigned* zero) {
    WinBeginArgs a;
    a.noisy = noisy; a.basic = basic; a.num = num; a.den = den; a.w_noisy = w_noisy; a.w_basic = w_basic; a.w_num = w_num; a.w_den = w_den;
    a.est = est; a.lf_stride = lf_stride; a.w_stride = w_stride; a.L = L; a.W = (int)W; a.H = (int)H; a.C = (int)C; a.N = (int)N; a.zero = zero;
    hipLaunchKernelGGL(k_window_begin, dim3((W + 2 * N + 63) / 64, (H + 2 * N + 3) / 4, L.n), dim3(64, 4), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_window_end(hipStream_t s, float* num, float* den, size_t lf_stride, const float* w_num, const float* w_den, size_t w_stride,
                             const SaiList& L, unsigned W, unsigned H, unsigned C, unsigned N, unsigned k, unsigned* count) {
    WinEndArgs a;
    a.num = num; a.den = den; a.w_num = w_num; a.w_den = w_den; a.lf_stride = lf_stride; a.w_stride = w_stride; a.L = L;
    a.W = (int)W; a.H = (int)H; a.C = (int)C; a.N = (int)N; a.k = (int)k; a.count = count;
    hipLaunchKernelGGL(k_window_end, dim3((W + 63) / 64, (H + kWinEndRows - 1) / kWinEndRows, L.n), dim3(64, 4), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_estimate_multi(hipStream_t s, const float* num, const float* den, const float* sub, float* est,
                                 size_t plane, unsigned C, unsigned A, const SaiMask& mask_bits) {
    hipLaunchKernelGGL(k_estimate_multi, dim3(grid1d(plane).x, A), dim3(256), 0, s, num, den, sub, est, plane, C, mask_bits);
    return hipGetLastError();
}
hipError_t launch_estimate(hipStream_t s, const float* num, const float* den, const float* sub, float* est, size_t n) {
    hipLaunchKernelGGL(k_estimate, grid1d(n), dim3(256), 0, s, num, den, sub, est, n);
    return hipGetLastError();
}
hipError_t launch_fill_f32(hipStream_t s, float* p, float v, size_t n) {
    hipLaunchKernelGGL(k_fill_f32, grid1d(n), dim3(256), 0, s, p, v, n);
    return hipGetLastError();
}
hipError_t launch_add(hipStream_t s, float* dst, const float* src, size_t n) {
    if (n) hipLaunchKernelGGL(k_add, grid1d(n), dim3(256), 0, s, dst, src, n);
    return hipGetLastError();
}
hipError_t launch_fill_i32(hipStream_t s, int* p, int v, size_t n) {
    hipLaunchKernelGGL(k_fill_i32, grid1d(n), dim3(256), 0, s, p, v, n);
    return hipGetLastError();
}
hipError_t launch_count_zeros(hipStream_t s, const float* den, size_t seg, unsigned n_seg, unsigned* counts) {
    unsigned gx = (unsigned)((seg + 256 * 8 - 1) / (256 * 8));
    if (gx > 256) gx = 256;
    hipLaunchKernelGGL(k_count_zeros, dim3(gx, n_seg), dim3(256), 0, s, den, seg, counts);
    return hipGetLastError();
}
hipError_t launch_count_denoised(hipStream_t s, const float* den, size_t sai_stride, unsigned n_slots, const SaiMask& mask_bits,
                                 unsigned W, unsigned H, unsigned C, unsigned N, unsigned k, unsigned* count) {
    hipLaunchKernelGGL(k_count_denoised, dim3(128, n_slots), dim3(256), 0, s, den, sai_stride, mask_bits, (int)W, (int)H, (int)C, (int)N, (int)k, count);
    return hipGetLastError();
}

/* Kernels whose LDS stack can exceed the 64 KiB a launch gets by default: raise the limit once per device
 * (called from lfbm5d_create after hipSetDevice; the attribute belongs to the device's code object). */
constexpr int kGenericLdsLimit = 160 * 1024 - 8192;   /* dynamic LDS of k_group: its static part (positions of up to 32 x 49 patches) is 6.3 KB */
hipError_t prepare_group_kernels() {
    const int lim = 160 * 1024 - 4096;
    const void* fns[] = {
        reinterpret_cast<const void*>(&k_group<1>), reinterpret_cast<const void*>(&k_group<2>),
        reinterpret_cast<const void*>(&k_group_dct8<1>), reinterpret_cast<const void*>(&k_group_dct8<2>),
        reinterpret_cast<const void*>(&k_group_dct8w<true, false>), reinterpret_cast<const void*>(&k_group_dct8w<false, false>),
        reinterpret_cast<const void*>(&k_group_dct8w2<true>), reinterpret_cast<const void*>(&k_group_dct8w2<false>),
        reinterpret_cast<const void*>(&k_group_dct8w<true, true>), reinterpret_cast<const void*>(&k_group_dct8w<false, true>),
        reinterpret_cast<const void*>(&k_group_bior16_haar), reinterpret_cast<const void*>(&k_group_bior16_any),
        reinterpret_cast<const void*>(&k_group_dct16_haar), reinterpret_cast<const void*>(&k_group_dct16_any),
        reinterpret_cast<const void*>(&k_group_bior16_n1), reinterpret_cast<const void*>(&k_group_dct16_n1),
        reinterpret_cast<const void*>(&k_group_bior16_n1_sa), reinterpret_cast<const void*>(&k_group_dct16_n1_sa),
        reinterpret_cast<const void*>(&k_group_bm3d8<2, true>), reinterpret_cast<const void*>(&k_group_bm3d8<2, false>)};
    for (const void* f : fns) {
        const bool generic = f == reinterpret_cast<const void*>(&k_group<1>) || f == reinterpret_cast<const void*>(&k_group<2>);
        const hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, generic ? kGenericLdsLimit : lim);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

constexpr unsigned kBigBlocks = 1024;   /* persistent workgroups of k_group_big (four per CU) */
static size_t group_tmp_floats(const GroupArgs& a) {
    /* the 2-D stage's work area: one patch per wave-quarter for the generic path, [patch][k][k+1] for bior1.5 */
    return (a.tau2 == 7 && (a.k == 8 || a.k == 16)) ? (size_t)(kThreads / a.k) * a.k * (a.k + 1) : std::max<size_t>(256, (size_t)a.k * a.k);
}
size_t group_lds_bytes(const GroupArgs& a) {
    const size_t stack = (size_t)a.N * a.A * a.k * a.k;
    return ((a.step == 2 ? 2 : 1) * stack + group_tmp_floats(a)) * sizeof(float);
}
static bool group_uses_generic(const GroupArgs& a) {   /* mirrors launch_group's dispatch */
    if (getenv("LFBM5D_GROUP_GENERIC") != nullptr) return true;
    if (a.A != 9) return true;
    if (a.tau2 == 4 && a.N <= 8 && a.k * a.k <= 256 && a.step == 1) return false;
    if ((a.tau2 == 7 || a.tau2 == 5) && a.k == 16 && a.N <= 8 && a.step == 1) return false;
    if (a.tau2 == 7 && a.k == 8 && a.step == 2 && a.N <= (unsigned)kMaxN) return false;
    if (a.tau2 == 5 && a.k == 8 && a.N <= (unsigned)kMaxN) return false;
    if (a.bm3d && a.A == 1 && a.k == 8 && a.tau5 == 8 && (a.tau2 == 5 || a.tau2 == 7)) return false;
    return true;
}
size_t group_scratch_bytes(const GroupArgs& a) {
    if (!group_uses_generic(a) || (group_lds_bytes(a) <= (size_t)kGenericLdsLimit && a.A <= (unsigned)kMaxA)) return 0;
    return (size_t)kBigBlocks * (a.step == 2 ? 2 : 1) * a.N * a.A * a.k * a.k * sizeof(float);
}
hipError_t launch_group(hipStream_t s, const GroupArgs& a) {
    /* geometry pre-pass: patch positions, aggregation positions, angular shapes */
    hipLaunchKernelGGL(k_group_pos, grid1d((size_t)a.n_groups * a.N), dim3(256), 0, s, a);
    const bool bigA = a.A > (unsigned)kMaxA;
    if (bigA) hipLaunchKernelGGL(k_group_shape<true>, grid1d(a.n_groups), dim3(256), 0, s, a);
    else      hipLaunchKernelGGL(k_group_shape<false>, grid1d(a.n_groups), dim3(256), 0, s, a);
    /* a window with an empty SAI: tau_4D is the shape-adaptive transform (bm5d.cpp:276-280) and every group uses it */
    const bool all_sa = a.tau4 == 6 && !a.mask_bits.holds_all(a.A) && getenv("LFBM5D_NO_SA_KERNELS") == nullptr;
    /* LFBM5D_GROUP_GENERIC: test hook, every configuration through the generic LDS kernel (the dedicated kernels' cross-check) */
    const bool generic_only = getenv("LFBM5D_GROUP_GENERIC") != nullptr;
    /* no 2-D transform and a stack small enough for registers: register-resident kernel */
    if (generic_only) {}
    else if (a.tau2 == 4 && a.N <= 8 && a.k * a.k <= 256 && a.step == 1 && a.A == 9 && (size_t)a.A * a.C * a.Wb * a.Hb * 4 < 0x7fffffffull) {   /* 32-bit byte offsets into the window */
        const unsigned threads = ((a.k * a.k + 63) / 64) * 64;
        const unsigned gx = ((a.n_groups + 7) / 8) * 8;   /* xcd_group_index */
        if (all_sa) {
            if (a.tau5 == 9) hipLaunchKernelGGL(k_group_id_haar_sa, dim3(gx, a.C), dim3(threads), 0, s, a);
            else             hipLaunchKernelGGL(k_group_id_any_sa, dim3(gx, a.C), dim3(threads), 0, s, a);
        }
        else if (a.tau5 == 9) hipLaunchKernelGGL(k_group_id_haar, dim3(gx, a.C), dim3(threads), 0, s, a);
        else                  hipLaunchKernelGGL(k_group_id_any, dim3(gx, a.C), dim3(threads), 0, s, a);
        return hipGetLastError();
    }
    else if ((a.tau2 == 7 || a.tau2 == 5) && a.k == 16 && a.N <= 8 && a.step == 1 && a.A == 9) {   /* bior1.5 / DCT on 16x16 patches, HT step */
        const size_t lb = (size_t)a.N * 9 * kT16Patch * sizeof(float);
        const dim3 grid(a.n_groups, a.C), block(256);
        if (a.N == 1 && a.tau5 != 5) {   /* nine patches per group: a few groups share a workgroup (Haar / Hadamard of one patch: identity) */
            const dim3 grid8((a.n_groups + kT16Groups - 1) / kT16Groups, a.C);
            const size_t l1 = (size_t)kT16Groups * 9 * kT16Patch * sizeof(float);
            if (all_sa) {
                if (a.tau2 == 7) hipLaunchKernelGGL(k_group_bior16_n1_sa, grid8, block, l1, s, a);
                else             hipLaunchKernelGGL(k_group_dct16_n1_sa, grid8, block, l1, s, a);
            }
            else if (a.tau2 == 7) hipLaunchKernelGGL(k_group_bior16_n1, grid8, block, l1, s, a);
            else                  hipLaunchKernelGGL(k_group_dct16_n1, grid8, block, l1, s, a);
            return hipGetLastError();
        }
        if (a.tau2 == 7) {
#ifndef LFBM5D_T16_NOSPLIT   /* two rounds through a work area of 40 patches: groups of fewer than eight matches fit it whole (N <= 4: 36 patches) */
            if (a.tau5 == 9) hipLaunchKernelGGL(k_group_bior16_haar, grid, block, std::min(lb, (size_t)kT16Half * kT16Patch * sizeof(float)), s, a);
#else
            if (a.tau5 == 9) hipLaunchKernelGGL(k_group_bior16_haar, grid, block, lb, s, a);
#endif
            else             hipLaunchKernelGGL(k_group_bior16_any, grid, block, lb, s, a);
        } else {
#ifndef LFBM5D_T16_NOSPLIT
            if (a.tau5 == 9) hipLaunchKernelGGL(k_group_dct16_haar, grid, block, std::min(lb, (size_t)kT16Half * kT16Patch * sizeof(float)), s, a);
#else
            if (a.tau5 == 9) hipLaunchKernelGGL(k_group_dct16_haar, grid, block, lb, s, a);
#endif
            else             hipLaunchKernelGGL(k_group_dct16_any, grid, block, lb, s, a);
        }
        return hipGetLastError();
    }
    else if (a.tau2 == 7 && a.k == 8 && a.A == 9 && a.step == 2 && a.N <= (unsigned)kMaxN) {   /* 8x8 bior1.5, Wiener step: the DCT kernel with the wavelet in its 2-D stage */
        const size_t l8 = (size_t)2 * 64 * ((a.N * 9) | 1) * sizeof(float);
        if (a.tau5 == 9) hipLaunchKernelGGL((k_group_dct8w<true, true>), dim3(a.n_groups, a.C), dim3(kDct8wThreads), l8, s, a);
        else             hipLaunchKernelGGL((k_group_dct8w<false, true>), dim3(a.n_groups, a.C), dim3(kDct8wThreads), l8, s, a);
        return hipGetLastError();
    }
    else if (a.tau2 == 5 && a.k == 8 && a.A == 9 && a.N <= (unsigned)kMaxN) {   /* 8x8 DCT: one thread per patch for the 2-D stage */
        const size_t l8 = (size_t)(a.step == 2 ? 2 : 1) * 64 * ((a.N * 9) | 1) * sizeof(float);
        if (a.step == 2 && !getenv("LFBM5D_DCT8_SCALAR") && !getenv("LFBM5D_DCT8W_V1")) {   /* packed noisy/pilot pair, 2-D stages dealt to all threads */
            const unsigned gx = ((a.n_groups + 7) / 8) * 8;   /* xcd_group_index */
#ifndef LFBM5D_NO_DCT8W3
            /* LFBM5D_DCT8W_V2: test hook, round 2's two-image kernel (the path of windows of 1.9 GB and more) */
            if (a.tau5 == 9 && (size_t)9 * a.C * a.Wb * a.Hb * 4 < 0x70000000ull && !getenv("LFBM5D_DCT8W_V2")) { if (all_sa) hipLaunchKernelGGL(k_group_dct8w3<true>, dim3(gx, a.C), dim3(kDct8w3Threads), kW3Lds, s, a); else hipLaunchKernelGGL(k_group_dct8w3<false>, dim3(gx, a.C), dim3(kDct8w3Threads), kW3Lds, s, a); } else
#endif
            if (a.tau5 == 9) hipLaunchKernelGGL((k_group_dct8w2<true>), dim3(gx, a.C), dim3(kDct8w2Threads), l8, s, a);
            else             hipLaunchKernelGGL((k_group_dct8w2<false>), dim3(gx, a.C), dim3(kDct8w2Threads), l8, s, a);
            return hipGetLastError();
        }
        if (a.step == 2 && !getenv("LFBM5D_DCT8_SCALAR")) {   /* round 1's packed kernel (one thread per patch in the 2-D stages) */
            if (a.tau5 == 9) hipLaunchKernelGGL((k_group_dct8w<true, false>), dim3(a.n_groups, a.C), dim3(kDct8wThreads), l8, s, a);
            else             hipLaunchKernelGGL((k_group_dct8w<false, false>), dim3(a.n_groups, a.C), dim3(kDct8wThreads), l8, s, a);
            return hipGetLastError();
        }
        if (a.step == 2) hipLaunchKernelGGL(k_group_dct8<2>, dim3(a.n_groups, a.C), dim3(kDct8Threads), l8, s, a);
        else             hipLaunchKernelGGL(k_group_dct8<1>, dim3(a.n_groups, a.C), dim3(kDct8Threads), l8, s, a);
        return hipGetLastError();
    }
    else if (a.bm3d && a.A == 1 && a.k == 8 && a.tau5 == 8 && (a.tau2 == 5 || a.tau2 == 7)) {   /* per-SAI BM3D, 8x8 patches: a group per wavefront */
        const dim3 grid((a.n_groups + kBm3dWaves - 1) / kBm3dWaves, a.C), block(64 * kBm3dWaves);
        const size_t lb = (size_t)kBm3dWaves * 64 * (kMaxN3 + 1) * (a.step == 2 ? sizeof(v2f) : sizeof(float));
        if (a.step == 2) { if (a.tau2 == 7) hipLaunchKernelGGL((k_group_bm3d8<2, true>), grid, block, lb, s, a); else hipLaunchKernelGGL((k_group_bm3d8<2, false>), grid, block, lb, s, a); }
        else             { if (a.tau2 == 7) hipLaunchKernelGGL((k_group_bm3d8<1, true>), grid, block, lb, s, a); else hipLaunchKernelGGL((k_group_bm3d8<1, false>), grid, block, lb, s, a); }
        return hipGetLastError();
    }
    const size_t lds = group_lds_bytes(a);
    if (lds > (size_t)kGenericLdsLimit || bigA) {   /* stacks in HBM scratch slices, persistent workgroups */
        const unsigned long long slice = (unsigned long long)(a.step == 2 ? 2 : 1) * a.N * a.A * a.k * a.k;
        if (!a.scratch || a.scratch_floats < slice * kBigBlocks) return hipErrorInvalidValue;
        const unsigned blocks = std::min<unsigned>(kBigBlocks, a.n_groups * a.C);
        const unsigned tf = (unsigned)group_tmp_floats(a);
        const size_t ltmp = (size_t)tf * sizeof(float) + (bigA ? (size_t)a.N * a.A * sizeof(unsigned) : 0);
        if (bigA) {
            if (a.step == 2) hipLaunchKernelGGL((k_group_big<2, true>), dim3(blocks), dim3(kThreads), ltmp, s, a, a.scratch, slice, tf);
            else             hipLaunchKernelGGL((k_group_big<1, true>), dim3(blocks), dim3(kThreads), ltmp, s, a, a.scratch, slice, tf);
        }
        else if (a.step == 2) hipLaunchKernelGGL((k_group_big<2, false>), dim3(blocks), dim3(kThreads), ltmp, s, a, a.scratch, slice, tf);
        else                  hipLaunchKernelGGL((k_group_big<1, false>), dim3(blocks), dim3(kThreads), ltmp, s, a, a.scratch, slice, tf);
        return hipGetLastError();
    }
    if (a.step == 2) hipLaunchKernelGGL(k_group<2>, dim3(a.n_groups, a.C), dim3(kThreads), lds, s, a);
    else             hipLaunchKernelGGL(k_group<1>, dim3(a.n_groups, a.C), dim3(kThreads), lds, s, a);
    return hipGetLastError();
}
hipError_t launch_aggregate(hipStream_t s, const AggArgs& a) {
    const bool wide = a.k >= 12;
    const unsigned tw = wide ? 16 : 8, th = wide ? 4 : 8;
    const unsigned tiles = ((a.Wb + tw - 1) / tw) * ((a.Hb + th - 1) / th) * a.A;
    const dim3 grid(((tiles + 7) / 8) * 8), block(64);
    const bool big = a.filt_bytes > 0xfffff000ull || getenv("LFBM5D_AGG_64BIT") != nullptr;   /* env: exercise the 64-bit path in tests */
    /* four candidates per lane and 16-byte position loads when a reference patch's N matches come in fours (LFBM5D_AGG_SCALAR_SCAN: the
     * one-candidate-per-lane scan of rounds 1-3, for A/B runs; N = 1, 2 always take it) */
    /* Measured at the headline window (same box, rounds of tools/pass_time.py; instruction counts: tools/pmc_agg_ab.sh): VALU instructions
     * -14 % (k = 8) / -11 % (k = 16), scalar -51 % / -43 %, time 0.648 against 0.658 ms (k = 8), 0.973 against 0.964 (k = 16) -- the
     * kernel's time is its consume phase, not the scan -- so only the 8 x 8 tiles take it. */
    const bool vec4 = a.N % 4 == 0 && !wide && getenv("LFBM5D_AGG_SCALAR_SCAN") == nullptr;
#define LFBM5D_AGG(W_, TW_, TH_, PF_, U_) \
    do { if (big && vec4) hipLaunchKernelGGL((k_aggregate<W_, TW_, TH_, PF_, U_, true, true>), grid, block, 0, s, a); \
         else if (big)    hipLaunchKernelGGL((k_aggregate<W_, TW_, TH_, PF_, U_, true, false>), grid, block, 0, s, a); \
         else if (vec4)   hipLaunchKernelGGL((k_aggregate<W_, TW_, TH_, PF_, U_, false, true>), grid, block, 0, s, a); \
         else             hipLaunchKernelGGL((k_aggregate<W_, TW_, TH_, PF_, U_, false, false>), grid, block, 0, s, a); } while (0)
    if (a.k == 12)      LFBM5D_AGG(true, 16, 4, 3, 12);
    else if (a.k == 8)  LFBM5D_AGG(true, 8, 8, 2, 6);
    else if (wide)      LFBM5D_AGG(false, 16, 4, 3, 12);
    else                LFBM5D_AGG(false, 8, 8, 2, 6);
#undef LFBM5D_AGG
    return hipGetLastError();
}

} /* namespace lfbm5d */
