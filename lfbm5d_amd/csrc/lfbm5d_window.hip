/*
 * lfbm5d_window.hip -- the small kernels of the window schedule (bm5d.cpp:165-407 / :861-1106) for gfx950: colour transforms,
 * mirror padding and cropping, estimates, coverage counts, the two ends of a window of the graph form (k_window_begin /
 * k_window_end), the per-SAI finalisation and output kernels of the two-step job and of the streamed host seam, and the gating
 * words of the two-processes-on-one-GPU transport.  Elementwise or block reductions, all SAIs of a window (or of the light
 * field) per launch.  Split from lfbm5d_kernels.hip in round 5.
 */
#include "lfbm5d_kernels.h"

#include <algorithm>

namespace lfbm5d {

namespace {

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
/* rows [src_row0, src_row0 + n_rows) of every plane of `src` (planes of src_H rows) to rows dst_row0 ... of `dst` (planes of dst_H rows):
 * the crop of a light field to a horizontal band and back (spatial bands, lfbm5d_steps.hip) */
__global__ void k_copy_rows(const float* __restrict__ src, unsigned src_H, unsigned src_row0, float* __restrict__ dst, unsigned dst_H,
                            unsigned dst_row0, unsigned n_rows, unsigned W) {
    const size_t n = (size_t)n_rows * W;
    const float* sp = src + ((size_t)blockIdx.y * src_H + src_row0) * W;
    float* dp = dst + ((size_t)blockIdx.y * dst_H + dst_row0) * W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dp[i] = sp[i];
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
hipError_t launch_copy_rows(hipStream_t s, const float* src, unsigned src_H, unsigned src_row0, float* dst, unsigned dst_H, unsigned dst_row0,
                            unsigned n_rows, unsigned W, unsigned planes) {
    if (!n_rows || !planes) return hipSuccess;
    const size_t n = (size_t)n_rows * W;
    const unsigned gx = (unsigned)std::min<size_t>(256, (n + 1023) / 1024);
    hipLaunchKernelGGL(k_copy_rows, dim3(gx, planes), dim3(256), 0, s, src, src_H, src_row0, dst, dst_H, dst_row0, n_rows, W);
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

} /* namespace lfbm5d */
