// Microbenchmark (round 6; the round-5 review's item 1): the 8-point DCT of the Wiener group kernel as it is -- IN-THREAD butterfly, a lane
// holds a row of two patches as packed pairs (dct8_fwd_t, lfbm5d_group_wiener.hip) -- against the CROSS-LANE form the review proposed
// ("row pass in registers, exchange by DPP within 8-lane groups"): lane = pixel of a patch, the eight values of a row sit in eight
// neighbouring lanes, stage 1 by row_half_mirror, then every lane's output as four DPP quad-broadcast multiply-adds with per-lane
// cosines.  Both forms transform 1 024 values per wave and call (64 lanes x 8 x 2 / 16 registers x 64 lanes); the loop feeds each
// result into the next call.  Prints shader cycles per call and wave at 1 / 2 / 4 waves per SIMD.
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/dct8_forms tools/dct8_forms.hip && /tmp/dct8_forms
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

typedef float v2f __attribute__((ext_vector_type(2)));
#define N_ITER 512

template <class T> __device__ __forceinline__ void dct8_fwd_t(T* x) {   // the product kernel's butterfly
    const float a0 = 0.35355339059327376f;
    const float c1 = 0.5f * 0.98078528040323044f, c2 = 0.5f * 0.92387953251128674f, c3 = 0.5f * 0.83146961230254524f,
                c4 = 0.5f * 0.70710678118654752f, c5 = 0.5f * 0.55557023301960222f, c6 = 0.5f * 0.38268343236508977f,
                c7 = 0.5f * 0.19509032201612827f;
    const T s0 = x[0] + x[7], s1 = x[1] + x[6], s2 = x[2] + x[5], s3 = x[3] + x[4];
    const T d0 = x[0] - x[7], d1 = x[1] - x[6], d2 = x[2] - x[5], d3 = x[3] - x[4];
    const T p0 = s0 + s3, p1 = s1 + s2, m0 = s0 - s3, m1 = s1 - s2;
    x[0] = a0 * (p0 + p1); x[4] = c4 * (p0 - p1);
    x[2] = c2 * m0 + c6 * m1; x[6] = c6 * m0 - c2 * m1;
    x[1] = c1 * d0 + c3 * d1 + c5 * d2 + c7 * d3; x[3] = c3 * d0 - c7 * d1 - c1 * d2 - c5 * d3;
    x[5] = c5 * d0 - c1 * d1 + c7 * d2 + c3 * d3; x[7] = c7 * d0 - c5 * d1 + c3 * d2 - c1 * d3;
}

__global__ void k_in_thread(float* out, long long* cyc) {
    v2f x[8];
#pragma unroll
    for (int i = 0; i < 8; i++) x[i] = v2f{out[threadIdx.x] + i, 1.0f + (threadIdx.x & 63) + i};
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < N_ITER; it++) { dct8_fwd_t(x); asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7])); }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; i++) s += x[i].x + x[i].y;
    out[threadIdx.x + blockIdx.x * blockDim.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

// cross-lane: one register = 64 pixels of a patch, the row's eight values in lanes 8r .. 8r + 7.
//   stage 1: t = x + sign * mirror(x) within the eight lanes (lanes 0-3: sums s0..s3, lanes 4-7: differences d3..d0)
//   stage 2: out(lane) = sum over the four values of the lane's quad of w[q](lane) * t(quad lane q)   (the 4 x 4 even / odd matrices)
__device__ __forceinline__ float dct8_cross(float x, float sgn, const float (&w)[4]) {
    const float m = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x141 /* row_half_mirror */, 0xf, 0xf, false));
    const float t = __builtin_fmaf(sgn, x, m);
    float r = w[0] * __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0x00 /* quad_perm [0,0,0,0] */, 0xf, 0xf, false));
    r = __builtin_fmaf(w[1], __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0x55 /* [1,1,1,1] */, 0xf, 0xf, false)), r);
    r = __builtin_fmaf(w[2], __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0xaa /* [2,2,2,2] */, 0xf, 0xf, false)), r);
    r = __builtin_fmaf(w[3], __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0xff /* [3,3,3,3] */, 0xf, 0xf, false)), r);
    return r;
}
__global__ void k_cross_lane(float* out, long long* cyc) {
    float v[16];
    const int lane = threadIdx.x & 63, j = lane & 7;
#pragma unroll
    for (int i = 0; i < 16; i++) v[i] = out[threadIdx.x] + i + lane;
    const float sgn = j < 4 ? 1.0f : -1.0f;
    float w[4];
#pragma unroll
    for (int q = 0; q < 4; q++) w[q] = 0.25f + 0.01f * (float)((j * 4 + q) % 7);   // stand-ins for the per-lane cosines: the values do not change the timing
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < N_ITER; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) v[i] = dct8_cross(v[i], sgn, w);
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; i++) s += v[i];
    out[threadIdx.x + blockIdx.x * blockDim.x] = s;
    if (lane == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <class K> void run(const char* name, K kern, int waves_per_simd) {
    const int threads = 64 * 4 * std::min(waves_per_simd, 4), blocks = 256 * std::max(1, waves_per_simd / 4);
    float* d; long long* c;
    (void)hipMalloc(&d, 1 << 24); (void)hipMalloc(&c, 1 << 20);
    (void)hipMemset(d, 0, 1 << 24);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d, c);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d, c);
    (void)hipDeviceSynchronize();
    const int nw = blocks * threads / 64;
    std::vector<long long> h(nw);
    (void)hipMemcpy(h.data(), c, nw * sizeof(long long), hipMemcpyDeviceToHost);
    double sum = 0; for (long long x : h) sum += (double)x;
    // a SIMD runs `waves_per_simd` such waves at once: cycles of the SIMD per call of ONE wave = wave's cycles / waves per SIMD
    printf("%-44s %d waves/SIMD: %7.1f cycles per call and wave, %6.1f SIMD cycles per 1 024 values\n", name, waves_per_simd,
           sum / nw / N_ITER, sum / nw / N_ITER / waves_per_simd);
    (void)hipFree(d); (void)hipFree(c);
}

int main() {
    for (int w : {1, 2, 4}) {
        run("in-thread butterfly, packed pairs (product)", k_in_thread, w);
        run("cross-lane: half-mirror + 4 quad broadcasts", k_cross_lane, w);
    }
    return 0;
}
