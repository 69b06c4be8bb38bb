// Calibration of rocprofv3's FETCH_SIZE on gfx950 for the aggregation's access shape (round 6; the round-5 review's question: does the
// counter halve 64-byte row gathers the way it halves 16-byte-per-lane streaming reads?).  Every kernel reads a buffer far larger than
// the Infinity Cache exactly once, so bytes-from-memory is known; run each under the counter and compare:
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/fetch_calib tools/fetch_calib.hip
//   rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -- /tmp/fetch_calib        (tools/fetch_calib.sh does this and prints the table)
// Kernels (names appear in the counter dump):
//   k_stream16      16 bytes per lane, consecutive lanes consecutive addresses (the shape the guide calibrated: reported at 1/2)
//   k_stream4       4 bytes per lane, 256 contiguous bytes per wave-instruction
//   k_rows64_t<0>   the aggregation's shape: a wave-instruction = 4 rows of 64 bytes of one 1-KB "patch" (lane = pixel of a 16 x 4 tile),
//                   patches visited in a scattered order, every row once
//   k_rows64_t<1>   the same, but only the first 32 bytes of every 64-byte row are read (tile overlapping half of a patch row)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float v4f __attribute__((ext_vector_type(4)));

__global__ void k_stream16(const v4f* __restrict__ p, size_t n16, float* out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    v4f acc = {0, 0, 0, 0};
    for (; i < n16; i += stride) acc += p[i];
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[0] = 1.0f;
}
__global__ void k_stream4(const float* __restrict__ p, size_t n4, float* out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    float acc = 0;
    for (; i < n4; i += stride) acc += p[i];
    if (acc == 12345.678f) out[0] = 1.0f;
}
// n_patches a power of two; wave w visits patches (w + it * n_waves) * 40503 mod n_patches (odd multiplier: a permutation)
template <bool HALF>
__global__ void k_rows64_t(const float* __restrict__ p, unsigned n_patches, float* out) {
    const unsigned lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = (gridDim.x * blockDim.x) >> 6;
    const unsigned px = lane & 15, row = lane >> 4;
    float acc = 0;
    for (unsigned q = wave; q < n_patches; q += n_waves) {
        const unsigned patch = (q * 40503u) & (n_patches - 1);
        const float* b = p + (size_t)patch * 256;
#pragma unroll
        for (int r4 = 0; r4 < 4; r4++) {
            const unsigned col = HALF ? (px & 7) : px;
            acc += b[(r4 * 4 + row) * 16 + col];
        }
    }
    if (acc == 12345.678f) out[0] = 1.0f;
}

int main(int argc, char** argv) {
    const size_t bytes = (size_t)2 << 30;   // 2 GiB: eight times the Infinity Cache
    float* buf; float* out;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(buf, 0, bytes);
    hipDeviceSynchronize();
    const unsigned n_patches = (unsigned)(bytes / 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto timeit = [&](const char* name, auto launch, double b) {
        hipEventRecord(e0, 0); launch(); hipEventRecord(e1, 0); hipDeviceSynchronize();
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        printf("%-14s read %.3f GB once in %.3f ms = %.2f TB/s\n", name, b / 1e9, ms, b / ms / 1e9);
    };
    timeit("k_stream16", [&] { hipLaunchKernelGGL(k_stream16, dim3(256 * 16), dim3(256), 0, 0, (const v4f*)buf, bytes / 16, out); }, (double)bytes);
    timeit("k_stream4", [&] { hipLaunchKernelGGL(k_stream4, dim3(256 * 16), dim3(256), 0, 0, buf, bytes / 4, out); }, (double)bytes);
    timeit("k_rows64", [&] { hipLaunchKernelGGL(k_rows64_t<false>, dim3(256 * 16), dim3(256), 0, 0, buf, n_patches, out); }, (double)bytes);
    timeit("k_rows64_half", [&] { hipLaunchKernelGGL(k_rows64_t<true>, dim3(256 * 16), dim3(256), 0, 0, buf, n_patches, out); }, (double)bytes / 2);
    printf("bytes of the buffer: %zu\n", bytes);
    return 0;
}
