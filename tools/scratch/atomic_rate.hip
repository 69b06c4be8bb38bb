// microbenchmark: cost of ~10 M scattered 32-bit atomicOr (no return) / byte stores into a 34 MB bitmap, the access pattern an
// aggregation pre-pass would have (instance -> up to 4 tiles, words a few KB apart).  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k_or(unsigned* bm, size_t words, unsigned n_inst, int scope_agent) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_inst) return;
    unsigned h = i * 2654435761u;
    const unsigned tile = (i / 37) % 44100u;          // neighbouring instances land in neighbouring tiles
#pragma unroll
    for (int t = 0; t < 4; t++) {
        const unsigned tl = (tile + (t & 1) + (t >> 1) * 70) % 44100u;
        const size_t w = (size_t)tl * 192 + ((h >> 7) % 146u);
        const unsigned bit = 1u << (h & 31);
        if (scope_agent) __hip_atomic_fetch_or(bm + w, bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else __hip_atomic_fetch_or(bm + w, bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        h = h * 1664525u + 1013904223u;
    }
}
__global__ void k_bytes(unsigned char* bm, size_t bytes, unsigned n_inst) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_inst) return;
    unsigned h = i * 2654435761u;
    const unsigned tile = (i / 37) % 44100u;
#pragma unroll
    for (int t = 0; t < 4; t++) {
        const unsigned tl = (tile + (t & 1) + (t >> 1) * 70) % 44100u;
        bm[(size_t)tl * 4672 + (h >> 7) % 4672u] = 1;
        h = h * 1664525u + 1013904223u;
    }
}
int main() {
    const size_t words = (size_t)44100 * 192, bytes = (size_t)44100 * 4672;
    unsigned* bm; unsigned char* bb;
    hipMalloc(&bm, words * 4); hipMalloc(&bb, bytes);
    const unsigned n_inst = 16129u * 16 * 9;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 4; mode++) {
        float best = 1e9f;
        for (int rep = 0; rep < 5; rep++) {
            hipMemset(bm, 0, words * 4); hipMemset(bb, 0, bytes); hipDeviceSynchronize();
            hipEventRecord(e0);
            if (mode == 0) k_or<<<(n_inst + 255) / 256, 256>>>(bm, words, n_inst, 1);
            else if (mode == 1) k_or<<<(n_inst + 255) / 256, 256>>>(bm, words, n_inst, 0);
            else if (mode == 2) k_bytes<<<(n_inst + 255) / 256, 256>>>(bb, bytes, n_inst);
            else hipMemsetAsync(bm, 0, words * 4);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
        }
        const char* nm[] = {"atomicOr agent scope", "atomicOr workgroup scope (L2-local, not a valid scope for this use)", "byte stores", "memset 34 MB"};
        printf("%s: %.3f ms for %u instances x 4\n", nm[mode], best, n_inst);
    }
    return 0;
}
