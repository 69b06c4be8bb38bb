// How fast can a CU gather 64-byte patch-row pieces at 4-byte alignment?  (the wide-window group kernel's input: 200 patches x 4 rows
// x 16 floats per workgroup, 3 workgroups per CU through 51 KB of LDS each).  Variants: 0 = one dword per lane (lanes along the row),
// 1 = 16-byte loads at 4-byte alignment (4 lanes per row piece), 2 = ALIGNED 16-byte loads (5 lanes per row piece, the piece's
// aligned hull), 3 = 8-byte loads at 4-byte alignment.   build: hipcc -O3 --offload-arch=gfx950 -o bin/gather_probe gather_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
struct __attribute__((packed, aligned(4))) f4u { float v[4]; };
struct __attribute__((packed, aligned(4))) f2u { float v[2]; };
constexpr int Wb = 340, Hb = 340, NSAI = 25, NM = 8, NP = NSAI * NM;   // 200 patches per workgroup
__device__ __forceinline__ unsigned hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
template <int V>
__global__ __launch_bounds__(256) void k_gather(const float* img, float* out, int groups_per_row, int slab, int passes, unsigned long long* clk) {
    extern __shared__ float S[];
    __shared__ unsigned pos[NP];
    const int tid = threadIdx.x, g = blockIdx.x;
    const int gy = g / groups_per_row, gx = g % groups_per_row;
    for (int i = tid; i < NP; i += 256) {
        const unsigned h = hash(g * 977u + i);
        const int y = 4 * gy + 12 + (int)(h % 25) - 12 + 4 * slab, x = 4 * gx + 12 + (int)((h >> 8) % 25) - 12;
        pos[i] = (unsigned)(i % NSAI) * (3u * Wb * Hb) + (unsigned)y * Wb + (unsigned)x;
    }
    __syncthreads();
    float acc = 0.0f;
    const long long t0 = (long long)__builtin_readcyclecounter();
    if (V == 0) {
        float v[50];
#pragma unroll
        for (int u = 0; u < 50; u++) { const int e = tid + u * 256, px = e % 64, ns = e / 64; v[u] = img[pos[ns] + (px / 16) * Wb + (px % 16)]; }
#pragma unroll
        for (int u = 0; u < 50; u++) S[tid + u * 256] = v[u];
    } else if (V == 1) {
        f4u v[13];
#pragma unroll
        for (int u = 0; u < 13; u++) { const int e = tid + u * 256; v[u] = f4u{{0, 0, 0, 0}}; if (e < 3200) { const int q = e % 16, ns = e / 16; v[u] = *reinterpret_cast<const f4u*>(img + pos[ns] + (q / 4) * Wb + 4 * (q % 4)); } }
#pragma unroll
        for (int u = 0; u < 13; u++) { const int e = tid + u * 256; if (e < 3200) *reinterpret_cast<float4*>(S + 4 * e) = make_float4(v[u].v[0], v[u].v[1], v[u].v[2], v[u].v[3]); }
    } else if (V == 2) {
        float4 v[16];   // 200 patches x 4 rows x 5 aligned quads = 4000
#pragma unroll
        for (int u = 0; u < 16; u++) { const int e = tid + u * 256; v[u] = make_float4(0, 0, 0, 0); if (e < 4000) { const int q = e % 5, rs = e / 5, row = rs % 4, ns = rs / 4; const unsigned a0 = pos[ns] + row * Wb; v[u] = *reinterpret_cast<const float4*>(img + (a0 & ~3u) + 4 * q); } }
#pragma unroll
        for (int u = 0; u < 16; u++) {
            const int e = tid + u * 256;
            if (e < 4000) {
                const int q = e % 5, rs = e / 5, row = rs % 4, ns = rs / 4; const int o = (int)((pos[ns] + row * Wb) & 3u);
                const float w[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
                for (int i = 0; i < 4; i++) { const int px = 4 * q + i - o; if (px >= 0 && px < 16) S[ns * 64 + row * 16 + px] = w[i]; }
            }
        }
    } else {
        f2u v[25];
#pragma unroll
        for (int u = 0; u < 25; u++) { const int e = tid + u * 256; const int q = e % 32, ns = e / 32; v[u] = *reinterpret_cast<const f2u*>(img + pos[ns] + (q / 8) * Wb + 2 * (q % 8)); }
#pragma unroll
        for (int u = 0; u < 25; u++) { const int e = tid + u * 256; *reinterpret_cast<float2*>(S + 2 * e) = make_float2(v[u].v[0], v[u].v[1]); }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t1 = (long long)__builtin_readcyclecounter();
    __syncthreads();
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f* S2 = reinterpret_cast<v2f*>(S);
    for (int ps = 0; ps < passes; ps++) {   /* stand-in for the transform passes: 5 values of pairs in, 25 packed FMAs, 5 out */
        for (int e = tid; e < 8 * 5 * 32; e += 256) {
            v2f* row = S2 + (e / 32) * 5 * 32 + e % 32;
            v2f x[5], t[5];
#pragma unroll
            for (int j = 0; j < 5; j++) x[j] = row[j * 32];
#pragma unroll
            for (int u = 0; u < 5; u++) { v2f a2 = {0.0f, 0.0f};
#pragma unroll
                for (int j = 0; j < 5; j++) a2 += x[j] * (0.1f * (float)(u * 5 + j + 1));
                t[u] = a2; }
#pragma unroll
            for (int u = 0; u < 5; u++) row[u * 32] = t[u];
        }
        __syncthreads();
    }
    if (tid == 0 && blockIdx.x % 64 == 5) { atomicAdd(&clk[V * 4], (unsigned long long)(t1 - t0)); atomicAdd(&clk[V * 4 + 1], (unsigned long long)((long long)__builtin_readcyclecounter() - t0)); atomicAdd(&clk[V * 4 + 2], 1ull); }
    for (int i = tid; i < 12800; i += 256) acc += S[i];
    if (acc == 123.456f) out[g] = acc;
}
int main(int argc, char** argv) {
    const int passes = argc > 1 ? atoi(argv[1]) : 0;
    unsigned long long* clk; hipMalloc(&clk, 256); hipMemset(clk, 0, 256);
    const size_t n = (size_t)NSAI * 3 * Wb * Hb + 4096;
    float *img, *out; hipMalloc(&img, n * 4); hipMalloc(&out, 1 << 20);
    std::vector<float> h(n); for (size_t i = 0; i < n; i++) h[i] = (float)(i % 251);
    hipMemcpy(img, h.data(), n * 4, hipMemcpyHostToDevice);
    const int gpr = 61, groups = 61 * 61;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int v = 0; v < 4; v++) {
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(e0);
            for (int c = 0; c < 3; c++) for (int slab = 0; slab < 4; slab++) {
                const float* p = img + (size_t)c * Wb * Hb;
                if (v == 0) hipLaunchKernelGGL(k_gather<0>, dim3(groups), dim3(256), 51200, 0, p, out, gpr, slab, passes, clk);
                if (v == 1) hipLaunchKernelGGL(k_gather<1>, dim3(groups), dim3(256), 51200, 0, p, out, gpr, slab, passes, clk);
                if (v == 2) hipLaunchKernelGGL(k_gather<2>, dim3(groups), dim3(256), 51200, 0, p, out, gpr, slab, passes, clk);
                if (v == 3) hipLaunchKernelGGL(k_gather<3>, dim3(groups), dim3(256), 51200, 0, p, out, gpr, slab, passes, clk);
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep == 2) printf("variant %d: %.3f ms for 12 launches x %d workgroups x 51 KB (%.2f TB/s)\n", v, ms, groups, 12.0 * groups * 51200 / ms / 1e9);
        }
    }
    unsigned long long h2[32]; hipMemcpy(h2, clk, 256, hipMemcpyDeviceToHost);
    for (int v = 0; v < 4; v++) printf("variant %d: gather %.0f cycles, workgroup %.0f cycles (averages, %d passes)\n", v, (double)h2[v * 4] / h2[v * 4 + 2], (double)h2[v * 4 + 1] / h2[v * 4 + 2], passes);
    return 0;
}
