// Microbenchmark (round 6): what a SIMD of gfx950 sustains per cycle for the instruction kinds the group kernels are made of --
// plain and PACKED fp32 arithmetic, selects, DPP moves -- at 1, 2, 3, 4 and 8 waves per SIMD.  The question it settles: is a
// v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 wave-instruction a 2-cycle or a 4-cycle instruction once a SIMD has more than one wave,
// i.e. does `VALU wave-instructions x 2 cycles` (profiles/valu_latest.json, roofline.valu_frac) under-count packed kernels?
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/valu_rate tools/valu_rate.hip && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define N_ITER 4096
typedef float v2f __attribute__((ext_vector_type(2)));

// 16 independent accumulators per lane, 16 instructions per iteration, no dependence closer than 16 instructions
template <int MODE>
__global__ void k_rate(float* out, long long* cyc) {
    const int lane = threadIdx.x & 63;
    v2f a[16];
#pragma unroll
    for (int i = 0; i < 16; i++) a[i] = v2f{out[threadIdx.x] + i, 1.0f + lane + i};
    const v2f m = v2f{1.0000001f, 0.9999999f}, c = v2f{1e-9f, -1e-9f};
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < N_ITER; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) {
            if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i].x) : "v"(m.x), "v"(c.x));
            else if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
            else if (MODE == 2) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            else if (MODE == 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            else if (MODE == 4) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i].x) : "v"(c.x));
            else if (MODE == 5) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i].x) : "v"(m.x));
            else if (MODE == 6) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i].x) : "v"(m.x) : );
            else if (MODE == 7) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i].x));
            else if (MODE == 8) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i].x));
            else if (MODE == 9) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a[i].x) : "v"(m.x), "v"(c.x));
            else if (MODE == 10) asm volatile("v_cmp_gt_f32 vcc, %0, %1" : : "v"(a[i].x), "v"(m.x) : "vcc");
            else if (MODE == 11) asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel_hi:[1,0,1]" : "+v"(a[i]) : "v"(m), "v"(c));   // scalar-broadcast operand form
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; i++) s += a[i].x + a[i].y;
    out[threadIdx.x + blockIdx.x * blockDim.x] = s;
    if (lane == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

static double g_clk_ratio = 1.0;   // shader cycles per s_memtime tick

template <int MODE> void run(const char* name, int threads, int blocks, int waves_per_simd) {
    float* d; long long* c;
    hipMalloc(&d, 1 << 22); hipMalloc(&c, 1 << 20);
    hipMemset(d, 0, 1 << 22);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_rate<MODE>, dim3(blocks), dim3(threads), 0, 0, d, c);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k_rate<MODE>, dim3(blocks), dim3(threads), 0, 0, d, c);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(blocks * threads / 64);
    hipMemcpy(h.data(), c, h.size() * 8, hipMemcpyDeviceToHost);
    double sum = 0; for (long long v : h) sum += (double)v;
    const double ticks_per_instr_wave = sum / h.size() / N_ITER / 16;   // per wave
    // wall-clock: wave-instructions per SIMD / time
    const double instr_per_simd = (double)N_ITER * 16 * waves_per_simd;
    const double ns_per_instr_simd = ms * 1e6 / instr_per_simd;
    printf("  %-34s %d wave(s)/SIMD: %6.2f counter ticks per wave-instruction of a wave; %6.3f ns per wave-instruction per SIMD (event time %.3f ms)\n",
           name, waves_per_simd, ticks_per_instr_wave, ns_per_instr_simd, ms);
    hipFree(d); hipFree(c); hipEventDestroy(e0); hipEventDestroy(e1);
}

int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("device %s, %d CUs, clock %d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
    const int blocks = p.multiProcessorCount;   // one workgroup per CU, threads = 256 x waves per SIMD
    for (int w : {1, 2, 3, 4, 8}) {
        const int threads = 256 * w > 1024 ? 1024 : 256 * w;
        const int blk = 256 * w > 1024 ? blocks * (256 * w / 1024) : blocks;
        printf("--- %d wave(s) per SIMD (%d threads x %d blocks)\n", w, threads, blk);
        if (256 * w > 1024) { printf("  (two workgroups of 1024 threads per CU: waves per SIMD as the heading says if both are resident)\n"); }
        run<0>("v_fma_f32", threads, blk, w);
        run<1>("v_pk_fma_f32", threads, blk, w);
        run<11>("v_pk_fma_f32 op_sel_hi:[1,0,1]", threads, blk, w);
        run<2>("v_pk_mul_f32", threads, blk, w);
        run<3>("v_pk_add_f32", threads, blk, w);
        run<4>("v_add_f32", threads, blk, w);
        run<5>("v_mul_f32", threads, blk, w);
        run<6>("v_cndmask_b32", threads, blk, w);
        run<10>("v_cmp_gt_f32", threads, blk, w);
        run<7>("v_mov_b32_dpp row_shr:1", threads, blk, w);
        run<8>("v_rcp_f32", threads, blk, w);
        run<9>("v_mad_u32_u24", threads, blk, w);
    }
    return 0;
}
