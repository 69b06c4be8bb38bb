// Microbenchmark (round 5): what a dependent VALU chain costs on gfx950 with and without the DPP wave shift the table kernel's
// recurrence uses, LDS read latencies, and the cost of a workgroup barrier -- one wave alone on its SIMD and several per SIMD.
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/chain_latency tools/scratch/chain_latency.hip && /tmp/chain_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define N_ITER 2048

template <int MODE>
__global__ void k_chain(float* out, long long* cyc, int stride) {
    __shared__ float lds[16384];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = (float)i;
    __syncthreads();
    float s = out[threadIdx.x], e1 = 1.0f + lane, e2 = 0.5f, f1 = 0.25f, f2 = 0.125f, lp = 0.0f;
    const float* p = lds + lane * 4;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < N_ITER; it++) {
        if (MODE == 0) {           // 7 dependent adds
            s = s + e1; s = s - lp; s = s + e1; s = s - e2; s = s - f1; s = s + f2; s = s + e2;
        } else if (MODE == 1) {    // the recurrence: wave_shr:1 DPP + 6 adds
            const float left = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(e2), __float_as_int(s), 0x138, 0xf, 0xf, false));
            float S = left + s; S = S - lp; S = S + e1; S = S - e2; S = S - f1; S = S + f2; lp = left; s = S;
        } else if (MODE == 2) {    // row_shr:1 DPP + 6 adds
            const float left = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(e2), __float_as_int(s), 0x111, 0xf, 0xf, false));
            float S = left + s; S = S - lp; S = S + e1; S = S - e2; S = S - f1; S = S + f2; lp = left; s = S;
        } else if (MODE == 3) {    // dependent ds_read_b128 chain (address depends on the value read)
            const float4 v = *reinterpret_cast<const float4*>(p);
            p = lds + ((((int)v.x) & 1023) * 4);
            s += v.y;
        } else if (MODE == 4) {    // barrier only
            __syncthreads();
        } else if (MODE == 5) {    // ds_bpermute dependent chain
            s = __int_as_float(__builtin_amdgcn_ds_bpermute(((lane + 1) & 63) << 2, __float_as_int(s))) + 1.0f;
        } else if (MODE == 6) {    // recurrence with two independent chains (two tables per wave)
            const float left = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(e2), __float_as_int(s), 0x138, 0xf, 0xf, false));
            const float left2 = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(e2), __float_as_int(e1), 0x138, 0xf, 0xf, false));
            float S = left + s; float T = left2 + e1;
            S = S - lp; T = T - f1; S = S + f2; T = T - e2; S = S - e2; T = T + f2; S = S - f1; T = T - lp; S = S + f2; T = T + e2;
            lp = left; s = S; e1 = T;
        } else if (MODE == 7) {    // v_readlane dependent chain (the first-row prologue)
            s = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(s + e1), it & 63));
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x + blockIdx.x * blockDim.x] = s + lp + e1 + p[0];
    if (lane == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
    (void)stride;
}

template <int MODE> void run(const char* name, int threads, int blocks) {
    float* d; long long* c;
    hipMalloc(&d, 1 << 22); hipMalloc(&c, 1 << 20);
    hipMemset(d, 0, 1 << 22);
    hipLaunchKernelGGL(k_chain<MODE>, dim3(blocks), dim3(threads), 0, 0, d, c, 1);
    hipLaunchKernelGGL(k_chain<MODE>, dim3(blocks), dim3(threads), 0, 0, d, c, 1);
    hipDeviceSynchronize();
    std::vector<long long> h(blocks * threads / 64);
    hipMemcpy(h.data(), c, h.size() * 8, hipMemcpyDeviceToHost);
    double sum = 0; for (long long v : h) sum += (double)v;
    printf("%-44s %4d threads x %4d blocks: %8.1f shader-clock cycles per iteration\n", name, threads, blocks, sum / h.size() / N_ITER);
    hipFree(d); hipFree(c);
}

int main() {
    // __builtin_readcyclecounter = s_memtime: constant 100 MHz?  print the conversion with s_memrealtime
    for (int cfg = 0; cfg < 3; cfg++) {
        const int threads = cfg == 0 ? 64 : (cfg == 1 ? 256 : 768), blocks = cfg == 0 ? 1 : 256;
        printf("--- %d wave(s) per SIMD, %d block(s)\n", cfg == 0 ? 1 : (cfg == 1 ? 1 : 3), blocks);
        run<0>("7 dependent v_add/v_sub", threads, blocks);
        run<1>("wave_shr:1 DPP + 6 dependent adds (one step)", threads, blocks);
        run<2>("row_shr:1 DPP + 6 dependent adds", threads, blocks);
        run<6>("two interleaved chains (wave_shr + 5 ops each)", threads, blocks);
        run<3>("dependent ds_read_b128", threads, blocks);
        run<5>("dependent ds_bpermute + add", threads, blocks);
        run<7>("add + v_readlane dependent", threads, blocks);
        run<4>("s_barrier (whole workgroup)", threads, blocks);
    }
    return 0;
}
