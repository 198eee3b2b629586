// micro-benchmark: LDS atomic throughput on gfx950 as a function of address conflicts (design input for the tiled
// backward kernels): ds_add_f32 / ds_add_u32 / ds_cmpst_rtn with G lanes per address, and global float atomics for scale.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>   // 0 = ds_add_f32, 1 = ds_add_u32, 2 = ds_cmpst_rtn (CAS), 3 = global atomicAdd f32, 4 = ds_add_u64, 5 = ds_add_rtn_u32
__global__ __launch_bounds__(256) void k(float* gout, int group, int iters) {
    __shared__ float sf[4096];
    __shared__ unsigned su[4096];
    __shared__ unsigned long long sl[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) { sf[i] = 0.f; su[i] = 0u; sl[i] = 0ull; }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // lanes lane/group share an address; different groups hit different banks
    int idx = (wave * 64 + (lane / group)) & 4095;
    float acc = 0.f;
    for (int it = 0; it < iters; it++) {
        const int a = (idx + it * 67) & 4095;
        if (MODE == 0) atomicAdd(&sf[a], 1.0f + lane);
        else if (MODE == 1) atomicAdd(&su[a], 1u + lane);
        else if (MODE == 4) atomicAdd(&sl[a], (unsigned long long)(1u + lane) << 20);
        else if (MODE == 5) acc += (float)atomicAdd(&su[a], 1u + lane);
        else if (MODE == 2) acc += (float)atomicCAS(&su[a], 0xffffffffu, (unsigned)lane);
        else atomicAdd(&gout[(size_t)blockIdx.x * 4096 + a], 1.0f + lane);
    }
    __syncthreads();
    if (threadIdx.x == 0) gout[blockIdx.x] += sf[1] + (float)su[2] + acc + (float)sl[3];
}
int main() {
    float* out; hipMalloc(&out, sizeof(float) * 4096 * 2048); hipMemset(out, 0, sizeof(float) * 4096 * 2048);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int wg = 2048, iters = 256;   // 8 workgroups (32 waves) per CU
    for (int mode : {4, 5}) for (int group : {1, 2, 4, 8, 16, 64}) {
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            if (mode == 0) k<0><<<wg, 256>>>(out, group, iters); else if (mode == 1) k<1><<<wg, 256>>>(out, group, iters);
            else if (mode == 2) k<2><<<wg, 256>>>(out, group, iters); else if (mode == 4) k<4><<<wg, 256>>>(out, group, iters); else if (mode == 5) k<5><<<wg, 256>>>(out, group, iters); else k<3><<<wg, 256>>>(out, group, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double wave_instr = (double)wg * 4 * iters;              // wave-level atomic instructions issued
        const double per_cu_ns = ms * 1e6 / (wave_instr / 256.0);        // ns per wave-instruction per CU
        printf("mode=%d lanes/address=%2d : %8.1f us  %6.2f ns per wave-instruction per CU (%.1f cycles @2.1GHz)  %.1f G lane-atomics/s\n", mode, group,
               ms * 1e3, per_cu_ns, per_cu_ns * 2.1, wave_instr * 64 / (ms * 1e6));
    }
    return 0;
}
