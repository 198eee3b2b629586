// micro-benchmark: cost of broadcasting per-triangle data via v_readlane vs LDS broadcast reads
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef short short2_t __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(int* out, const int* in, int iters) {
    __shared__ int4 sd[4][64][3];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int v[12];
    for (int i = 0; i < 12; i++) v[i] = in[(threadIdx.x * 12 + i) & 1023];
    sd[wave][lane][0] = make_int4(v[0], v[1], v[2], v[3]);
    sd[wave][lane][1] = make_int4(v[4], v[5], v[6], v[7]);
    sd[wave][lane][2] = make_int4(v[8], v[9], v[10], v[11]);
    const short2_t dxy = __builtin_bit_cast(short2_t, (lane & 7) * 16 | ((lane >> 3) * 16) << 16);
    unsigned long long best = ~0ull;
    unsigned long long mask0 = __ballot(v[0] & 1) | 0xffff;
    for (int it = 0; it < iters; it++) {
        unsigned long long mask = mask0;
        while (mask) {
            const int j = __builtin_ctzll(mask);
            mask &= mask - 1;
            int a[10];
            if (MODE == 0) {
                for (int i = 0; i < 10; i++) a[i] = __builtin_amdgcn_readlane(v[i], j);
            } else if (MODE == 1) {
                const int4 x = sd[wave][j][0], y = sd[wave][j][1], z = sd[wave][j][2];
                a[0] = x.x; a[1] = x.y; a[2] = x.z; a[3] = x.w; a[4] = y.x; a[5] = y.y; a[6] = y.z; a[7] = y.w; a[8] = z.x; a[9] = z.y;
            } else {
                for (int i = 0; i < 10; i++) a[i] = v[i] + j;  // no broadcast at all (VALU only)
            }
            const int e0 = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2_t, a[0]), dxy, a[3], false);
            const int e1 = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2_t, a[1]), dxy, a[4], false);
            const int e2 = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2_t, a[2]), dxy, a[5], false);
            const float zt = __fmaf_rn(__int_as_float(a[6]), 1.5f, __fmaf_rn(__int_as_float(a[7]), 2.5f, __int_as_float(a[8])));
            const bool inside = ((e0 | e1 | e2) >= 0) && (zt >= -1.0f && zt <= 1.0f);
            const unsigned u = __float_as_uint(zt);
            const unsigned long long key = ((unsigned long long)((u & 0x80000000u) ? ~u : (u | 0x80000000u)) << 32) | (unsigned)a[9];
            if (inside && key < best) best = key;
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = (int)best;
}
int main() {
    int *out, *in;
    hipMalloc(&out, 4 * 256 * 2048); hipMalloc(&in, 4096);
    std::vector<int> h(1024); for (int i = 0; i < 1024; i++) h[i] = (i * 2654435761u) >> 8;
    hipMemcpy(in, h.data(), 4096, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wg : {256, 1024, 2048}) for (int mode = 0; mode < 3; mode++) {
        const int iters = 200;
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            if (mode == 0) k<0><<<wg, 256>>>(out, in, iters); else if (mode == 1) k<1><<<wg, 256>>>(out, in, iters); else k<2><<<wg, 256>>>(out, in, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        // each wave: iters * popc(mask) iterations; popc>=16
        printf("wg=%d (waves/SIMD=%.1f) mode=%d: %.1f us  -> %.1f ns per wave-iteration (assuming 16..64 hits)\n", wg, wg * 4 / 1024.0, mode, ms * 1e3, ms * 1e6 / (iters * 16.0));
    }
    return 0;
}
