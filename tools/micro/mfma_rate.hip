// Developer micro-benchmark: issue rate of v_mfma_f32_32x32x16_bf16 on gfx950 with the conv kernel's
// occupancy (2 waves per SIMD) for several independent-accumulator counts, with and without LDS
// fragment reads in the loop.  Build: hipcc --offload-arch=gfx950 -O3 mfma_rate.hip -o mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;

template <int NACC, int WITH_LDS>
__global__ __launch_bounds__(512) void k(float* out, long long* cyc, int iters) {
    __shared__ bf16x8 lds[4096];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 4096; i += 512) { bf16x8 v; for (int j = 0; j < 8; ++j) v[j] = (__bf16)(0.001f * (i + j)); lds[i] = v; }
    __syncthreads();
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    bf16x8 fa = lds[lane], fb = lds[64 + lane];
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (WITH_LDS) {
            fa = lds[(it & 31) * 64 + lane];
            fb = lds[2048 + (it & 31) * 64 + lane];
        }
#pragma unroll
        for (int rep = 0; rep < 12 / NACC; ++rep)
#pragma unroll
            for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[a], 0, 0, 0);
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NACC, int WITH_LDS>
void run(const char* name, int threads) {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8);
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NACC, WITH_LDS>), dim3(256), dim3(threads), 0, 0, out, cyc, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NACC, WITH_LDS>), dim3(256), dim3(threads), 0, 0, out, cyc, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    const double waves_per_simd = threads / 256.0;
    const double mf = 12.0 * iters * waves_per_simd;                  // MFMAs per SIMD
    printf("%-28s threads %d: %.1f us, %.1f memtime ticks per MFMA (per SIMD), %.1f ns per MFMA -> %.0f TFLOP/s bf16, tick rate %.2f GHz\n", name, threads,
           ms * 1e3, h[0] / mf, ms * 1e6 / mf, 256.0 * 4 * mf * 32768 / (ms * 1e-3) / 1e12, h[0] / (ms * 1e6));
    hipFree(out); hipFree(cyc);
}

int main() {
    run<4, 0>("4 accumulators", 512);
    run<2, 0>("2 accumulators", 512);
    run<1, 0>("1 accumulator", 512);
    run<4, 1>("4 accumulators + ds_read", 512);
    run<4, 0>("4 accumulators", 256);
    run<4, 0>("4 accumulators", 1024);
    return 0;
}
