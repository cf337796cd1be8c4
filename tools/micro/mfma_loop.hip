// Developer micro-benchmark: the conv kernel's K loop reduced to its skeleton -- 8 waves per CU, 12 MFMAs per wave and
// step on 4 accumulators in three groups of 4, 9 steps per phase -- with the loop's other ingredients switched on one
// at a time: the barrier per phase, the per-step s_setprio alternation, the 8 fragment reads per step.
// Build: hipcc --offload-arch=gfx950 -O3 mfma_loop.hip -o mfma_loop
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;

template <int BARRIER, int PRIO, int FRAG, int LAYOUT, int NVALU>
__global__ __launch_bounds__(512) void k(float* out, long long* cyc, int phases) {
    extern __shared__ __attribute__((aligned(16))) unsigned char raw[];
    bf16x8* lds = reinterpret_cast<bf16x8*>(raw);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = __builtin_amdgcn_readfirstlane(wave >> 2);
    for (int i = threadIdx.x; i < (LAYOUT ? 9600 : 8192); i += 512) { bf16x8 v; for (int j = 0; j < 8; ++j) v[j] = (__bf16)(0.001f * ((i + j) & 255)); lds[i] = v; }
    __syncthreads();
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    bf16x8 f[2][8];
    for (int i = 0; i < 8; ++i) { f[0][i] = lds[i * 64 + lane]; f[1][i] = lds[512 + i * 64 + lane]; }
    float vx[4] = {1.f + lane, 2.f, 3.f, 4.f}; const float vy = 1.0001f;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int ph = 0; ph < phases; ++ph) {
#pragma unroll
        for (int s = 0; s < 9; ++s) {
            if (BARRIER && s == 8) __syncthreads();
            if (FRAG) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    if (LAYOUT == 0) f[(s + 1) & 1][i] = lds[((s * 8 + i) & 63) * 64 + wave * 512 + lane];
                    else if (i < 4) {      // B: [part][khalf][18 rows][34 px]; rows wave + ky (+ 8), column px + kx
                        const int NPX = 18 * 34, half = lane >> 5, px = lane & 31, ky = s / 3, kx = s % 3;
                        f[(s + 1) & 1][i] = lds[(ph & 1) * 4 * NPX + (i & 1) * 2 * NPX + half * NPX + (wave + ky + 8 * (i >> 1)) * 34 + px + kx];
                    } else                 // A: [step][t][part][64 lanes], the same for every wave
                        f[(s + 1) & 1][i] = lds[2 * 4 * 18 * 34 + (ph & 1) * 2304 + (s * 4 + (i - 4)) * 64 + lane];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (PRIO) { if (((s ^ grp) & 1) != 0) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
            const bf16x8* fr = f[s & 1];
#pragma unroll
            for (int part = 0; part < 3; ++part) {
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[(a & 1) * 2 + (part & 1)], fr[4 + (a >> 1) * 2 + (part >> 1)], acc[a], 0, 0, 0);
#pragma unroll
                    for (int v = 0; v < NVALU; ++v) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(vx[v & 3]) : "v"(vy));   // independent vector work behind every MFMA
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float sum = vx[0] + vx[1] + vx[2] + vx[3];
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) sum += acc[a][r];
    out[blockIdx.x * 512 + threadIdx.x] = sum;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int BARRIER, int PRIO, int FRAG, int LAYOUT, int NVALU = 0>
void run(const char* name) {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8);
    const int phases = 200;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k<BARRIER, PRIO, FRAG, LAYOUT, NVALU>), hipFuncAttributeMaxDynamicSharedMemorySize, 153600);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<BARRIER, PRIO, FRAG, LAYOUT, NVALU>), dim3(256), dim3(512), LAYOUT ? 153600 : 131072, 0, out, cyc, phases);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<BARRIER, PRIO, FRAG, LAYOUT, NVALU>), dim3(256), dim3(512), LAYOUT ? 153600 : 131072, 0, out, cyc, phases);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    const double mf = 12.0 * 9 * phases * 2;                         // MFMAs per SIMD
    printf("%-44s %.1f us, %.1f ticks per MFMA per SIMD (%.0f per step), tick rate %.2f GHz\n", name, ms * 1e3, h[0] / mf, 24.0 * h[0] / mf, h[0] / (ms * 1e6));
    hipFree(out); hipFree(cyc);
}

int main() {
    run<0, 0, 0, 0>("MFMAs only");
    run<1, 0, 0, 0>("+ barrier per phase");
    run<0, 1, 0, 0>("+ setprio alternation");
    run<1, 1, 0, 0>("+ barrier + setprio");
    run<0, 0, 1, 0>("+ 8 fragment reads per step");
    run<1, 0, 1, 0>("+ barrier + fragment reads");
    run<1, 1, 1, 0>("+ barrier + setprio + fragment reads");
    run<1, 0, 1, 1>("+ barrier + fragment reads, conv layout");
    run<1, 1, 1, 1>("+ barrier + setprio + reads, conv layout");
    run<1, 1, 1, 1, 1>("... + 1 VALU per MFMA");
    run<1, 1, 1, 1, 2>("... + 2 VALU per MFMA");
    run<1, 1, 1, 1, 3>("... + 3 VALU per MFMA");
    run<1, 1, 1, 1, 4>("... + 4 VALU per MFMA");
    run<1, 1, 1, 1, 6>("... + 6 VALU per MFMA");
    run<1, 0, 0, 0, 3>("MFMAs + barrier + 3 VALU per MFMA");
    run<1, 0, 0, 0, 6>("MFMAs + barrier + 6 VALU per MFMA");
    return 0;
}
