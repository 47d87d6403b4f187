// Developer probe: f16 MFMA throughput of a register-only loop, per instruction shape, at the package power cap.
//   hipcc -O3 --offload-arch=gfx950 mfma_energy_probe.hip -o mfma_energy_probe ;  ./mfma_energy_probe [seconds per arm]
// The split GEMM runs at the 1 400 W cap (profiles/r03_power_probe.txt), so what a shape sustains HERE -- no LDS, no memory, the same
// 128x64 wave tile and register footprint as gemm_split_big.hip -- is a direct reading of its energy per FLOP.  tools/attic/dbg_mfma_energy.py
// samples power and clock beside it.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ h8 frag(unsigned seed)
{
    h8 v;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        seed = seed * 1664525u + 1013904223u;
        v[i] = (_Float16)(((int)(seed >> 20) - 2048) * (1.0f / 2048.0f));
    }
    return v;
}

// arm 0: 16x16x32, 8 x 4 tiles;  arm 1: 32x32x16, 4 x 2 tiles x 2 k-steps;  both 262 144 MACs per (wave, inner pass)
// ORDER (arm 0 only): 0 = A operand shared by four consecutive MFMAs (the shipped order), 1 = both operands change on every MFMA,
// 2 = ONE operand pair for all of them (accumulators still differ), 3 = as 0 with all-zero operands (no data toggling)
template <int ARM, int WAVES, int ORDER = 0>
__global__ __launch_bounds__(WAVES * 64) void mfma_loop(float *out, int iters)
{
    const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
    h8 a[8], b[4];
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = frag(t * 31u + i);
#pragma unroll
    for (int j = 0; j < 4; j++) b[j] = frag(t * 17u + 1000u + j);
    if (ORDER == 3) {
#pragma unroll
        for (int i = 0; i < 8; i++) a[i] = a[i] * (_Float16)(iters < 0 ? 1.f : 0.f);
#pragma unroll
        for (int j = 0; j < 4; j++) b[j] = b[j] * (_Float16)(iters < 0 ? 1.f : 0.f);
    }
    float s = 0.f;
    if (ARM == 0) {
        f4 acc[8][4];
#pragma unroll
        for (int i = 0; i < 8; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) acc[i][j] = f4{0, 0, 0, 0};
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int p = 0; p < 3; p++)
#pragma unroll
                for (int i = 0; i < 8; i++)
#pragma unroll
                    for (int j = 0; j < 4; j++)
                        acc[i][j] = ORDER == 1   ? __builtin_amdgcn_mfma_f32_16x16x32_f16(a[(i + j + p) & 7], b[(j + p) & 3], acc[i][j], 0, 0, 0)
                                    : ORDER == 2 ? __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0], b[0], acc[i][j], 0, 0, 0)
                                                 : __builtin_amdgcn_mfma_f32_16x16x32_f16(a[(i + p) & 7], b[(j + p) & 3], acc[i][j], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 8; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    } else {
        f16v acc[4][2];
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int p = 0; p < 3; p++)
#pragma unroll
                for (int k = 0; k < 2; k++)
#pragma unroll
                    for (int i = 0; i < 4; i++)
#pragma unroll
                        for (int j = 0; j < 2; j++)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(i * 2 + k + p) & 7], b[(j * 2 + k + p) & 3], acc[i][j], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int e = 0; e < 16; e++) s += acc[i][j][e];
    }
    if (s == 123.456f) out[t] = s;
}

template <int ARM, int WAVES, int ORDER = 0>
static void run(const char *name, double seconds, float *d)
{
    const int iters = 2000, wgs = 256 * (8 / WAVES) * 4;
    const double flops_per_launch = 2.0 * 262144.0 * 3.0 * iters * (double)wgs * WAVES;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const size_t lds = WAVES == 8 ? 0 : 120 * 1024;          // 4-wave arms: one workgroup per CU
    if (lds) hipFuncSetAttribute((const void *)mfma_loop<ARM, WAVES, ORDER>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL((mfma_loop<ARM, WAVES, ORDER>), dim3(wgs), dim3(WAVES * 64), lds, 0, d, iters);
    hipDeviceSynchronize();
    auto t0 = std::chrono::steady_clock::now();
    double last = 0;
    int rounds = 0;
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
        hipEventRecord(e0, 0);
        for (int i = 0; i < 20; i++) hipLaunchKernelGGL((mfma_loop<ARM, WAVES, ORDER>), dim3(wgs), dim3(WAVES * 64), lds, 0, d, iters);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        last = flops_per_launch * 20 / (ms * 1e-3) / 1e12;
        rounds++;
    }
    printf("%-44s %8.1f TFLOP/s (last of %d rounds)  t=%.1f s\n", name, last, rounds,
           std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count());
    fflush(stdout);
}

int main(int argc, char **argv)
{
    const double seconds = argc > 1 ? atof(argv[1]) : 4.0;
    float *d;
    hipMalloc(&d, 256 * 32 * 512 * sizeof(float));
    run<0, 8>("v_mfma_f32_16x16x32_f16, 8 waves/CU", seconds, d);
    run<1, 8>("v_mfma_f32_32x32x16_f16, 8 waves/CU", seconds, d);
    run<0, 4>("v_mfma_f32_16x16x32_f16, 4 waves/CU", seconds, d);
    run<1, 4>("v_mfma_f32_32x32x16_f16, 4 waves/CU", seconds, d);
    run<0, 8, 1>("16x16x32, 8 waves, both operands change", seconds, d);
    run<0, 8, 2>("16x16x32, 8 waves, one operand pair", seconds, d);
    run<0, 8, 3>("16x16x32, 8 waves, zero operands", seconds, d);
    return 0;
}
