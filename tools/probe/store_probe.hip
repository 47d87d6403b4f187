// Developer probe (never the product): how fast can a GEMM-epilogue-shaped store stream leave the chip?
//   hipcc -O3 --offload-arch=gfx950 tools/probe/store_probe.hip -o /tmp/store_probe && /tmp/store_probe
// Each wave instruction stores 64 x 16 B.  SHAPE 0: 4 rows x 256 B of a [M, 2048] fp32 matrix (the shipped epilogue);
// 1: 2 rows x 512 B; 2: 1 row x 1 KB (what a 256-column tile would give); variants: waves per workgroup, workgroups per CU
// (by LDS), nt / default policy, and the same with an equally shaped load stream beside it (the residual).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE, int AUX, bool LOAD>
__global__ __launch_bounds__(256) void store_kernel(float *__restrict__ out, const float *__restrict__ res, long M, int N, int lds_pad)
{
    extern __shared__ char pad[];
    // a workgroup owns a 128 x 128 output tile (as the GEMM's): 64 wave-instructions of 1 KB, 16 per wave
    const int tiles_n = N / 128;
    const long tile = blockIdx.x;
    const long m0 = (tile / tiles_n) * 128;
    const int n0 = (tile % tiles_n) * 128;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lds_pad < 0) pad[threadIdx.x] = 0;
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(out + m0 * N, 0, 0xffffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(res) + m0 * N, 0, 0xffffffff, 0x00020000);
    // SHAPE 0: wave = 64x64 sub-tile, lane -> row lane/16 (4 rows), 16 B chunk lane%16 (256 B per row)
    f32x4 v = {1.f, 2.f, 3.f, (float)lane};
#pragma unroll
    for (int it = 0; it < 16; it++) {
        unsigned off;
        if (SHAPE == 0) {
            const int row = (wave >> 1) * 64 + it * 4 + (lane >> 4), col = n0 + (wave & 1) * 64 + (lane & 15) * 4;
            off = (unsigned)((row * (long)N + col) * 4);
        } else if (SHAPE == 1) {          // 2 rows x 512 B (a 128-column tile row = 512 B)
            const int row = wave * 32 + it * 2 + (lane >> 5), col = n0 + (lane & 31) * 4;
            off = (unsigned)((row * (long)N + col) * 4);
        } else {                          // 1 row x 1 KB: the workgroup's tile is 64 rows x 256 columns here
            const long t2 = tile;
            const int tn2 = N / 256;
            const long mm = (t2 / tn2) * 64 - m0;   // relative to the descriptor base
            const int nn = (int)(t2 % tn2) * 256;
            const int row = wave * 16 + it;
            off = (unsigned)(((mm + row) * (long)N + nn + lane * 4) * 4);
        }
        if (LOAD) {
            const f32x4 r = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rr, off, 0, 2));
            v += r;
        }
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ro, off, 0, AUX);
    }
}

template <int SHAPE, int AUX, bool LOAD>
static void run(const char *name, float *out, const float *res, long M, int N, int lds)
{
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    const long tiles = (M / 128) * (N / 128);
    float best = 1e9f;
    for (int rep = 0; rep < 4; rep++) {
        hipEventRecord(a);
        for (int i = 0; i < 3; i++) hipLaunchKernelGGL((store_kernel<SHAPE, AUX, LOAD>), dim3((unsigned)tiles), dim3(256), lds, 0, out, res, M, N, 0);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        if (rep > 0 && ms / 3 < best) best = ms / 3;
    }
    const double gb = (double)M * N * 4 / 1e9;
    printf("%-46s lds/wg %6d B: %.3f ms  write %.2f TB/s%s\n", name, lds, best, gb / best, LOAD ? "  (+ equal read stream)" : "");
    fflush(stdout);
}

int main()
{
    const long M = 392000 - 392000 % 128;
    const int N = 2048;
    float *out, *res;
    hipMalloc(&out, M * N * 4);
    hipMalloc(&res, M * N * 4);
    hipMemset(res, 0, M * N * 4);
    for (int lds : {0, 36 * 1024, 72 * 1024}) {        // 4 / 4 / 2 workgroups per CU (LDS-limited), i.e. 16 / 16 / 8 waves
        run<0, 2, false>("4 rows x 256 B, nt", out, res, M, N, lds);
        run<0, 0, false>("4 rows x 256 B, default", out, res, M, N, lds);
        run<1, 2, false>("2 rows x 512 B, nt", out, res, M, N, lds);
        run<2, 2, false>("1 row x 1 KB, nt", out, res, M, N, lds);
        run<2, 0, false>("1 row x 1 KB, default", out, res, M, N, lds);
        run<0, 2, true>("4 rows x 256 B, nt", out, res, M, N, lds);
        run<2, 2, true>("1 row x 1 KB, nt", out, res, M, N, lds);
    }
    run<0, 2, false>("4 rows x 256 B, nt (1 WG/CU)", out, res, M, N, 150 * 1024);
    run<2, 2, false>("1 row x 1 KB, nt (1 WG/CU)", out, res, M, N, 150 * 1024);
    run<0, 2, true>("4 rows x 256 B, nt (1 WG/CU)", out, res, M, N, 150 * 1024);
    return 0;
}
