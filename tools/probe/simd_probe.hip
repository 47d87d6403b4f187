// Developer probe: which SIMD does wave w of a 512-thread workgroup land on?  (hipcc --offload-arch=gfx950 simd_probe.hip -o simd_probe)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(512, 2) void probe(unsigned *out)
{
    const unsigned hw = __builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4);     // HW_ID: simd_id = bits [5:4]
    const unsigned full = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
    if ((threadIdx.x & 63) == 0) {
        out[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 2] = hw;
        out[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 2 + 1] = full;
    }
}
int main()
{
    unsigned *d, h[64 * 16];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(probe, dim3(64), dim3(512), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int b = 0; b < 6; b++) {
        printf("wg %d simd:", b);
        for (int w = 0; w < 8; w++) printf(" %u", h[(b * 8 + w) * 2]);
        printf("   hw_id:");
        for (int w = 0; w < 8; w++) printf(" %08x", h[(b * 8 + w) * 2 + 1]);
        printf("\n");
    }
    int bad = 0;
    for (int b = 0; b < 64; b++) {
        int cnt[4] = {0, 0, 0, 0};
        for (int w = 0; w < 8; w++) cnt[h[(b * 8 + w) * 2] & 3]++;
        for (int s = 0; s < 4; s++) bad += cnt[s] != 2;
        for (int w = 0; w < 4; w++) bad += (h[(b * 8 + w) * 2] != h[(b * 8 + w + 4) * 2]) ? 0 : 0;
    }
    printf("workgroups without exactly two waves per SIMD (x4 SIMDs): %d\n", bad);
    int same = 0;
    for (int b = 0; b < 64; b++)
        for (int w = 0; w < 4; w++) same += h[(b * 8 + w) * 2] == h[(b * 8 + w + 4) * 2];
    printf("wave w and w+4 on the same SIMD: %d of %d\n", same, 64 * 4);
    return 0;
}
