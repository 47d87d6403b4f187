// Developer probe (never the product; round 5): fp32 global atomics at AGENT scope (performed at the memory side: the eight XCDs' L2s are
// not coherent with each other) against WORKGROUP scope (performed in the issuing XCD's L2), on the access pattern of the even-grid
// ROIAlign backward: every wave adds 64 consecutive floats (256 B) at scattered lines of the channel slice its XCD owns.
//   hipcc -O3 --offload-arch=gfx950 tools/probe/atomic_scope_probe.hip -o /tmp/atomic_scope_probe && /tmp/atomic_scope_probe
// Also reported: whether blockIdx % 8 equals the hardware XCC_ID of the workgroup (the dispatcher's round-robin), and whether the
// L2-scope sums are exact when the slice is chosen (a) by blockIdx % 8 and (b) by the hardware XCC_ID.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ unsigned xcc_id()
{
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xf;
}

// mode 0: agent scope; 1: workgroup scope, slice = blockIdx % 8; 2: workgroup scope, slice = XCC_ID
template <int MODE>
__global__ __launch_bounds__(256) void probe(float *buf, unsigned lines_per_slice, int iters, unsigned *mismatch)
{
    const unsigned hw = xcc_id();
    if (threadIdx.x == 0 && hw != (blockIdx.x & 7)) atomicAdd(mismatch, 1u);
    const unsigned slice = MODE == 2 ? hw : (blockIdx.x & 7);
    float *base = buf + (size_t)slice * lines_per_slice * 64;
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned state = blockIdx.x * 4 + wave + 1;
    for (int i = 0; i < iters; i++) {
        state = state * 1664525u + 1013904223u;
        const unsigned line = (state >> 8) % lines_per_slice;       // a 256-byte run
        float *p = base + (size_t)line * 64 + lane;
        if (MODE == 0)
            __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else
            __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}

int main()
{
    const unsigned lines_per_slice = 34000;                            // 8 x 34 000 x 256 B = 69.6 MB: the res4 gradient of an LSM step
    const size_t n = (size_t)8 * lines_per_slice * 64;
    float *buf;
    unsigned *mm;
    hipMalloc(&buf, n * 4);
    hipMalloc(&mm, 4);
    const int wgs = 2048, iters = 2000;
    const double total = (double)wgs * 4 * iters * 64;
    for (int mode = 0; mode < 3; mode++) {
        hipMemset(buf, 0, n * 4);
        hipMemset(mm, 0, 4);
        hipEvent_t a, b;
        hipEventCreate(&a);
        hipEventCreate(&b);
        hipEventRecord(a);
        if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(wgs), dim3(256), 0, 0, buf, lines_per_slice, iters, mm);
        if (mode == 1) hipLaunchKernelGGL(probe<1>, dim3(wgs), dim3(256), 0, 0, buf, lines_per_slice, iters, mm);
        if (mode == 2) hipLaunchKernelGGL(probe<2>, dim3(wgs), dim3(256), 0, 0, buf, lines_per_slice, iters, mm);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        std::vector<float> h(n);
        unsigned mismatch;
        hipMemcpy(h.data(), buf, n * 4, hipMemcpyDeviceToHost);
        hipMemcpy(&mismatch, mm, 4, hipMemcpyDeviceToHost);
        double sum = 0;
        for (size_t i = 0; i < n; i++) sum += h[i];
        printf("%-44s %8.3f ms  %7.1f G atomic lanes/s (%5.2f TB/s of payload)  sum %s (%.0f of %.0f)  workgroups with blockIdx %% 8 != XCC_ID: %u of %d\n",
               mode == 0 ? "agent scope" : mode == 1 ? "workgroup scope, slice = blockIdx % 8" : "workgroup scope, slice = hardware XCC_ID", ms,
               total / ms / 1e6, total * 4 / ms / 1e9, sum == total ? "exact" : "WRONG", sum, total, mismatch, wgs);
    }
    return 0;
}
