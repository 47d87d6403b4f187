// Error handling, device info, and small element-wise kernels.
#include "common.h"

#include <atomic>
#include <cstring>

namespace locov {

static std::atomic<long long> g_launch_checks{0};

static thread_local char g_err[512] = "";

char *err_buf() { return g_err; }

int set_error(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int check_launch(const char *what)
{
    g_launch_checks.fetch_add(1, std::memory_order_relaxed);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess)
        return set_error(LOCOV_ERR_LAUNCH, "%s: launch failed: %s", what, hipGetErrorString(e));
    return LOCOV_OK;
}

// fp32 -> bf16, 4 elements per lane (16-byte loads, 8-byte stores).  A plain cast lowers
// to v_cvt_pk_bf16_f32 on gfx950 (round-to-nearest-even, NaN stays NaN).
__global__ __launch_bounds__(256) void f32_to_bf16_kernel(const float *__restrict__ x, int64_t n,
                                                          __bf16 *__restrict__ y)
{
    const int64_t n4 = n >> 2;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const float4 v = reinterpret_cast<const float4 *>(x)[i];
        typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
        bf16x4 o;
        o[0] = (__bf16)v.x; o[1] = (__bf16)v.y; o[2] = (__bf16)v.z; o[3] = (__bf16)v.w;
        reinterpret_cast<bf16x4 *>(y)[i] = o;
    }
    // tail
    const int64_t t = (n4 << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) y[t] = (__bf16)x[t];
}

}  // namespace locov

extern "C" {

int locov_abi_version(void) { return LOCOV_ABI_VERSION; }

const char *locov_last_error(void) { return locov::err_buf(); }

int64_t locov_launch_count(void) { return (int64_t)locov::g_launch_checks.load(std::memory_order_relaxed); }

int locov_device_info(int *cu_count, int *wave_size, int *lds_bytes_per_cu)
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return locov::set_error(LOCOV_ERR_LAUNCH, "hipGetDevice: %s", hipGetErrorString(e));
    hipDeviceProp_t p;
    e = hipGetDeviceProperties(&p, dev);
    if (e != hipSuccess)
        return locov::set_error(LOCOV_ERR_LAUNCH, "hipGetDeviceProperties: %s", hipGetErrorString(e));
    if (cu_count) *cu_count = p.multiProcessorCount;
    if (wave_size) *wave_size = p.warpSize;
    if (lds_bytes_per_cu) *lds_bytes_per_cu = (int)p.maxSharedMemoryPerMultiProcessor;
    return LOCOV_OK;
}

int locov_f32_to_bf16(const float *x, int64_t n, uint16_t *y, locov_stream_t stream)
{
    LOCOV_REQUIRE(n >= 0, "locov_f32_to_bf16: n < 0");
    if (n == 0) return LOCOV_OK;
    LOCOV_REQUIRE(x && y, "locov_f32_to_bf16: null pointer");
    LOCOV_REQUIRE(((uintptr_t)x % 16 == 0) && ((uintptr_t)y % 8 == 0), "locov_f32_to_bf16: misaligned pointer");
    const int64_t blocks = locov::ceil_div(locov::ceil_div(n, 4) + 3, 256);
    const int grid = (int)(blocks < 2048 ? blocks : 2048);
    hipLaunchKernelGGL(locov::f32_to_bf16_kernel, dim3(grid), dim3(256), 0, locov::as_stream(stream), x, n,
                       reinterpret_cast<__bf16 *>(y));
    return locov::check_launch("locov_f32_to_bf16");
}

}  // extern "C"
