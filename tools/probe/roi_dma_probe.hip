// Developer probe (never the product; round 5): what would the pooler-contract ROIAlign gain from taking a LARGE
// proposal's taps out of an LDS ring filled by LDS DMA (buffer_load ... lds: in-flight bytes that cost no registers)?
//   hipcc -O3 --offload-arch=gfx950 tools/probe/roi_dma_probe.hip -o /tmp/roi_dma_probe && /tmp/roi_dma_probe
// The probe keeps the STRUCTURE of that kernel and none of its arithmetic details: one workgroup = one proposal x 32 channels
// (128 B per pixel); the proposal's pixel rows stream through a ring in LDS, each row fetched ONCE by 1 KB wave instructions
// (8 pixels x 128 B, lane data contiguous in LDS); per stage (two bin rows = 28 bins x 8 channel quads = 224 threads) every thread
// accumulates S samples x 4 taps (ds_read_b128, sequential accumulation as torchvision's order demands) while the next stage's rows
// are in flight; the 7 results per thread go through the [32][196] transpose tile and leave as one 25 KB run.  LDS per workgroup
// = tile + ring => ONE workgroup (4 waves) per CU for the 448-800 px class.
// Reported per size class: ms per 8 000 proposals, to be read against tools/attic/ab_t2_sizes.py's figures of the product on the same box
// (round 4: 224-448 px 2.88 ms, 448-800 px 5.40 ms).  The probe is an UPPER bound on what the real kernel could reach: no sampling
// validity rules, perfectly regular rows (the sampling tables are the real kernel's: one 16-byte LDS entry per axis sample).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int kC = 1024, kW = 84, kH = 50, kBins = 196, kTileStride = 197;

// ww: pixels per ring row; rows_res: pixel rows a stage's taps span; rows_new: rows appended per stage; gh x gw samples per bin
__global__ __launch_bounds__(256) void probe_kernel(const float *__restrict__ feat, float *__restrict__ out, int ww, int rows_res, int rows_new,
                                                    int gh, int gw, int ring_rows, int stages)
{
    extern __shared__ char lds[];
    float *tile = reinterpret_cast<float *>(lds);                     // [32][197]
    char *ring = lds + 32 * kTileStride * 4 + 16 - (32 * kTileStride * 4) % 16;
    const int r = blockIdx.x, c0 = blockIdx.y * 32;
    const int img = r & 7;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int chunks = (ww + 7) / 8, pitch = chunks * 1024;
    const int x0 = (r * 7) % (kW - ww + 1), y0 = (r * 3) % (kH - (rows_res + rows_new * stages) > 0 ? kH - (rows_res + rows_new * stages) : 1);
    const char *base = reinterpret_cast<const char *>(feat + (int64_t)img * kH * kW * kC);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(base), 0, (unsigned)(kH * kW * kC * 4), 0x00020000);
    auto dma_rows = [&](int row_lo, int n_rows) {      // rows [row_lo, row_lo + n_rows) of the proposal's footprint -> their ring slots
        const int total = n_rows * chunks;
        for (int j = wave; j < total; j += 4) {
            const int row = row_lo + j / chunks, ch = j % chunks;
            int y = y0 + row, x = x0 + ch * 8 + (lane >> 3);
            y = y < kH ? y : kH - 1;
            x = x < kW ? x : kW - 1;
            const unsigned voff = (unsigned)(((y * kW + x) * kC + c0) * 4 + (lane & 7) * 16);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)(ring + (row % ring_rows) * pitch + ch * 1024), 16,
                                                     voff, 0, 0, 0);
        }
    };
    dma_rows(0, rows_res);
    // sampling tables as in the real kernel (one 16-byte entry per axis sample: two byte offsets into the ring / the row, two weights)
    struct Tab { int lo, hi; float wl, wh; };
    Tab *ytab = reinterpret_cast<Tab *>(ring + ring_rows * pitch);        // [14 bin rows][gh]
    Tab *xtab = ytab + 14 * 4;                                              // [14 bin columns][gw]
    for (int t = tid; t < 14 * gh; t += 256) {
        const int ph = t / gh, iy = t % gh;
        const int row = (ph / 2) * rows_new + (ph & 1) * ((rows_res - 1) / 2) + (iy * (rows_res / 2)) / gh;
        ytab[ph * 4 + iy] = Tab{(row % ring_rows) * pitch, ((row + 1) % ring_rows) * pitch, 0.3f + 0.01f * iy, 0.7f - 0.01f * iy};
    }
    for (int t = tid; t < 14 * gw; t += 256) {
        const int pxx = t / gw, ix = t % gw;
        int xl = (pxx * (ww - 1)) / 14 + (ix * ((ww - 1) / 14 + 1)) / gw;
        xl = xl < ww - 1 ? xl : ww - 2;
        xtab[pxx * 4 + ix] = Tab{xl * 128, (xl + 1) * 128, 0.4f, 0.6f};
    }
    const int bin = tid >> 3, q = tid & 7;                // 28 bins x 8 quads = 224 active threads
    const bool active = bin < 28;
    const int px = bin % 14;
    f32x4 res[7];
#pragma unroll
    for (int s = 0; s < 7; s++) res[s] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int S = gh * gw;
#pragma unroll 1
    for (int s = 0; s < stages; s++) {
        __builtin_amdgcn_s_waitcnt(0x0070);               // vmcnt(0): this stage's rows have landed (this wave's share)
        __syncthreads();
        if (s + 1 < stages) dma_rows(rows_res + s * rows_new, rows_new);       // next stage's new rows, in flight under the taps
        if (active) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            const int ph = 2 * s + bin / 14;
            int iy = 0, ix = 0;
            const char *p = ring + q * 16;
            for (int k = 0; k < S; k++) {
                const Tab ys = ytab[ph * 4 + iy], xs = xtab[px * 4 + ix];
                const float w1 = ys.wh * xs.wh, w2 = ys.wh * xs.wl, w3 = ys.wl * xs.wh, w4 = ys.wl * xs.wl;
                const f32x4 v1 = *reinterpret_cast<const f32x4 *>(p + ys.lo + xs.lo);
                const f32x4 v2 = *reinterpret_cast<const f32x4 *>(p + ys.lo + xs.hi);
                const f32x4 v3 = *reinterpret_cast<const f32x4 *>(p + ys.hi + xs.lo);
                const f32x4 v4 = *reinterpret_cast<const f32x4 *>(p + ys.hi + xs.hi);
                acc = acc + (((w1 * v1 + w2 * v2) + w3 * v3) + w4 * v4);
                if (++ix == gw) {
                    ix = 0;
                    iy++;
                }
            }
            // (stage s's result register: selected without dynamic indexing)
#pragma unroll
            for (int u = 0; u < 7; u++)
                if (u == s) res[u] = acc;
        }
        __syncthreads();                                  // the taps are done: rows below the next stage's window may be overwritten
    }
    if (active) {
#pragma unroll
        for (int s = 0; s < 7; s++) {
            const int b = s * 28 + bin;
            float *t = tile + (4 * q) * kTileStride + b;
            t[0] = res[s][0];
            t[kTileStride] = res[s][1];
            t[2 * kTileStride] = res[s][2];
            t[3 * kTileStride] = res[s][3];
        }
    }
    __syncthreads();
    float *dst = out + ((int64_t)r * kC + c0) * kBins;
    for (int i = tid; i < 32 * 49; i += 256) {             // 32 channels x 49 quads of bins
        const int c = i / 49, b4 = i % 49;
        const float *t = tile + c * kTileStride + 4 * b4;
        const f32x4 v = {t[0], t[1], t[2], t[3]};
        __builtin_nontemporal_store(v, reinterpret_cast<f32x4 *>(dst + c * kBins + 4 * b4));
    }
}

int main()
{
    const int R = 8000;
    float *feat, *out;
    hipMalloc(&feat, (size_t)8 * kH * kW * kC * 4);
    hipMalloc(&out, (size_t)R * kC * kBins * 4);
    hipMemset(feat, 0, (size_t)8 * kH * kW * kC * 4);
    struct Cls { const char *name; int ww, rows_res, rows_new, gh, gw; };
    // side s px -> s/16 map pixels; bin = s/16/14; grid = ceil(bin); a stage = 2 bin rows: spans 2 bin + 2 rows, appends 2 bin rows
    const Cls cls[] = {{"112-224 px (avg 168)", 13, 4, 2, 1, 1}, {"224-448 px (avg 336)", 23, 5, 3, 2, 2}, {"448-800 px (avg 624)", 41, 8, 6, 3, 3},
                       {"800 px", 52, 10, 7, 4, 4}};
    for (const Cls &c : cls) {
        const int ring_rows = c.rows_res + c.rows_new;
        const int pitch = (c.ww + 7) / 8 * 1024;
        const size_t lds = 32 * kTileStride * 4 + 16 + (size_t)ring_rows * pitch + 2 * 14 * 4 * 16;
        hipFuncSetAttribute((const void *)probe_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipEvent_t a, b;
        hipEventCreate(&a);
        hipEventCreate(&b);
        float best = 1e9f;
        for (int rep = 0; rep < 4; rep++) {
            hipEventRecord(a);
            for (int i = 0; i < 5; i++)
                hipLaunchKernelGGL(probe_kernel, dim3(R, kC / 32), dim3(256), lds, 0, feat, out, c.ww, c.rows_res, c.rows_new, c.gh, c.gw, ring_rows, 7);
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms;
            hipEventElapsedTime(&ms, a, b);
            if (ms / 5 < best) best = ms / 5;
        }
        if (hipGetLastError() != hipSuccess) printf("launch error\n");
        const double taps_gb = (double)R * 32 * kBins * c.gh * c.gw * 4 * 128 / 1e9, fetch_gb = (double)R * 32 * (c.rows_res + 6 * c.rows_new) * c.ww * 128 / 1e9;
        printf("%-22s LDS %6.1f KB/workgroup (%d per CU)  %7.3f ms per 8000 proposals   taps %6.1f GB from LDS (%5.1f TB/s)  ring fill %5.1f GB  stores 6.4 GB\n", c.name,
               lds / 1024.0, (int)(160 * 1024 / lds), best, taps_gb, taps_gb / best, fetch_gb);
    }
    return 0;
}
