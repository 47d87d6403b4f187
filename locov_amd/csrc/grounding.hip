// LSM grounding: word<->region alignment and the [caption, image] cost matrices, fused.
//
// Replaces the B^2-replicated elementwise chain of GroundingHead.forward
// (ovr/modeling/mmss_heads/grounding_head.py:116-243: repeat x6, bmm, /temperature, masked fill,
// softmax over regions and over words, attention * distance, masked sums, /num_words|regions)
// for LOCAL_METRIC "dot", ALIGNMENT "softmax", GLOBAL_METRIC "aligned_local" -- the only
// combination configs/coco_lsm.yaml selects.
//
// Input is ONE similarity matrix S = caption_tokens . region_embeddings^T of shape [B*T, B*NR]
// (a single NT GEMM of this library instead of a bmm over B^2 materialised copies); workgroup
// (c, i) owns the TxNR block of caption c vs image i, stages it in LDS, and reduces it to the two
// scalars cost_w2r[c,i], cost_r2w[c,i] with wavefront shuffles.  The backward kernel recomputes
// the softmaxes (cheaper than storing them) and emits dS.
//
// Masked pairs: the reference fills them with (global min - 100) before the softmax; their
// weight is then <= e^-100 (below fp32 resolution), so they are simply excluded here.  A row /
// column with NO valid entry is a uniform distribution over all entries in the reference (every
// entry equals the fill value) and is reproduced as such.
#include "common.h"

namespace locov {

constexpr int kGroundThreads = 256;

__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
    return v;
}
__device__ __forceinline__ float wave_add(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// BWD = false: cost_w2r[c,i], cost_r2w[c,i].   BWD = true: dS block from (g_w2r[c,i], g_r2w[c,i]).
template <bool BWD>
__global__ __launch_bounds__(kGroundThreads) void grounding_kernel(
    const float *__restrict__ S, int B, int T, int NR, const float *__restrict__ cmask,
    const float *__restrict__ rmask, float inv_temp, float *__restrict__ cost_w2r, float *__restrict__ cost_r2w,
    const float *__restrict__ g_w2r, const float *__restrict__ g_r2w, float *__restrict__ dS)
{
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float *P = sm;                       // [T][NR]  similarities / temperature
    float *red = sm + T * NR;            // [T] w2r per-word terms (or stats), then [NR] r2w per-region terms
    float *rowf = red;                   // BWD: f_t per word            [T]
    float *rowz = red + T;               // BWD: 1/sum exp per word      [T]
    float *rowm = red + 2 * T;           // BWD: max per word            [T]

    const int c = blockIdx.x / B, i = blockIdx.x % B;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t ld = (int64_t)B * NR;
    const float *Sblk = S + ((int64_t)c * T) * ld + (int64_t)i * NR;
    const float *cm = cmask + (int64_t)c * T;
    const float *rm = rmask + (int64_t)i * NR;

    for (int idx = tid; idx < T * NR; idx += kGroundThreads) {
        const int t = idx / NR, r = idx - t * NR;
        P[idx] = Sblk[(int64_t)t * ld + r] * inv_temp;
    }
    // valid words of caption c / regions of image i (block-uniform counts)
    float nw = 0.f, nr = 0.f;
    for (int t = 0; t < T; t++) nw += cm[t] > 0.f ? 1.f : 0.f;
    for (int r = 0; r < NR; r++) nr += rm[r] > 0.f ? 1.f : 0.f;
    __syncthreads();

    const float gw = BWD ? g_w2r[blockIdx.x] / fmaxf(nw, 1.f) : 0.f;
    const float gr = BWD ? g_r2w[blockIdx.x] / fmaxf(nr, 1.f) : 0.f;

    // ---- words -> regions: one wave per word row, lanes over regions ---------------------------
    float acc_w = 0.f;
    for (int t = wave; t < T; t += kGroundThreads / 64) {
        const bool wv = cm[t] > 0.f;
        const float *row = P + t * NR;
        float m = -3.0e38f;
        for (int r = lane; r < NR; r += 64)
            if (nr == 0.f || rm[r] > 0.f) m = fmaxf(m, row[r]);
        m = wave_max(m);
        float z = 0.f, f = 0.f;
        for (int r = lane; r < NR; r += 64) {
            if (nr == 0.f || rm[r] > 0.f) {          // no valid region at all -> uniform over every region
                const float e = nr == 0.f ? 1.f : expf(row[r] - m);
                z += e;
                f += e * (-row[r]);
            }
        }
        z = wave_add(z);
        f = wave_add(f) / z;                         // sum_r att[t,r] * (-s[t,r])
        if (!BWD) {
            if (wv) acc_w += f;
        } else if (lane == 0) {
            rowf[t] = f;
            rowz[t] = wv ? 1.f / z : 0.f;            // zero -> a masked word contributes no gradient
            rowm[t] = m;
        }
    }
    if (!BWD) {
        if (lane == 0) red[wave] = acc_w;            // 4 partial sums
    }
    __syncthreads();
    float cw = 0.f;
    if (!BWD) {
        cw = (red[0] + red[1] + red[2] + red[3]) / fmaxf(nw, 1.f);
        __syncthreads();
    }

    // ---- regions -> words: one lane per region column, loop over words ---------------------------
    float acc_r = 0.f;
    for (int r = tid; r < NR; r += kGroundThreads) {
        const bool rv = rm[r] > 0.f;
        float m = -3.0e38f;
        for (int t = 0; t < T; t++)
            if (nw == 0.f || cm[t] > 0.f) m = fmaxf(m, P[t * NR + r]);
        float z = 0.f, f = 0.f;
        for (int t = 0; t < T; t++) {
            if (nw == 0.f || cm[t] > 0.f) {
                const float s = P[t * NR + r];
                const float e = nw == 0.f ? 1.f : expf(s - m);
                z += e;
                f += e * (-s);
            }
        }
        f /= z;
        if (!BWD) {
            if (rv) acc_r += f;
        } else {
            // dS[t,r] = gw * a_w[t,r] * (-s - f_t - 1)  +  gr * a_r[t,r] * (-s - f_r - 1), times 1/temperature
            const float invz = rv ? 1.f / z : 0.f;
            for (int t = 0; t < T; t++) {
                const float s = P[t * NR + r];
                float d = 0.f;
                // f = sum_r a_r * (-s_r): a valid entry gets a_r * (-s_r - f - 1) (softmax + distance
                // paths); in the uniform (nothing valid) case only the distance path exists: -a_r
                if (nr == 0.f) {
                    d -= gw * rowz[t];
                } else if (rv) {
                    const float aw = expf(s - rowm[t]) * rowz[t];
                    d += gw * aw * (-s - rowf[t] - 1.f);
                }
                if (nw == 0.f) {
                    d -= gr * invz;
                } else if (cm[t] > 0.f) {
                    const float ar = expf(s - m) * invz;
                    d += gr * ar * (-s - f - 1.f);
                }
                dS[((int64_t)c * T + t) * ld + (int64_t)i * NR + r] = d * inv_temp;
            }
        }
    }
    if (!BWD) {
        acc_r = wave_add(acc_r);
        if (lane == 0) red[wave] = acc_r;
        __syncthreads();
        if (tid == 0) {
            cost_w2r[blockIdx.x] = cw;
            cost_r2w[blockIdx.x] = (red[0] + red[1] + red[2] + red[3]) / fmaxf(nr, 1.f);
        }
    }
}

static int check_grounding(const char *what, int B, int T, int NR, size_t *lds)
{
    if (B <= 0 || T <= 0 || NR <= 0) return set_error(LOCOV_ERR_INVALID_ARG, "%s: bad shape B=%d T=%d NR=%d", what, B, T, NR);
    *lds = ((size_t)T * NR + 3 * (size_t)T + 8) * sizeof(float);
    if (*lds > 150 * 1024)
        return set_error(LOCOV_ERR_UNSUPPORTED, "%s: T*NR = %d does not fit the LDS tile", what, T * NR);
    return LOCOV_OK;
}

}  // namespace locov

using namespace locov;

extern "C" {

int locov_grounding_fwd(const float *S, int B, int T, int NR, const float *caption_mask, const float *region_mask,
                        float temperature, float *cost_w2r, float *cost_r2w, locov_stream_t stream)
{
    size_t lds;
    int rc = check_grounding("locov_grounding_fwd", B, T, NR, &lds);
    if (rc) return rc;
    LOCOV_REQUIRE(S && caption_mask && region_mask && cost_w2r && cost_r2w, "locov_grounding_fwd: null pointer");
    LOCOV_REQUIRE(temperature > 0.f, "locov_grounding_fwd: temperature must be > 0");
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void *>(grounding_kernel<false>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return set_error(LOCOV_ERR_LAUNCH, "locov_grounding_fwd: cannot raise the dynamic LDS limit to %zu bytes", (size_t)lds);
    hipLaunchKernelGGL(grounding_kernel<false>, dim3((unsigned)(B * B)), dim3(kGroundThreads), lds, as_stream(stream), S,
                       B, T, NR, caption_mask, region_mask, 1.f / temperature, cost_w2r, cost_r2w, nullptr, nullptr,
                       nullptr);
    return check_launch("locov_grounding_fwd");
}

int locov_grounding_bwd(const float *S, int B, int T, int NR, const float *caption_mask, const float *region_mask,
                        float temperature, const float *grad_w2r, const float *grad_r2w, float *grad_S,
                        locov_stream_t stream)
{
    size_t lds;
    int rc = check_grounding("locov_grounding_bwd", B, T, NR, &lds);
    if (rc) return rc;
    LOCOV_REQUIRE(S && caption_mask && region_mask && grad_w2r && grad_r2w && grad_S, "locov_grounding_bwd: null pointer");
    LOCOV_REQUIRE(temperature > 0.f, "locov_grounding_bwd: temperature must be > 0");
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void *>(grounding_kernel<true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return set_error(LOCOV_ERR_LAUNCH, "locov_grounding_bwd: cannot raise the dynamic LDS limit to %zu bytes", (size_t)lds);
    hipLaunchKernelGGL(grounding_kernel<true>, dim3((unsigned)(B * B)), dim3(kGroundThreads), lds, as_stream(stream), S,
                       B, T, NR, caption_mask, region_mask, 1.f / temperature, nullptr, nullptr, grad_w2r, grad_r2w,
                       grad_S);
    return check_launch("locov_grounding_bwd");
}

}  // extern "C"
