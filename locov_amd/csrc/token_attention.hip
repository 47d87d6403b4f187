// Multi-token class scoring of the grounding predictor:
// ovr/modeling/roi_heads/box_emb_grounding_head.py:60-256 GroundingModule.forward after its
// token_score Linear (that GEMM is locov_gemm_nt_f32 on the concatenated token bank).
//
// Per region r and class k with n_k real tokens (the reference pads every class to Tmax slots):
//   s_t = sim[r, off_k + t] / temperature                      (cosine: NaN -> 0 first)
//   d_t = -s_t (dot)  |  (1 - sim) / temperature (cosine)
//   masked s_t = s_t for t < n_k, else gmin - 100               (gmin = min of the padded tensor)
//   a = softmax_t(masked s)  |  one_hot(argmax_t masked s)      over all Tmax slots
//   att_t = a_t * [t < n_k];   score = -sum_t att_t * d_t
// A class without tokens (the background row) keeps the one slot the reference's in-place
// `split_sizes[split_sizes == 0] = 1` gives it, but its mask row is zero: attention 0, score -0.
// One lane per (r, k); Tmax is small (<= 32), everything stays in registers.
#include "common.h"

namespace locov {

constexpr int kTokMax = 32;

__global__ __launch_bounds__(256) void token_attention_kernel(const float *__restrict__ sim, int64_t R, int Ttot,
                                                              const int *__restrict__ tok_off,
                                                              const int *__restrict__ num_tok, int K1, int Tmax,
                                                              float temp, int cosine, int hardmax,
                                                              const float *__restrict__ gmin,
                                                              float *__restrict__ scores, float *__restrict__ att)
{
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= R * K1) return;
    const int64_t r = idx / K1;
    const int k = (int)(idx - r * K1);
    const int n = num_tok[k], off = tok_off[k];
    const float fill = gmin[0] - 100.0f;
    const float *row = sim + r * Ttot + off;
    float s[kTokMax], d[kTokMax];
    float mx = -INFINITY;
    int arg = 0;
#pragma unroll
    for (int t = 0; t < kTokMax; t++) {
        if (t >= Tmax) break;
        float v = t < n ? row[t] : 0.f;
        if (cosine && v != v) v = 0.f;
        d[t] = (cosine ? (1.0f - v) : -v) / temp;
        s[t] = t < n ? v / temp : fill;
        if (s[t] > mx) {                                      // first maximum, as torch.argmax
            mx = s[t];
            arg = t;
        }
    }
    float denom = 0.f;
    if (!hardmax) {
#pragma unroll
        for (int t = 0; t < kTokMax; t++) {
            if (t >= Tmax) break;
            s[t] = expf(s[t] - mx);
            denom += s[t];
        }
    }
    float dist = 0.f;
#pragma unroll
    for (int t = 0; t < kTokMax; t++) {
        if (t >= Tmax) break;
        float a = hardmax ? (t == arg ? 1.f : 0.f) : s[t] / denom;
        a = t < n ? a : 0.f;
        dist += a * d[t];
        if (att) att[idx * Tmax + t] = a;
    }
    scores[idx] = -dist;
}

}  // namespace locov

using namespace locov;

extern "C" int locov_token_attention_fwd(const float *sim, int64_t R, int Ttot, const int *tok_off, const int *num_tok,
                                         int K1, int Tmax, float temperature, int cosine, int hardmax,
                                         const float *gmin, float *scores, float *att, locov_stream_t stream)
{
    LOCOV_REQUIRE(R >= 0 && Ttot > 0 && K1 > 0, "locov_token_attention_fwd: bad shape R=%lld Ttot=%d K1=%d", (long long)R, Ttot,
                  K1);
    LOCOV_REQUIRE(Tmax > 0 && Tmax <= kTokMax, "locov_token_attention_fwd: Tmax must be in [1, %d] (got %d)", kTokMax, Tmax);
    LOCOV_REQUIRE(temperature > 0.f, "locov_token_attention_fwd: temperature must be > 0");
    if (R == 0) return LOCOV_OK;
    LOCOV_REQUIRE(sim && tok_off && num_tok && gmin && scores, "locov_token_attention_fwd: null pointer");
    const int64_t total = R * K1;
    hipLaunchKernelGGL(token_attention_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, as_stream(stream), sim, R,
                       Ttot, tok_off, num_tok, K1, Tmax, temperature, cosine, hardmax, gmin, scores, att);
    return check_launch("locov_token_attention_fwd");
}
