// NT GEMM  y[M,N] = epi(x[M,K] . W[N,K]^T)  on the gfx950 matrix cores, plus its
// implicit-GEMM form for 3x3 convolutions on channels-last tiles.
//
//   fp32 : v_mfma_f32_32x32x2_f32  -- exact fp32 products / fp32 accumulate (no TF32 on
//          gfx950); used for emb_pred (box_emb_head.py:206), bbox_pred (:196), the fp32
//          similarity GEMM cls_score (:211; parity gate 1e-4 on the logits) and the Res5
//          convolutions (roi_emb_heads.py:217-245) as GEMMs over [R*7*7, C] pixel rows.
//   bf16 : v_mfma_f32_32x32x16_bf16 -- bf16 operands / fp32 accumulate; the LVIS-size bank
//          similarity GEMM (BASELINE.json config 3).
//
// Both operands are K-contiguous (nn.Linear / conv weights are [out, in...]), so A and B tiles
// are staged the same way: 16-byte global loads -> registers -> ds_write_b128 into LDS rows
// padded to 144 B (stride 9 x 16 B: any 16 distinct rows hit 16 distinct 16-byte slots, so the
// ds_read_b128 fragment reads are conflict-free), register double-buffered so the next
// K-tile's loads are in flight under the MFMAs.  A wave owns a (BM/WM)x(BN/WN) sub-tile as
// 32x32 accumulators.  Lane l supplies row (l&31), 16 bytes at k-offset 16B*(l>>5):
//   fp32 -> 4 consecutive 32x32x2 MFMAs use .x .y .z .w (lane-half h covers k = 4h+j),
//   bf16 -> one 32x32x16 MFMA (lane-half h covers k = 8h..8h+7)          [guide section 3].
//
// CONV3 (3x3, pad 1, stride 1): A is the [R*H*W, Cin] pixel matrix of R independent HxW
// tiles; GEMM column k = tap*Cin + c reads pixel (y+dy, x+dx) of the same tile, zero outside
// it.  A K-tile never straddles a tap (Cin % BK == 0), so (dy,dx) is uniform per tile and the
// gather is just a row offset plus a per-row validity mask.
#include "gemm_nt.h"

namespace locov {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <typename T>
struct Frag;  // one 16-byte fragment per lane
template <>
struct Frag<float> {
    typedef f32x4 type;
    static constexpr int kPer16B = 4;
};
template <>
struct Frag<__bf16> {
    typedef bf16x8 type;
    static constexpr int kPer16B = 8;
};

__device__ __forceinline__ void mma_step(const f32x4 &a, const f32x4 &b, f32x16 &acc)
{
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], b[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], b[3], acc, 0, 0, 0);
}
__device__ __forceinline__ void mma_step(const bf16x8 &a, const bf16x8 &b, f32x16 &acc)
{
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
}

__device__ __forceinline__ void store_out(float *p, float v) { *p = v; }
__device__ __forceinline__ void store_out(__bf16 *p, float v) { *p = (__bf16)v; }

// XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs, so give each XCD a
// contiguous run of tiles (bijective for any tile count; guide 5, "XCD swizzle must be
// bijective").  Consecutive tiles share the same A row panel -> it stays in that XCD's L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg)
{
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (bid >> 3);
}

template <typename T, typename TOut, int BM, int BN, int WM, int WN, bool CONV3>
__global__ __launch_bounds__(64 * WM *WN) void gemm_nt_kernel(const T *__restrict__ A, int64_t lda,
                                                               const T *__restrict__ B, int64_t ldb,
                                                               TOut *__restrict__ Cout, int64_t ldc, int64_t M,
                                                               int N, int K, Epilogue epi, ConvGeom cg)
{
    typedef typename Frag<T>::type frag_t;
    constexpr int E = Frag<T>::kPer16B;    // elements per 16 B
    constexpr int BK16 = 8;                // 16-byte chunks per tile row: BK = 8*E (32 f32 / 64 bf16)
    constexpr int BK = BK16 * E;
    constexpr int LDS16 = BK16 + 1;        // padded row length in 16-byte units (144 B)
    constexpr int NT = 64 * WM * WN;
    constexpr int TM = BM / WM, TN = BN / WN;
    constexpr int MI = TM / 32, NI = TN / 32;
    constexpr int A_CH = BM * BK16 / NT;   // 16-byte chunks per thread per tile
    constexpr int B_CH = BN * BK16 / NT;
    static_assert(BM * BK16 % NT == 0 && BN * BK16 % NT == 0, "tile must divide evenly over threads");

    __shared__ frag_t lds[(BM + BN) * LDS16];
    frag_t *As = lds, *Bs = lds + BM * LDS16;

    const int tiles_n = (N + BN - 1) / BN;
    const int nwg = gridDim.x;
    const int tile = xcd_remap(blockIdx.x, nwg);
    const int64_t m0 = (int64_t)(tile / tiles_n) * BM;
    const int n0 = (tile % tiles_n) * BN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave / WN) * TM, wn = (wave % WN) * TN;

    // per-thread A rows are the same for every K-tile: decode their tile coordinates once
    int a_y[A_CH], a_x[A_CH];
    if (CONV3) {
#pragma unroll
        for (int i = 0; i < A_CH; i++) {
            const int64_t gm = m0 + (tid + i * NT) / BK16;
            const int rem = (int)(gm % (cg.H * cg.W));
            a_y[i] = rem / cg.W;
            a_x[i] = rem - a_y[i] * cg.W;
        }
    }

    frag_t ra[A_CH], rb[B_CH];
    const frag_t zero = {};

    auto load_tiles = [&](int k0) {
        int dy = 0, dx = 0, kc = k0;
        if (CONV3) {
            const int tap = k0 / cg.Cin;
            kc = k0 - tap * cg.Cin;
            dy = tap / 3 - 1;
            dx = tap - (tap / 3) * 3 - 1;
        }
#pragma unroll
        for (int i = 0; i < A_CH; i++) {
            const int idx = tid + i * NT, row = idx / BK16, ch = idx % BK16;
            const int64_t gm = m0 + row;
            if (CONV3) {
                const bool ok = gm < M && (unsigned)(a_y[i] + dy) < (unsigned)cg.H &&
                                (unsigned)(a_x[i] + dx) < (unsigned)cg.W;
                ra[i] = ok ? *reinterpret_cast<const frag_t *>(A + (gm + dy * cg.W + dx) * lda + kc + ch * E) : zero;
            } else {
                const int gk = k0 + ch * E;
                ra[i] = (gm < M && gk < K) ? *reinterpret_cast<const frag_t *>(A + gm * lda + gk) : zero;
            }
        }
#pragma unroll
        for (int i = 0; i < B_CH; i++) {
            const int idx = tid + i * NT, row = idx / BK16, ch = idx % BK16;
            const int gn = n0 + row;
            const int gk = k0 + ch * E;
            rb[i] = (gn < N && gk < K) ? *reinterpret_cast<const frag_t *>(B + (int64_t)gn * ldb + gk) : zero;
        }
    };
    auto store_tiles = [&]() {
#pragma unroll
        for (int i = 0; i < A_CH; i++) {
            const int idx = tid + i * NT;
            As[(idx / BK16) * LDS16 + idx % BK16] = ra[i];
        }
#pragma unroll
        for (int i = 0; i < B_CH; i++) {
            const int idx = tid + i * NT;
            Bs[(idx / BK16) * LDS16 + idx % BK16] = rb[i];
        }
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; i++)
#pragma unroll
        for (int j = 0; j < NI; j++) acc[i][j] = f32x16{};

    load_tiles(0);
    store_tiles();
    __syncthreads();

    const int frow = lane & 31, fch = lane >> 5;
    for (int k0 = 0; k0 < K; k0 += BK) {
        const bool more = k0 + BK < K;
        if (more) load_tiles(k0 + BK);  // in flight under the MFMAs below
#pragma unroll
        for (int kk = 0; kk < BK16; kk += 2) {
            frag_t a[MI], b[NI];
#pragma unroll
            for (int i = 0; i < MI; i++) a[i] = As[(wm + i * 32 + frow) * LDS16 + kk + fch];
#pragma unroll
            for (int j = 0; j < NI; j++) b[j] = Bs[(wn + j * 32 + frow) * LDS16 + kk + fch];
#pragma unroll
            for (int i = 0; i < MI; i++)
#pragma unroll
                for (int j = 0; j < NI; j++) mma_step(a[i], b[j], acc[i][j]);
        }
        __syncthreads();
        if (more) {
            store_tiles();
            __syncthreads();
        }
    }

    // epilogue: C/D layout of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
    for (int j = 0; j < NI; j++) {
        const int n = n0 + wn + j * 32 + (lane & 31);
        if (n >= N) continue;
        const float sc = epi.scale ? epi.scale[n] : 1.f;
        const float sh = epi.shift ? epi.shift[n] : 0.f;
#pragma unroll
        for (int i = 0; i < MI; i++) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int64_t m = m0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (m >= M) continue;
                float v = acc[i][j][r];
                if (epi.scale) v *= sc;
                v += sh;
                if (epi.residual) v += epi.residual[m * ldc + n];
                if (epi.flags & LOCOV_EPI_RELU) v = fmaxf(v, 0.f);
                store_out(Cout + m * ldc + n, v);
            }
        }
    }
}

template <typename T, typename TOut, int BM, int BN, int WM, int WN>
static int launch_cfg(const T *A, int64_t lda, const T *B, int64_t ldb, TOut *C, int64_t ldc, int64_t M, int N,
                      int K, const Epilogue &epi, const ConvGeom &cg, hipStream_t s, const char *what)
{
    const int64_t tiles = ceil_div(M, BM) * ceil_div(N, BN);
    if (tiles > 0x7fffffffLL) return set_error(LOCOV_ERR_INVALID_ARG, "%s: problem too large", what);
    if (cg.H > 0)
        hipLaunchKernelGGL((gemm_nt_kernel<T, TOut, BM, BN, WM, WN, true>), dim3((unsigned)tiles), dim3(64 * WM * WN),
                           0, s, A, lda, B, ldb, C, ldc, M, N, K, epi, cg);
    else
        hipLaunchKernelGGL((gemm_nt_kernel<T, TOut, BM, BN, WM, WN, false>), dim3((unsigned)tiles),
                           dim3(64 * WM * WN), 0, s, A, lda, B, ldb, C, ldc, M, N, K, epi, cg);
    return check_launch(what);
}

template <typename T, typename TOut>
int launch_gemm_nt(const T *A, int64_t lda, const T *B, int64_t ldb, TOut *C, int64_t ldc, int64_t M, int N, int K,
                   const Epilogue &epi, hipStream_t s, const char *what, const ConvGeom &cg)
{
    if (N <= 32) return launch_cfg<T, TOut, 128, 32, 4, 1>(A, lda, B, ldb, C, ldc, M, N, K, epi, cg, s, what);
    if (N <= 64 || (N <= 192 && N % 128 != 0 && N % 128 <= 64))
        return launch_cfg<T, TOut, 128, 64, 4, 1>(A, lda, B, ldb, C, ldc, M, N, K, epi, cg, s, what);
    return launch_cfg<T, TOut, 128, 128, 2, 2>(A, lda, B, ldb, C, ldc, M, N, K, epi, cg, s, what);
}

#define LOCOV_INST(T, TOut)                                                                                        \
    template int launch_gemm_nt<T, TOut>(const T *, int64_t, const T *, int64_t, TOut *, int64_t, int64_t, int, int, \
                                         const Epilogue &, hipStream_t, const char *, const ConvGeom &);
LOCOV_INST(float, float)
LOCOV_INST(__bf16, float)
LOCOV_INST(__bf16, __bf16)
#undef LOCOV_INST

// conv weight [N, Cin, 3, 3] -> GEMM operand [N, 9*Cin] with k = (ky*3+kx)*Cin + c
template <typename TOut>
__global__ __launch_bounds__(256) void pack_conv3x3_kernel(const float *__restrict__ w, int N, int Cin,
                                                           TOut *__restrict__ out)
{
    const int64_t total = (int64_t)N * Cin * 9;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % Cin);
        const int tap = (int)((i / Cin) % 9);
        const int64_t n = i / ((int64_t)Cin * 9);
        out[i] = (TOut)w[(n * Cin + c) * 9 + tap];
    }
}

// FrozenBatchNorm2d fold: scale = weight * rsqrt(var + eps), shift = bias - mean * scale
__global__ __launch_bounds__(256) void bn_fold_kernel(const float *__restrict__ weight, const float *__restrict__ bias,
                                                      const float *__restrict__ mean, const float *__restrict__ var,
                                                      float eps, int C, float *__restrict__ scale,
                                                      float *__restrict__ shift)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float s = weight[c] * (1.0f / sqrtf(var[c] + eps));
    scale[c] = s;
    shift[c] = bias[c] - mean[c] * s;
}

}  // namespace locov

using namespace locov;

extern "C" {

int locov_gemm_nt_f32(const float *x, int64_t lda, const float *W, const float *scale, const float *shift,
                      const float *residual, float *y, int64_t ldc, int64_t M, int N, int K, unsigned flags,
                      locov_stream_t stream)
{
    LOCOV_REQUIRE(M >= 0 && N > 0 && K > 0, "locov_gemm_nt_f32: bad shape M=%lld N=%d K=%d", (long long)M, N, K);
    if (M == 0) return LOCOV_OK;
    LOCOV_REQUIRE(x && W && y, "locov_gemm_nt_f32: null pointer");
    LOCOV_REQUIRE(K % 4 == 0 && lda % 4 == 0, "locov_gemm_nt_f32: K and lda must be multiples of 4 (got %d, %lld)",
                  K, (long long)lda);
    LOCOV_REQUIRE(lda >= K && ldc >= N, "locov_gemm_nt_f32: lda < K or ldc < N");
    LOCOV_REQUIRE((uintptr_t)x % 16 == 0 && (uintptr_t)W % 16 == 0, "locov_gemm_nt_f32: x / W must be 16-byte aligned");
    Epilogue epi{scale, shift, residual, flags};
    return launch_gemm_nt<float, float>(x, lda, W, (int64_t)K, y, ldc, M, N, K, epi, as_stream(stream),
                                        "locov_gemm_nt_f32");
}

int locov_conv3x3_nhwc_f32(const float *x, int64_t R, int H, int W, int Cin, const float *w_packed,
                           const float *scale, const float *shift, const float *residual, float *y, int N,
                           unsigned flags, locov_stream_t stream)
{
    LOCOV_REQUIRE(R >= 0 && H > 0 && W > 0 && Cin > 0 && N > 0, "locov_conv3x3_nhwc_f32: bad shape");
    if (R == 0) return LOCOV_OK;
    LOCOV_REQUIRE(x && w_packed && y, "locov_conv3x3_nhwc_f32: null pointer");
    LOCOV_REQUIRE(Cin % 32 == 0, "locov_conv3x3_nhwc_f32: Cin must be a multiple of 32 (got %d)", Cin);
    LOCOV_REQUIRE((uintptr_t)x % 16 == 0 && (uintptr_t)w_packed % 16 == 0, "locov_conv3x3_nhwc_f32: misaligned pointer");
    Epilogue epi{scale, shift, residual, flags};
    ConvGeom cg{H, W, Cin};
    return launch_gemm_nt<float, float>(x, (int64_t)Cin, w_packed, (int64_t)9 * Cin, y, (int64_t)N, R * H * W, N,
                                        9 * Cin, epi, as_stream(stream), "locov_conv3x3_nhwc_f32", cg);
}

int locov_pack_conv3x3_weight(const float *w, int N, int Cin, void *out, int out_dtype, locov_stream_t stream)
{
    LOCOV_REQUIRE(N > 0 && Cin > 0, "locov_pack_conv3x3_weight: bad shape");
    LOCOV_REQUIRE(w && out, "locov_pack_conv3x3_weight: null pointer");
    LOCOV_REQUIRE(out_dtype == LOCOV_F32 || out_dtype == LOCOV_BF16, "locov_pack_conv3x3_weight: bad dtype");
    const int64_t total = (int64_t)N * Cin * 9;
    const int grid = (int)(ceil_div(total, 256) < 4096 ? ceil_div(total, 256) : 4096);
    if (out_dtype == LOCOV_F32)
        hipLaunchKernelGGL(pack_conv3x3_kernel<float>, dim3(grid), dim3(256), 0, as_stream(stream), w, N, Cin, (float *)out);
    else
        hipLaunchKernelGGL(pack_conv3x3_kernel<__bf16>, dim3(grid), dim3(256), 0, as_stream(stream), w, N, Cin,
                           (__bf16 *)out);
    return check_launch("locov_pack_conv3x3_weight");
}

int locov_frozen_bn_fold(const float *weight, const float *bias, const float *running_mean,
                         const float *running_var, float eps, int C, float *scale, float *shift,
                         locov_stream_t stream)
{
    LOCOV_REQUIRE(C > 0, "locov_frozen_bn_fold: C <= 0");
    LOCOV_REQUIRE(weight && bias && running_mean && running_var && scale && shift, "locov_frozen_bn_fold: null pointer");
    hipLaunchKernelGGL(bn_fold_kernel, dim3((unsigned)ceil_div(C, 256)), dim3(256), 0, as_stream(stream), weight, bias,
                       running_mean, running_var, eps, C, scale, shift);
    return check_launch("locov_frozen_bn_fold");
}

int locov_sim_gemm_bf16(const uint16_t *emb, const uint16_t *bank, int64_t R, int D, int K1, float *logits,
                        int64_t ldc, locov_stream_t stream)
{
    LOCOV_REQUIRE(R >= 0 && D > 0 && K1 > 0, "locov_sim_gemm_bf16: bad shape R=%lld D=%d K1=%d", (long long)R, D, K1);
    if (R == 0) return LOCOV_OK;
    LOCOV_REQUIRE(emb && bank && logits, "locov_sim_gemm_bf16: null pointer");
    LOCOV_REQUIRE(D % 8 == 0, "locov_sim_gemm_bf16: D must be a multiple of 8 (got %d)", D);
    LOCOV_REQUIRE(ldc >= K1, "locov_sim_gemm_bf16: ldc < K1");
    LOCOV_REQUIRE((uintptr_t)emb % 16 == 0 && (uintptr_t)bank % 16 == 0,
                  "locov_sim_gemm_bf16: emb / bank must be 16-byte aligned");
    Epilogue epi{nullptr, nullptr, nullptr, 0u};
    return launch_gemm_nt<__bf16, float>(reinterpret_cast<const __bf16 *>(emb), (int64_t)D,
                                         reinterpret_cast<const __bf16 *>(bank), (int64_t)D, logits, ldc, R, K1, D,
                                         epi, as_stream(stream), "locov_sim_gemm_bf16");
}

}  // extern "C"
