// Persistent, epilogue-pipelined form of the split-operand NT GEMM for PRE-SPLIT activations at K = 512 (gemm_split.hip,
// ASPLIT: same operand layouts, same LDS images, same three f16 MFMAs per product in the same order, same epilogue
// arithmetic) -- the shapes Res5 spends most of its launches on: the three Winograd-domain batched GEMMs [R,512]x[512,512]
// and the last 1x1 convolution of blocks 0-1 [49R,512]x[2048,512] + residual (roi_emb_heads.py:217-245 as GEMMs).
//
// Why.  In-kernel stamps of gemm_split_kernel (tools/attic/dbg_ktrace.py) put a K = 512 tile at 44 k cycles of K-loop, 10.5-12 k
// cycles of epilogue (64 KB of stores per workgroup, store-issue bound) and 1.3-1.6 k before the first DMA: a fifth of every
// workgroup slot is spent with that workgroup's MFMAs stopped, and what the K-loop itself could still give up is returned
// as a lower clock by the power-limited chip, while an epilogue saving is not.  A wave-specialised
// kernel that moved the epilogue to other waves lost more than it gained (one MFMA wave per SIMD cannot hide its LDS
// latency: tools/attic/experiments/).  This kernel keeps what works -- two independent 4-wave workgroups per CU, every wave doing
// the whole K-tile step -- and changes only WHEN the epilogue happens:
//   * workgroups are persistent (2 per CU) and walk the tile list in the hardware dispatcher's order; the operand stream is
//     continuous across tiles (the last K-tile step of tile t requests K-tile 0 of tile t+1);
//   * the MFMA operands are swapped (W rows as the A operand): the accumulator of a 16x16 block then holds, per lane, FOUR
//     CONSECUTIVE COLUMNS of one row -- 16 bytes that can be stored (and whose residual can be loaded) straight from / to
//     registers, no LDS re-layout, no barrier;
//   * a finished tile's accumulators move to a second register set, and its sixteen 16x16 blocks are finished and stored ONE
//     PER K-TILE STEP of the next tile (K = 512 has exactly 16 steps), each block's residual / scale / shift requested a step
//     ahead.  Per step and wave that is three 16-byte loads, ~12 vector-ALU instructions and one 16-byte store beside 48
//     MFMAs; the MFMAs never stop for an epilogue.
// Results are bit-identical to gemm_split_kernel<false, false, true> (tools/attic/dbg_asplit.py, tests/test_gpu_split_gemm.py).
#include "gemm_nt.h"

#include <type_traits>

namespace locov {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BN = 128, NT = 256, BK = 32, CH = 4;
constexpr int TM = 64, TN = 64, MB = 4, NB = 4;          // a wave's 64x64 sub-tile = 4x4 blocks of 16x16
constexpr int ROWB = 128;                                 // a K-tile row of either operand: 32 columns x (hi, lo) halves
constexpr int OPB = BM * ROWB, STAGEB = 2 * OPB;          // 16 KB per operand, 32 KB per stage
constexpr int KT = 16;                                    // K-tile steps per tile: K == 512

__device__ __forceinline__ int wswz(int row) { return (int)((0x75642031u >> (4 * ((row >> 1) & 7))) & 7u); }

__device__ __forceinline__ int xcd_remap(int bid, int nwg)
{
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (bid >> 3);
}

struct PPArgs {
    const float *A;
    int64_t lda;
    const float *B;
    float *C;
    int64_t ldc, M;
    int N;
    Epilogue epi;
    Batch bt;
    float out_scale;
    int total;           // tiles of the launch (all problems)
};

struct Tile {            // wave-uniform description of one output tile
    const char *a, *b;   // operand panels (K-tile 0)
    float *c;            // C + m0 * ldc of the tile's problem
    const float *res;    // residual + m0 * ldc, or null
    int64_t m0;
    int n0;
    unsigned nrec;       // bytes of the tile's valid rows: (rows_here * ldc) * 4
};

__global__ __launch_bounds__(NT, 2) void gemm_split_pp_kernel(PPArgs p)
{
    __shared__ u32x4 lds[2 * STAGEB / 16];
    char *const ldsb = reinterpret_cast<char *>(lds);
    constexpr int K = KT * BK;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave >> 1) * TM, wn = (wave & 1) * TN;
    const int l16 = lane & 15, kg = lane >> 4;
    const int P = gridDim.x;
    const int count = p.bt.count > 1 ? p.bt.count : 1;
    const int tiles_n = p.N / BN, per = p.total / count, tiles_m = per / tiles_n;

    // tile v of the launch in gemm_split_kernel's order (xcd_remap of the block index, N tiles in groups of 8, inside a group
    // M-tile outer / N-tile inner): what the hardware dispatcher would hand out next is what a persistent workgroup takes next
    auto locate = [&](int v, Tile &t) __attribute__((always_inline)) {
        int tile = xcd_remap(v, p.total);
        const int b = tile / per;
        tile -= b * per;
        constexpr int NG = 8;
        const int full = (tiles_n / NG) * NG, per_group = tiles_m * NG;
        int64_t m0;
        int n0;
        if (tiles_n <= NG) {
            m0 = (int64_t)(tile / tiles_n) * BM;
            n0 = (tile % tiles_n) * BN;
        } else if (tile < tiles_m * full) {
            const int g = tile / per_group, rem = tile - g * per_group;
            m0 = (int64_t)(rem / NG) * BM;
            n0 = (g * NG + rem % NG) * BN;
        } else {
            const int gs = tiles_n - full, rem = tile - tiles_m * full;
            m0 = (int64_t)(rem / gs) * BM;
            n0 = (full + rem % gs) * BN;
        }
        t.m0 = m0;
        t.n0 = n0;
        t.a = reinterpret_cast<const char *>(p.A + b * p.bt.sa + m0 * p.lda);
        t.b = reinterpret_cast<const char *>(p.B + b * p.bt.sb + (int64_t)n0 * K);
        t.c = p.C + b * p.bt.sc + m0 * p.ldc;
        t.res = p.epi.residual ? p.epi.residual + m0 * p.ldc : nullptr;
        const int64_t rows = p.M - m0 < BM ? p.M - m0 : BM;
        t.nrec = (unsigned)(rows * p.ldc * 4);
    };

    // DMA lane offsets (gemm_split.hip: instruction i of wave w brings rows (4w + i)*8 .. +7; lane l supplies row l/8 and fetches
    // the global chunk (l%8) ^ wswz(row)).  W's never change (N % 128 == 0); A's clamp rows past M and are rebuilt per tile.
    unsigned b_voff[CH], a_voff[CH];
#pragma unroll
    for (int i = 0; i < CH; i++) {
        const int row = (wave * CH + i) * 8 + (lane >> 3);
        b_voff[i] = (unsigned)(row * K * 4 + (((lane & 7) ^ wswz(row)) * 16));
    }
    auto set_a_voff = [&](int64_t m0) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < CH; i++) {
            const int row = (wave * CH + i) * 8 + (lane >> 3);
            const int64_t gm = m0 + row;
            a_voff[i] = (unsigned)((((gm < p.M ? gm : p.M - 1) - m0) * p.lda * 4) + (((lane & 7) ^ wswz(row)) * 16));
        }
    };
    auto dma = [&](const Tile &t, int kt, int stage) __attribute__((always_inline)) {
        const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(t.b + kt * (BK * 4)), 0, 0xffffffff, 0x00020000);
#pragma unroll
        for (int i = 0; i < CH; i++)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(
                rb, (__attribute__((address_space(3))) void *)(ldsb + stage * STAGEB + OPB + (wave * CH + i) * 8 * ROWB), 16, b_voff[i], 0, 0, 0);
        const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(t.a + kt * (BK * 4)), 0, 0xffffffff, 0x00020000);
#pragma unroll
        for (int i = 0; i < CH; i++)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(
                ra, (__attribute__((address_space(3))) void *)(ldsb + stage * STAGEB + (wave * CH + i) * 8 * ROWB), 16, a_voff[i], 0, 0, 0);
    };

    // fragments (gemm_split.hip): lane l holds row l%16 and the 8 halves of k-group l/16, [block][0 = hi, 1 = lo]; four
    // groups GA0 / GA1 = A row blocks {0,1} / {2,3}, GB0 / GB1 = W column blocks {0,1} / {2,3}
    f16x8 fa[MB][2], fb[NB][2];
    int bfo[2];
#pragma unroll
    for (int hl = 0; hl < 2; hl++) bfo[hl] = l16 * ROWB + (((2 * kg + hl) ^ wswz(l16)) * 16);
    auto rd_a = [&](int stage, int ga) __attribute__((always_inline)) {
        const char *As = ldsb + stage * STAGEB + (wm + ga * 32) * ROWB;
#pragma unroll
        for (int i = 0; i < 2; i++) {
            fa[2 * ga + i][0] = *reinterpret_cast<const f16x8 *>(As + i * 16 * ROWB + bfo[0]);
            fa[2 * ga + i][1] = *reinterpret_cast<const f16x8 *>(As + i * 16 * ROWB + bfo[1]);
        }
    };
    auto rd_b = [&](int stage, int gb) __attribute__((always_inline)) {
        const char *Bs = ldsb + stage * STAGEB + OPB + (wn + gb * 32) * ROWB;
#pragma unroll
        for (int j = 0; j < 2; j++) {
            fb[2 * gb + j][0] = *reinterpret_cast<const f16x8 *>(Bs + j * 16 * ROWB + bfo[0]);
            fb[2 * gb + j][1] = *reinterpret_cast<const f16x8 *>(Bs + j * 16 * ROWB + bfo[1]);
        }
    };
    // acc[i][j] holds block (row block i, column block j) TRANSPOSED: the W fragment is the MFMA's A operand, so lane l owns
    // row l%16 of the block and its columns 4*(l/16) .. +3 (C/D layout: col = lane&15, row = 4*(lane>>4) + reg, of W.A^T)
    f32x4 acc[MB][NB], prev[MB][NB];
#pragma unroll
    for (int i = 0; i < MB; i++)
#pragma unroll
        for (int j = 0; j < NB; j++) acc[i][j] = prev[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr int NQM = 12;
    auto quarter = [&](int ga, int gb, int p0, int p1) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < NQM; q++) {
            if (q < p0 || q >= p1) continue;
            const int t = q / 3, i = 2 * ga + t / 2, j = 2 * gb + t % 2, w = q % 3;   // hi.hi, (A hi)(W lo), (A lo)(W hi)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[j][w == 1 ? 1 : 0], fa[i][w == 2 ? 1 : 0], acc[i][j], 0, 0, 0);
        }
    };

    // ---- epilogue of the PREVIOUS tile, one 16x16 block per K-tile step ------------------------------------------------
    const bool relu = (p.epi.flags & LOCOV_EPI_RELU) != 0;
    const bool has_scale = p.epi.scale != nullptr;
    const __amdgpu_buffer_rsrc_t r_scale = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(has_scale ? p.epi.scale : p.A), 0, has_scale ? (unsigned)p.N * 4u : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_shift = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(p.epi.shift ? p.epi.shift : p.A), 0, p.epi.shift ? (unsigned)p.N * 4u : 0u, 0x00020000);
    struct Emit {                       // where the tile being drained lives (wave-uniform but for lane_off)
        float *c;
        const float *res;
        unsigned nrec;
        int n0;
    };
    // byte offsets inside the tile's rows: everything goes into the VECTOR offset (the scalar offset of a buffer instruction is
    // not range-checked, and rows past M must fall outside num_records)
    const unsigned lane_col = (unsigned)((wn + 4 * kg) * 4);                 // + tile column n0 * 4, + block column j * 64
    const unsigned lane_base = (unsigned)((wm + l16) * p.ldc * 4) + lane_col;   // + block row i: i * 16 * ldc * 4
    const unsigned row16 = 16u * (unsigned)p.ldc * 4u;
    f32x4 e_res = {0.f, 0.f, 0.f, 0.f}, e_sc = {0.f, 0.f, 0.f, 0.f}, e_sh = {0.f, 0.f, 0.f, 0.f};
    // the three loads block e of tile `t` needs (a null residual / scale / shift reads through an empty descriptor: zeros, no
    // memory access, and the wave's count of outstanding vector-memory operations stays what the barrier waits assume)
    auto emit_loads = [&](const Emit &t, int e) __attribute__((always_inline)) {
        const int i = e >> 2, j = e & 3;
        const __amdgpu_buffer_rsrc_t r_res =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(t.res ? t.res : p.A), 0, t.res ? t.nrec : 0u, 0x00020000);
        e_res = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                      r_res, lane_base + ((unsigned)(t.n0 * 4 + j * 64) + (unsigned)i * row16), 0, 2));
        e_sc = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_scale, lane_col + (unsigned)(t.n0 * 4 + j * 64), 0, 0));
        e_sh = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_shift, lane_col + (unsigned)(t.n0 * 4 + j * 64), 0, 0));
    };
    // finish block e (scale / shift / residual / ReLU) ...
    auto emit_value = [&](const Emit &t, int e) __attribute__((always_inline)) {
        const int i = e >> 2, j = e & 3;
        f32x4 sc = has_scale ? e_sc : f32x4{1.f, 1.f, 1.f, 1.f};
        sc *= p.out_scale;                                   // undo the operand scales
        f32x4 v = prev[i][j] * sc + e_sh;
        if (t.res) v += e_res;
        if (relu) {
            v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f);
            v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
        }
        return v;
    };
    // ... and store it.  Between the two, the NEXT block's loads are issued (into the registers emit_value has just read): the
    // wave's vector-memory operations complete in issue order, and a store's acknowledgement takes longer than a K-tile step
    // under this load -- with the store as the YOUNGEST of the three, the next step's wait for those loads does not include it,
    // and the first wait that does (the DMA wait of the next step, before its barrier) comes 1.6 steps later.  (Store first,
    // loads second: every step stood ~1 700 cycles waiting for its predecessor's store; 1.61 instead of 0.97 ms.)
    auto emit_store = [&](const Emit &t, int e, bool valid, const f32x4 &v) __attribute__((always_inline)) {
        const int i = e >> 2, j = e & 3;
        const __amdgpu_buffer_rsrc_t r_out = __builtin_amdgcn_make_buffer_rsrc(t.c, 0, valid ? t.nrec : 0u, 0x00020000);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r_out,
                                               lane_base + ((unsigned)(t.n0 * 4 + j * 64) + (unsigned)i * row16), 0, 2);   // aux 2 = nt: streamed once
    };

    // ---- one K-tile step E of the current tile (stage E & 1); quarters as in gemm_split.hip's tile_step ------------------
    Tile cur, nxt;
    Emit drain{p.C, nullptr, 0u, 0};
    bool drain_valid = false, has_next = false;
    auto step = [&](auto e_tag) __attribute__((always_inline)) {
        constexpr int E = decltype(e_tag)::value;
        constexpr int s = E & 1, x = s, y = s ^ 1;
        rd_b(s, y);
        rd_a(s, 1);
        __builtin_amdgcn_sched_barrier(0);
        // the previous tile's block E goes out FIRST, while the texture-address path is idle (every DMA has landed before the
        // barrier just passed): issued behind this step's DMA burst instead, the store queued behind the 8 KB pieces of all eight
        // waves of the CU and held its wave's issue -- MFMAs included -- for ~1 700 cycles per step
#ifndef LOCOV_PP_NOEMIT
        const f32x4 val = emit_value(drain, E);              // block E of the previous tile
        __builtin_amdgcn_sched_barrier(0);
        if (E + 1 < KT) {
            emit_loads(drain, E + 1);                        // block E+1's residual / scale / shift, a step ahead
        } else {                                             // block 0 of THIS tile, which the next tile's steps (or the flush) drain
            const Emit me{cur.c, cur.res, cur.nrec, cur.n0};
            emit_loads(me, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
#ifndef LOCOV_PP_NOSTORE
        emit_store(drain, E, drain_valid, val);
#endif
#endif
        if (E + 1 < KT) {
            dma(cur, E + 1, s ^ 1);
        } else if (has_next) {                               // the stream runs on into the next tile
            set_a_voff(nxt.m0);
            dma(nxt, 0, s ^ 1);
        }
        __builtin_amdgcn_sched_barrier(0);
        quarter(0, x, 0, NQM);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_sched_barrier(0);
        quarter(0, y, 0, NQM);
        __builtin_amdgcn_sched_barrier(0);
        quarter(1, y, 0, NQM);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0x0F70);                  // vmcnt(0): the DMA (the youngest operations) has landed
        __syncthreads();
        rd_a(s ^ 1, 0);
        rd_b(s ^ 1, y);
        __builtin_amdgcn_sched_barrier(0);
        quarter(1, x, 0, NQM);
        __builtin_amdgcn_sched_barrier(0);
    };

    int v = blockIdx.x;
    locate(v, cur);
    set_a_voff(cur.m0);
    dma(cur, 0, 0);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    rd_a(0, 0);
    rd_b(0, 0);
    {
        const Emit none{p.C, nullptr, 0u, 0};
        emit_loads(none, 0);                                 // (keeps the first step's counts like every other step's)
    }
    for (;;) {
        has_next = v + P < p.total;
        if (has_next) locate(v + P, nxt);
        step(std::integral_constant<int, 0>{});
        step(std::integral_constant<int, 1>{});
        step(std::integral_constant<int, 2>{});
        step(std::integral_constant<int, 3>{});
        step(std::integral_constant<int, 4>{});
        step(std::integral_constant<int, 5>{});
        step(std::integral_constant<int, 6>{});
        step(std::integral_constant<int, 7>{});
        step(std::integral_constant<int, 8>{});
        step(std::integral_constant<int, 9>{});
        step(std::integral_constant<int, 10>{});
        step(std::integral_constant<int, 11>{});
        step(std::integral_constant<int, 12>{});
        step(std::integral_constant<int, 13>{});
        step(std::integral_constant<int, 14>{});
        step(std::integral_constant<int, 15>{});
        // the tile is complete: its accumulators become the set the next tile's steps drain
#pragma unroll
        for (int i = 0; i < MB; i++)
#pragma unroll
            for (int j = 0; j < NB; j++) {
                prev[i][j] = acc[i][j];
                acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        drain = Emit{cur.c, cur.res, cur.nrec, cur.n0};
        drain_valid = true;
        if (!has_next) break;
        cur = nxt;
        v += P;
    }
    // flush: the last tile has no successor to hide under
#pragma unroll
    for (int e = 0; e < KT; e++) {
        const f32x4 val = emit_value(drain, e);
        if (e + 1 < KT) emit_loads(drain, e + 1);
        emit_store(drain, e, true, val);
    }
}

}  // namespace

bool gemm_split_pp_applicable(int64_t M, int N, int K, const Epilogue &epi, const Batch &bt, int64_t lda, int64_t ldc)
{
    if (K != KT * BK || N % BN != 0 || epi.mask != nullptr || M < BM) return false;
    if ((int64_t)BM * (lda > ldc ? lda : ldc) * 4 > 0x7fffffffLL || (int64_t)N * K * 4 > 0xffffffffLL) return false;
    return true;
}

int launch_gemm_split_pp(const float *A, int64_t lda, const void *Wsplit, float *C, int64_t ldc, int64_t M, int N, int K,
                         const Epilogue &epi, float a_scale, float w_scale, hipStream_t s, const char *what, const Batch &bt,
                         int cu_count)
{
    const int count = bt.count > 1 ? bt.count : 1;
    const int64_t tiles = ceil_div(M, BM) * (N / BN) * count;
    if (tiles > 0x7fffffffLL) return set_error(LOCOV_ERR_INVALID_ARG, "%s: problem too large", what);
    PPArgs p{A, lda, reinterpret_cast<const float *>(Wsplit), C, ldc, M, N, epi, bt, 1.f / (a_scale * w_scale), (int)tiles};
    if (count == 1) p.bt = Batch{1, 0, 0, 0};
    const int64_t slots = 2LL * cu_count;
    const unsigned grid = (unsigned)(tiles < slots ? tiles : slots);
    hipLaunchKernelGGL(gemm_split_pp_kernel, dim3(grid), dim3(NT), 0, s, p);
    return check_launch(what);
}

}  // namespace locov
