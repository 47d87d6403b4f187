// Wave-specialised, persistent form of the split-operand NT GEMM for PRE-SPLIT activations (gemm_split.hip, ASPLIT: same
// arithmetic, same LDS images, same MFMA sequence per K-tile -- the results are bit-identical), for the large plain / batched
// launches of the Res5 stage whose A operand leaves its producer in the split layout (the Winograd-domain batched GEMMs, the
// 1x1 convolutions behind the Winograd output transform; roi_emb_heads.py:217-245 as GEMMs).
//
// Why.  In gemm_split_kernel every wave does everything: staging, fragment reads, 48 MFMAs per K-tile and, at the end of the
// tile, the epilogue (residual reads, 64 KB of stores).  The two workgroups of a CU run in lockstep, so the memory-bound
// epilogue does not overlap the other workgroup's MFMAs: for the K = 512 shapes (16 K-tiles per output tile) the launch time is
// T_mem + T_mfma, not max(T_mem, T_mfma).  A first wave-specialised kernel that still CONVERTED A in
// the kernel was slower than the plain one: measured with in-kernel cycle stamps, its four staging waves needed ~1 500 cycles
// per K-tile (4 LDS DMAs 330, 4 buffer loads 170, the split of 4 chunks + 8 LDS stores 560-680, waiting for A 200-400)
// against 768 cycles of MFMA work -- the plain kernel is bound by the same staging instruction stream, spread over eight waves
// (tools/attic/experiments/).  With A pre-split the staging of a K-tile is eight LDS DMAs per wave and nothing else, and the roles
// separate cleanly.
//
// ONE 768-thread workgroup per CU stays resident and walks over its tiles; its 12 waves have fixed roles (waves w, w+4, w+8
// share a SIMD, waves 0-3 sit on four different SIMDs -- tools/probe/simd_probe.hip):
//   waves 0-3   MFMA      one per SIMD, a 64x64 sub-tile each: fragment reads + MFMAs, nothing else; at the end of a tile
//                         the accumulators go to an LDS buffer and the next tile starts at once
//   waves 4-7   staging   the operand stream, continuous across tile boundaries: both operands wait two K-tiles ahead in
//                         registers (one buffer load + one ds_write_b128 per 16-byte chunk, no conversion)
//   waves 8-11  epilogue  drain tile t-1's accumulators from the LDS buffer DURING tile t's K-loop: 8 rows per step, the
//                         residual rows requested a whole tile ahead (64 KB per CU in flight), stores never waited for
// One s_barrier per K-tile (all 12 waves) hands the LDS stages over.
//
// LDS: two stages per operand (16 KB each) + the 66 KB accumulator buffer = 130 KB.  Used by launch_gemm_split for
// LOCOV_GEMM_A_SPLIT launches with K % 64 == 0, K >= 512, N % 128 == 0, no mask, at least two tiles per CU.
#include "gemm_nt.h"

#include <type_traits>

namespace locov {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int WROWB = 128;
constexpr int ASTB = BM * WROWB;                        // one A stage: 16 384 B (pre-split A: unpadded, swizzled rows like W)
constexpr int WSTB = BN * WROWB;                        // one W stage: 16 384 B
constexpr int WBASE = 2 * ASTB;                         // two stages each; both operands wait two K-tiles ahead in REGISTERS
constexpr int EPBASE = WBASE + 2 * WSTB;                // 65 536
constexpr int EPS = BN + 4;                             // floats per row of the accumulator buffer
constexpr int EPB = BM * EPS * 4;                       // 67 584   (total 133 120 B)
constexpr int NTHREADS = 768;                          // 12 waves: 4 MFMA + 4 staging + 4 epilogue (three per SIMD)
constexpr int NSTEPS = BM / 8;                          // epilogue steps per tile (8 rows each)

__device__ __forceinline__ int wswz(int row) { return (int)((0x75642031u >> (4 * ((row >> 1) & 7))) & 7u); }

__device__ __forceinline__ int xcd_remap(int bid, int nwg)
{
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (bid >> 3);
}

__device__ __forceinline__ void split4(const f32x4 &x, float s, u32x2 &hi, u32x2 &lo)
{
#pragma unroll
    for (int e = 0; e < 2; e++) {
        unsigned h, l;
        asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h) : "v"(x[2 * e]), "s"(s));
        asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h) : "v"(x[2 * e + 1]), "s"(s));
        asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(l) : "v"(x[2 * e]), "s"(s), "v"(h));
        asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(x[2 * e + 1]), "s"(s), "v"(h));
        hi[e] = h;
        lo[e] = l;
    }
}

// s_waitcnt immediate (gfx9 encoding): vmcnt = vm (6 bits, split), expcnt / lgkmcnt = their "no wait" maxima unless lgkm0
constexpr int waitcnt_imm(int vm, bool lgkm0) { return (vm & 15) | ((vm >> 4) << 14) | 0x0070 | (lgkm0 ? 0 : 0x0F00); }

#ifdef LOCOV_WS_TRACE
__device__ unsigned long long g_ws_dbg[16];
#define LOCOV_WS_BARRIER()                                                    \
    do {                                                                      \
        const unsigned long long t0_ = __builtin_readcyclecounter();          \
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");       \
        ws_wait_ += __builtin_readcyclecounter() - t0_;                       \
    } while (0)
#else
#define LOCOV_WS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#endif

struct TileGeom {
    int tiles_n, tiles_m, per_batch, NG;
};

// tile index -> (batch, m0, n0): the order of gemm_split_kernel (N tiles in groups of NG, inside a group M-tile outer)
__device__ __forceinline__ void tile_coords(const TileGeom &g, int tile, int &b, int64_t &m0, int &n0)
{
    b = tile / g.per_batch;
    tile -= b * g.per_batch;
    const int full = (g.tiles_n / g.NG) * g.NG, per_group = g.tiles_m * g.NG;
    if (g.tiles_n <= g.NG) {
        m0 = (int64_t)(tile / g.tiles_n) * BM;
        n0 = (tile % g.tiles_n) * BN;
    } else if (tile < g.tiles_m * full) {
        const int gi = tile / per_group, rem = tile - gi * per_group;
        m0 = (int64_t)(rem / g.NG) * BM;
        n0 = (gi * g.NG + rem % g.NG) * BN;
    } else {
        const int gs = g.tiles_n - full, rem = tile - g.tiles_m * full;
        m0 = (int64_t)(rem / gs) * BM;
        n0 = (full + rem % gs) * BN;
    }
}

}  // namespace

// SPS = epilogue steps per K-tile slot (2 for 16 K-tiles per tile, 1 from 17 on); HASRES = the epilogue adds a residual
template <int SPS, bool HASRES>
__global__ __launch_bounds__(NTHREADS, 3) void gemm_split_ws_kernel(const float *__restrict__ A, int64_t lda,
                                                                    const float *__restrict__ B, float *__restrict__ Cout,
                                                                    int64_t ldc, int64_t M, int N, int K, Epilogue epi, Batch bt,
                                                                    float a_scale, float out_scale, unsigned *overflow,
                                                                    int total_tiles)
{
    extern __shared__ __attribute__((aligned(16))) char ldsb[];
    float *const epbuf = reinterpret_cast<float *>(ldsb + EPBASE);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef LOCOV_WS_TRACE
    unsigned long long ws_wait_ = 0;
    const unsigned long long ws_t0_ = __builtin_readcyclecounter();
    auto ws_report = [&](int role) {
        if (blockIdx.x == 8 && lane == 0 && (wave & 3) == 0) {
            g_ws_dbg[role * 2] = ws_wait_;
            g_ws_dbg[role * 2 + 1] = __builtin_readcyclecounter() - ws_t0_;
        }
    };
#else
    auto ws_report = [&](int) {};
#endif
    const int KT = K / BK;
    TileGeom tg;
    tg.tiles_n = N / BN;
    tg.tiles_m = (int)((M + BM - 1) / BM);
    tg.per_batch = tg.tiles_m * tg.tiles_n;
    tg.NG = (int64_t)K * 4 * BN * 8 <= (2 << 20) ? 8 : 4;
    // my tiles: xcd_remap(blockIdx.x + k * gridDim.x), k = 0 .. T-1   (gridDim.x is a multiple of 8)
    const int G = gridDim.x;
    const int T = (total_tiles - (int)blockIdx.x + G - 1) / G;
    const int Q = T * KT;                                     // K-tiles this workgroup walks through
    auto my_tile = [&](int k) { return xcd_remap((int)blockIdx.x + k * G, total_tiles); };

    if (wave < 4) {
        // ------------------------------------------------------------------ MFMA waves
        __builtin_amdgcn_s_setprio(3);
        const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
        const int l16 = lane & 15, kg = lane >> 4;
        int bfo[2];
#pragma unroll
        for (int hl = 0; hl < 2; hl++) bfo[hl] = l16 * WROWB + (((2 * kg + hl) ^ wswz(l16)) * 16);
        f16x8 fa[4][2], fb[4][2];
        f32x4 acc[4][4];
        auto rd_a = [&](int astage, int ga) {                     // stage = q & 1 (a literal: the K-loop is unrolled by two)
            const char *As = ldsb + astage * ASTB + (wm + ga * 32) * WROWB;
#pragma unroll
            for (int i = 0; i < 2; i++) {
                fa[2 * ga + i][0] = *reinterpret_cast<const f16x8 *>(As + i * 16 * WROWB + bfo[0]);
                fa[2 * ga + i][1] = *reinterpret_cast<const f16x8 *>(As + i * 16 * WROWB + bfo[1]);
            }
        };
        auto rd_b = [&](int wstage, int gb) {                     // wstage = q & 1 (a literal: the K-loop is unrolled by two)
            const char *Bs = ldsb + WBASE + wstage * WSTB + (wn + gb * 32) * WROWB;
#pragma unroll
            for (int j = 0; j < 2; j++) {
                fb[2 * gb + j][0] = *reinterpret_cast<const f16x8 *>(Bs + j * 16 * WROWB + bfo[0]);
                fb[2 * gb + j][1] = *reinterpret_cast<const f16x8 *>(Bs + j * 16 * WROWB + bfo[1]);
            }
        };
        // 12 MFMAs of a quarter, product-major (hi.hi of the four blocks, then hi.lo, then lo.hi): with ONE wave per SIMD
        // feeding the pipe, consecutive MFMAs must not chain on the same accumulator (the per-block sum is the same three
        // products in the same order as gemm_split_kernel: bit-identical)
        auto quarter = [&](int ga, int gb) {
#pragma unroll
            for (int w = 0; w < 3; w++)
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    const int i = 2 * ga + t / 2, j = 2 * gb + t % 2;
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[i][w == 2 ? 1 : 0], fb[j][w == 1 ? 1 : 0], acc[i][j], 0, 0, 0);
                }
        };
        // one K-tile from LDS stage s (= its parity): (GA0,GBx) (GA0,GBy) (GA1,GBy) | barrier | (GA1,GBx), x = s, y = 1 - s;
        // the fragments of the next K-tile's first quarter are read right behind the barrier (gemm_split.hip)
        auto ktile = [&](const int s, const bool has_next) __attribute__((always_inline)) {
            const int x = s, y = s ^ 1;
            rd_b(s, y);
            rd_a(s, 1);
            quarter(0, x);
            quarter(0, y);
            quarter(1, y);
            __builtin_amdgcn_sched_barrier(0);
            LOCOV_WS_BARRIER();
            if (has_next) {
                rd_a(s ^ 1, 0);
                rd_b(s ^ 1, y);
            }
            __builtin_amdgcn_sched_barrier(0);
            quarter(1, x);
            __builtin_amdgcn_sched_barrier(0);
        };
        LOCOV_WS_BARRIER();                                   // barrier P: K-tile 0 is in A stage 0 / W stage 0
        rd_a(0, 0);
        rd_b(0, 0);
        int q = 0;
        for (int t = 0; t < T; t++) {
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            for (int kt = 0; kt < KT; kt += 2, q += 2) {
                ktile(0, true);
                ktile(1, q + 2 < Q);
            }
            // hand the finished accumulators to the epilogue waves (they read them behind the next barrier; the buffer is
            // free: the previous tile's last step was read before the barrier inside this tile's last K-tile)
            // C/D layout of the 16x16 MFMA: col = lane & 15, row = 4 * (lane >> 4) + reg
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++)
#pragma unroll
                    for (int r = 0; r < 4; r++) epbuf[(wm + i * 16 + 4 * kg + r) * EPS + wn + j * 16 + l16] = acc[i][j][r];
        }
        LOCOV_WS_BARRIER();                                   // barrier D
        ws_report(0);
        return;
    }

    if (wave < 8) {
        // ------------------------------------------------------------------ staging waves (256 threads)
        // Neither operand needs a conversion, so a staged 16-byte chunk is one buffer load and one ds_write_b128.  (LDS DMA was
        // tried first: a dedicated wave ISSUES a 1 KB DMA in 80-150 cycles -- the guide's 25 GB/s per loader wave -- and four
        // waves x 8 DMAs per K-tile then take ~1 250 cycles against 768 of MFMA work; a load + a store issue in ~60.)  Chunks
        // wait two K-tiles ahead in registers, as A did in gemm_split_kernel: two LDS stages per operand suffice.
        constexpr int CH = 4;                                     // chunks per thread, operand and K-tile
        const int ts = tid - 256;
        unsigned a_off[CH], b_off[CH];
        int s_lds[CH];
#pragma unroll
        for (int i = 0; i < CH; i++) {
            const int idx = ts + i * 256, row = idx >> 3, ch = idx & 7;
            a_off[i] = (unsigned)(((int64_t)row * lda * 4) + ch * 16);
            b_off[i] = (unsigned)(((int64_t)row * K * 4) + ch * 16);
            s_lds[i] = row * WROWB + ((ch ^ wswz(row)) * 16);       // chunk c of row r sits at slot c ^ wswz(r) (gemm_split.hip)
        }
        // operand cursor: K-tile index -> (tile, kt) -> base pointers; rows of A past M are outside the descriptor's
        // num_records and read as zero (they feed rows that are never stored)
        struct Cursor {
            int k, kt;
            const char *a, *b;
            unsigned a_rec;
        };
        auto open_tile = [&](Cursor &c) {
            int bb, n0;
            int64_t m0;
            tile_coords(tg, my_tile(c.k), bb, m0, n0);
            c.a = reinterpret_cast<const char *>(A + (bt.count > 1 ? bb * bt.sa : 0) + m0 * lda);
            c.b = reinterpret_cast<const char *>(B + (bt.count > 1 ? bb * bt.sb : 0) + (int64_t)n0 * K);
            const int64_t rows = M - m0 < BM ? M - m0 : BM;
            c.a_rec = (unsigned)(((rows - 1) * lda + K) * 4);
            c.kt = 0;
        };
        Cursor cur{0, 0, nullptr, nullptr, 0u};
        open_tile(cur);
        // TWO register sets: while K-tile q+1 waits in one, K-tile q+2 is in flight into the other (with one set the chain
        // "request, wait out the whole memory latency, store, request" bounded the K-tile period at ~1 450 cycles)
        u32x4 ra[2][CH], rb[2][CH];
        auto load = [&](auto set_tag) __attribute__((always_inline)) {       // the K-tile the cursor points at -> register set
            constexpr int SET = decltype(set_tag)::value;
            const unsigned koff = (unsigned)cur.kt * (BK * 4);
            const __amdgpu_buffer_rsrc_t r_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(cur.a), 0, cur.a_rec, 0x00020000);
            const __amdgpu_buffer_rsrc_t r_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(cur.b), 0, 0xffffffff, 0x00020000);
#pragma unroll
            for (int i = 0; i < CH; i++) ra[SET][i] = __builtin_amdgcn_raw_buffer_load_b128(r_a, a_off[i] + koff, 0, 0);
#pragma unroll
            for (int i = 0; i < CH; i++) rb[SET][i] = __builtin_amdgcn_raw_buffer_load_b128(r_b, b_off[i] + koff, 0, 0);
            if (++cur.kt == KT) {
                cur.k++;
                if (cur.k < T) open_tile(cur);
            }
        };
        auto store = [&](auto set_tag, int stage) __attribute__((always_inline)) {
            constexpr int SET = decltype(set_tag)::value;
#pragma unroll
            for (int i = 0; i < CH; i++) *reinterpret_cast<u32x4 *>(ldsb + stage * ASTB + s_lds[i]) = ra[SET][i];
#pragma unroll
            for (int i = 0; i < CH; i++) *reinterpret_cast<u32x4 *>(ldsb + WBASE + stage * WSTB + s_lds[i]) = rb[SET][i];
        };
        std::integral_constant<int, 0> S0;
        std::integral_constant<int, 1> S1;
        // prologue: K-tile 0 -> stage 0; K-tile 1 -> set 1, K-tile 2 -> set 0 (in flight)
        load(S0);
        __builtin_amdgcn_s_waitcnt(waitcnt_imm(0, false));
        store(S0, 0);
        if (Q > 1) load(S1);
        if (Q > 2) load(S0);
        LOCOV_WS_BARRIER();                                          // barrier P
        // iteration q (the MFMA waves consume K-tile q from stage q & 1): K-tile q+1 (requested two iterations ago, register set
        // (q+1) & 1) -> stage (q+1) & 1, free since barrier q-1; K-tile q+3 is requested into the set just emptied.  The wait
        // leaves K-tile q+2's 2*CH loads in flight.
        auto iter = [&](auto set_tag, int q) __attribute__((always_inline)) {
            constexpr int SET = decltype(set_tag)::value;           // = (q + 1) & 1
            if (q + 1 < Q) {
                if (q + 2 < Q) __builtin_amdgcn_s_waitcnt(waitcnt_imm(2 * CH, false));
                else __builtin_amdgcn_s_waitcnt(waitcnt_imm(0, false));
                store(set_tag, SET);
                if (q + 3 < Q) load(set_tag);
            }
            LOCOV_WS_BARRIER();                                       // barrier q
        };
        for (int q = 0; q < Q; q += 2) {                              // Q is even (K % 64 == 0)
            iter(S1, q);
            iter(S0, q + 1);
        }
        LOCOV_WS_BARRIER();                                           // barrier D
        ws_report(1);
        return;
    }

    // ---------------------------------------------------------------------- epilogue waves (256 threads)
    {
        constexpr int EJ = 1;                                     // 16-byte pieces per thread and step (8 rows x 32 chunks = 256)
        const int te = tid - 512;
        const int ch = te & 31;                                   // 16-byte column chunk of this thread
        const int rsub[EJ] = {te >> 5};                           // its row inside an 8-row step
        const bool relu = (epi.flags & LOCOV_EPI_RELU) != 0;
        // per tile: output / residual descriptors, scale / shift of this thread's four columns
        __amdgpu_buffer_rsrc_t r_out = __builtin_amdgcn_make_buffer_rsrc(Cout, 0, 0u, 0x00020000);
        __amdgpu_buffer_rsrc_t r_res = r_out;
        f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
        auto open_out = [&](int k, __amdgpu_buffer_rsrc_t &ro, __amdgpu_buffer_rsrc_t &rr) {
            int bb, n0;
            int64_t m0;
            tile_coords(tg, my_tile(k), bb, m0, n0);
            const int64_t rows = M - m0 < BM ? M - m0 : BM;
            const unsigned nrec = (unsigned)(((rows - 1) * ldc + BN) * 4);
            float *Cb = Cout + (bt.count > 1 ? bb * bt.sc : 0) + m0 * ldc + n0;
            ro = __builtin_amdgcn_make_buffer_rsrc(Cb, 0, nrec, 0x00020000);
            if (HASRES) rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(epi.residual) + m0 * ldc + n0, 0, nrec, 0x00020000);
            return n0;
        };
        auto step_off = [&](int step, int j) {                    // byte offset of (row, chunk) in the tile; steps past the tile fall outside
            return step < NSTEPS ? (unsigned)(((int64_t)(step * 8 + rsub[j]) * ldc + ch * 4) * 4) : 0xffffffffu;
        };
        // Residual pipeline, one whole tile deep.  The epilogue of tile u runs during tile u+1's K-loop, 8 rows (one 16-byte piece
        // per thread) per step; its residual rows are requested one tile EARLIER, in the same slots of tile u's own K-loop, into
        // the other of two register sets of NSTEPS pieces -- 64 KB per CU in flight, which is what it takes to stream the
        // residual at HBM rate (with two K-tiles of lookahead the 16 KB in flight bounded the whole kernel).  Every slot issues
        // exactly SPS loads and SPS stores (out-of-range offsets where there is nothing to do), so "the piece requested one
        // tile ago has arrived" is the constant s_waitcnt vmcnt(2*NSTEPS - 2*SPS): the other NSLOT-1 slots' loads and stores are younger.
        // Stores are never waited for sooner than a tile after their issue.
        f32x4 res[2][NSTEPS];
        auto open_scale = [&](int n0) {
            sc = f32x4{1.f, 1.f, 1.f, 1.f};
            sh = f32x4{0.f, 0.f, 0.f, 0.f};
            if (epi.scale) sc = *reinterpret_cast<const f32x4 *>(epi.scale + n0 + ch * 4);
            sc *= out_scale;
            if (epi.shift) sh = *reinterpret_cast<const f32x4 *>(epi.shift + n0 + ch * 4);
        };
        constexpr int NSLOT_C = (NSTEPS + SPS - 1) / SPS;         // slots that carry steps (<= KT - 1 by the launch conditions)
        // one slot: process steps [slot*SPS, +SPS) of the tile in the buffer (descriptor ro, register set `set`), then request the
        // same steps of the tile whose K-loop is running (descriptor rr) into the OTHER set
        auto slot_work = [&](int slot, int set, const __amdgpu_buffer_rsrc_t &ro, bool drain, const __amdgpu_buffer_rsrc_t &rr, bool req) {
            f32x4 v[SPS];
            if (drain) {
#pragma unroll
                for (int s = 0; s < SPS; s++) v[s] = *reinterpret_cast<const f32x4 *>(epbuf + ((slot * SPS + s) * 8 + rsub[0]) * EPS + ch * 4);
            }
            if (HASRES) __builtin_amdgcn_s_waitcnt(waitcnt_imm(2 * NSTEPS - 2 * SPS, true));   // younger: NSLOT-1 slots of SPS stores + SPS loads
            else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int s = 0; s < SPS; s++) {
                f32x4 o = {0.f, 0.f, 0.f, 0.f};
                if (drain) {
                    o = v[s] * sc + sh;
                    if (HASRES) o += res[set][slot * SPS + s];
                    if (relu) {
                        o[0] = fmaxf(o[0], 0.f); o[1] = fmaxf(o[1], 0.f);
                        o[2] = fmaxf(o[2], 0.f); o[3] = fmaxf(o[3], 0.f);
                    }
                }
                if (drain || HASRES)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), ro, drain ? step_off(slot * SPS + s, 0) : 0xffffffffu, 0, 2);
            }
            if (HASRES) {
#pragma unroll
                for (int s = 0; s < SPS; s++)
                    res[set ^ 1][slot * SPS + s] = __builtin_bit_cast(
                        f32x4, __builtin_amdgcn_raw_buffer_load_b128(rr, req ? step_off(slot * SPS + s, 0) : 0xffffffffu, 0, 2));
            }
        };
        __amdgpu_buffer_rsrc_t ro_cur = r_out;                    // tile being drained (t - 1)
        __amdgpu_buffer_rsrc_t rr_cur = r_res;                    // tile whose K-loop is running (t): its residual is requested now
        LOCOV_WS_BARRIER();                                        // barrier P
        // (slots and register sets are compile-time indices: the tile loop is unrolled by two, the slots of a tile fully)
        auto tile_body = [&](auto set_tag, int t) __attribute__((always_inline)) {
            constexpr int SET = decltype(set_tag)::value;
#pragma unroll
            for (int slot = 0; slot < NSLOT_C; slot++) {
                LOCOV_WS_BARRIER();                                // barrier q = t*KT + slot
                if (slot == 0) {
                    if (t > 0) {
                        __amdgpu_buffer_rsrc_t unused = r_res;
                        open_scale(open_out(t - 1, ro_cur, unused));   // tile t-1 is in the buffer now
                    }
                    if (HASRES) {
                        __amdgpu_buffer_rsrc_t unused = r_out;
                        open_out(t, unused, rr_cur);
                    }
                }
                slot_work(slot, SET, t > 0 ? ro_cur : r_out, t > 0, rr_cur, true);
            }
            for (int kt = NSLOT_C; kt < KT; kt++) LOCOV_WS_BARRIER();
        };
        for (int t = 0; t < T; t += 2) {
            tile_body(std::integral_constant<int, 0>{}, t);
            if (t + 1 < T) tile_body(std::integral_constant<int, 1>{}, t + 1);
        }
        LOCOV_WS_BARRIER();                                        // barrier D: the last tile is in the buffer
        {
            __amdgpu_buffer_rsrc_t unused = r_res;
            open_scale(open_out(T - 1, ro_cur, unused));
            if (T & 1) {
#pragma unroll
                for (int slot = 0; slot < NSLOT_C; slot++) slot_work(slot, 1, ro_cur, true, r_res, false);
            } else {
#pragma unroll
                for (int slot = 0; slot < NSLOT_C; slot++) slot_work(slot, 0, ro_cur, true, r_res, false);
            }
        }
        ws_report(2);
    }
}

int launch_gemm_split_ws(const float *A, int64_t lda, const void *Wsplit, float *C, int64_t ldc, int64_t M, int N, int K,
                         const Epilogue &epi, float a_scale, float w_scale, hipStream_t s, const char *what, const Batch &bt,
                         unsigned *overflow, int cu_count)
{
    const int count = bt.count > 1 ? bt.count : 1;
    const int64_t tiles = ceil_div(M, BM) * (N / BN) * count;
    const int KT = K / BK;
    const int grid = cu_count - cu_count % 8;
    const size_t lds = EPBASE + EPB;
    static bool attr_done = false;
    if (!attr_done) {
        const void *fns[4] = {reinterpret_cast<const void *>(gemm_split_ws_kernel<1, false>), reinterpret_cast<const void *>(gemm_split_ws_kernel<1, true>),
                              reinterpret_cast<const void *>(gemm_split_ws_kernel<2, false>), reinterpret_cast<const void *>(gemm_split_ws_kernel<2, true>)};
        for (const void *f : fns)
            if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
                return set_error(LOCOV_ERR_LAUNCH, "%s: cannot raise the dynamic LDS limit to %zu bytes", what, lds);
        attr_done = true;
    }
    const int trec = timing_begin(s, 5, 2.0 * (double)M * N * K * count);
    const float out_scale = 1.f / (a_scale * w_scale);
    const float *Bf = reinterpret_cast<const float *>(Wsplit);
#define LOCOV_WS_LAUNCH(SPS, HR)                                                                                        \
    hipLaunchKernelGGL((gemm_split_ws_kernel<SPS, HR>), dim3((unsigned)grid), dim3(NTHREADS), lds, s, A, lda, Bf, C, ldc, M, N, K, epi, \
                       bt, a_scale, out_scale, overflow, (int)tiles)
    if (KT >= 17) {
        if (epi.residual) LOCOV_WS_LAUNCH(1, true);
        else LOCOV_WS_LAUNCH(1, false);
    } else {
        if (epi.residual) LOCOV_WS_LAUNCH(2, true);
        else LOCOV_WS_LAUNCH(2, false);
    }
#undef LOCOV_WS_LAUNCH
    timing_end(trec, s);
    return check_launch(what);
}

#ifdef LOCOV_WS_TRACE
extern "C" int locov_ws_debug_read(unsigned long long *out_host)
{
    return hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_ws_dbg), sizeof(unsigned long long) * 16) == hipSuccess ? 0 : -1;
}
#endif

}  // namespace locov
