// Wave-specialised, persistent form of the split-operand NT GEMM (gemm_split.hip: same arithmetic, same LDS images, same
// MFMA sequence per K-tile -- the results are bit-identical), for the large plain / batched launches of the Res5 stage
// (roi_emb_heads.py:217-245 as GEMMs).
//
// Why.  In gemm_split_kernel every wave does everything: it stages its share of A (buffer loads, the (hi, lo) split in
// the vector ALU, LDS stores) and of W (LDS DMA), reads its fragments, issues its 48 MFMAs per K-tile and, at the end of
// the tile, runs the epilogue (residual reads, 64 KB of stores).  Two workgroups per CU run in lockstep -- they start, finish
// their K-loops and reach their epilogues together -- so the memory-bound epilogue does not overlap the other workgroup's
// MFMAs: for the K = 512 shapes (16 K-tiles per output tile: the 1x1 convolutions into 2 048 channels, the Winograd-domain
// batched GEMMs) the launch time is T_mem + T_mfma, not max(T_mem, T_mfma), and inside the K-loop the
// ~80 staging instructions per K-tile compete with the 48 MFMAs for the wave's single issue stream.
//
// Here ONE 512-thread workgroup per CU stays resident and walks over its tiles; its 8 waves have fixed roles (a 512-thread
// workgroup places waves w and w+4 on the same SIMD, waves 0-3 on four different SIMDs -- tools/probe/simd_probe.hip):
//   waves 0-3  MFMA      one per SIMD, a 64x64 sub-tile each: fragment reads + MFMAs, nothing else; at the end of a tile
//                        the accumulators go to an LDS buffer and the next tile starts at once
//   waves 4-5  staging   the operand stream, continuous across tile boundaries: A two K-tiles ahead in registers, split and
//                        stored one K-tile ahead; W by LDS DMA one K-tile ahead
//   waves 6-7  epilogue  drain tile t-1's accumulators from the LDS buffer DURING tile t's K-loop: 8 rows per step, the
//                        residual rows requested two K-tiles before they are used, stores never waited for
// One s_barrier per K-tile (all 8 waves) hands the LDS stages over, exactly as in gemm_split_kernel.
//
// LDS: two A stages (20 KB), three W stages (16 KB), the 66 KB accumulator buffer = 154 KB.  Used by launch_gemm_split for K % 64 == 0, K >= 512,
// N % 128 == 0, no mask, at least 512 tiles; everything else (and the mean-fused last convolution) stays on gemm_split_kernel.
#include "gemm_nt.h"

namespace locov {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int ROWB = 160, WROWB = 128;
constexpr int ASTB = BM * ROWB;                         // one A stage: 20 480 B (two stages: A waits two K-tiles ahead in registers)
constexpr int WSTB = BN * WROWB;                        // one W stage: 16 384 B (THREE stages: the DMA of K-tile q+2 is in flight while
                                                        // q is consumed -- with one workgroup per CU nothing else hides its latency)
constexpr int WBASE = 2 * ASTB;                         // W stages behind the A stages
constexpr int EPBASE = WBASE + 3 * WSTB;                // 90 112
constexpr int EPS = BN + 4;                             // floats per row of the accumulator buffer
constexpr int EPB = BM * EPS * 4;                       // 67 584   (total 157 696 B of the CU's 160 KB)
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
        const int afo = l16 * ROWB + ((kg >> 1) * 4 + (kg & 1)) * 16;
        int bfo[2];
#pragma unroll
        for (int hl = 0; hl < 2; hl++) bfo[hl] = l16 * WROWB + (((2 * kg + hl) ^ wswz(l16)) * 16);
        f16x8 fa[4][2], fb[4][2];
        f32x4 acc[4][4];
        auto rd_a = [&](int stage, int ga) {
            const char *As = ldsb + stage * ASTB + (wm + ga * 32) * ROWB + afo;
#pragma unroll
            for (int i = 0; i < 2; i++) {
                fa[2 * ga + i][0] = *reinterpret_cast<const f16x8 *>(As + i * 16 * ROWB);
                fa[2 * ga + i][1] = *reinterpret_cast<const f16x8 *>(As + i * 16 * ROWB + 32);
            }
        };
        auto rd_b = [&](int wstage_off, int gb) {                 // wstage_off = byte offset of the W stage (q % 3, a run-time value)
            const char *Bs = ldsb + WBASE + wstage_off + (wn + gb * 32) * WROWB;
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
        int wcur = 0;                                             // byte offset of the W stage of the current K-tile
        auto ktile = [&](const int s, const bool has_next) __attribute__((always_inline)) {
            const int x = s, y = s ^ 1;
            const int wnext = wcur == 2 * WSTB ? 0 : wcur + WSTB;
            rd_b(wcur, y);
            rd_a(s, 1);
            quarter(0, x);
            quarter(0, y);
            quarter(1, y);
            __builtin_amdgcn_sched_barrier(0);
            LOCOV_WS_BARRIER();
            if (has_next) {
                rd_a(s ^ 1, 0);
                rd_b(wnext, y);
            }
            wcur = wnext;
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
        // ------------------------------------------------------------------ staging waves (256 threads: gemm_split_kernel's share per thread)
        constexpr int CH = 4;
        const int ts = tid - 256, sw = wave - 4;
        unsigned a_off[CH];
        int a_lds[CH];
#pragma unroll
        for (int i = 0; i < CH; i++) {
            const int idx = ts + i * 256, row = idx >> 3, ch = idx & 7;
            a_off[i] = (unsigned)(((int64_t)row * lda + ch * 4) * 4);
            a_lds[i] = row * ROWB + (((ch >> 2) * 4 + ((ch >> 1) & 1)) * 16) + (ch & 1) * 8;
        }
        unsigned b_voff[CH];
#pragma unroll
        for (int i = 0; i < CH; i++) {
            const int row = (sw * CH + i) * 8 + (lane >> 3);
            b_voff[i] = (unsigned)(((int64_t)row * K * 4) + (((lane & 7) ^ wswz(row)) * 16));
        }
        // operand cursors: K-tile index -> (tile, kt) -> base pointers.  `ld` runs two K-tiles ahead (A loads), `st` one
        // (W DMA); rows past M are outside the A descriptor's num_records and read as zero.
        struct Cursor {
            int k, kt;
            const char *a, *b;
            unsigned a_rec;
        };
        auto open_tile = [&](Cursor &c) {
            int bb, n0;
            int64_t m0;
            tile_coords(tg, my_tile(c.k), bb, m0, n0);
            const float *Ab = A + (bt.count > 1 ? bb * bt.sa : 0) + m0 * lda;
            c.a = reinterpret_cast<const char *>(Ab);
            c.b = reinterpret_cast<const char *>(B + (bt.count > 1 ? bb * bt.sb : 0) + (int64_t)n0 * K);
            const int64_t rows = M - m0 < BM ? M - m0 : BM;
            c.a_rec = (unsigned)(((rows - 1) * lda + K) * 4);
            c.kt = 0;
        };
        auto advance = [&](Cursor &c) {
            if (++c.kt == KT) {
                c.k++;
                if (c.k < T) open_tile(c);
            }
        };
        Cursor ld{0, 0, nullptr, nullptr, 0u}, st{0, 0, nullptr, nullptr, 0u};
        open_tile(ld);
        open_tile(st);
        f32x4 ra[CH];
        float amax = 0.f;
        auto ld_a = [&]() {
            const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(ld.a), 0, ld.a_rec, 0x00020000);
#pragma unroll
            for (int i = 0; i < CH; i++)
                ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, a_off[i] + (unsigned)ld.kt * (BK * 4), 0, 0));
            advance(ld);
        };
        auto st_a = [&](int stage) {
#pragma unroll
            for (int i = 0; i < CH; i++) {
                u32x2 hi, lo;
                amax = fmaxf(fmaxf(amax, fabsf(ra[i][0])), fabsf(ra[i][1]));
                amax = fmaxf(fmaxf(amax, fabsf(ra[i][2])), fabsf(ra[i][3]));
                split4(ra[i], a_scale, hi, lo);
                char *p = ldsb + stage * ASTB + a_lds[i];
                *reinterpret_cast<u32x2 *>(p) = hi;
                *reinterpret_cast<u32x2 *>(p + 32) = lo;
            }
        };
        auto dma_b = [&](int wstage) {
            const __amdgpu_buffer_rsrc_t r =
                __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(st.b) + (int64_t)st.kt * (BK * 4), 0, 0xffffffff, 0x00020000);
#pragma unroll
            for (int i = 0; i < CH; i++)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(
                    r, (__attribute__((address_space(3))) void *)(ldsb + WBASE + wstage * WSTB + (sw * CH + i) * 8 * WROWB), 16,
                    b_voff[i], 0, 0, 0);
            advance(st);
        };
        // prologue: K-tile 0 -> A stage 0 / W stage 0, W of K-tile 1 -> W stage 1 (in flight), A of K-tile 1 -> registers
        dma_b(0);
        ld_a();
        __builtin_amdgcn_s_waitcnt(waitcnt_imm(0, false));
        st_a(0);
        if (Q > 1) {
            dma_b(1);
            ld_a();
        }
        LOCOV_WS_BARRIER();                                          // barrier P
        // iteration q (the MFMA waves consume K-tile q): W of K-tile q+2 is requested, A of K-tile q+1 (requested one iteration
        // ago, as was W of q+1) goes from the registers into A stage (q+1) & 1, A of K-tile q+2 is requested.  The one wait --
        // "A of q+1 has arrived" = everything but the DMAs just issued -- also covers W of q+1: nothing on this path waits
        // for an operation younger than one whole iteration.
        int wreq = 2;                                                // W stage of K-tile q+2
#ifdef LOCOV_WS_TRACE
        unsigned long long seg_[4] = {0, 0, 0, 0}, ts_;
#define WS_SEG(i) do { asm volatile("" ::: "memory"); const unsigned long long n_ = __builtin_readcyclecounter(); seg_[i] += n_ - ts_; ts_ = n_; } while (0)
#else
#define WS_SEG(i)
#endif
        for (int q = 0; q < Q; q++) {
#ifdef LOCOV_WS_TRACE
            ts_ = __builtin_readcyclecounter();
#endif
            if (q + 2 < Q) dma_b(wreq);
            WS_SEG(0);
            wreq = wreq == 2 ? 0 : wreq + 1;
            if (q + 1 < Q) {
                if (q + 2 < Q) __builtin_amdgcn_s_waitcnt(waitcnt_imm(CH, false));
                else __builtin_amdgcn_s_waitcnt(waitcnt_imm(0, false));
                WS_SEG(1);
                st_a((q + 1) & 1);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                WS_SEG(2);
                if (q + 2 < Q) ld_a();
                WS_SEG(3);
            }
            LOCOV_WS_BARRIER();                                       // barrier q
        }
#ifdef LOCOV_WS_TRACE
        if (blockIdx.x == 8 && lane == 0 && wave == 4)
            for (int i = 0; i < 4; i++) g_ws_dbg[8 + i] = seg_[i];
#endif
        if (overflow != nullptr && amax * a_scale >= 65504.f) atomicOr(overflow, 1u);
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
        auto step_off = [&](int step, int j) {                    // byte offset of (row, chunk) in the tile; inactive steps fall outside
            return step < NSTEPS ? (unsigned)(((int64_t)(step * 8 + rsub[j]) * ldc + ch * 4) * 4) : 0xffffffffu;
        };
        // Residual pipeline: a slot's residual rows are requested two iterations before the slot is processed, into a ring of
        // two register sets indexed by the parity of the processing iteration.  Per iteration the order is
        //     process(slot q)  [reads set q & 1, stores]   then   request(slot q + 2)  [loads into set q & 1],
        // and EVERY iteration issues exactly 2*SPS stores and 2*SPS loads (with out-of-range offsets when there is nothing to
        // do), so that "the loads issued two iterations ago have arrived" is the constant s_waitcnt vmcnt(4*SPS): younger
        // than them are only the previous iteration's stores and loads.  Stores are never waited for sooner than two
        // iterations after their issue.
        f32x4 res[2][SPS][EJ];
        auto request = [&](int set, const __amdgpu_buffer_rsrc_t &rr, int slot) {
            if (!HASRES) return;
#pragma unroll
            for (int s = 0; s < SPS; s++)
#pragma unroll
                for (int j = 0; j < EJ; j++)
                    res[set][s][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rr, step_off(slot * SPS + s, j), 0, 2));
        };
        auto process = [&](int set, const __amdgpu_buffer_rsrc_t &ro, int slot, bool active) {
            f32x4 v[SPS][EJ];
            if (active) {
#pragma unroll
                for (int s = 0; s < SPS; s++)
#pragma unroll
                    for (int j = 0; j < EJ; j++) {
                        const int step = slot * SPS + s;
                        v[s][j] = *reinterpret_cast<const f32x4 *>(epbuf + ((step < NSTEPS ? step : 0) * 8 + rsub[j]) * EPS + ch * 4);
                    }
            }
            if (HASRES) __builtin_amdgcn_s_waitcnt(waitcnt_imm(2 * EJ * SPS, true));
            if (!active && !HASRES) return;
#pragma unroll
            for (int s = 0; s < SPS; s++)
#pragma unroll
                for (int j = 0; j < EJ; j++) {
                    f32x4 o = {0.f, 0.f, 0.f, 0.f};
                    if (active) {
                        o = v[s][j] * sc + sh;
                        if (HASRES) o += res[set][s][j];
                        if (relu) {
                            o[0] = fmaxf(o[0], 0.f); o[1] = fmaxf(o[1], 0.f);
                            o[2] = fmaxf(o[2], 0.f); o[3] = fmaxf(o[3], 0.f);
                        }
                    }
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), ro, active ? step_off(slot * SPS + s, j) : 0xffffffffu, 0, 2);
                }
        };
        auto open_scale = [&](int n0) {
            sc = f32x4{1.f, 1.f, 1.f, 1.f};
            sh = f32x4{0.f, 0.f, 0.f, 0.f};
            if (epi.scale) sc = *reinterpret_cast<const f32x4 *>(epi.scale + n0 + ch * 4);
            sc *= out_scale;
            if (epi.shift) sh = *reinterpret_cast<const f32x4 *>(epi.shift + n0 + ch * 4);
        };
        const int NSLOT = (NSTEPS + SPS - 1) / SPS;               // slots that carry steps (<= KT - 1 by the launch conditions)
        // Slot schedule in terms of this workgroup's K-tile counter q = t * KT + kt: the accumulators of tile t-1 are in the
        // buffer behind barrier (t*KT); slot kt of tile t-1's epilogue runs behind barrier (t*KT + kt), kt < NSLOT; the MFMA
        // waves overwrite the buffer only behind barrier (t*KT + KT - 1).  The last tile is drained behind barrier D.
        __amdgpu_buffer_rsrc_t ro_cur = r_out;                    // tile being drained
        __amdgpu_buffer_rsrc_t rr_req = r_res;                    // tile whose residual rows are being requested
        LOCOV_WS_BARRIER();                                        // barrier P
        int t = 0, kt = 0;                                         // (tile, K-tile) of iteration q
        int t2 = 0, kt2 = 2;                                       // ... of iteration q + 2   (KT >= 16)
        for (int q = 0; q < Q; q++) {
            LOCOV_WS_BARRIER();                                    // barrier q
            if (kt == 0 && t > 0) {
                __amdgpu_buffer_rsrc_t unused = r_res;
                open_scale(open_out(t - 1, ro_cur, unused));
            }
            const bool cur = t > 0 && kt < NSLOT;
            process(q & 1, cur ? ro_cur : r_out, kt, cur);         // (r_out has num_records 0: its stores are dropped)
            // request the residual rows of the slot processed two iterations from now (tile t2 - 1; for q + 2 >= Q that is the
            // last tile, drained behind barrier D)
            const bool req = t2 > 0 && kt2 < NSLOT;
            if (HASRES && req && kt2 == 0) {
                __amdgpu_buffer_rsrc_t unused = r_out;
                open_out(t2 - 1, unused, rr_req);
            }
            request(q & 1, req ? rr_req : r_res, req ? kt2 : NSLOT);   // (r_res has num_records 0: the loads return 0)
            if (++kt == KT) { kt = 0; t++; }
            if (++kt2 == KT) { kt2 = 0; t2++; }
        }
        LOCOV_WS_BARRIER();                                        // barrier D: the last tile is in the buffer
        {
            __amdgpu_buffer_rsrc_t rr_last = r_res;
            open_scale(open_out(T - 1, ro_cur, rr_last));
            for (int slot = 0; slot < NSLOT; slot++) {             // iteration Q + slot of the same pipeline
                process((Q + slot) & 1, ro_cur, slot, true);
                const bool req = slot + 2 < NSLOT;
                request((Q + slot) & 1, req ? rr_last : r_res, req ? slot + 2 : NSLOT);
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
