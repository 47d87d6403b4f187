// The split-operand NT GEMM of gemm_split.hip on a 256x256 tile: y[M,N] = epi(x[M,K] . W[N,K]^T), BOTH operands already in the
// split (hi, lo) f16 layout (LOCOV_GEMM_A_SPLIT launches: the Winograd-domain batched GEMMs and the 1x1 convolutions behind a
// Winograd output transform -- roi_emb_heads.py:217-245 as GEMMs).
//
// Why a second tile.  On MI355X these launches sit at the 1 400 W package power cap (profiles/r03_power_probe.txt): their time is
// (dynamic energy) / (cap - idle), and the energy ablation of the 128x128 kernel (profiles/r03_energy_ablation.txt) puts 57 % of
// the dynamic energy into the MFMAs, 31 % (K = 2048) / 19 % (K = 512) into staging the operand tiles L2 -> LDS, 6 % into the
// fragment reads.  The MFMA count is fixed; what a kernel can still choose is how often an operand byte crosses L2 -> LDS.
// A 256x256 tile stages (256 + 256) rows per K-tile for 4x the products of a 128x128 tile's (128 + 128): HALF the staged
// bytes per product, and a wave's 128x64 sub-tile reads (128 + 64) fragment rows per 96 MFMAs instead of (64 + 64) per 48.
//
// Shape of the kernel: ONE workgroup of 8 waves per CU (2 x 4 waves of 128 x 64: acc = 128 VGPRs), two LDS stages of 64 KB, every
// operand byte by LDS DMA (`buffer_load ... lds`, XOR-swizzled 128-byte rows, as gemm_split.hip).  A K-tile is eight "eighths" of
// 12 MFMAs -- an A row-block PAIR (32 rows) against a W column-block PAIR (32 columns) -- ordered so that only two A pairs
// (double buffer) and the two W pairs are ever live: 64 fragment VGPRs, as in the small kernel:
//
//     (A0,Bf) (A0,Bg) (A1,Bg) (A1,Bf) (A2,Bf) (A2,Bg) (A3,Bg) | wait DMA(t+1), barrier, issue DMA(t+2) | (A3,Bf)      g = 1 - f
//
// The tile ends on Bf, so Bg's registers take the next tile's first W pair behind the barrier and the next tile starts on
// (A0', Bg'): f flips every tile.  The DMA of tile t+2 goes out right behind the barrier of tile t (all reads of its stage
// are done by then) -- a full K-tile of matrix work (3 072 cycles per SIMD) ahead of its first use.
// With one workgroup per CU nothing overlaps a tile's prologue and epilogue: 10.5-11.4 us per tile whatever K (tools/attic/dbg_big_fixed_cost.py),
// a fifth of a K = 512 tile.  Resident workgroups, a staggered start and an earlier residual request were each measured and bought
// nothing: hiding it takes matrix work running beside the epilogue, which 160 KB of LDS and 512 VGPRs
// per SIMD do not leave room for at this tile size.
#include "gemm_split_common.h"
#include "winograd_transform.h"

#include <cstdlib>
#include <type_traits>

namespace locov {

namespace {

constexpr int GBM = 256, GBN = 256, GNW = 8, GNT = 64 * GNW;
constexpr int GTM = 128, GTN = 64;                  // a wave's sub-tile: 8 x 4 blocks of 16 x 16
constexpr int GSTAGE = (GBM + GBN) * WROWB;         // 64 KB per stage
// Epilogue staging: a wave's 64 x 64 half in its private LDS area, UNPADDED rows of 64 floats with the 16-float column blocks of
// rows 4..7 (mod 8) swapped pairwise (column ^ 16): the accumulator dump (ds_write_b32; lanes 0-15 / 16-31 of a service group hold
// rows 4 apart) and the row reads (ds_read_b128; a 16-lane service group -- lanes {0-3, 12-15, 20-27} -- spans two adjacent rows
// of one 4-row group) are then both bank-conflict-free.  The padded pitch of 68 floats this replaces made every one of those reads
// 2-way conflicted on 4 banks (rows r and r + 1 shifted by 4 banks: SQ_LDS_BANK_CONFLICT = tiles x 8 waves x 32 reads x 4 =
// 12.5 M per conv3 launch, profiles/r03_pmc_mfma_util.json; the Winograd-input form, which does not use this staging, showed 0).
constexpr int GEPS = GTN;                           // epilogue staging pitch (floats)
__device__ __forceinline__ int eps_swz(int row) { return ((row >> 2) & 1) * 16; }
constexpr int GLDS_WINO = GBM * (GBN / 2 + 4) * 4;   // MODE_WINO's finished half tile [256][128 + 4] (below)
constexpr int GLDS_KE = 2 * GSTAGE > GNW * 64 * GEPS * 4 ? 2 * GSTAGE : GNW * 64 * GEPS * 4;
constexpr int GLDS = GLDS_KE > GLDS_WINO ? GLDS_KE : GLDS_WINO;
constexpr int GCH = 4;                              // LDS-DMA pieces (8 rows each) per wave, operand and K-tile
static_assert(GLDS <= 160 * 1024, "one workgroup per CU: all of the LDS, no more");

constexpr int MODE_PLAIN = 0, MODE_SEGSUM = 1, MODE_WINO = 2;
// MODE_WINO: an M tile holds the rows of whole ROIs only (5 x 49 = 245 of the 256; the last 11 compute on clamped rows and are dropped)
constexpr int WSEG = 49, WROIS = GBM / WSEG, WROWS = WROIS * WSEG;
constexpr int WYP = GBN / 2 + 4;                    // pitch (floats) of the finished half tile [256][128] in LDS
static_assert(GBM * WYP * 4 <= GLDS, "the finished half tile (all 256 rows: the dump is branch-free) fits the K-loop's LDS");

}  // namespace

// SEGSUM (the stage's last 1x1 convolution + the spatial mean behind it, roi_emb_heads.py:262,344,356): rows are ROI-major
// (m = roi * seg + position), the residual too; instead of storing the finished [M, N] values every wave sums them per ROI over
// each 64-row chunk of its sub-tile: partial[(chunk * 3 + slot) * N + n] with chunk = (global row) / 64 and slot = the ROI's
// index among the (at most three, seg >= 43) ROIs the chunk touches.  segsum64_finish_kernel adds the one or two chunks of a
// ROI in a fixed order: deterministic, and the [M, N] tensor is neither written nor re-read.
//
// MODE_WINO (a bottleneck's first 1x1 convolution + FrozenBN + ReLU, and the input transform of the Winograd-domain 3x3 behind it:
// roi_emb_heads.py:217-245's conv1 -> conv2): rows are ROI-major with 49 per ROI and an M tile is 5 whole ROIs.  The finished tile
// never leaves the CU as pixels: half of its columns at a time it is laid out in LDS ([256][128] fp32, pitch 132), and the 8 waves share the
// (ROI, fy) units of wino_in_fy -- lane = channel pair, the 3-4 patch rows fy needs read from LDS, 11 transform-domain values
// written straight into V [121][R][N] (`partial`) in the split layout, the bits wino_input_kernel<false, true> would have written
// from the stored pixels.  Saves the pixel tensor's write and re-read (2 x 0.8 GB per block at 8 000 proposals) and a launch.
// BATCHED: bt.count problems of the same shape in one grid (the 121 Winograd-domain GEMMs) -- a template parameter only so that the
// two launch kinds of MODE_PLAIN carry different kernel names in profiler output (per-kind traffic / duration, bench.py `instances`)
template <int MODE, bool BATCHED = false>
__global__ __launch_bounds__(GNT, 1) void gemm_split_big_kernel(const float *__restrict__ A, int64_t lda, const float *__restrict__ B,
                                                               float *__restrict__ Cout, int64_t ldc, int64_t M, int N, int K,
                                                               Epilogue epi, Batch bt, float a_scale, float out_scale,
                                                               unsigned *overflow, int seg, float *__restrict__ partial, float v_scale)
{
    constexpr bool SEGSUM = MODE == MODE_SEGSUM;
    constexpr int TILE_ROWS = MODE == MODE_WINO ? WROWS : GBM;
    __shared__ u32x4 lds[GLDS / 16];
    char *const ldsb = reinterpret_cast<char *>(lds);

    const int tiles_n = (N + GBN - 1) / GBN;
    const int nwg = gridDim.x;
    int tile = xcd_remap(blockIdx.x, nwg);
    if (BATCHED) {
        const int per = nwg / bt.count, b = tile / per;
        tile -= b * per;
        A += b * bt.sa;
        B += b * bt.sb;
        Cout += b * bt.sc;
    }
    // N tiles in groups of NG, inside a group M-tile outer / N-tile inner: the workgroups an XCD runs back to back share their
    // A panels and stream a W slice (NG x 256 rows x K) that stays in its 4 MB L2 (gemm_split.hip's order)
    int64_t m0;
    int n0;
    {
#ifndef LOCOV_BIG_NG
#define LOCOV_BIG_NG 0
#endif
        const int NG = LOCOV_BIG_NG > 0 ? LOCOV_BIG_NG : ((int64_t)K * 4 * GBN * 8 <= (2 << 20) ? 8 : 4);
        const int tiles_m = (int)((BATCHED ? nwg / bt.count : nwg) / tiles_n);
        const int full = (tiles_n / NG) * NG, per_group = tiles_m * NG;
        if (tiles_n <= NG) {
            m0 = (int64_t)(tile / tiles_n) * TILE_ROWS;
            n0 = (tile % tiles_n) * GBN;
        } else if (tile < tiles_m * full) {
            const int g = tile / per_group, rem = tile - g * per_group;
            m0 = (int64_t)(rem / NG) * TILE_ROWS;
            n0 = (g * NG + rem % NG) * GBN;
        } else {
            const int gs = tiles_n - full, rem = tile - tiles_m * full;
            m0 = (int64_t)(rem / gs) * TILE_ROWS;
            n0 = (full + rem % gs) * GBN;
        }
    }

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = (wave >> 2) * GTM, wn = (wave & 3) * GTN;

    // staging: wave w brings rows 32 w .. 32 w + 31 of both operand tiles, four 1 KB pieces (8 rows) each; lane l supplies row
    // l / 8 of a piece and fetches the global 16-byte chunk (l % 8) ^ wswz(row): chunk c of row r then sits at slot c ^ wswz(r)
    const char *a_base = reinterpret_cast<const char *>(A + m0 * lda);
    const char *b_base = reinterpret_cast<const char *>(B + (int64_t)n0 * K);
    unsigned a_voff[GCH], b_voff[GCH];
#pragma unroll
    for (int i = 0; i < GCH; i++) {
        const int row = (wave * GCH + i) * 8 + (lane >> 3);
        const int64_t gm = m0 + row;
        const int gn = n0 + row;
        const bool a_ok = gm < M && row < TILE_ROWS;            // rows past M (or past the tile's whole ROIs): clamped, never stored
        a_voff[i] = (unsigned)((((a_ok ? gm : m0) - m0) * lda * 4) + (((lane & 7) ^ wswz(row)) * 16));
        b_voff[i] = (unsigned)(((int64_t)((gn < N ? gn : N - 1) - n0) * K * 4) + (((lane & 7) ^ wswz(row)) * 16));
    }
#ifndef LOCOV_BIG_A_AUX
#define LOCOV_BIG_A_AUX 0                                  // developer A/B: cache policy bits of the operand DMAs (2 = nt)
#endif
#ifndef LOCOV_BIG_B_AUX
#define LOCOV_BIG_B_AUX 0
#endif
    auto dma = [&](int stage) {
        const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(a_base), 0, 0xffffffff, 0x00020000);
        const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(b_base), 0, 0xffffffff, 0x00020000);
#pragma unroll
        for (int i = 0; i < GCH; i++) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(
                ra, (__attribute__((address_space(3))) void *)(ldsb + stage * GSTAGE + (wave * GCH + i) * 8 * WROWB), 16, a_voff[i], 0, 0, LOCOV_BIG_A_AUX);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(
                rb, (__attribute__((address_space(3))) void *)(ldsb + stage * GSTAGE + GBM * WROWB + (wave * GCH + i) * 8 * WROWB), 16,
                b_voff[i], 0, 0, LOCOV_BIG_B_AUX);
        }
        a_base += BK * 4;
        b_base += BK * 4;
    };

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // fragments: lane l holds row l % 16 and the 8 halves of k-group l / 16 of a 16-row block for the whole 32-wide K-tile
    f16x8 fa[2][2][2];                      // [buffer][block of the pair][hi, lo]
    f16x8 fb[4][2];                         // [column block][hi, lo]; pairs {0,1} and {2,3}
    const int l16 = lane & 15, kg = lane >> 4;
    int bfo[2];
#pragma unroll
    for (int hl = 0; hl < 2; hl++) bfo[hl] = l16 * WROWB + (((2 * kg + hl) ^ wswz(l16)) * 16);
    auto rd_a = [&](int stage, int ap, int buf) {
        const char *As = ldsb + stage * GSTAGE + (wm + ap * 32) * WROWB;
#pragma unroll
        for (int i = 0; i < 2; i++) {
            fa[buf][i][0] = *reinterpret_cast<const f16x8 *>(As + i * 16 * WROWB + bfo[0]);
            fa[buf][i][1] = *reinterpret_cast<const f16x8 *>(As + i * 16 * WROWB + bfo[1]);
        }
    };
    auto rd_b = [&](int stage, int bp) {
        const char *Bs = ldsb + stage * GSTAGE + GBM * WROWB + (wn + bp * 32) * WROWB;
#pragma unroll
        for (int j = 0; j < 2; j++) {
            fb[2 * bp + j][0] = *reinterpret_cast<const f16x8 *>(Bs + j * 16 * WROWB + bfo[0]);
            fb[2 * bp + j][1] = *reinterpret_cast<const f16x8 *>(Bs + j * 16 * WROWB + bfo[1]);
        }
    };
    // the 12 MFMAs of (A pair ap in buffer buf) x (W pair bp): per 16x16 block hi.hi, hi.lo, lo.hi
#ifndef LOCOV_BIG_MFMA_ORDER
#define LOCOV_BIG_MFMA_ORDER 0
#endif
    auto eighth = [&](int ap, int buf, int bp) __attribute__((always_inline)) {
#pragma unroll
        for (int ii = 0; ii < 2; ii++) {
            const int i = 2 * ap + ii, j0 = 2 * bp, j1 = 2 * bp + 1;
            if (LOCOV_BIG_MFMA_ORDER == 1) {
                // developer A/B: the six products of an A block grouped by A operand (hi x {W0 hi, W0 lo, W1 hi, W1 lo}, lo x {W0 hi, W1 hi}):
                // every accumulator still sees hi.hi, hi.lo, lo.hi in that order -- same bits, half the A-operand changes
                acc[i][j0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[buf][ii][0], fb[j0][0], acc[i][j0], 0, 0, 0);
                acc[i][j1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[buf][ii][0], fb[j1][0], acc[i][j1], 0, 0, 0);
                acc[i][j0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[buf][ii][0], fb[j0][1], acc[i][j0], 0, 0, 0);
                acc[i][j1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[buf][ii][0], fb[j1][1], acc[i][j1], 0, 0, 0);
                acc[i][j0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[buf][ii][1], fb[j0][0], acc[i][j0], 0, 0, 0);
                acc[i][j1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[buf][ii][1], fb[j1][0], acc[i][j1], 0, 0, 0);
            } else {
#pragma unroll
                for (int jj = 0; jj < 2; jj++) {
                    const int j = 2 * bp + jj;
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[buf][ii][0], fb[j][0], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[buf][ii][0], fb[j][1], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[buf][ii][1], fb[j][0], acc[i][j], 0, 0, 0);
                }
            }
        }
    };

    const int T = K / BK;
#ifndef LOCOV_BIG_STAGGER
#define LOCOV_BIG_STAGGER 0
#endif
    if (LOCOV_BIG_STAGGER && blockIdx.x < 256u) {
        // developer experiment: the first workgroup of every CU starts up to one tile-time late (golden-ratio spread), so that the
        // CUs -- whose tiles all take the same time -- do not reach their HBM-bound epilogues together
        const unsigned frac = (blockIdx.x * 0x9E3779B9u) >> 24;                 // 0..255
        const int units = (int)((frac * (unsigned)(T * LOCOV_BIG_STAGGER + 160)) >> 8);      // in 64-clock s_sleep units: ~T*48 clocks per K-tile + epilogue
        for (int i = 0; i < units; i += 64) __builtin_amdgcn_s_sleep(64);
    }
    __builtin_amdgcn_s_setprio(3);
    dma(0);
    dma(1);
    __builtin_amdgcn_s_waitcnt(0x0F70 | (2 * GCH));         // vmcnt(8): tile 0 has landed, tile 1's pieces stay in flight
    __syncthreads();
    rd_a(0, 0, 0);
    rd_b(0, 0);
    __builtin_amdgcn_s_setprio(0);

    // one K-tile computing from LDS stage s; f = the W pair it starts on (= its parity); more: a successor exists
    auto tile_step = [&](const int s, const bool more, const bool more2) __attribute__((always_inline)) {
        const int f = s, g = s ^ 1;
        rd_b(s, g);
        rd_a(s, 1, 1);
        __builtin_amdgcn_sched_barrier(0);
        eighth(0, 0, f);
        __builtin_amdgcn_sched_barrier(0);
        eighth(0, 0, g);
        __builtin_amdgcn_sched_barrier(0);
        rd_a(s, 2, 0);
        __builtin_amdgcn_sched_barrier(0);
        eighth(1, 1, g);
        __builtin_amdgcn_sched_barrier(0);
        eighth(1, 1, f);
        __builtin_amdgcn_sched_barrier(0);
        rd_a(s, 3, 1);
        __builtin_amdgcn_sched_barrier(0);
        eighth(2, 0, f);
        __builtin_amdgcn_sched_barrier(0);
        eighth(2, 0, g);
        __builtin_amdgcn_sched_barrier(0);
        eighth(3, 1, g);
        __builtin_amdgcn_sched_barrier(0);
        if (more) {
            __builtin_amdgcn_s_waitcnt(0x0F70);             // vmcnt(0): the next tile (requested a whole tile ago) has landed
            __syncthreads();                                // ... for every wave, and every wave is done reading stage s
            if (more2) dma(s);                              // tile t+2 -> the stage this tile is leaving
            rd_a(s ^ 1, 0, 0);
            rd_b(s ^ 1, g);
            __builtin_amdgcn_sched_barrier(0);
        }
        eighth(3, 1, f);
        __builtin_amdgcn_sched_barrier(0);
    };
    // T is even (the launcher requires K % 64 == 0): pairs of tiles, ONE straight-line tail -- alternative tails would meet in a
    // merge of two versions of the 128 accumulator registers, which the register allocator answers with spills
    for (int t = 0; t + 2 < T; t += 2) {
        tile_step(0, true, true);
        tile_step(1, true, true);
    }
    tile_step(0, true, false);
    tile_step(1, false, false);

    // Epilogue: the wave's 128 x 64 sub-tile in two halves of 64 x 64 through its private LDS area (C/D layout of the 16x16 MFMA:
    // col = lane & 15, row = 4 (lane >> 4) + reg), 16 bytes per lane and row-contiguous from there -- gemm_split.hip's.
    __builtin_amdgcn_sched_barrier(0);                     // (nothing of the epilogue is hoisted into the last K-tile: it has no registers to spare)
    __builtin_amdgcn_s_setprio(3);
    const bool relu = (epi.flags & LOCOV_EPI_RELU) != 0;
    if constexpr (MODE == MODE_WINO) {
        float *yt = reinterpret_cast<float *>(lds);
        float scj[4], shj[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int nn = n0 + wn + j * 16 + l16;
            scj[j] = (epi.scale ? epi.scale[nn] : 1.f) * out_scale;
            shj[j] = epi.shift ? epi.shift[nn] : 0.f;
        }
        const int64_t roi0 = m0 / WSEG, Rtot = M / WSEG;
        const int rois_here = Rtot - roi0 < WROIS ? (int)(Rtot - roi0) : WROIS;
        const int64_t fstride = Rtot * N;
        float amax = 0.f;
#pragma unroll 1
        for (int h = 0; h < 2; h++) {
            __syncthreads();                               // every wave has left the K-loop's LDS / the previous half's patch rows
            if (((wave & 3) >> 1) == h) {
                const int cl = (wave & 1) * GTN;
#pragma unroll
                for (int i = 0; i < 8; i++)
#pragma unroll
                    for (int j = 0; j < 4; j++)
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            const int row = wm + i * 16 + 4 * kg + r;
                            float v = acc[i][j][r] * scj[j] + shj[j];
                            if (relu) v = fmaxf(v, 0.f);
                            yt[row * WYP + cl + j * 16 + l16] = v;
                        }
            }
            __syncthreads();
            const int c = n0 + h * (GBN / 2) + 2 * lane;   // this lane's channel pair
            const int c_off = split_pair_offset(c);
#ifndef LOCOV_BIG_WINO_ABLATE
#define LOCOV_BIG_WINO_ABLATE 0                            // developer timing: 1 = no (ROI, fy) units at all, 2 = units without their stores
#endif
            for (int u = wave; u < (LOCOV_BIG_WINO_ABLATE == 1 ? 0 : rois_here * wino::NF); u += GNW) {
                const int roi = u / wino::NF, fy = u - roi * wino::NF;
                const float *patch = yt + roi * WSEG * WYP + 2 * lane;
                // V row (roi0 + roi) of transform-domain plane fy * 11: a buffer whose planes fx are a scalar offset apart
                const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(
                    partial + (roi0 + roi) * N + (int64_t)fy * wino::NF * fstride, 0, 0xffffffff, 0x00020000);
                const unsigned fbytes = (unsigned)(fstride * 4);
                auto load = [&](int y, int xx) __attribute__((always_inline)) {
                    return *reinterpret_cast<const f32x2 *>(patch + (y * 7 + xx) * WYP);
                };
                auto emit = [&](int fx, f32x2 a) __attribute__((always_inline)) {
                    amax = fmaxf(fmaxf(amax, fabsf(a[0])), fabsf(a[1]));
                    if (LOCOV_BIG_WINO_ABLATE != 2 || a[0] == 1234.5f)
                        __builtin_amdgcn_raw_buffer_store_b64(split_pair_words(c, a, v_scale), rv, c_off, fx * fbytes, 0);
                };
                switch (fy) {
                case 0: wino_in_fy<false, 0>(load, emit); break;
                case 1: wino_in_fy<false, 1>(load, emit); break;
                case 2: wino_in_fy<false, 2>(load, emit); break;
                case 3: wino_in_fy<false, 3>(load, emit); break;
                case 4: wino_in_fy<false, 4>(load, emit); break;
                case 5: wino_in_fy<false, 5>(load, emit); break;
                case 6: wino_in_fy<false, 6>(load, emit); break;
                case 7: wino_in_fy<false, 7>(load, emit); break;
                case 8: wino_in_fy<false, 8>(load, emit); break;
                case 9: wino_in_fy<false, 9>(load, emit); break;
                default: wino_in_fy<false, 10>(load, emit); break;
                }
            }
        }
        if (overflow != nullptr && amax * v_scale >= 65504.f) atomicOr(overflow, 1u);
        return;
    }
    const bool out_split = (epi.flags & LOCOV_EPI_OUT_SPLIT) != 0, res_split = (epi.flags & LOCOV_EPI_RES_SPLIT) != 0;
    const bool odd_lane = (lane & 1) != 0;
    const float inv_a_scale = 1.f / a_scale;
    float omax = 0.f;
    constexpr int LPR = GTN / 4, RPI = 64 / LPR, NIT = 64 / RPI;      // 16 lanes per row, 4 rows per instruction, 16 instructions per half
    static_assert(RPI == 4, "eps_swz: one read instruction covers one 4-row group (a uniform column swap per instruction)");
    const int c4 = (lane % LPR) * 4, rr = lane / LPR;
    const int n = n0 + wn + c4;
    const bool n_ok = n < N;
    const int64_t rows_here = M - m0 < GBM ? M - m0 : GBM;
    const unsigned nrec = (unsigned)(rows_here * ldc * 4);
    const __amdgpu_buffer_rsrc_t r_out = __builtin_amdgcn_make_buffer_rsrc(SEGSUM ? const_cast<float *>(epi.residual) + m0 * ldc : Cout + m0 * ldc, 0, nrec, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_res =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(epi.residual ? epi.residual + m0 * ldc : Cout + m0 * ldc), 0, nrec, 0x00020000);
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
    if (n_ok && epi.scale) sc = *reinterpret_cast<const f32x4 *>(epi.scale + n);
    sc *= out_scale;                                       // undo the operand scales
    if (n_ok && epi.shift) sh = *reinterpret_cast<const f32x4 *>(epi.shift + n);
    float *ep = reinterpret_cast<float *>(lds) + wave * (64 * GEPS);
    const unsigned vstep = (unsigned)(RPI * ldc * 4);
#pragma unroll
    for (int h = 0; h < 2; h++) {
#ifndef LOCOV_BIG_EPI_FEWBAR
#define LOCOV_BIG_EPI_FEWBAR 1                             // 0 = a workgroup barrier around every dump (the round's first form: conv3 +1 %, conv1 +1.5 %)
#endif
        // ONE workgroup barrier: every wave has left the K-loop's LDS.  From there on a wave only touches its own staging area, and the
        // LDS executes a wave's instructions in order: the dump of half 1 cannot pass the reads of half 0, the reads of a half cannot
        // pass its dump.  (The compiler is held to the program order by the memory clobbers.)
        if (h == 0 || !LOCOV_BIG_EPI_FEWBAR) __syncthreads();
        asm volatile("" ::: "memory");
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int r = 0; r < 4; r++) ep[(i * 16 + 4 * kg + r) * GEPS + ((j * 16 + l16) ^ eps_swz(4 * kg))] = acc[4 * h + i][j][r];
        if (!LOCOV_BIG_EPI_FEWBAR) __syncthreads();
        asm volatile("" ::: "memory");
        if (n_ok) {
            const unsigned voff = (unsigned)(((int64_t)(wm + 64 * h + rr) * ldc + n) * 4);      // rows past M: outside num_records
            // (requesting the residual a whole half -- 16 x 16 bytes per lane -- ahead of its use instead of four at a time was tried
            //  twice: conv3 2.55 -> 2.53 / 2.58 -> 2.64 ms, i.e. nothing.  Without the residual loads the launch is 0.5 ms shorter
            //  (LOCOV_BIG_EPI_ABLATE=1: 2.58 -> 2.08 ms) -- the 3.2 GB they read are what costs, not the round trips)
#pragma unroll
            for (int q4 = 0; q4 < NIT; q4 += 4) {
                f32x4 res[4];
#ifndef LOCOV_BIG_EPI_ABLATE
#define LOCOV_BIG_EPI_ABLATE 0                             // developer timing: 1 = no residual loads, 2 = no per-ROI column walk (SEGSUM), 3 = both
#endif
                if (epi.residual) {
#pragma unroll
                    for (int u = 0; u < 4; u++)
                        res[u] = (LOCOV_BIG_EPI_ABLATE & 1) ? f32x4{0.f, 0.f, 0.f, 0.f}
                                                            : __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_res, voff, (q4 + u) * vstep, 2));
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int it = q4 + u;
                    f32x4 v = *reinterpret_cast<const f32x4 *>(ep + (it * RPI + rr) * GEPS + (c4 ^ eps_swz(it * RPI)));
                    v = v * sc + sh;
                    if (epi.residual) v += res_split ? unsplit4(res[u], odd_lane, inv_a_scale) : res[u];
                    if (relu) {
                        v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f);
                        v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
                    }
                    if (SEGSUM) {
                        *reinterpret_cast<f32x4 *>(ep + (it * RPI + rr) * GEPS + (c4 ^ eps_swz(it * RPI))) = v;        // finished value back in place
                    } else if (out_split) {
                        omax = fmaxf(fmaxf(omax, fabsf(v[0])), fabsf(v[1]));
                        omax = fmaxf(fmaxf(omax, fabsf(v[2])), fabsf(v[3]));
                        __builtin_amdgcn_raw_buffer_store_b128(split4_pair(v, odd_lane, a_scale), r_out, voff, it * vstep, 2);
                    } else {
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r_out, voff, it * vstep, 2);     // nt: streamed once
                    }
                }
            }
        }
        if constexpr (SEGSUM) {
            // per-ROI column sums of this wave's 64 x 64 chunk: lane = column, a walk down the rows (the wave's own LDS area:
            // its writes above are ordered before these reads, no workgroup barrier needed -- the loop's next one covers re-use)
            const int64_t grow0 = m0 + wm + 64 * h;
            const int64_t left = M - grow0;
            const int rows_valid = left < 0 ? 0 : left < 64 ? (int)left : 64;
            const int col = n0 + wn + lane;
            if (rows_valid > 0 && col < N && !(LOCOV_BIG_EPI_ABLATE & 2)) {
                const int64_t chunk = grow0 >> 6;
                int pos = (int)(grow0 % seg), slot = 0;
                float sum = 0.f;
                for (int i0 = 0; i0 < 64; i0 += 8) {              // eight LDS reads in flight, then the (order-preserving) adds
                    float rv[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) rv[u] = ep[(i0 + u) * GEPS + (lane ^ eps_swz(u))];        // (i0 is a multiple of 8)
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        if (i0 + u < rows_valid) {
                            sum += rv[u];
                            if (++pos == seg) {
                                partial[(chunk * 3 + slot) * (int64_t)N + col] = sum;
                                sum = 0.f;
                                pos = 0;
                                slot++;
                            }
                        }
                    }
                }
                if (pos != 0) partial[(chunk * 3 + slot) * (int64_t)N + col] = sum;
            }
        }
    }
    if (!SEGSUM && out_split && overflow != nullptr && omax * a_scale >= 65504.f) atomicOr(overflow, 1u);
}

// out[q, n] = (the partial sums of ROI q's rows in the one or two 64-row chunks they fall into) / seg, in chunk order
__global__ __launch_bounds__(256) void segsum64_finish_kernel(const float *__restrict__ partial, int64_t R, int N, int seg, float inv_seg,
                                                              float *__restrict__ out)
{
    const int64_t total = R * (N / 4);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t q = i / (N / 4);
        const int n = (int)(i - q * (N / 4)) * 4;
        const int64_t c0 = (q * seg) >> 6, c1 = (q * seg + seg - 1) >> 6;
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
        for (int64_t c = c0; c <= c1; c++) {
            const int64_t slot = q - (c << 6) / seg;                 // ROI q among the ROIs chunk c touches
            a += *reinterpret_cast<const f32x4 *>(partial + (c * 3 + slot) * N + n);
        }
        *reinterpret_cast<f32x4 *>(out + q * N + n) = a * inv_seg;
    }
}

// launches that qualify: pre-split A, no mask / device scale, shapes that fill the chip with 256 x 256 tiles and row pitches
// whose tile offsets fit 32 bits
bool gemm_split_big_applicable(int64_t lda, int64_t ldc, int64_t M, int N, int K, const Epilogue &epi, const Batch &bt, const float *a_scale_dev)
{
    const char *fe = getenv("LOCOV_SPLIT_BIG");            // developer A/B (read per launch: tools/attic/dbg_bigtile.py flips it): 0 never, 1 whenever legal
    const int forced = fe ? atoi(fe) : -1;
    if (forced == 0) return false;
    if (!(epi.flags & LOCOV_GEMM_A_SPLIT) || epi.mask || epi.amax_out || a_scale_dev) return false;
    if (K % (2 * BK) != 0 || N % 8 != 0 || (int64_t)GBM * (lda > ldc ? lda : ldc) * 4 > 0x7fffffffLL || (int64_t)GBN * K * 4 > 0x7fffffffLL)
        return false;
    const int count = bt.count > 1 ? bt.count : 1;
    const int64_t tiles = ceil_div(M, GBM) * ceil_div(N, GBN) * count;
    if (tiles > 0x7fffffffLL) return false;
    if (forced > 0) return true;
    // enough tiles for four rounds over 256 CUs, and N a whole number of 256-wide tiles (a half-empty N tile wastes a quarter of the CU)
#ifndef LOCOV_BIG_BATCHED_MIN_TILES
#define LOCOV_BIG_BATCHED_MIN_TILES 1024       // (tools/make_variant.py A/B: the batched launches' own threshold, R6.6)
#endif
    return tiles >= (count > 1 ? LOCOV_BIG_BATCHED_MIN_TILES : 1024) && N % GBN == 0 && M >= (LOCOV_BIG_BATCHED_MIN_TILES < 1024 && count > 1 ? 3 : 4) * GBM;
}

int launch_gemm_split_big(const float *A, int64_t lda, const void *Wsplit, float *C, int64_t ldc, int64_t M, int N, int K,
                          const Epilogue &epi, float a_scale, float w_scale, hipStream_t s, const char *what, const Batch &bt,
                          unsigned *overflow)
{
    const int count = bt.count > 1 ? bt.count : 1;
    const int64_t tiles = ceil_div(M, GBM) * ceil_div(N, GBN) * count;
    // classes 9 / 10: the 256x256 split GEMM, one problem (the 1x1 convolutions) / batched (the Winograd-domain GEMMs).
    // algorithmic bytes: A and W once, the result once, the residual once
    const double abytes = 4.0 * count * ((double)M * K + (double)N * K + (double)M * N * (epi.residual ? 2.0 : 1.0));
    const int trec = timing_begin(s, count > 1 ? 10 : 9, 2.0 * (double)M * N * K * count, abytes);
    if (count > 1)
        hipLaunchKernelGGL((gemm_split_big_kernel<MODE_PLAIN, true>), dim3((unsigned)tiles), dim3(GNT), 0, s, A, lda, reinterpret_cast<const float *>(Wsplit), C,
                           ldc, M, N, K, epi, bt, a_scale, 1.f / (a_scale * w_scale), overflow, 0, static_cast<float *>(nullptr), 0.f);
    else
        hipLaunchKernelGGL((gemm_split_big_kernel<MODE_PLAIN, false>), dim3((unsigned)tiles), dim3(GNT), 0, s, A, lda, reinterpret_cast<const float *>(Wsplit), C,
                           ldc, M, N, K, epi, bt, a_scale, 1.f / (a_scale * w_scale), overflow, 0, static_cast<float *>(nullptr), 0.f);
    timing_end(trec, s);
    return check_launch(what);
}

// the mean-fused form: ROI-major rows and residual, seg rows per ROI (43 <= seg: at most three ROIs per 64-row chunk)
bool gemm_split_big_segmean_applicable(int64_t lda, int64_t M, int N, int K, const Epilogue &epi, int seg)
{
    if (!(epi.flags & LOCOV_SEGMEAN_RES_ROI_MAJOR) || seg < 43) return false;
    return gemm_split_big_applicable(lda, (int64_t)N, M, N, K, Epilogue{epi.scale, epi.shift, epi.residual, epi.flags & ~(unsigned)LOCOV_SEGMEAN_RES_ROI_MAJOR},
                                     Batch{1, 0, 0, 0}, nullptr);
}

int64_t gemm_split_big_segmean_workspace_bytes(int64_t M, int N) { return ceil_div(M, 64) * 3 * (int64_t)N * (int64_t)sizeof(float); }

int launch_gemm_split_big_segmean(const float *A, int64_t lda, const void *Wsplit, int64_t M, int N, int K, const Epilogue &epi, int seg,
                                  float a_scale, float w_scale, float *partial, float *out, hipStream_t s, const char *what, unsigned *overflow)
{
    const int64_t tiles = ceil_div(M, GBM) * ceil_div(N, GBN);
    // class 11: the mean-fused form -- A, W and the residual once; the per-chunk column sums instead of the [M, N] result
    const int trec = timing_begin(s, 11, 2.0 * (double)M * N * K,
                                  4.0 * ((double)M * K + (double)N * K + (double)M * N) + (double)gemm_split_big_segmean_workspace_bytes(M, N));
    hipLaunchKernelGGL((gemm_split_big_kernel<MODE_SEGSUM, false>), dim3((unsigned)tiles), dim3(GNT), 0, s, A, lda, reinterpret_cast<const float *>(Wsplit),
                       static_cast<float *>(nullptr), (int64_t)N, M, N, K, epi, Batch{1, 0, 0, 0}, a_scale, 1.f / (a_scale * w_scale), overflow, seg,
                       partial, 0.f);
    timing_end(trec, s);
    int rc = check_launch(what);
    if (rc) return rc;
    const int64_t R = M / seg, total = R * (N / 4);
    const unsigned blocks = (unsigned)(ceil_div(total, 256) < 65536 ? ceil_div(total, 256) : 65536);
    hipLaunchKernelGGL(segsum64_finish_kernel, dim3(blocks), dim3(256), 0, s, partial, R, N, seg, 1.f / (float)seg, out);
    return check_launch(what);
}

// the 1x1 convolution whose finished rows go straight into the Winograd domain (MODE_WINO): ROI-major rows, 49 per ROI, N a whole
// number of 256-wide tiles; V [121][R][N] in the split layout scaled by v_scale
bool gemm_split_big_wino_applicable(int64_t lda, int64_t M, int N, int K, const Epilogue &epi)
{
    const char *fe = getenv("LOCOV_WINO_FUSE");            // developer A/B / tests (read per launch): 0 = never
    if (fe && atoi(fe) == 0) return false;
    if (M % WSEG != 0 || N % GBN != 0 || epi.residual || (epi.flags & (LOCOV_EPI_OUT_SPLIT | LOCOV_EPI_RES_SPLIT))) return false;
    if ((M / WSEG) * (int64_t)N * 4 * wino::NF > 0xffffffffLL) return false;        // (a row of 11 transform-domain planes: 32-bit offsets)
    return gemm_split_big_applicable(lda, (int64_t)N, M, N, K, epi, Batch{1, 0, 0, 0}, nullptr);
}

int launch_gemm_split_big_wino(const float *A, int64_t lda, const void *Wsplit, int64_t M, int N, int K, const Epilogue &epi, float a_scale,
                               float w_scale, float *V, float v_scale, hipStream_t s, const char *what, unsigned *overflow)
{
    const int64_t tiles = ceil_div(M / WSEG, WROIS) * ceil_div(N, GBN);
    // class 8: A and W once, the transform-domain tensor V [121][M / 49][N] instead of the [M, N] pixels
    const int trec = timing_begin(s, 8, 2.0 * (double)M * N * K, 4.0 * ((double)M * K + (double)N * K + 121.0 * (double)(M / WSEG) * N));
    hipLaunchKernelGGL((gemm_split_big_kernel<MODE_WINO, false>), dim3((unsigned)tiles), dim3(GNT), 0, s, A, lda, reinterpret_cast<const float *>(Wsplit),
                       static_cast<float *>(nullptr), (int64_t)N, M, N, K, epi, Batch{1, 0, 0, 0}, a_scale, 1.f / (a_scale * w_scale), overflow, WSEG,
                       V, v_scale);
    timing_end(trec, s);
    return check_launch(what);
}

}  // namespace locov
