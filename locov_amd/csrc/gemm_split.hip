// fp32 NT GEMM  y[M,N] = epi(x[M,K] . W[N,K]^T)  on the gfx950 16-bit matrix pipe with SPLIT operands
// ("f16x2" arithmetic of the Res5 GEMMs -- the default; roi_emb_heads.py:217-245 of the reference as GEMMs):
//
//     s x = hi + lo,      hi = fp16(s x) (round to nearest),   lo = fp16(s x - hi)        (s = a power of two)
//     a.b ~= (hi_a.hi_b + hi_a.lo_b + lo_a.hi_b) / (s_a s_b)
//
// hi + lo carries 22 significant bits of x (s x - hi is exact in fp32; lo is an fp16 NORMAL number, i.e. keeps
// its 11 bits, for every |s x| >= 2^-3, and below that its absolute error is <= 2^-25 -- a fixed-point floor far
// under the rounding of the larger entries of the same row).  The operand scales s_a, s_w exist to place the
// data in fp16's range: s_w is chosen per weight matrix at pack time (max |s_w w| ~ 2^13), s_a is a launch
// parameter (2^4 for activations: exact up to |x| < 4094; 2^-2 for Winograd-domain data).  The dropped lo.lo term is
// 2^-24 relative; all three products are accumulated in fp32 by v_mfma_f32_16x16x32_f16 into ONE accumulator.
// The result differs from an fp32-MFMA GEMM by about two fp32 roundings per operand -- the order of the fp32
// accumulation error itself at K >= 512, and measured at or below it -- while the matrix pipe runs 3 f16 MFMAs
// (48 cycles) per 16x16x32 block instead of 16 f32 16x16x4 ones (256 cycles): 5.3x fewer matrix-pipe cycles.
//
// Data movement is the fp32 kernel's (gemm_nt.hip), byte for byte: 128x128 tile, BK = 32 fp32 columns, 4 waves
// (2x2, 64x64 each = 4x4 MFMA blocks of 16x16), two LDS stages, one barrier per K-tile, two workgroups per CU; tile
// order, epilogue (LDS re-layout, 16-byte buffer stores, residual prefetch) and the batched form are gemm_nt.hip's.
//   A is read as fp32 (16-byte buffer loads, two tiles ahead) and split in registers on its way into LDS: rows of
//     160 B = 8 data slots of 16 B + 2 pad, k-group c (8 columns) with its hi halves in slot {0,1,4,5}[c] and its lo
//     halves two slots further.  Even k-groups in even slots, odd ones in odd slots, pitch 10 slots: the 16x16x32
//     fragment read (lane l: row l%16, k-group l/16) and the 8-byte hi / lo staging stores (two rows x eight
//     4-column chunks per 16-lane group) are both bank-conflict-free.
//   W is split ONCE (locov_split_f16x2_pack) into a buffer with the size and row pitch of the fp32 matrix -- per
//     group of 8 columns: 8 hi halves, then 8 lo halves -- and needs no conversion, so it bypasses the registers:
//     `buffer_load_dwordx4 ... lds` (LDS DMA), one tile ahead, into unpadded 128-byte rows whose eight chunks are
//     XOR-permuted per row (wswz) for conflict-free fragment reads.
#include "gemm_split_common.h"

#include <cstdlib>
#include <type_traits>

#ifndef LOCOV_RES_PREFETCH
#define LOCOV_RES_PREFETCH 4
#endif
#ifndef LOCOV_SPLIT_MINWG
#define LOCOV_SPLIT_MINWG 2   // launch-bounds occupancy hint of the 128x128 tile: 2 workgroups per CU (3: a 168-register budget, see DESIGN)
#endif
#ifndef LOCOV_SPLIT_ABLATE
#define LOCOV_SPLIT_ABLATE 0  // ENERGY-ablation builds only (tools/attic/dbg_power.py; results are wrong on purpose): 1 no staging DMA in the
#endif                        // K-loop, 2 no fragment reads in the K-loop, 4 no MFMAs, 8 no epilogue; operands stay real data
#ifndef LOCOV_STORE_AUX
#define LOCOV_STORE_AUX 2     // cache policy bits of the epilogue's stores: 2 = nt (A/B: tools/make_variant.py ... -DLOCOV_STORE_AUX=0)
#endif

namespace locov {

namespace {

constexpr int BM = 128, BN = 128, WM = 2, WN = 2, NT = 256;
constexpr int TM = BM / WM, TN = BN / WN, MB = TM / 16, NB = TN / 16;   // a wave's 64x64 sub-tile = 4x4 blocks of 16x16
constexpr int ROWB = 160;                   // LDS pitch of an A row: 8 data slots of 16 B + 2 pad (layout below)
constexpr int CH = 4;                       // 16-byte chunks per thread, operand and tile

}  // namespace

// SEGSUM form (the last 1x1 convolution of Res5 + the spatial mean that follows it, roi_emb_heads.py:262,344,356): the
// rows are ROI-major (m = roi * seg + position, seg = 49), the residual is gathered from the position-major tensor
// (row position * R + roi), and instead of the [M,N] output the kernel leaves, per M-tile and per ROI the tile touches
// (at most four), the column sums of the finished values: partial[(tile_m * 4 + slot) * N + n].  segsum_finish_kernel
// adds the one or two partials of each ROI in a fixed order -- the [M,N] tensor is never written or re-read.
struct SegSum {
    int seg;
    int64_t R;
    float *partial;
};

// EMASK: the epilogue zeroes every value whose `epi.mask` entry is <= 0 (ReLU backward of a saved activation: the data
// gradients of the training step); a_scale_dev: when non-null the operand scale of A is read from device memory
// (a_scale_dev[0], a power of two chosen on the device from the tensor's max |x| by split_scale_kernel -- gradients have
// no a-priori range) and `out_scale` holds 1 / w_scale only.
// ASPLIT: A is ALREADY in the split layout of locov_split_f16x2_pack (written that way by its producer) with scale a_scale:
// it then needs no conversion and takes W's road -- LDS DMA into unpadded, XOR-swizzled 128-byte rows -- instead of
// buffer loads, 8 v_fma_mix + 2 v_max3 and two 8-byte LDS stores per 16-byte chunk.
// LOCOV_KTRACE (tools/make_variant.py ktrace gemm_split.hip -DLOCOV_KTRACE=1; never the product): s_memtime stamps inside the
// K-tile step of a few workgroups of the ASPLIT form, read back by tools/attic/dbg_ktrace.py
#ifdef LOCOV_KTRACE
__device__ unsigned long long g_ktrace[4 * 16];
#define KSTAMP(i) asm volatile("s_memtime %0" : "=s"(kt_[i]))
#else
#define KSTAMP(i)
#endif

// WM_: waves along M.  2 = the 128x128 tile, two workgroups per CU; 4 = a 256x128 tile of 8 waves, ONE workgroup per CU: the W
// tile is shared by twice the rows, so the CU stages 48 KB per K-tile through L2 -> LDS for the MFMA work it staged 64 KB for.
template <bool SEGSUM, bool EMASK = false, bool ASPLIT = false, int WM_ = 2>
__global__ __launch_bounds__(128 * WM_, WM_ == 2 ? LOCOV_SPLIT_MINWG : 1) void gemm_split_kernel(const float *__restrict__ A, int64_t lda,
                                                           const float *__restrict__ B, float *__restrict__ Cout,
                                                           int64_t ldc, int64_t M, int N, int K, Epilogue epi, Batch bt,
                                                           float a_scale, float out_scale, SegSum ss, unsigned *overflow,
                                                           const float *__restrict__ a_scale_dev)
{
    constexpr int WM = WM_, BM = TM * WM, NT = 64 * WM * WN, NW = WM * WN;
    constexpr int STAGEB = BM * ROWB + BN * WROWB;        // bytes per stage
    constexpr int CHB = BN / 8 / NW;                      // LDS-DMA pieces (8 rows each) of the W tile per wave
    constexpr int EPS = TN + 4;                           // epilogue staging pitch (floats)
    constexpr int LDSB = 2 * STAGEB > NW * TM * EPS * 4 ? 2 * STAGEB : NW * TM * EPS * 4;
    static_assert(!SEGSUM || WM_ == 2, "the mean-fused form is written for the 128-row tile");
    __shared__ u32x4 lds[LDSB / 16];
    char *const ldsb = reinterpret_cast<char *>(lds);
#ifdef LOCOV_KTRACE
    const unsigned long long kw0_ = __builtin_amdgcn_s_memtime();
#endif
    if (a_scale_dev != nullptr) {
        float inv;
        split_scale_of(a_scale_dev, a_scale, inv);
        out_scale *= inv;                     // = 1 / a_scale (exact: powers of two)
    }

    __builtin_amdgcn_s_setprio(3);
    const int tiles_n = (N + BN - 1) / BN;
    const int nwg = gridDim.x;
    int tile = xcd_remap(blockIdx.x, nwg);
    if (bt.count > 1) {
        const int per = nwg / bt.count, b = tile / per;
        tile -= b * per;
        A += b * bt.sa;
        B += b * bt.sb;
        Cout += b * bt.sc;
    }
    // tile order of gemm_nt.hip: N tiles in groups of NG, inside a group M-tile outer / N-tile inner, so that the
    // workgroups resident on an XCD stream a W slice that stays in its L2
    int64_t m0;
    int n0;
    {
        const int NG = (int64_t)K * 4 * BN * 8 <= (2 << 20) ? 8 : 4;
        const int tiles_m = (int)((bt.count > 1 ? nwg / bt.count : nwg) / tiles_n);
        const int full = (tiles_n / NG) * NG, per_group = tiles_m * NG;
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
    }

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (scalar: the LDS-DMA destinations are M0 values)
    const int wm = (wave / WN) * TM, wn = (wave % WN) * TN;

    // A staging: chunk i of a thread = row (tid + i*NT) / 8, 16-byte chunk (tid + i*NT) % 8 of the tile row; rows past
    // M read a clamped in-bounds row (they only feed outputs that are never stored)
    const char *a_base = reinterpret_cast<const char *>(A + m0 * lda);
    const char *b_base = reinterpret_cast<const char *>(B + (int64_t)n0 * K);
    unsigned a_off[CH];
    int a_lds[CH];
#pragma unroll
    for (int i = 0; i < CH; i++) {
        const int idx = tid + i * NT, row = idx >> 3, ch = idx & 7;
        const int64_t gm = m0 + row;
        a_off[i] = (unsigned)((((gm < M ? gm : M - 1) - m0) * lda + ch * 4) * 4);
        // k-group c = ch/2 (8 columns): hi halves in 16-byte slot {0,1,4,5}[c], lo halves two slots further
        a_lds[i] = row * ROWB + (((ch >> 2) * 4 + ((ch >> 1) & 1)) * 16) + (ch & 1) * 8;
    }
    auto ld_a = [&](int i) {
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(a_base), 0, 0xffffffff, 0x00020000);
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, a_off[i], 0, 0));
    };
    // W staging needs no conversion, so it bypasses the registers: `buffer_load_dwordx4 ... lds` (LDS DMA) writes the 64
    // lanes' 16 bytes to 1 KB of consecutive LDS = 8 tile rows of 128 B.  Instruction i of wave w brings rows
    // (4w + i)*8 .. +7; lane l supplies row l/8 and fetches the GLOBAL chunk (l%8) ^ x(row), x = (row/2) % 8, so that
    // chunk c of row r sits at slot c ^ x(r): the fragment reads below (32 consecutive rows, same c) then touch all
    // 16 sixteen-byte slots of the 256-byte bank window exactly once per 16-lane group.
    unsigned b_voff[CH];              // (CHB <= CH entries used; a template-dependent array size captured by the lambdas below loses hipcc the host stub)
#pragma unroll
    for (int i = 0; i < CHB; i++) {
        const int row = (wave * CHB + i) * 8 + (lane >> 3), gn = n0 + row;
        b_voff[i] = (unsigned)(((int64_t)((gn < N ? gn : N - 1) - n0) * K * 4) + (((lane & 7) ^ wswz(row)) * 16));
    }
    auto dma_b = [&](int stage) {
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(b_base), 0, 0xffffffff, 0x00020000);
#pragma unroll
        for (int i = 0; i < CHB; i++)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(
                r, (__attribute__((address_space(3))) void *)(ldsb + stage * STAGEB + BM * ROWB + (wave * CHB + i) * 8 * WROWB), 16,
                b_voff[i], 0, 0, 0);
    };
    // ASPLIT: the same DMA for A (rows past M are clamped to the last row: they only feed outputs that are never stored)
    const char *a_dbase = reinterpret_cast<const char *>(A + m0 * lda);
    unsigned a_voff[CH];
#pragma unroll
    for (int i = 0; i < CH; i++) {
        const int row = (wave * CH + i) * 8 + (lane >> 3);
        const int64_t gm = m0 + row;
        a_voff[i] = (unsigned)((((gm < M ? gm : M - 1) - m0) * lda * 4) + (((lane & 7) ^ wswz(row)) * 16));
    }
    auto dma_a = [&](int stage) {
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(a_dbase), 0, 0xffffffff, 0x00020000);
#pragma unroll
        for (int i = 0; i < CH; i++)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(
                r, (__attribute__((address_space(3))) void *)(ldsb + stage * STAGEB + (wave * CH + i) * 8 * WROWB), 16, a_voff[i], 0, 0, 0);
    };
    f32x4 ra[CH];
    // Range guard of the split arithmetic: |a_scale * x| must stay below fp16's largest finite value.  Every A value passes
    // through this thread's registers exactly once on its way into LDS; two v_max3_f32 per chunk keep the running maximum of
    // |x|, and a launch that saw a value outside the range raises the caller's flag word (locov_hip.h: `overflow`) -- the
    // outputs fed by that value are inf / NaN, the flag is how the host learns of it without looking at them.
    float amax = 0.f;
    auto st_a = [&](int i, int stage) {
        u32x2 hi, lo;
        amax = fmaxf(fmaxf(amax, fabsf(ra[i][0])), fabsf(ra[i][1]));
        amax = fmaxf(fmaxf(amax, fabsf(ra[i][2])), fabsf(ra[i][3]));
        split4(ra[i], a_scale, hi, lo);
        char *p = ldsb + stage * STAGEB + a_lds[i];
        *reinterpret_cast<u32x2 *>(p) = hi;
        *reinterpret_cast<u32x2 *>(p + 32) = lo;
    };

    f32x4 acc[MB][NB];
#pragma unroll
    for (int i = 0; i < MB; i++)
#pragma unroll
        for (int j = 0; j < NB; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // v_mfma_f32_16x16x32_f16 (it runs ~15 % faster than the 32x32x16 form under the chip's power limit): lane l holds
    // row l%16 and the 8 halves of k-group l/16 of a 16-row block, for the WHOLE 32-wide K-tile.  One register set of
    // fragments, in four groups: GA0 / GA1 = A row blocks {0,1} / {2,3}, GB0 / GB1 = W column blocks {0,1} / {2,3}
    // (each [block][0 = hi, 1 = lo]).  A K-tile is four quarters of 12 MFMAs, one (GA, GB) pair each, visited so that
    // a group is free for the next tile's data as early as possible (see tile_step).
    f16x8 fa[MB][2], fb[NB][2];
    const int l16 = lane & 15, kg = lane >> 4;
    const int afo = l16 * ROWB + ((kg >> 1) * 4 + (kg & 1)) * 16;                   // A: hi; lo 32 B further
    int bfo[2];                                                                      // W: [hi / lo]
#pragma unroll
    for (int hl = 0; hl < 2; hl++) bfo[hl] = l16 * WROWB + (((2 * kg + hl) ^ wswz(l16)) * 16);
    auto rd_a = [&](int stage, int ga) {
        if constexpr (ASPLIT) {
            const char *As = ldsb + stage * STAGEB + (wm + ga * 32) * WROWB;
#pragma unroll
            for (int i = 0; i < 2; i++) {
                fa[2 * ga + i][0] = *reinterpret_cast<const f16x8 *>(As + i * 16 * WROWB + bfo[0]);
                fa[2 * ga + i][1] = *reinterpret_cast<const f16x8 *>(As + i * 16 * WROWB + bfo[1]);
            }
        } else {
            const char *As = ldsb + stage * STAGEB + (wm + ga * 32) * ROWB + afo;
#pragma unroll
            for (int i = 0; i < 2; i++) {
                fa[2 * ga + i][0] = *reinterpret_cast<const f16x8 *>(As + i * 16 * ROWB);
                fa[2 * ga + i][1] = *reinterpret_cast<const f16x8 *>(As + i * 16 * ROWB + 32);
            }
        }
    };
    auto rd_b = [&](int stage, int gb) {
        const char *Bs = ldsb + stage * STAGEB + BM * ROWB + (wn + gb * 32) * WROWB;
#pragma unroll
        for (int j = 0; j < 2; j++) {
            fb[2 * gb + j][0] = *reinterpret_cast<const f16x8 *>(Bs + j * 16 * WROWB + bfo[0]);
            fb[2 * gb + j][1] = *reinterpret_cast<const f16x8 *>(Bs + j * 16 * WROWB + bfo[1]);
        }
    };
    // MFMAs p0..p1-1 of the quarter (GA ga, GB gb): per 16x16 block hi.hi, hi.lo, lo.hi
    constexpr int NQM = 12;
    auto quarter = [&](int ga, int gb, int p0, int p1) {
#pragma unroll
        for (int p = 0; p < NQM; p++) {
            if (p < p0 || p >= p1) continue;
            const int t = p / 3, i = 2 * ga + t / 2, j = 2 * gb + t % 2, w = p % 3;
            if (LOCOV_SPLIT_ABLATE & 4) {        // keep the fragments alive without the matrix pipe
                asm volatile("" :: "v"(fa[i][w == 2 ? 1 : 0]), "v"(fb[j][w == 1 ? 1 : 0]));
                continue;
            }
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[i][w == 2 ? 1 : 0], fb[j][w == 1 ? 1 : 0], acc[i][j], 0, 0, 0);
        }
    };

#ifdef LOCOV_KTRACE
    unsigned long long kt_[7], kd_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, kprev_ = 0;
    const unsigned long long kc0_ = __builtin_amdgcn_s_memtime(), kr0_ = __builtin_amdgcn_s_memrealtime();   // shader clock / 100 MHz
#endif
    const int k_last = K - BK;                              // k0 of the last K-tile (K % BK == 0)
    // prologue: tile 0 -> LDS stage 0 (W by DMA), A tile 1 -> staging registers, fragments of the first quarter
    dma_b(0);
    if (LOCOV_SPLIT_ABLATE & 1) dma_b(1);                   // (ablation without staging: both stages hold real data)
    b_base += BK * 4;                                       // W is fetched ONE tile ahead: b_base addresses tile t+1
    int k_ptr = 0;
    if constexpr (ASPLIT) {
        dma_a(0);
        if (LOCOV_SPLIT_ABLATE & 1) dma_a(1);
        a_dbase += BK * 4;
        __builtin_amdgcn_s_waitcnt(0x0F70);                 // vmcnt(0)
    } else {
#pragma unroll
        for (int i = 0; i < CH; i++) ra[i] = ld_a(i);
#pragma unroll
        for (int i = 0; i < CH; i++) st_a(i, 0);
        k_ptr = BK < k_last ? BK : k_last;                  // the A tile a_base addresses (two ahead in the loop)
        a_base += (int64_t)k_ptr * 4;
#pragma unroll
        for (int i = 0; i < CH; i++) ra[i] = ld_a(i);
        __builtin_amdgcn_s_waitcnt(0x0F70 | CH);            // vmcnt(CH): everything but the A loads just issued has landed
    }
    __syncthreads();
    rd_a(0, 0);
    rd_b(0, 0);
    if (LOCOV_SPLIT_ABLATE & 2) {
        rd_a(0, 1);
        rd_b(0, 1);
    }
    __builtin_amdgcn_s_setprio(0);

    // One K-tile that has a successor, computing from LDS stage s (= the tile's parity).  Quarters: (GA0,GBx) (GA0,GBy)
    // (GA1,GBy) | barrier | (GA1,GBx) with x = s, y = 1 - s: the tile ends on GBx, so GA0 and GBy are free before its
    // last quarter and receive the next tile's data behind the barrier -- and the next tile (parity 1 - s) starts on exactly
    // (GA0, GBy).  The first half also carries the staging: the W rows of tile t+1 by DMA, the A chunks of tile t+1 from
    // registers (split here) with the refill loads of tile t+2.
    auto tile_step = [&](const int s, const int k0) __attribute__((always_inline)) {
        const int kn = k0 + 2 * BK < k_last ? k0 + 2 * BK : k_last;
        a_base += (int64_t)(kn - k_ptr) * 4;
        k_ptr = kn;
        const int x = s, y = s ^ 1;
        KSTAMP(0);
        if (!(LOCOV_SPLIT_ABLATE & 2)) {
            rd_b(s, y);
            rd_a(s, 1);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (!(LOCOV_SPLIT_ABLATE & 1)) dma_b(s ^ 1);
        b_base += BK * 4;
        if constexpr (ASPLIT) {
            if (!(LOCOV_SPLIT_ABLATE & 1)) dma_a(s ^ 1);
            a_dbase += BK * 4;
        }
        __builtin_amdgcn_sched_barrier(0);
        KSTAMP(1);
#pragma unroll
        for (int g = 0; g < CH; g++) {
            if constexpr (!ASPLIT) {
                st_a(g, s ^ 1);
                ra[g] = ld_a(g);
            }
            if (g < CH / 2)
                quarter(0, x, g * 2 * NQM / CH, (g + 1) * 2 * NQM / CH);
            else
                quarter(0, y, (g - CH / 2) * 2 * NQM / CH, (g - CH / 2 + 1) * 2 * NQM / CH);
            __builtin_amdgcn_sched_barrier(0);
        }
        KSTAMP(2);
        quarter(1, y, 0, NQM);
        __builtin_amdgcn_sched_barrier(0);
        KSTAMP(3);
        // the barrier sits three quarters into the tile (measured 1 % better than the middle: the DMA and the LDS stores
        // get more time, the two fragment groups read behind it are still a full quarter ahead of their use)
        __builtin_amdgcn_s_waitcnt(ASPLIT ? 0x0F70 : (0x0F70 | CH));   // vmcnt(CH): the DMA is older than the CH A loads
        KSTAMP(4);
        __syncthreads();
        KSTAMP(5);
        if (!(LOCOV_SPLIT_ABLATE & 2)) {
            rd_a(s ^ 1, 0);
            rd_b(s ^ 1, y);
        }
        __builtin_amdgcn_sched_barrier(0);
        quarter(1, x, 0, NQM);
        __builtin_amdgcn_sched_barrier(0);
        KSTAMP(6);
#ifdef LOCOV_KTRACE
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(kt_[0]), "+s"(kt_[1]), "+s"(kt_[2]), "+s"(kt_[3]), "+s"(kt_[4]), "+s"(kt_[5]), "+s"(kt_[6])::"memory");
#pragma unroll
        for (int i = 0; i < 6; i++) kd_[i] += kt_[i + 1] - kt_[i];
        if (kprev_ != 0) kd_[6] += kt_[0] - kprev_;
        kprev_ = kt_[6];
        kd_[7] += 1;
#endif
    };
    auto last_tile = [&](const int s) __attribute__((always_inline)) {
        const int x = s, y = s ^ 1;
        rd_b(s, y);
        rd_a(s, 1);
        quarter(0, x, 0, NQM);
        quarter(0, y, 0, NQM);
        quarter(1, y, 0, NQM);
        quarter(1, x, 0, NQM);
    };
    int k0 = 0;
    for (; k0 + BK < k_last; k0 += 2 * BK) {
        tile_step(0, k0);
        tile_step(1, k0 + BK);
    }
    // residual rows of the first NPRE row groups are requested before the last K-tile (gemm_nt.hip)
    constexpr int NPRE = LOCOV_RES_PREFETCH;
    constexpr int LPR = TN / 4, RPI = 64 / LPR, NIT = TM / RPI;
    f32x4 res_pre[NPRE > 0 ? NPRE : 1];
    const int c4 = (lane % LPR) * 4, rr = lane / LPR;
    const int n = n0 + wn + c4;
    const bool n_ok = n < N;
    const int64_t rows_here = M - m0 < BM ? M - m0 : BM;
    const unsigned nrec = (unsigned)(rows_here * ldc * 4);
    const unsigned voff = (unsigned)(((int64_t)(wm + rr) * ldc + n) * 4);
    const unsigned vstep = (unsigned)(RPI * ldc * 4);
    const __amdgpu_buffer_rsrc_t r_out =
        __builtin_amdgcn_make_buffer_rsrc(SEGSUM ? const_cast<float *>(epi.residual) : Cout + m0 * ldc, 0, nrec, 0x00020000);
    // (SEGSUM: the residual descriptor spans the whole position-major tensor, rows are gathered by a computed offset)
    const __amdgpu_buffer_rsrc_t r_res = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(SEGSUM ? epi.residual : epi.residual ? epi.residual + m0 * ldc : Cout + m0 * ldc), 0,
        SEGSUM ? (unsigned)(M * ldc * 4) : nrec, 0x00020000);
    auto prefetch_residual = [&]() __attribute__((always_inline)) {
        if (SEGSUM || NPRE == 0 || !epi.residual || !n_ok) return;
#pragma unroll
        for (int it = 0; it < NPRE; it++)
            res_pre[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_res, voff + it * vstep, 0, 2));
    };
    if (k0 < k_last) {
        tile_step(0, k0);
        prefetch_residual();
        last_tile(1);
    } else {
        prefetch_residual();
        last_tile(0);
    }

#ifdef LOCOV_KTRACE
    if (ASPLIT && lane == 0 && blockIdx.x >= 3000 && blockIdx.x < 3064)
#pragma unroll
        for (int i = 0; i < 8; i++) atomicAdd(&g_ktrace[wave * 16 + i], kd_[i]);
    const unsigned long long kw2_ = __builtin_amdgcn_s_memtime();
    if (ASPLIT && lane == 0 && blockIdx.x >= 3000 && blockIdx.x < 3064) {
        atomicAdd(&g_ktrace[wave * 16 + 8], kw2_ - kc0_);
        atomicAdd(&g_ktrace[wave * 16 + 9], __builtin_amdgcn_s_memrealtime() - kr0_);
        atomicAdd(&g_ktrace[wave * 16 + 10], kc0_ - kw0_);          // entry -> first DMA issued (index math)
    }
#endif
    __builtin_amdgcn_s_setprio(3);
    if (!ASPLIT && overflow != nullptr && amax * a_scale >= 65504.f) atomicOr(overflow, 1u);
    // Epilogue (C/D layout of the 16x16 MFMA: col = lane&15, row = 4*(lane>>4) + reg): re-lay the wave's sub-tile out
    // through LDS, 16 bytes per lane and row-contiguous from there.
    const bool relu = (epi.flags & LOCOV_EPI_RELU) != 0;
    const bool out_split = (epi.flags & LOCOV_EPI_OUT_SPLIT) != 0, res_split = (epi.flags & LOCOV_EPI_RES_SPLIT) != 0;
    const bool odd_lane = (lane & 1) != 0;
    const float inv_a_scale = 1.f / a_scale;
    float omax = 0.f;                         // range guard of a split-layout OUTPUT: max |finished value|
    static_assert(NW * TM * EPS * 4 <= LDSB && LDSB <= 160 * 1024, "epilogue staging must fit the LDS");
    float *ep = reinterpret_cast<float *>(lds) + wave * (TM * EPS);
    const int64_t seg_q0 = SEGSUM ? m0 / ss.seg : 0;                 // first ROI of the tile, and the position its first row holds
    const int seg_r0 = SEGSUM ? (int)(m0 - seg_q0 * ss.seg) : 0;
    const float seg_inv = SEGSUM ? 1.0f / (float)ss.seg : 0.f;
    const __amdgpu_buffer_rsrc_t r_msk = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(EMASK && epi.mask ? epi.mask + m0 * ldc : Cout + m0 * ldc), 0, nrec, 0x00020000);
    auto tail = [&](auto full_tag) __attribute__((always_inline)) {
        constexpr bool FULL = decltype(full_tag)::value;
        f32x4 res[NIT];
        if (epi.residual && n_ok) {
#pragma unroll
            for (int it = 0; it < NIT; it++) {
                if (SEGSUM) {
                    // row m = m0 + t of the tile is (ROI q, position pos): small-integer division by a float reciprocal
                    // (t + m0 % seg < 256, exact), 32-bit offsets (M * N * 4 < 2^32 is checked by the launcher)
                    const int u = seg_r0 + wm + it * RPI + rr;
                    const int dq = (int)(((float)u + 0.5f) * seg_inv), pos = u - dq * ss.seg;
                    const unsigned row = ss.R ? (unsigned)pos * (unsigned)ss.R + (unsigned)(seg_q0 + dq)     // position-major residual
                                              : (unsigned)(m0 + wm + it * RPI + rr);                        // ROI-major residual
                    const unsigned off = wm + it * RPI + rr < rows_here ? (row * (unsigned)ldc + (unsigned)n) * 4u : 0xffffffffu;
                    res[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_res, off, 0, 2));
                } else if (it < NPRE && NPRE > 0)
                    res[it] = res_pre[it];
                else
                    res[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                            r_res, FULL ? voff : voff + it * vstep, FULL ? it * vstep : 0u, 2));
            }
        }
        f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
        if (n_ok && epi.scale) sc = *reinterpret_cast<const f32x4 *>(epi.scale + n);
        sc *= out_scale;                                  // undo the operand scales
        if (n_ok && epi.shift) sh = *reinterpret_cast<const f32x4 *>(epi.shift + n);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < MB; i++)
#pragma unroll
            for (int j = 0; j < NB; j++)
#pragma unroll
                for (int r = 0; r < 4; r++) ep[(i * 16 + 4 * kg + r) * EPS + j * 16 + l16] = acc[i][j][r];
        __syncthreads();
        if (n_ok) {
#pragma unroll
            for (int it = 0; it < NIT; it++) {
                f32x4 v = *reinterpret_cast<const f32x4 *>(ep + (it * RPI + rr) * EPS + c4);
                v = v * sc + sh;
                if (epi.residual) v += res_split ? unsplit4(res[it], odd_lane, inv_a_scale) : res[it];
                if (relu) {
                    v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f);
                    v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
                }
                if (EMASK && !SEGSUM && epi.mask) {
                    const f32x4 mk = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                                   r_msk, FULL ? voff : voff + it * vstep, FULL ? it * vstep : 0u, 2));
                    v[0] = mk[0] > 0.f ? v[0] : 0.f; v[1] = mk[1] > 0.f ? v[1] : 0.f;
                    v[2] = mk[2] > 0.f ? v[2] : 0.f; v[3] = mk[3] > 0.f ? v[3] : 0.f;
                }
                if (EMASK && !SEGSUM && (FULL || wm + it * RPI + rr < (int)rows_here)) {   // (the gradient this launch writes is a later GEMM's operand: its max |.|)
                    omax = fmaxf(fmaxf(omax, fabsf(v[0])), fabsf(v[1]));
                    omax = fmaxf(fmaxf(omax, fabsf(v[2])), fabsf(v[3]));
                }
                if (SEGSUM)
                    *reinterpret_cast<f32x4 *>(ep + (it * RPI + rr) * EPS + c4) = v;    // finished value back in place
                else if (out_split) {
                    omax = fmaxf(fmaxf(omax, fabsf(v[0])), fabsf(v[1]));
                    omax = fmaxf(fmaxf(omax, fabsf(v[2])), fabsf(v[3]));
                    __builtin_amdgcn_raw_buffer_store_b128(split4_pair(v, odd_lane, a_scale), r_out, FULL ? voff : voff + it * vstep,
                                                           FULL ? it * vstep : 0u, LOCOV_STORE_AUX);
                } else
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r_out, FULL ? voff : voff + it * vstep,
                                                           FULL ? it * vstep : 0u, LOCOV_STORE_AUX);  // aux 2 = nt: streamed once
            }
        }
        if (!SEGSUM && out_split && overflow != nullptr && omax * a_scale >= 65504.f) atomicOr(overflow, 1u);
        if (EMASK && !SEGSUM && epi.amax_out != nullptr) amax_fold(epi.amax_out, omax);
        if (SEGSUM) {
            // column sums per ROI: thread t owns column t % 128 and the ROI slots {t / 128, t / 128 + 2} of this tile
            __syncthreads();
            const float *base = reinterpret_cast<const float *>(lds);
            const int col = tid & (BN - 1), cw = (col >> 6) * (TM * EPS) + (col & 63);
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const int slot = (tid >> 7) + 2 * k;
                int lo = slot * ss.seg - seg_r0, hi = lo + ss.seg;            // tile rows of ROI q0 + slot
                lo = lo > 0 ? lo : 0;
                hi = hi < (int)rows_here ? hi : (int)rows_here;
                // rows lo..hi-1 of column `col`: the part in each 64-row half of the tile is a fixed-stride walk through that
                // half's staging area; four independent partial sums (fixed order) keep the LDS reads in flight
                float ps[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int half = 0; half < 2; half++) {
                    const int a = lo > half * 64 ? lo : half * 64, b = hi < half * 64 + 64 ? hi : half * 64 + 64;
                    if (a >= b) continue;
                    const float *p = base + half * (WN * TM * EPS) + (a & 63) * EPS + cw;
                    int r = a;
                    for (; r + 4 <= b; r += 4, p += 4 * EPS) {
                        ps[0] += p[0];
                        ps[1] += p[EPS];
                        ps[2] += p[2 * EPS];
                        ps[3] += p[3 * EPS];
                    }
                    for (; r < b; r++, p += EPS) ps[r & 3] += p[0];
                }
                const float acc_s = (ps[0] + ps[1]) + (ps[2] + ps[3]);
                if (n0 + col < N) ss.partial[((m0 / BM) * 4 + slot) * (int64_t)N + n0 + col] = acc_s;
            }
        }
    };
    if (LOCOV_SPLIT_ABLATE & 8) {
#pragma unroll
        for (int i = 0; i < MB; i++)
#pragma unroll
            for (int j = 0; j < NB; j++) asm volatile("" :: "v"(acc[i][j]));
    } else if (rows_here == BM)
        tail(std::true_type{});
    else
        tail(std::false_type{});
#ifdef LOCOV_KTRACE
    {
        const unsigned long long ke_ = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long kf_ = __builtin_amdgcn_s_memtime();
        if (ASPLIT && lane == 0 && blockIdx.x >= 3000 && blockIdx.x < 3064) {
            atomicAdd(&g_ktrace[wave * 16 + 11], ke_ - kw2_);          // epilogue, stores issued
            atomicAdd(&g_ktrace[wave * 16 + 12], kf_ - ke_);           // stores drained
            atomicAdd(&g_ktrace[wave * 16 + 13], 1ull);
        }
    }
#endif
}

// W [rows, K] fp32 (row pitch ld) -> the split layout of s*W: per row and group of 8 columns, 8 hi halves then 8 lo halves
__global__ __launch_bounds__(256) void split_pack_kernel(const float *__restrict__ w, int64_t rows, int K, int64_t ld,
                                                         float s, _Float16 *__restrict__ out, unsigned *overflow)
{
    const int64_t total = rows * K;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / K;
        const int k = (int)(i - r * K), t = k >> 5, e = k & 31;
        const float x = w[r * ld + k] * s;
        if (overflow != nullptr && fabsf(x) >= 65504.f) atomicOr(overflow, 1u);     // a re-used scale no longer covers the data
        const _Float16 h = (_Float16)x;
        const _Float16 l = (_Float16)(x - (float)h);
        _Float16 *o = out + r * 2 * K + t * 64 + (e >> 3) * 16 + (e & 7);
        o[0] = h;
        o[8] = l;
    }
}

// The same packing, one group of 8 columns per thread: two 16-byte loads, one 32-byte run [8 hi | 8 lo] out -- consecutive threads
// write consecutive runs.  Same arithmetic per element (hi = f16(s x), lo = f16(s x - hi)): the same bits as split_pack_kernel,
// which stays for row pitches / pointers that are not 16-byte aligned.  (The element-wise form spent 22 us on a 2 M-element
// weight -- an integer division and two 2-byte scattered stores per element; a training step packs 26 weights.)
__global__ __launch_bounds__(256) void split_pack8_kernel(const float *__restrict__ w, int64_t rows, int K8, int64_t ld, float s,
                                                          u32x4 *__restrict__ out, unsigned *overflow)
{
    const int64_t total = rows * K8;
    bool over = false;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / K8;
        const int g = (int)(i - r * K8);
        const f32x4 *src = reinterpret_cast<const f32x4 *>(w + r * ld + g * 8);
        const f32x4 a = src[0], b = src[1];
        _Float16 h[8], l[8];
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const float x = (e < 4 ? a[e] : b[e - 4]) * s;
            over |= fabsf(x) >= 65504.f;
            h[e] = (_Float16)x;
            l[e] = (_Float16)(x - (float)h[e]);
        }
        u32x4 ho, lo;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            ho[e] = (unsigned)__builtin_bit_cast(unsigned short, h[2 * e]) | ((unsigned)__builtin_bit_cast(unsigned short, h[2 * e + 1]) << 16);
            lo[e] = (unsigned)__builtin_bit_cast(unsigned short, l[2 * e]) | ((unsigned)__builtin_bit_cast(unsigned short, l[2 * e + 1]) << 16);
        }
        out[2 * i] = ho;                                   // (row r, group g) sits at 32-byte slot r * K8 + g = i: the layout is dense
        out[2 * i + 1] = lo;
    }
    if (overflow != nullptr && over) atomicOr(overflow, 1u);        // a re-used scale no longer covers the data
}

// out[q, n] = (sum over the one or two M-tiles ROI q's rows fall into of its partial) * inv_seg
__global__ __launch_bounds__(256) void segsum_finish_kernel(const float *__restrict__ partial, int64_t R, int N, int seg,
                                                            float inv_seg, float *__restrict__ out)
{
    const int64_t total = R * (N / 4);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t q = i / (N / 4);
        const int n = (int)(i - q * (N / 4)) * 4;
        const int64_t t0 = (q * seg) / BM, t1 = (q * seg + seg - 1) / BM;
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
        for (int64_t t = t0; t <= t1; t++) {
            const int64_t slot = q - (t * BM) / seg;
            a += *reinterpret_cast<const f32x4 *>(partial + (t * 4 + slot) * N + n);
        }
        *reinterpret_cast<f32x4 *>(out + q * N + n) = a * inv_seg;
    }
}

// Operand scale of a tensor whose range is only known on the device (gradients): scale_out = {s, 1/s, amax bits, arrivals} with
// s = 2^(target_log2 - 1 - floor(log2(max |x|))), i.e. max |s x| in [2^(target-1), 2^target); an all-zero (or empty) tensor
// gets s = 1.  ONE kernel behind a 16-byte memset: every workgroup folds its max into the word with one atomic and takes an
// arrival ticket; the workgroup that draws the last ticket turns the max into the scale (the others' atomics are complete by
// then: each was issued before its workgroup's ticket, and both are device-scope atomics on the same cache line's L2).
__device__ __forceinline__ void split_scale_finish(float *out, float target_log2)
{
    const float amax = __uint_as_float(__hip_atomic_load(reinterpret_cast<unsigned *>(out) + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    float s = 1.f;
    if (amax > 0.f && amax < 3.0e38f) {
        int e;
        frexpf(amax, &e);                                 // amax = f * 2^e, f in [0.5, 1)  ->  amax <= 2^e
        s = ldexpf(1.f, (int)target_log2 - e);
    }
    out[0] = s;
    out[1] = 1.f / s;
}

__global__ __launch_bounds__(256) void split_amax_kernel(const float *__restrict__ x, int64_t n4, float *__restrict__ out, float target_log2)
{
    float m = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const f32x4 v = reinterpret_cast<const f32x4 *>(x)[i];
        m = fmaxf(fmaxf(m, fabsf(v[0])), fabsf(v[1]));
        m = fmaxf(fmaxf(m, fabsf(v[2])), fabsf(v[3]));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    __shared__ float wmax[4];
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {                    // ONE atomic per workgroup (thousands of same-address atomics serialise)
        unsigned *w = reinterpret_cast<unsigned *>(out);
        m = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
        if (m > 0.f) atomicMax(w + 2, __float_as_uint(m));                             // non-negative floats order like their bits
        __threadfence();
        if (atomicAdd(w + 3, 1u) == gridDim.x - 1) {
            split_scale_finish(out, target_log2);
            w[2] = 0u;                         // the scratch words are zero again: the 16 bytes can take the next reduction as they are
            w[3] = 0u;
        }
    }
}

// max_i (mul_i * max |x_i|) of up to LOCOV_AMAX_BOUND_MAX small tensors folded into word 2 of a zeroed operand-scale slot
struct AmaxBoundList {
    const float *x[LOCOV_AMAX_BOUND_MAX];
    int64_t n4[LOCOV_AMAX_BOUND_MAX];
    float mul[LOCOV_AMAX_BOUND_MAX];
    unsigned first_block[LOCOV_AMAX_BOUND_MAX + 1];
    int count;
};

__global__ __launch_bounds__(256) void amax_bound_kernel(AmaxBoundList L, float *__restrict__ slot)
{
    int ti = 0;
    while (ti + 1 < L.count && blockIdx.x >= L.first_block[ti + 1]) ti++;
    const unsigned blocks = L.first_block[ti + 1] - L.first_block[ti];
    const f32x4 *x = reinterpret_cast<const f32x4 *>(L.x[ti]);
    float m = 0.f;
    for (int64_t i = (int64_t)(blockIdx.x - L.first_block[ti]) * 256 + threadIdx.x; i < L.n4[ti]; i += (int64_t)blocks * 256) {
        const f32x4 v = x[i];
        m = fmaxf(fmaxf(m, fabsf(v[0])), fabsf(v[1]));
        m = fmaxf(fmaxf(m, fabsf(v[2])), fabsf(v[3]));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    __shared__ float wmax[4];
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {                    // ONE atomic per workgroup
        m = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3])) * L.mul[ti];
        if (m > 0.f) atomicMax(reinterpret_cast<unsigned *>(slot) + 2, __float_as_uint(m));      // non-negative floats order like their bits
    }
}

int launch_gemm_split(const float *A, int64_t lda, const void *Wsplit, float *C, int64_t ldc, int64_t M, int N, int K,
                      const Epilogue &epi, float a_scale, float w_scale, hipStream_t s, const char *what, const Batch &bt,
                      unsigned *overflow, const float *a_scale_dev)
{
    if (a_scale_dev) a_scale = 1.f;
    if (!(a_scale > 0.f) || !(w_scale > 0.f)) return set_error(LOCOV_ERR_INVALID_ARG, "%s: operand scales must be positive", what);
    if (epi.mask && ((uintptr_t)epi.mask % 16 != 0 || bt.count > 1))
        return set_error(LOCOV_ERR_UNSUPPORTED, "%s: the mask must be 16-byte aligned (and batched launches take none)", what);
    if (K % BK != 0 || K < BK) return set_error(LOCOV_ERR_UNSUPPORTED, "%s: K must be a positive multiple of %d", what, BK);
    if (N % 4 != 0 || ldc % 4 != 0 || lda % 4 != 0 || (uintptr_t)A % 16 != 0 || (uintptr_t)Wsplit % 16 != 0 ||
        (uintptr_t)C % 16 != 0 || (epi.residual && (uintptr_t)epi.residual % 16 != 0) ||
        (epi.scale && (uintptr_t)epi.scale % 16 != 0) || (epi.shift && (uintptr_t)epi.shift % 16 != 0))
        return set_error(LOCOV_ERR_UNSUPPORTED, "%s: N, lda, ldc must be multiples of 4 and every pointer 16-byte aligned", what);
    if (bt.count > 1 && epi.residual) return set_error(LOCOV_ERR_UNSUPPORTED, "%s: batched launches take no residual", what);
    if ((epi.flags & (LOCOV_EPI_OUT_SPLIT | LOCOV_EPI_RES_SPLIT)) && (N % 8 != 0 || ldc % 8 != 0 || epi.mask || a_scale_dev || bt.count > 1))
        return set_error(LOCOV_ERR_UNSUPPORTED, "%s: a split-layout output / residual needs N and ldc to be multiples of 8, no mask, no device scale, no batch", what);
    const int count = bt.count > 1 ? bt.count : 1;
    // tile choice: both operands pre-split and enough 256 x 256 tiles to fill the chip several times -> gemm_split_big.hip
    if (gemm_split_big_applicable(lda, ldc, M, N, K, epi, bt, a_scale_dev))
        return launch_gemm_split_big(A, lda, Wsplit, C, ldc, M, N, K, epi, a_scale, w_scale, s, what, bt, overflow);
    const int bm = BM;
    const int64_t tiles = ceil_div(M, bm) * ceil_div(N, BN) * count;
    if (tiles > 0x7fffffffLL) return set_error(LOCOV_ERR_INVALID_ARG, "%s: problem too large", what);
    if ((int64_t)bm * (lda > ldc ? lda : ldc) * 4 > 0x7fffffffLL)
        return set_error(LOCOV_ERR_INVALID_ARG, "%s: row pitch too large for 32-bit tile offsets", what);
    const int trec = timing_begin(s, 5, 2.0 * (double)M * N * K * count,       // class 5: the 128x128 split-operand GEMM
                                  4.0 * count * ((double)M * K + (double)N * K + (double)M * N * (epi.residual ? 2.0 : 1.0)));
    const float os = 1.f / (a_scale * w_scale);
    const float *Bw = reinterpret_cast<const float *>(Wsplit);
    const SegSum ss0{0, 0, nullptr};
#define LOCOV_LAUNCH_SPLIT(EM, AS, WMV)                                                                                         \
    hipLaunchKernelGGL((gemm_split_kernel<false, EM, AS, WMV>), dim3((unsigned)tiles), dim3(64 * WMV * WN), 0, s, A, lda, Bw, C, \
                       ldc, M, N, K, epi, bt, a_scale, os, ss0, overflow, a_scale_dev)
    if (epi.flags & LOCOV_GEMM_A_SPLIT) {
        // (amax_out too: the pre-split instance carries no fold, a silently untouched slot would read as scale 1 downstream)
        if (epi.mask || a_scale_dev || epi.amax_out)
            return set_error(LOCOV_ERR_UNSUPPORTED, "%s: pre-split A takes no mask / device scale / amax_out", what);
        LOCOV_LAUNCH_SPLIT(false, true, 2);
    } else if (epi.mask || epi.amax_out)
        LOCOV_LAUNCH_SPLIT(true, false, 2);
    else
        LOCOV_LAUNCH_SPLIT(false, false, 2);
#undef LOCOV_LAUNCH_SPLIT
    timing_end(trec, s);
    return check_launch(what);
}

#ifdef LOCOV_KTRACE
}  // namespace locov
extern "C" int locov_dbg_ktrace(unsigned long long *dst, int reset)
{
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(dst, HIP_SYMBOL(locov::g_ktrace), sizeof(unsigned long long) * 64) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[64] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(locov::g_ktrace), z, sizeof z) != hipSuccess) return -1;
    }
    return 0;
}
namespace locov {
#endif

// The SEGSUM form + its finishing pass: mean over the `seg` rows of every ROI of relu(scale * (x . W^T) + shift + residual)
int launch_gemm_split_segmean(const float *A, int64_t lda, const void *Wsplit, int64_t M, int N, int K, const Epilogue &epi,
                              int seg, float a_scale, float w_scale, float *partial, float *out, hipStream_t s, const char *what,
                              unsigned *overflow)
{
    if (!(a_scale > 0.f) || !(w_scale > 0.f)) return set_error(LOCOV_ERR_INVALID_ARG, "%s: operand scales must be positive", what);
    if (K % BK != 0 || K < BK) return set_error(LOCOV_ERR_UNSUPPORTED, "%s: K must be a positive multiple of %d", what, BK);
    // (a 128-row tile may touch at most the four ROIs its partial slots provide for: seg >= 43)
    if (seg < 43 || seg > BM || M % seg != 0)
        return set_error(LOCOV_ERR_UNSUPPORTED, "%s: M must be a multiple of seg, 43 <= seg <= %d (got %d)", what, BM, seg);
    if (N % 4 != 0 || lda % 4 != 0 || (uintptr_t)A % 16 != 0 || (uintptr_t)Wsplit % 16 != 0 || (uintptr_t)partial % 16 != 0 ||
        (uintptr_t)out % 16 != 0 || (epi.residual && (uintptr_t)epi.residual % 16 != 0) ||
        (epi.scale && (uintptr_t)epi.scale % 16 != 0) || (epi.shift && (uintptr_t)epi.shift % 16 != 0))
        return set_error(LOCOV_ERR_UNSUPPORTED, "%s: N, lda must be multiples of 4 and every pointer 16-byte aligned", what);
    if (!epi.residual) return set_error(LOCOV_ERR_INVALID_ARG, "%s: the residual is required", what);
    // 256 x 256 tile (gemm_split_big.hip): every tile addresses A, the residual and its partial sums from a 64-bit tile base, so
    // M * N * 4 may pass 2^32 (12 000 proposals: 4.8 GB); the 128 x 128 form below keeps 32-bit residual offsets
    if (gemm_split_big_segmean_applicable(lda, M, N, K, epi, seg))
        return launch_gemm_split_big_segmean(A, lda, Wsplit, M, N, K, epi, seg, a_scale, w_scale, partial, out, s, what, overflow);
    const int64_t tiles_m = ceil_div(M, BM), tiles = tiles_m * ceil_div(N, BN);
    if (tiles > 0x7fffffffLL || (double)M * N * 4 > 4294967295.0 || (int64_t)BM * lda * 4 > 0x7fffffffLL)
        return set_error(LOCOV_ERR_INVALID_ARG, "%s: problem too large for 32-bit residual offsets", what);
    const int trec = timing_begin(s, 5, 2.0 * (double)M * N * K, 4.0 * ((double)M * K + (double)N * K + (double)M * N));
    if (epi.flags & LOCOV_GEMM_A_SPLIT)
        hipLaunchKernelGGL((gemm_split_kernel<true, false, true>), dim3((unsigned)tiles), dim3(NT), 0, s, A, lda,
                           reinterpret_cast<const float *>(Wsplit), static_cast<float *>(nullptr), (int64_t)N, M, N, K, epi, Batch{1, 0, 0, 0},
                           a_scale, 1.f / (a_scale * w_scale),
                           SegSum{seg, (epi.flags & LOCOV_SEGMEAN_RES_ROI_MAJOR) ? (int64_t)0 : M / seg, partial}, overflow,
                           static_cast<const float *>(nullptr));
    else
        hipLaunchKernelGGL((gemm_split_kernel<true, false, false>), dim3((unsigned)tiles), dim3(NT), 0, s, A, lda,
                           reinterpret_cast<const float *>(Wsplit), static_cast<float *>(nullptr), (int64_t)N, M, N, K, epi, Batch{1, 0, 0, 0},
                           a_scale, 1.f / (a_scale * w_scale),
                           SegSum{seg, (epi.flags & LOCOV_SEGMEAN_RES_ROI_MAJOR) ? (int64_t)0 : M / seg, partial}, overflow,
                           static_cast<const float *>(nullptr));
    timing_end(trec, s);
    int rc = check_launch(what);
    if (rc) return rc;
    const int64_t R = M / seg, total = R * (N / 4);
    const unsigned blocks = (unsigned)(ceil_div(total, 256) < 65536 ? ceil_div(total, 256) : 65536);
    hipLaunchKernelGGL(segsum_finish_kernel, dim3(blocks), dim3(256), 0, s, partial, R, N, seg, 1.f / (float)seg, out);
    return check_launch(what);
}

}  // namespace locov

using namespace locov;

extern "C" {

int locov_split_f16x2_pack(const float *w, int64_t rows, int K, int64_t ld, float w_scale, void *out, unsigned *overflow,
                           locov_stream_t stream)
{
    LOCOV_REQUIRE(w_scale > 0.f, "locov_split_f16x2_pack: w_scale must be positive");
    LOCOV_REQUIRE(rows >= 0 && K > 0 && K % BK == 0 && ld >= K, "locov_split_f16x2_pack: K must be a multiple of %d, ld >= K", BK);
    if (rows == 0) return LOCOV_OK;
    LOCOV_REQUIRE(w && out, "locov_split_f16x2_pack: null pointer");
    const int64_t total = rows * K;
    if (ld % 4 == 0 && (uintptr_t)w % 16 == 0 && (uintptr_t)out % 16 == 0) {
        const int64_t groups = total / 8;
        const unsigned blocks = (unsigned)(ceil_div(groups, 256) < 16384 ? ceil_div(groups, 256) : 16384);
        hipLaunchKernelGGL(split_pack8_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), w, rows, K / 8, ld, w_scale,
                           reinterpret_cast<u32x4 *>(out), overflow);
        return check_launch("locov_split_f16x2_pack");
    }
    const unsigned blocks = (unsigned)(ceil_div(total, 256) < 65536 ? ceil_div(total, 256) : 65536);
    hipLaunchKernelGGL(split_pack_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), w, rows, K, ld, w_scale,
                       reinterpret_cast<_Float16 *>(out), overflow);
    return check_launch("locov_split_f16x2_pack");
}

int locov_gemm_nt_f32_split(const float *x, int64_t lda, const void *W_split, const float *scale, const float *shift,
                            const float *residual, float *y, int64_t ldc, int64_t M, int N, int K, unsigned flags,
                            float x_scale, float w_scale, unsigned *overflow, locov_stream_t stream)
{
    LOCOV_REQUIRE(M >= 0 && N > 0 && K > 0, "locov_gemm_nt_f32_split: bad shape M=%lld N=%d K=%d", (long long)M, N, K);
    if (M == 0) return LOCOV_OK;
    LOCOV_REQUIRE(x && W_split && y, "locov_gemm_nt_f32_split: null pointer");
    LOCOV_REQUIRE(lda >= K && ldc >= N, "locov_gemm_nt_f32_split: lda < K or ldc < N");
    Epilogue epi{scale, shift, residual, flags};
    return launch_gemm_split(x, lda, W_split, y, ldc, M, N, K, epi, x_scale, w_scale, as_stream(stream),
                             "locov_gemm_nt_f32_split", Batch{1, 0, 0, 0}, overflow);
}

static int split_scale_from_amax(const float *x, int64_t n, float target_log2, float *scale_out, bool zeroed, locov_stream_t stream)
{
    LOCOV_REQUIRE(n >= 0 && x && scale_out, "locov_split_scale_from_amax: bad arguments");
    LOCOV_REQUIRE((uintptr_t)x % 16 == 0 && n % 4 == 0, "locov_split_scale_from_amax: x must be 16-byte aligned, n a multiple of 4");
    hipStream_t s = as_stream(stream);
    if (!zeroed && hipMemsetAsync(scale_out, 0, 16, s) != hipSuccess)
        return set_error(LOCOV_ERR_LAUNCH, "locov_split_scale_from_amax: memset failed");
    const int64_t n4 = n / 4;
    const int64_t want = ceil_div(n4, 256 * 8);
    const unsigned blocks = (unsigned)(want < 1 ? 1 : want < 1024 ? want : 1024);       // (n == 0: one workgroup writes s = 1)
    hipLaunchKernelGGL(split_amax_kernel, dim3(blocks), dim3(256), 0, s, x, n4, scale_out, target_log2);
    return check_launch("locov_split_scale_from_amax");
}

int locov_split_scale_from_amax(const float *x, int64_t n, float target_log2, float *scale_out, locov_stream_t stream)
{
    return split_scale_from_amax(x, n, target_log2, scale_out, false, stream);
}

int locov_split_scale_from_amax_zeroed(const float *x, int64_t n, float target_log2, float *scale_out, locov_stream_t stream)
{
    return split_scale_from_amax(x, n, target_log2, scale_out, true, stream);
}

int locov_amax_bound(const float *const *xs, const int64_t *ns, const float *muls, int count, float *slot, locov_stream_t stream)
{
    LOCOV_REQUIRE(count >= 0 && count <= LOCOV_AMAX_BOUND_MAX, "locov_amax_bound: 0..%d tensors per call", LOCOV_AMAX_BOUND_MAX);
    if (count == 0) return LOCOV_OK;
    LOCOV_REQUIRE(xs && ns && muls && slot, "locov_amax_bound: null pointer");
    AmaxBoundList L{};
    unsigned next = 0;
    int used = 0;
    for (int i = 0; i < count; i++) {
        LOCOV_REQUIRE(ns[i] >= 0 && ns[i] % 4 == 0 && (ns[i] == 0 || (xs[i] && (uintptr_t)xs[i] % 16 == 0)) && muls[i] >= 0.f,
                      "locov_amax_bound: tensor %d: numel %% 4, 16-byte pointer, non-negative multiplier", i);
        if (ns[i] == 0) continue;
        const int64_t n4 = ns[i] / 4, want = ceil_div(n4, 256 * 8);
        L.x[used] = xs[i];
        L.n4[used] = n4;
        L.mul[used] = muls[i];
        L.first_block[used] = next;
        next += (unsigned)(want < 1 ? 1 : want < 512 ? want : 512);
        used++;
    }
    if (used == 0) return LOCOV_OK;
    L.first_block[used] = next;
    L.count = used;
    hipLaunchKernelGGL(amax_bound_kernel, dim3(next), dim3(256), 0, as_stream(stream), L, slot);
    return check_launch("locov_amax_bound");
}

int locov_gemm_nt_f32_split_ex(const float *x, int64_t lda, const void *W_split, const float *scale, const float *shift,
                               const float *residual, const float *mask, float *y, int64_t ldc, int64_t M, int N, int K,
                               unsigned flags, float x_scale, const float *x_scale_dev, float w_scale, unsigned *overflow,
                               float *amax_out, locov_stream_t stream)
{
    LOCOV_REQUIRE(M >= 0 && N > 0 && K > 0, "locov_gemm_nt_f32_split_ex: bad shape M=%lld N=%d K=%d", (long long)M, N, K);
    if (M == 0) return LOCOV_OK;
    LOCOV_REQUIRE(x && W_split && y, "locov_gemm_nt_f32_split_ex: null pointer");
    LOCOV_REQUIRE(lda >= K && ldc >= N, "locov_gemm_nt_f32_split_ex: lda < K or ldc < N");
    Epilogue epi{scale, shift, residual, flags, mask, amax_out};
    return launch_gemm_split(x, lda, W_split, y, ldc, M, N, K, epi, x_scale, w_scale, as_stream(stream),
                             "locov_gemm_nt_f32_split_ex", Batch{1, 0, 0, 0}, overflow, x_scale_dev);
}

int locov_gemm_segmean_supported(int64_t lda, int64_t M, int N, int K, int seg, unsigned flags)
{
    if (M <= 0 || N <= 0 || K < BK || K % BK != 0 || seg < 43 || seg > BM || M % seg != 0 || N % 4 != 0 || lda % 4 != 0 || lda < K) return 0;
    const Epilogue epi{nullptr, nullptr, reinterpret_cast<const float *>(16), flags};
    if (gemm_split_big_segmean_applicable(lda, M, N, K, epi, seg)) return 1;
    const int64_t tiles = ceil_div(M, BM) * ceil_div(N, BN);
    return !(tiles > 0x7fffffffLL || (double)M * N * 4 > 4294967295.0 || (int64_t)BM * lda * 4 > 0x7fffffffLL);
}

int64_t locov_gemm_segmean_workspace_bytes(int64_t M, int N)
{
    if (M <= 0 || N <= 0) return 0;
    const int64_t small = ceil_div(M, BM) * 4 * (int64_t)N * (int64_t)sizeof(float), big = gemm_split_big_segmean_workspace_bytes(M, N);
    return small > big ? small : big;             // (either kernel may take the launch)
}

int locov_gemm_nt_f32_split_segmean(const float *x, int64_t lda, const void *W_split, const float *scale, const float *shift,
                                    const float *residual, float *out, int64_t M, int N, int K, int seg, unsigned flags,
                                    float x_scale, float w_scale, void *workspace, int64_t workspace_bytes,
                                    unsigned *overflow, locov_stream_t stream)
{
    LOCOV_REQUIRE(M >= 0 && N > 0 && K > 0, "locov_gemm_nt_f32_split_segmean: bad shape M=%lld N=%d K=%d", (long long)M, N, K);
    if (M == 0) return LOCOV_OK;
    LOCOV_REQUIRE(x && W_split && residual && out && workspace, "locov_gemm_nt_f32_split_segmean: null pointer");
    LOCOV_REQUIRE(lda >= K, "locov_gemm_nt_f32_split_segmean: lda < K");
    LOCOV_REQUIRE(!(flags & LOCOV_EPI_RES_SPLIT) || N % 8 == 0, "locov_gemm_nt_f32_split_segmean: a split-layout residual needs N %% 8 == 0");
    LOCOV_REQUIRE(!(flags & ~(unsigned)(LOCOV_EPI_RELU | LOCOV_SEGMEAN_RES_ROI_MAJOR | LOCOV_GEMM_A_SPLIT | LOCOV_EPI_RES_SPLIT)),
                  "locov_gemm_nt_f32_split_segmean: unsupported flags 0x%x", flags);
    LOCOV_REQUIRE(workspace_bytes >= locov_gemm_segmean_workspace_bytes(M, N),
                  "locov_gemm_nt_f32_split_segmean: workspace too small (%lld bytes)", (long long)workspace_bytes);
    Epilogue epi{scale, shift, residual, flags};
    return launch_gemm_split_segmean(x, lda, W_split, M, N, K, epi, seg, x_scale, w_scale, static_cast<float *>(workspace), out,
                                     as_stream(stream), "locov_gemm_nt_f32_split_segmean", overflow);
}

int locov_gemm_nt_batched_f32_split(const float *x, int64_t lda, int64_t stride_x, const void *W_split, int64_t stride_w,
                                    float *y, int64_t ldc, int64_t stride_y, int64_t M, int N, int K, int batch,
                                    float x_scale, float w_scale, unsigned *overflow, locov_stream_t stream)
{
    LOCOV_REQUIRE(M >= 0 && N > 0 && K > 0 && batch > 0, "locov_gemm_nt_batched_f32_split: bad shape");
    if (M == 0) return LOCOV_OK;
    LOCOV_REQUIRE(x && W_split && y, "locov_gemm_nt_batched_f32_split: null pointer");
    LOCOV_REQUIRE(lda >= K && ldc >= N && stride_x % 4 == 0 && stride_w % 4 == 0 && stride_y % 4 == 0,
                  "locov_gemm_nt_batched_f32_split: lda < K, ldc < N or a stride that is not a multiple of 4");
    Epilogue epi{nullptr, nullptr, nullptr, 0u};
    return launch_gemm_split(x, lda, W_split, y, ldc, M, N, K, epi, x_scale, w_scale, as_stream(stream),
                             "locov_gemm_nt_batched_f32_split", Batch{batch, stride_x, stride_w, stride_y}, overflow);
}

}  // extern "C"
