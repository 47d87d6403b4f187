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
// ds_read_b128 fragment reads are conflict-free).  A wave owns a (BM/WM)x(BN/WN) sub-tile as
// 32x32 accumulators.  Lane l supplies row (l&31), 16 bytes at k-offset 16B*(l>>5):
//   fp32 -> 4 consecutive 32x32x2 MFMAs use .x .y .z .w (lane-half h covers k = 4h+j),
//   bf16 -> one 32x32x16 MFMA (lane-half h covers k = 8h..8h+7)          [guide section 3].
//
// K-loop = a 4-phase software pipeline per K-tile (one phase per quarter of the tile's MFMAs),
// built so that a single wave per SIMD keeps the matrix pipe fed (the f32 MFMA occupies the
// pipe 64 cycles; everything else has to fit in its shadow):
//   A: MFMA q0 | read fragments q1 from LDS
//   B: MFMA q1 | read fragments q2 | write tile t+1 (global data loaded one tile ago) to the OTHER LDS stage
//   C: MFMA q2 | read fragments q3 | issue the global loads of tile t+2
//      barrier (tile t+1 is now visible; nobody still reads the stage it went to)
//   D: MFMA q3 | read fragments q0 of tile t+1
// Two LDS stages -> one barrier per tile; fragments are double-buffered in registers; global
// data is prefetched two tiles ahead.  sched_barriers pin the phase order, inside a phase the
// compiler interleaves memory instructions with the MFMAs.
//
// CONV3 (3x3, pad 1, stride 1): A is the [R*H*W, Cin] pixel matrix of R independent HxW
// tiles; GEMM column k = tap*Cin + c reads pixel (y+dy, x+dx) of the same tile, zero outside
// it.  A K-tile never straddles a tap (Cin % BK == 0), so (dy,dx) is uniform per tile and the
// gather is just a row offset plus a per-row validity mask.
#include "gemm_nt.h"

#include <cstdlib>
#include <type_traits>
#include <mutex>
#include <vector>

#ifndef LOCOV_RES_PREFETCH
#define LOCOV_RES_PREFETCH 4     // residual row groups (of 16 per wave sub-tile) requested before the last K-tile
#endif

namespace locov {

// Optional per-launch timing of the GEMM kernels (locov_gemm_timing_*): HIP events recorded on the
// launch stream right around each kernel, so that a caller (bench.py) gets the kernels' own
// durations and FLOP counts over its timed region without a profiler attached.
struct TimingRec {
    hipEvent_t e0, e1;
    int cls;
    double flops, bytes;
};
static std::mutex g_timing_mutex;
static bool g_timing_on = false;
static std::vector<TimingRec> g_timing;

int timing_begin(hipStream_t s, int cls, double flops, double bytes)
{
    std::lock_guard<std::mutex> lock(g_timing_mutex);
    if (!g_timing_on) return -1;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) return -1;   // never inside a graph capture
    TimingRec r{nullptr, nullptr, cls, flops, bytes};
    if (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess) return -1;
    (void)hipEventRecord(r.e0, s);
    g_timing.push_back(r);
    return (int)g_timing.size() - 1;
}
void timing_end(int idx, hipStream_t s)
{
    if (idx < 0) return;
    std::lock_guard<std::mutex> lock(g_timing_mutex);
    if (idx < (int)g_timing.size()) (void)hipEventRecord(g_timing[idx].e1, s);
}

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

// OCC = waves per SIMD the register allocator must leave room for (= workgroups per CU for the
// 4-wave configurations; the two LDS stages allow as many).
// CONV: 0 = plain GEMM, 1 = 3x3 conv on ROI-major rows [r][pos], 2 = 3x3 conv on POSITION-major
// rows [pos][r] (cg.R rows per position).  In mode 2 an M-tile holds ONE tile position of up to
// BM different ROIs, so tap validity is uniform per workgroup: padding taps are skipped outright
// (361 of the 441 (position, tap) pairs of a 7x7 tile are real -> 18 % fewer MFMAs, no masking).
//
// MASKED = false is the fast staging path: K % BK == 0 and no per-row tap masks (CONV != 1).  Rows
// past M / N are simply read from a clamped in-bounds row -- they only feed output rows / columns
// that are never stored -- so no select is needed, and the per-lane global pointers just advance by
// a wave-uniform delta per K-tile.  MASKED = true keeps the general path (ragged K, ROI-major conv).
// EMASK: the epilogue additionally zeroes every value whose `epi.mask` entry is <= 0 (a separate instance, so that the
// inference kernels' register allocation is untouched).
template <typename T, typename TOut, int BM, int BN, int WM, int WN, int OCC, int CONV, bool MASKED, int BK16, bool EMASK = false>
__global__ __launch_bounds__(64 * WM *WN, OCC) void gemm_nt_kernel(const T *__restrict__ A, int64_t lda,
                                                                    const T *__restrict__ B, int64_t ldb,
                                                                    TOut *__restrict__ Cout, int64_t ldc, int64_t M_,
                                                                    int N, int K_, Epilogue epi, ConvGeom cg, Batch bt)
{
    constexpr bool CONV3 = CONV == 1;     // ROI-major: per-row tap masks
    constexpr bool CONVP = CONV == 2;     // position-major: per-workgroup tap list
    static_assert(MASKED || !CONV3, "the ROI-major convolution needs the masked staging path");
    typedef typename Frag<T>::type frag_t;
    constexpr int E = Frag<T>::kPer16B;    // elements per 16 B
    // BK16 = 16-byte chunks per tile row: BK = BK16*E (8 -> 32 f32 / 64 bf16; 16 -> 64 f32 / 128 bf16)
    static_assert(BK16 == 8 || BK16 == 16, "BK16 must be 8 or 16");
    constexpr int BK = BK16 * E;
    constexpr int LDS16 = BK16 + 1;        // padded row length in 16-byte units (odd: conflict-free b128 reads)
    constexpr int NT = 64 * WM * WN;
    constexpr int TM = BM / WM, TN = BN / WN;
    constexpr int MI = TM / 32, NI = TN / 32;
    constexpr int A_CH = BM * BK16 / NT;   // 16-byte chunks per thread per tile
    constexpr int B_CH = BN * BK16 / NT;
    static_assert(BM * BK16 % NT == 0 && BN * BK16 % NT == 0, "tile must divide evenly over threads");
    static_assert(A_CH <= 16 && B_CH <= 16, "ok_mask holds 16 chunks per operand");

    constexpr int STAGE = (BM + BN) * LDS16;
    __shared__ frag_t lds[2 * STAGE];

    __builtin_amdgcn_s_setprio(3);
    const int tiles_n = (N + BN - 1) / BN;
    const int nwg = gridDim.x;
    int tile = xcd_remap(blockIdx.x, nwg);
    if (CONV == 0 && bt.count > 1) {
        // batched: consecutive tiles (one XCD's run) belong to the same problem and share its B in L2
        const int per = nwg / bt.count, b = tile / per;
        tile -= b * per;
        A += b * bt.sa;
        B += b * bt.sb;
        Cout += b * bt.sc;
    }
    // Plain GEMM tile order: the N tiles are taken in groups of NG columns-of-tiles; inside a group M-tile outer,
    // N-tile inner.  The workgroups resident on an XCD then stream a B slice of NG*BN rows (2 MB at K = 512 / 1 MB
    // per 128 columns at K = 2048) that stays in the 4 MB L2 next to the A panels, instead of the whole weight
    // matrix (4 MB for the Res5 1x1 convolutions), which would be re-fetched from the Infinity Cache by every
    // M-tile group; the price is reading A once per group.
    int64_t m0;
    int n0;
    {
        const int NG = (int64_t)K_ * (int)sizeof(T) * BN * 8 <= (2 << 20) ? 8 : 4;      // <= 2 MB of B per group
        const int tiles_m = (int)((bt.count > 1 ? nwg / bt.count : nwg) / tiles_n);
        const int full = (tiles_n / NG) * NG, per_group = tiles_m * NG;
        if (tiles_n <= NG) {
            m0 = (int64_t)(tile / tiles_n) * BM;
            n0 = (tile % tiles_n) * BN;
        } else if (tile < tiles_m * full) {
            const int g = tile / per_group, rem = tile - g * per_group;
            m0 = (int64_t)(rem / NG) * BM;
            n0 = (g * NG + rem % NG) * BN;
        } else {                                            // ragged last group
            const int gs = tiles_n - full, rem = tile - tiles_m * full;
            m0 = (int64_t)(rem / gs) * BM;
            n0 = (full + rem % gs) * BN;
        }
    }
    int64_t M = M_;
    int K = K_;
    unsigned long long taps = 0;          // CONVP: valid tap ids, 4 bits each, in ascending order
    int64_t pos_stride = 0;               // CONVP: elements between the same ROI at adjacent positions
    if (CONVP) {
        // Tile order: groups of G ROI blocks; inside a group position-outer, then (ROI block, N tile).
        // The G*tiles_n workgroups of one position have the same tap list, hence the same length:
        // they start together, stay in step and finish together, so the weight slice they stream
        // (and, across the N tiles, the A rows) is fetched into the XCD's L2 once per group instead of
        // once per workgroup.  Positions of a group run back to back on the same XCD, which keeps the
        // 3x3 window's re-reads of those ROIs' rows close in time (served by the Infinity Cache).
        const int npos = cg.H * cg.W;
        const int nrb = (int)((cg.R + BM - 1) / BM);
        const int G = cg.group > 0 ? cg.group : 4;
        const int per_group = G * npos * tiles_n;
        const int rbg = tile / per_group;
        const int gsz = min(G, nrb - rbg * G);
        const int rem = tile - rbg * per_group;
        const int pos = rem / (gsz * tiles_n), rem2 = rem - pos * (gsz * tiles_n);
        const int rb = rbg * G + rem2 / tiles_n;
        n0 = (rem2 % tiles_n) * BN;
        m0 = (int64_t)pos * cg.R + (int64_t)rb * BM;
        M = (int64_t)(pos + 1) * cg.R;                    // rows of this position end here
        const int py = pos / cg.W, px = pos - py * cg.W;
        int nt = 0;
        for (int t = 0; t < 9; t++) {
            const int dy = t / 3 - 1, dx = t % 3 - 1;
            if ((unsigned)(py + dy) < (unsigned)cg.H && (unsigned)(px + dx) < (unsigned)cg.W)
                taps |= (unsigned long long)t << (4 * nt++);
        }
        K = nt * cg.Cin;                                  // virtual K: only the real taps
                pos_stride = (int64_t)cg.R * lda;
    }

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave / WN) * TM, wn = (wave % WN) * TN;

    // Per-thread staging rows are the same for every K-tile: resolve pointers / validity once.
    // Loads are branch-free: an out-of-range row reads a clamped (in-bounds) address and is zeroed
    // by a select at store time, so the K-loop has no exec-masked branches and no per-load waits.
    const T *a_ptr[A_CH];
    const T *b_ptr[B_CH];
    bool a_ok[A_CH], b_ok[B_CH];
    int a_y[A_CH], a_x[A_CH];
#pragma unroll
    for (int i = 0; i < A_CH; i++) {
        const int idx = tid + i * NT, row = idx / BK16, ch = idx % BK16;
        const int64_t gm = m0 + row;
        a_ok[i] = gm < M;
        const int64_t gmc = a_ok[i] ? gm : M - 1;
        a_ptr[i] = A + gmc * lda + ch * E;
        if (CONV3) {
            const int rem = (int)(gmc % (cg.H * cg.W));
            a_y[i] = rem / cg.W;
            a_x[i] = rem - a_y[i] * cg.W;
        }
    }
#pragma unroll
    for (int i = 0; i < B_CH; i++) {
        const int idx = tid + i * NT, row = idx / BK16, ch = idx % BK16;
        const int gn = n0 + row;
        b_ok[i] = gn < N;
        b_ptr[i] = B + (int64_t)(b_ok[i] ? gn : N - 1) * ldb + ch * E;
    }

    // Fast path addressing: a wave-uniform base (SGPR pair, advanced once per K-tile by scalar adds)
    // plus a per-lane 32-bit byte offset that never changes -> `global_load ... v_off, s[base]` with no
    // per-lane pointer arithmetic in the K-loop (the matrix pipe cannot start an MFMA while the
    // wave's VALU slot is taken by address math).
    const char *a_base = reinterpret_cast<const char *>(A + m0 * lda);
    const char *b_base = reinterpret_cast<const char *>(B + (int64_t)n0 * ldb);
    unsigned a_off[A_CH], b_off[B_CH];
#pragma unroll
    for (int i = 0; i < A_CH; i++) {
        const int idx = tid + i * NT, row = idx / BK16, ch = idx % BK16;
        const int64_t gm = m0 + row;
        a_off[i] = (unsigned)((((gm < M ? gm : M - 1) - m0) * lda + ch * E) * (int64_t)sizeof(T));
    }
#pragma unroll
    for (int i = 0; i < B_CH; i++) {
        const int idx = tid + i * NT, row = idx / BK16, ch = idx % BK16;
        const int gn = n0 + row;
        b_off[i] = (unsigned)(((int64_t)((gn < N ? gn : N - 1) - n0) * ldb + ch * E) * (int64_t)sizeof(T));
    }
    // raw buffer loads: descriptor (4 SGPRs) rebuilt from the running base by scalar ops, 32-bit VGPR offset
    auto ld_a = [&](int i) {
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(a_base), 0, 0xffffffff, 0x00020000);
        return __builtin_bit_cast(frag_t, __builtin_amdgcn_raw_buffer_load_b128(r, a_off[i], 0, 0));
    };
    auto ld_b = [&](int i) {
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(b_base), 0, 0xffffffff, 0x00020000);
        return __builtin_bit_cast(frag_t, __builtin_amdgcn_raw_buffer_load_b128(r, b_off[i], 0, 0));
    };

    frag_t ra[A_CH], rb[B_CH];
    unsigned ok_mask = 0;                 // bit i: A chunk i is data, bit 16+i: B chunk i is data
    const frag_t zero = {};

    // CONVP: K-tile k0 -> (index into the tap list, first channel); K runs tap-outer, channel-inner
    auto convp_split = [&](int k0, int &ti, int &kc) {
        ti = k0 / cg.Cin;
        kc = k0 - ti * cg.Cin;
    };
    // Raw 16-byte loads of the K-tile at k0 (nothing here depends on their results); records which
    // chunks are real data.  A ragged K tail is handled per 16-byte chunk (K % E == 0).
    // element offsets of K-tile k0 relative to a row start: into A (aoff) and into B's row (kb)
    auto tile_offsets = [&](int k0, int64_t &aoff, int &kb) {
        aoff = k0;
        kb = k0;
        if (CONVP) {
            int ti, kc;
            convp_split(k0, ti, kc);
            const int tap = (int)((taps >> (4 * ti)) & 15u);
            const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
            aoff = (int64_t)(dy * cg.W + dx) * pos_stride + kc;    // same ROI, neighbouring position
            kb = tap * cg.Cin + kc;
        }
    };
    int k_ptr = 0;                        // fast path: the K-tile a_ptr / b_ptr currently point at
    if (!MASKED) {
        int64_t ao;
        int bo;
        tile_offsets(0, ao, bo);
        a_base += ao * (int64_t)sizeof(T);
        b_base += bo * (int64_t)sizeof(T);
    }

    auto load_tiles = [&](int k0) {
        if (!MASKED) {
            // advance every per-lane pointer by the same wave-uniform delta, then 16-byte loads
            int64_t ao0, ao1;
            int bo0, bo1;
            tile_offsets(k_ptr, ao0, bo0);
            tile_offsets(k0, ao1, bo1);
            const int64_t da = ao1 - ao0;
            const int db = bo1 - bo0;
            k_ptr = k0;
            a_base += da * (int64_t)sizeof(T);
            b_base += db * (int64_t)sizeof(T);
#pragma unroll
            for (int i = 0; i < A_CH; i++) ra[i] = ld_a(i);
#pragma unroll
            for (int i = 0; i < B_CH; i++) rb[i] = ld_b(i);
            return;
        }
        ok_mask = 0;
        int64_t aoff = k0;
        int kb = k0;                      // column offset into B
        int dy = 0, dx = 0;
        if (CONV3) {
            const int tap = k0 / cg.Cin;
            dy = tap / 3 - 1;
            dx = tap - (tap / 3) * 3 - 1;
            aoff = (int64_t)(dy * cg.W + dx) * lda + (k0 - tap * cg.Cin);
        }
        if (CONVP) {
            int ti, kc;
            convp_split(k0, ti, kc);
            const int tap = (int)((taps >> (4 * ti)) & 15u);
            dy = tap / 3 - 1;
            dx = tap - (tap / 3) * 3 - 1;
            aoff = (int64_t)(dy * cg.W + dx) * pos_stride + kc;    // same ROI, neighbouring position
            kb = tap * cg.Cin + kc;
        }
#pragma unroll
        for (int i = 0; i < A_CH; i++) {
            const int ch = (tid + i * NT) % BK16;
            bool ok = a_ok[i];
            int64_t off = aoff;
            if (CONV3) {
                const bool in = (unsigned)(a_y[i] + dy) < (unsigned)cg.H && (unsigned)(a_x[i] + dx) < (unsigned)cg.W;
                ok = ok && in;
                off = in ? aoff : (int64_t)(k0 % cg.Cin);      // padding tap: stay inside the tensor
            } else {
                const bool kin = k0 + ch * E < K;
                ok = ok && kin;
                off = kin ? aoff : -(int64_t)(ch * E);       // chunk past K: read the row's first chunk (in bounds), store zero
            }
            ra[i] = *reinterpret_cast<const frag_t *>(a_ptr[i] + off);
            ok_mask |= ok ? (1u << i) : 0u;
        }
#pragma unroll
        for (int i = 0; i < B_CH; i++) {
            const int ch = (tid + i * NT) % BK16;
            const bool kin = k0 + ch * E < K;
            rb[i] = *reinterpret_cast<const frag_t *>(b_ptr[i] + (kin ? kb : -(ch * E)));
            ok_mask |= (b_ok[i] && kin) ? (1u << (16 + i)) : 0u;
        }
    };
    auto store_tiles = [&](int stage) {
        frag_t *As = lds + stage * STAGE, *Bs = As + BM * LDS16;
        if (!MASKED) {
#pragma unroll
            for (int i = 0; i < A_CH; i++) As[((tid + i * NT) / BK16) * LDS16 + (tid + i * NT) % BK16] = ra[i];
#pragma unroll
            for (int i = 0; i < B_CH; i++) Bs[((tid + i * NT) / BK16) * LDS16 + (tid + i * NT) % BK16] = rb[i];
            return;
        }
#pragma unroll
        for (int i = 0; i < A_CH; i++) {
            const int idx = tid + i * NT;
            As[(idx / BK16) * LDS16 + idx % BK16] = (ok_mask >> i) & 1u ? ra[i] : zero;
        }
#pragma unroll
        for (int i = 0; i < B_CH; i++) {
            const int idx = tid + i * NT;
            Bs[(idx / BK16) * LDS16 + idx % BK16] = (ok_mask >> (16 + i)) & 1u ? rb[i] : zero;
        }
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; i++)
#pragma unroll
        for (int j = 0; j < NI; j++) acc[i][j] = f32x16{};

    // register double buffer of MFMA fragments: sub-step q of a tile = 16-byte chunks 2q, 2q+1
    // (index [q & 1]; every use below is fully unrolled, so the indices are compile-time constants)
    frag_t fa[2][MI], fb[2][NI];
    const int frow = lane & 31, fch = lane >> 5;
    auto read_frags = [&](int stage, int q, int set) {
        const frag_t *As = lds + stage * STAGE, *Bs = As + BM * LDS16;
#pragma unroll
        for (int i = 0; i < MI; i++) fa[set][i] = As[(wm + i * 32 + frow) * LDS16 + 2 * q + fch];
#pragma unroll
        for (int j = 0; j < NI; j++) fb[set][j] = Bs[(wn + j * 32 + frow) * LDS16 + 2 * q + fch];
    };
    auto mma = [&](int set) {
#pragma unroll
        for (int i = 0; i < MI; i++)
#pragma unroll
            for (int j = 0; j < NI; j++) mma_step(fa[set][i], fb[set][j], acc[i][j]);
    };

    // MFMAs p0..p1-1 of a sub-step (fp32: 4 per accumulator tile, one per k-pair; bf16: 1 per tile)
    constexpr int MPT = sizeof(T) == 4 ? 4 : 1;
    constexpr int NMFMA = MI * NI * MPT;
    auto mma_range = [&](int set, int p0, int p1) {
#pragma unroll
        for (int p = 0; p < NMFMA; p++) {
            if (p < p0 || p >= p1) continue;
            const int t = p / MPT, i = t / NI, j = t % NI;
            if constexpr (sizeof(T) == 4) {
                const int ks = p % MPT;
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set][i][ks], fb[set][j][ks], acc[i][j], 0, 0, 0);
            } else {
                mma_step(fa[set][i], fb[set][j], acc[i][j]);
            }
        }
    };
    // one half of a sub-step's MFMAs (row tiles split in two; MI == 1: first half does everything)
    auto mma_half = [&](int set, int half) {
#pragma unroll
        for (int i = 0; i < MI; i++) {
            if ((MI == 1 ? 0 : (i * 2) / MI) != half) continue;
#pragma unroll
            for (int j = 0; j < NI; j++) mma_step(fa[set][i], fb[set][j], acc[i][j]);
        }
    };

    const int k_last = ((K + BK - 1) / BK - 1) * BK;        // k0 of the last K-tile
    // prologue: tile 0 -> LDS stage 0, tile 1 -> staging registers, fragments of sub-step 0
    load_tiles(0);
    store_tiles(0);
    load_tiles(BK < k_last ? BK : k_last);
    __syncthreads();
    read_frags(0, 0, 0);
    // The prologue above and the epilogue below run at raised wave priority, the K-loop at the lowest:
    // a co-resident workgroup that is streaming MFMAs otherwise starves this wave's vector / memory
    // instructions down to one issue per MFMA slot (measured: a 6k-cycle epilogue stretched to 60k+),
    // both workgroups then finish together and their MFMA-free phases coincide instead of overlapping
    // with the partner's matrix work.
    __builtin_amdgcn_s_setprio(0);
    constexpr int NCH = A_CH + B_CH;                        // staging chunks per thread and tile
    // Fast path: chunks [g0,g1) of the staged tile t+1 go registers -> LDS stage `ws` and the refill
    // loads of tile t+2 (a_base / b_base already address it) are issued, each chunk followed by its
    // share of the sub-step's MFMAs (hand-interleaved, so the staging traffic sits evenly in the
    // shadow of the matrix pipe).
    auto stage_and_mma = [&](int g0, int g1, int ws, int set) {
        frag_t *As = lds + ws * STAGE, *Bs = As + BM * LDS16;
        const int n = g1 - g0;
#pragma unroll
        for (int g = 0; g < NCH; g++) {
            if (g < g0 || g >= g1) continue;
            if (g < A_CH) {
                const int idx = tid + g * NT;
                As[(idx / BK16) * LDS16 + idx % BK16] = ra[g];
                ra[g] = ld_a(g);
            } else {
                const int h = g - A_CH, idx = tid + h * NT;
                Bs[(idx / BK16) * LDS16 + idx % BK16] = rb[h];
                rb[h] = ld_b(h);
            }
            mma_range(set, (g - g0) * NMFMA / n, (g - g0 + 1) * NMFMA / n);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // Sub-steps per tile and where the staging goes: the SQ sub-steps right before the last one
    // carry at most 8 chunks each (measured on MI355X: staging late in the tile beats an even spread
    // or an early burst by 5-8 %); the LAST sub-step holds the barrier.
    constexpr int NQ = BK16 / 2;
    constexpr int SQ = (NCH + 7) / 8;
    static_assert(NQ % 2 == 0 && NQ >= SQ + 2, "sub-step schedule needs room for the staging phases");

    // One K-tile that has a successor; `s` = LDS stage it computes from.  Always called with a literal
    // stage (the loop below is unrolled by two), so every LDS address is a per-lane base register plus
    // an immediate: no address arithmetic in the loop.
    auto tile_step = [&](const int s, const int k0) __attribute__((always_inline)) {
        const int kn = k0 + 2 * BK < k_last ? k0 + 2 * BK : k_last;   // tile to prefetch (clamped)
        if (!MASKED) {                                      // bases now address tile kn (scalar adds only)
            int64_t ao0, ao1;
            int bo0, bo1;
            tile_offsets(k_ptr, ao0, bo0);
            tile_offsets(kn, ao1, bo1);
            a_base += (ao1 - ao0) * (int64_t)sizeof(T);
            b_base += (bo1 - bo0) * (int64_t)sizeof(T);
            k_ptr = kn;
        }
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            const int cur = q & 1, nxt = cur ^ 1;
            if (q < NQ - 1) {
                // fragment reads of the NEXT sub-step first: their LDS latency sits under this one's MFMAs
                read_frags(s, q + 1, nxt);
                __builtin_amdgcn_sched_barrier(0);
                const int sq = q - (NQ - 1 - SQ);             // staging slot of this sub-step, if any
                if (!MASKED && sq >= 0) {
                    stage_and_mma(sq * NCH / SQ, (sq + 1) * NCH / SQ, s ^ 1, cur);
                } else if (MASKED && q == NQ - 2) {
                    store_tiles(s ^ 1);                        // staging registers (tile t+1) -> other stage,
                    load_tiles(kn);                            // then refill them with tile t+2
                    mma(cur);
                } else {
                    mma(cur);
                }
                __builtin_amdgcn_sched_barrier(0);
            } else {
                // last sub-step: the barrier sits in the MIDDLE of its MFMAs -- the first half is issued
                // from registers while the waves rendezvous, the second half covers the LDS latency of
                // the next tile's first fragments
                mma_half(cur, 0);
                __builtin_amdgcn_sched_barrier(0);
                __syncthreads();
                read_frags(s ^ 1, 0, nxt);
                __builtin_amdgcn_sched_barrier(0);
                mma_half(cur, 1);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    auto last_tile = [&](const int s) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            if (q < NQ - 1) read_frags(s, q + 1, (q & 1) ^ 1);
            mma(q & 1);
        }
    };
    int k0 = 0;
    for (; k0 + BK < k_last; k0 += 2 * BK) {
        tile_step(0, k0);
        tile_step(1, k0 + BK);
    }
    // Residual prefetch: the residual rows of the first NPRE row groups are requested BEFORE the last K-tile, into
    // the registers the (now idle) staging stream used, so that their HBM latency sits under that tile's MFMAs
    // instead of at the head of the epilogue.
    constexpr int NPRE = LOCOV_RES_PREFETCH;
    constexpr int P_LPR = TN / 4, P_RPI = 64 / P_LPR;
    f32x4 res_pre[NPRE > 0 ? NPRE : 1];
    const bool pre_ok = NPRE > 0 && sizeof(TOut) == 4 && epi.residual && !(epi.flags & 0x800u) && (N % 4 == 0) &&
                        (ldc % 4 == 0) && ((uintptr_t)Cout % 16 == 0) && ((uintptr_t)epi.residual % 16 == 0);
    auto prefetch_residual = [&]() __attribute__((always_inline)) {
        if (!pre_ok) return;
        const int pc4 = (lane % P_LPR) * 4, prr = lane / P_LPR, pn = n0 + wn + pc4;
        if (pn >= N) return;
        const int64_t rows_here = M - m0 < BM ? M - m0 : BM;
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float *>(epi.residual) + m0 * ldc, 0, (unsigned)(rows_here * ldc * (int64_t)sizeof(float)), 0x00020000);
        const unsigned pvoff = (unsigned)(((int64_t)(wm + prr) * ldc + pn) * (int64_t)sizeof(float));
        const unsigned pvstep = (unsigned)(P_RPI * ldc * (int64_t)sizeof(float));
#pragma unroll
        for (int it = 0; it < NPRE; it++)
            res_pre[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, pvoff + it * pvstep, 0, 0));
    };
    if (k0 < k_last) {
        tile_step(0, k0);
        prefetch_residual();
        last_tile(1);
    } else {
        prefetch_residual();
        last_tile(0);
    }

    __builtin_amdgcn_s_setprio(3);
    // Epilogue.  C/D layout of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
    // Runtime options are wave-uniform and hoisted; residual values of a 32x32 tile are fetched as
    // one batch of 16 independent loads before they are consumed.
    const bool relu = (epi.flags & LOCOV_EPI_RELU) != 0;

    // Fast path: a row-per-lane store tail is store-ISSUE-bound (64 dword stores per lane), so each
    // wave re-lays its sub-tile out through LDS (free after the K-loop) and writes 16 bytes per lane:
    // 4x fewer store (and residual load) instructions, 256-byte contiguous runs per row.
    const bool vec_ok = sizeof(TOut) == 4 && !(epi.flags & 0x800u) && (N % 4 == 0) && (ldc % 4 == 0) &&
                        ((uintptr_t)Cout % 16 == 0) && (!epi.residual || (uintptr_t)epi.residual % 16 == 0) &&
                        (!(EMASK && epi.mask) || (uintptr_t)epi.mask % 16 == 0);
    if (vec_ok) {
        constexpr int EPS = TN + 4;                       // padded row (floats): conflict-free b128 reads
        static_assert(WM * WN * TM * EPS * 4 <= 2 * STAGE * 16, "epilogue staging must fit the K-loop LDS");
        constexpr int LPR = TN / 4;                       // lanes per row
        constexpr int RPI = 64 / LPR;                     // rows per wave instruction
        constexpr int NIT = TM / RPI;
        const int c4 = (lane % LPR) * 4, rr = lane / LPR;
        const int n = n0 + wn + c4;
        const bool n_ok = n < N;
        // Residual loads and output stores go through raw buffer descriptors based at the tile's first row,
        // addressed by ONE per-lane 32-bit offset: a co-resident workgroup that streams MFMAs leaves this
        // wave only one vector-ALU issue per MFMA slot, so every v_* instruction here costs ~64 cycles --
        // the epilogue must be (nearly) free of vector address arithmetic and compares.
        //   full tile    : the row step goes into the scalar offset operand (no vector math at all)
        //   partial tile : the row step is added to the lane offset, so that rows >= M fall outside
        //                  num_records and the hardware drops the store / zero-fills the load
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        const int64_t rows_here = M - m0 < BM ? M - m0 : BM;
        const unsigned nrec = (unsigned)(rows_here * ldc * (int64_t)sizeof(float));
        const unsigned voff = (unsigned)(((int64_t)(wm + rr) * ldc + n) * (int64_t)sizeof(float));
        const unsigned vstep = (unsigned)(RPI * ldc * (int64_t)sizeof(float));
        const __amdgpu_buffer_rsrc_t r_out =
            __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float *>(Cout) + m0 * ldc, 0, nrec, 0x00020000);
        const __amdgpu_buffer_rsrc_t r_res = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float *>(epi.residual ? epi.residual + m0 * ldc : reinterpret_cast<float *>(Cout) + m0 * ldc), 0, nrec,
            0x00020000);
        float *ep = reinterpret_cast<float *>(lds) + wave * (TM * EPS);
        const __amdgpu_buffer_rsrc_t r_msk = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float *>(EMASK && epi.mask ? epi.mask + m0 * ldc : reinterpret_cast<float *>(Cout) + m0 * ldc), 0, nrec, 0x00020000);
        auto tail = [&](auto full_tag) __attribute__((always_inline)) {
            constexpr bool FULL = decltype(full_tag)::value;
            // residual rows are fetched FIRST (16-byte loads, all in flight) so that their latency sits
            // under the LDS re-layout below
            f32x4 res[NIT];
            if (epi.residual && n_ok) {
#pragma unroll
                for (int it = 0; it < NIT; it++) {
                    if (it < NPRE && NPRE > 0) {
                        res[it] = res_pre[it];                 // requested before the last K-tile (pre_ok holds here)
                    } else {
                        res[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                                r_res, FULL ? voff : voff + it * vstep, FULL ? it * vstep : 0u, 0));
                    }
                }
            }
            // (scale / shift of this lane's four columns: requested here, consumed after the re-layout)
            f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
            if (n_ok && epi.scale) sc = *reinterpret_cast<const f32x4 *>(epi.scale + n);
            if (n_ok && epi.shift) sh = *reinterpret_cast<const f32x4 *>(epi.shift + n);
            __syncthreads();                              // every wave is done reading the last stage
#pragma unroll
            for (int i = 0; i < MI; i++)
#pragma unroll
                for (int j = 0; j < NI; j++)
#pragma unroll
                    for (int r = 0; r < 16; r++)
                        ep[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * EPS + j * 32 + (lane & 31)] = acc[i][j][r];
            __syncthreads();
            if (n_ok) {
#pragma unroll
                for (int it = 0; it < NIT; it++) {
                    f32x4 v = *reinterpret_cast<const f32x4 *>(ep + (it * RPI + rr) * EPS + c4);
                    v = v * sc + sh;
                    if (epi.residual) v += res[it];
                    if (relu) {
                        v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f);
                        v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
                    }
                    if (EMASK && epi.mask) {
                        const f32x4 mk = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                                       r_msk, FULL ? voff : voff + it * vstep, FULL ? it * vstep : 0u, 0));
                        v[0] = mk[0] > 0.f ? v[0] : 0.f; v[1] = mk[1] > 0.f ? v[1] : 0.f;
                        v[2] = mk[2] > 0.f ? v[2] : 0.f; v[3] = mk[3] > 0.f ? v[3] : 0.f;
                    }
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r_out, FULL ? voff : voff + it * vstep,
                                                           FULL ? it * vstep : 0u, 0);
                }
            }
        };
        if (rows_here == BM)
            tail(std::true_type{});
        else
            tail(std::false_type{});
        return;
    }

    // General path (odd N / ldc, bf16 output): one element per lane.
#pragma unroll
    for (int j = 0; j < NI; j++) {
        const int n = n0 + wn + j * 32 + (lane & 31);
        const bool n_ok = n < N;
        const int nc = n_ok ? n : N - 1;
        const float sc = epi.scale ? epi.scale[nc] : 1.f;
        const float sh = epi.shift ? epi.shift[nc] : 0.f;
#pragma unroll
        for (int i = 0; i < MI; i++) {
            const int64_t mb = m0 + wm + i * 32 + 4 * (lane >> 5);
            float res[16];
            if (epi.residual) {
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int64_t m = mb + (r & 3) + 8 * (r >> 2);
                    res[r] = epi.residual[(m < M ? m : M - 1) * ldc + nc];
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; r++) res[r] = 0.f;
            }
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int64_t m = mb + (r & 3) + 8 * (r >> 2);
                float v = acc[i][j][r] * sc + sh + res[r];
                if (relu) v = fmaxf(v, 0.f);
                if (EMASK && epi.mask && !(epi.mask[(m < M ? m : M - 1) * ldc + nc] > 0.f)) v = 0.f;
                if (n_ok && m < M) store_out(Cout + m * ldc + n, v);
            }
        }
    }
}

// developer experiment: extra dynamic LDS per workgroup (lowers the workgroups-per-CU count)
static unsigned dyn_lds_dbg()
{
    static const unsigned v = [] { const char *e = getenv("LOCOV_GEMM_DYNLDS"); return e ? (unsigned)atoi(e) : 0u; }();
    return v;
}

template <typename T, typename TOut, int BM, int BN, int WM, int WN, int OCC, int BK16 = 8>
static int launch_cfg(const T *A, int64_t lda, const T *B, int64_t ldb, TOut *C, int64_t ldc, int64_t M, int N,
                      int K, const Epilogue &epi, const ConvGeom &cg, const Batch &bt, hipStream_t s, const char *what)
{
    const bool posm = cg.H > 0 && cg.R > 0;
    const int64_t tiles_m = posm ? (int64_t)cg.H * cg.W * ceil_div(cg.R, BM) : ceil_div(M, BM);
    const int64_t tiles = tiles_m * ceil_div(N, BN) * (bt.count > 1 ? bt.count : 1);
    if (bt.count > 1 && (cg.H > 0 || epi.residual))
        return set_error(LOCOV_ERR_UNSUPPORTED, "%s: batched launches are plain GEMMs without residual", what);
    if (tiles > 0x7fffffffLL) return set_error(LOCOV_ERR_INVALID_ARG, "%s: problem too large", what);
    // timing class: 0 = the 128x128 plain / batched GEMM, 1 = position-major 3x3 conv, 2 = everything else,
    // 3 / 4 = classes 0 / 1 with bf16 operands;
    // FLOPs = what the kernel executes (the position-major conv skips its padding taps)
    const int tcls0 = posm ? 1 : (cg.H == 0 && BM == 128 && BN == 128 ? 0 : 2);
    const int tcls = (sizeof(T) == 2 && tcls0 < 2) ? tcls0 + 3 : tcls0;
    const double tflops = posm ? 2.0 * cg.R * (3.0 * cg.H - 2) * (3.0 * cg.W - 2) * cg.Cin * N
                               : 2.0 * (double)M * N * K * (bt.count > 1 ? bt.count : 1);
    const int trec = timing_begin(s, tcls, tflops);
    constexpr int BK = BK16 * Frag<T>::kPer16B;
    const bool ragged_k = (posm ? cg.Cin : K) % BK != 0;
    const dim3 grid((unsigned)tiles), block(64 * WM * WN);

    if (epi.mask && !(std::is_same<T, float>::value && std::is_same<TOut, float>::value))
        return set_error(LOCOV_ERR_UNSUPPORTED, "%s: the epilogue mask needs fp32 operands and output", what);
#define LOCOV_LAUNCH(CONV, MASKED)                                                                                 \
    do {                                                                                                            \
        if constexpr (std::is_same<T, float>::value && std::is_same<TOut, float>::value) {                          \
            if (epi.mask) {                                                                                         \
                hipLaunchKernelGGL((gemm_nt_kernel<T, TOut, BM, BN, WM, WN, OCC, CONV, MASKED, BK16, true>), grid, block, dyn_lds_dbg(), \
                                   s, A, lda, B, ldb, C, ldc, M, N, K, epi, cg, bt);                               \
                break;                                                                                              \
            }                                                                                                       \
        }                                                                                                           \
        hipLaunchKernelGGL((gemm_nt_kernel<T, TOut, BM, BN, WM, WN, OCC, CONV, MASKED, BK16>), grid, block, dyn_lds_dbg(), s, A, lda, B, \
                           ldb, C, ldc, M, N, K, epi, cg, bt);                                                      \
    } while (0)
    if (posm) {
        if (ragged_k) return set_error(LOCOV_ERR_UNSUPPORTED, "%s: Cin must be a multiple of %d", what, BK);
        LOCOV_LAUNCH(2, false);
    } else if (cg.H > 0) {
        LOCOV_LAUNCH(1, true);
    } else if (ragged_k) {
        LOCOV_LAUNCH(0, true);
    } else {
        LOCOV_LAUNCH(0, false);
    }
#undef LOCOV_LAUNCH
    timing_end(trec, s);
    return check_launch(what);
}

// developer knob (tools/bench_gemm.py): LOCOV_GEMM_CFG=1 forces the small-tile configuration
static int forced_cfg()
{
    static const int v = [] {
        const char *e = getenv("LOCOV_GEMM_CFG");
        return e ? atoi(e) : -1;
    }();
    return v;
}

template <typename T, typename TOut>
int launch_gemm_nt(const T *A, int64_t lda, const T *B, int64_t ldb, TOut *C, int64_t ldc, int64_t M, int N, int K,
                   const Epilogue &epi, hipStream_t s, const char *what, const ConvGeom &cg, const Batch &bt)
{
    if (N <= 32) return launch_cfg<T, TOut, 128, 32, 4, 1, 2>(A, lda, B, ldb, C, ldc, M, N, K, epi, cg, bt, s, what);
    if (N <= 64 || (N <= 192 && N % 128 != 0 && N % 128 <= 64))
        return launch_cfg<T, TOut, 128, 64, 4, 1, 2>(A, lda, B, ldb, C, ldc, M, N, K, epi, cg, bt, s, what);
    if (forced_cfg() == 1)
        return launch_cfg<T, TOut, 64, 64, 2, 2, 4>(A, lda, B, ldb, C, ldc, M, N, K, epi, cg, bt, s, what);
    if (forced_cfg() == 3)   // 8 waves per workgroup (64x32 per wave): 4 waves per SIMD with two resident workgroups
        return launch_cfg<T, TOut, 128, 128, 2, 4, 4>(A, lda, B, ldb, C, ldc, M, N, K, epi, cg, bt, s, what);
    if (forced_cfg() == 2)   // deep tile: BK = 64 (f32) / 128 (bf16), 136 KiB LDS, one workgroup per CU
        return launch_cfg<T, TOut, 128, 128, 2, 2, 1, 16>(A, lda, B, ldb, C, ldc, M, N, K, epi, cg, bt, s, what);
    // a plain GEMM whose 128 x 128 tiles would leave most of the 256 CUs idle (the predictor's FCs of a training step: 800 sampled
    // proposals x 768 / 1204 columns = 42-70 tiles, each walking K = 2048 alone): 64 x 64 tiles, four times the workgroups.  Every
    // output element accumulates the same products in the same order in both configurations (same BK steps, same MFMA), so the
    // choice does not change a bit of the result (tests/test_gpu_kernels.py).
    if (forced_cfg() != 0 && cg.H == 0 && (bt.count <= 1) && ceil_div(M, 128) * ceil_div(N, 128) < 128 && M > 64)
        return launch_cfg<T, TOut, 64, 64, 2, 2, 4>(A, lda, B, ldb, C, ldc, M, N, K, epi, cg, bt, s, what);
    return launch_cfg<T, TOut, 128, 128, 2, 2, 2>(A, lda, B, ldb, C, ldc, M, N, K, epi, cg, bt, s, what);
}

#define LOCOV_INST(T, TOut)                                                                                        \
    template int launch_gemm_nt<T, TOut>(const T *, int64_t, const T *, int64_t, TOut *, int64_t, int64_t, int, int, \
                                         const Epilogue &, hipStream_t, const char *, const ConvGeom &, const Batch &);
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

int locov_gemm_nt_f32_ex(const float *x, int64_t lda, const float *W, int64_t ldb, const float *scale, const float *shift,
                         const float *residual, const float *mask, float *y, int64_t ldc, int64_t M, int N, int K,
                         unsigned flags, locov_stream_t stream)
{
    LOCOV_REQUIRE(M >= 0 && N > 0 && K > 0, "locov_gemm_nt_f32_ex: bad shape M=%lld N=%d K=%d", (long long)M, N, K);
    if (M == 0) return LOCOV_OK;
    LOCOV_REQUIRE(x && W && y, "locov_gemm_nt_f32_ex: null pointer");
    LOCOV_REQUIRE(K % 4 == 0 && lda % 4 == 0 && ldb % 4 == 0, "locov_gemm_nt_f32_ex: K, lda and ldb must be multiples of 4");
    LOCOV_REQUIRE(lda >= K && ldb >= K && ldc >= N, "locov_gemm_nt_f32_ex: lda < K, ldb < K or ldc < N");
    LOCOV_REQUIRE((uintptr_t)x % 16 == 0 && (uintptr_t)W % 16 == 0, "locov_gemm_nt_f32_ex: x / W must be 16-byte aligned");
    Epilogue epi{scale, shift, residual, flags, mask};
    return launch_gemm_nt<float, float>(x, lda, W, ldb, y, ldc, M, N, K, epi, as_stream(stream), "locov_gemm_nt_f32_ex");
}

int locov_conv3x3_nhwc_f32_ex(const float *x, int64_t R, int H, int W, int Cin, int pos_major, const float *w_packed,
                              const float *scale, const float *shift, const float *residual, const float *mask, float *y,
                              int N, unsigned flags, locov_stream_t stream)
{
    LOCOV_REQUIRE(R >= 0 && H > 0 && W > 0 && Cin > 0 && N > 0, "locov_conv3x3_nhwc_f32_ex: bad shape");
    if (R == 0) return LOCOV_OK;
    LOCOV_REQUIRE(x && w_packed && y, "locov_conv3x3_nhwc_f32_ex: null pointer");
    LOCOV_REQUIRE(Cin % 32 == 0, "locov_conv3x3_nhwc_f32_ex: Cin must be a multiple of 32 (got %d)", Cin);
    LOCOV_REQUIRE((uintptr_t)x % 16 == 0 && (uintptr_t)w_packed % 16 == 0, "locov_conv3x3_nhwc_f32_ex: misaligned pointer");
    LOCOV_REQUIRE(R <= 0x7fffffffLL / (H * W), "locov_conv3x3_nhwc_f32_ex: R too large");
    Epilogue epi{scale, shift, residual, flags, mask};
    ConvGeom cg{H, W, Cin, pos_major ? (int)R : 0, 0};
    return launch_gemm_nt<float, float>(x, (int64_t)Cin, w_packed, (int64_t)9 * Cin, y, (int64_t)N, R * H * W, N, 9 * Cin, epi,
                                        as_stream(stream), "locov_conv3x3_nhwc_f32_ex", cg);
}

int locov_gemm_timing_enable(int on)
{
    std::lock_guard<std::mutex> lock(g_timing_mutex);
    for (auto &r : g_timing) {
        (void)hipEventDestroy(r.e0);
        (void)hipEventDestroy(r.e1);
    }
    g_timing.clear();
    g_timing_on = on != 0;
    return LOCOV_OK;
}

int locov_gemm_timing_read_ex(int cls, int64_t *launches, double *ms, double *flops, double *bytes)
{
    LOCOV_REQUIRE(launches && ms && flops, "locov_gemm_timing_read: null pointer");
    std::lock_guard<std::mutex> lock(g_timing_mutex);
    *launches = 0;
    *ms = 0.0;
    *flops = 0.0;
    if (bytes) *bytes = 0.0;
    for (auto &r : g_timing) {
        if (r.cls != cls) continue;
        float t = 0.f;
        if (hipEventSynchronize(r.e1) != hipSuccess || hipEventElapsedTime(&t, r.e0, r.e1) != hipSuccess)
            return set_error(LOCOV_ERR_LAUNCH, "locov_gemm_timing_read: event query failed");
        *launches += 1;
        *ms += (double)t;
        *flops += r.flops;
        if (bytes) *bytes += r.bytes;
    }
    return LOCOV_OK;
}

int locov_gemm_timing_read(int cls, int64_t *launches, double *ms, double *flops)
{
    return locov_gemm_timing_read_ex(cls, launches, ms, flops, nullptr);
}

int locov_conv3x3_nhwc_f32(const float *x, int64_t R, int H, int W, int Cin, int pos_major, const float *w_packed,
                           const float *scale, const float *shift, const float *residual, float *y, int N,
                           unsigned flags, locov_stream_t stream)
{
    LOCOV_REQUIRE(R >= 0 && H > 0 && W > 0 && Cin > 0 && N > 0, "locov_conv3x3_nhwc_f32: bad shape");
    if (R == 0) return LOCOV_OK;
    LOCOV_REQUIRE(x && w_packed && y, "locov_conv3x3_nhwc_f32: null pointer");
    LOCOV_REQUIRE(Cin % 32 == 0, "locov_conv3x3_nhwc_f32: Cin must be a multiple of 32 (got %d)", Cin);
    LOCOV_REQUIRE((uintptr_t)x % 16 == 0 && (uintptr_t)w_packed % 16 == 0, "locov_conv3x3_nhwc_f32: misaligned pointer");
    LOCOV_REQUIRE(R <= 0x7fffffffLL / (H * W), "locov_conv3x3_nhwc_f32: R too large");
    Epilogue epi{scale, shift, residual, flags};
    static const int group = [] { const char *e = getenv("LOCOV_CONV_GROUP"); return e ? atoi(e) : 0; }();
    ConvGeom cg{H, W, Cin, pos_major ? (int)R : 0, group};
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

int locov_gemm_nt_bf16(const uint16_t *x, int64_t lda, const uint16_t *W, const float *scale, const float *shift,
                       const float *residual, float *y, int64_t ldc, int64_t M, int N, int K, unsigned flags,
                       locov_stream_t stream)
{
    LOCOV_REQUIRE(M >= 0 && N > 0 && K > 0, "locov_gemm_nt_bf16: bad shape M=%lld N=%d K=%d", (long long)M, N, K);
    if (M == 0) return LOCOV_OK;
    LOCOV_REQUIRE(x && W && y, "locov_gemm_nt_bf16: null pointer");
    LOCOV_REQUIRE(K % 8 == 0 && lda % 8 == 0, "locov_gemm_nt_bf16: K and lda must be multiples of 8 (got %d, %lld)", K,
                  (long long)lda);
    LOCOV_REQUIRE(lda >= K && ldc >= N, "locov_gemm_nt_bf16: lda < K or ldc < N");
    LOCOV_REQUIRE((uintptr_t)x % 16 == 0 && (uintptr_t)W % 16 == 0, "locov_gemm_nt_bf16: x / W must be 16-byte aligned");
    Epilogue epi{scale, shift, residual, flags};
    return launch_gemm_nt<__bf16, float>(reinterpret_cast<const __bf16 *>(x), lda, reinterpret_cast<const __bf16 *>(W),
                                         (int64_t)K, y, ldc, M, N, K, epi, as_stream(stream), "locov_gemm_nt_bf16");
}

int locov_conv3x3_nhwc_bf16(const uint16_t *x, int64_t R, int H, int W, int Cin, int pos_major, const uint16_t *w_packed,
                            const float *scale, const float *shift, const float *residual, float *y, int N,
                            unsigned flags, locov_stream_t stream)
{
    LOCOV_REQUIRE(R >= 0 && H > 0 && W > 0 && Cin > 0 && N > 0, "locov_conv3x3_nhwc_bf16: bad shape");
    if (R == 0) return LOCOV_OK;
    LOCOV_REQUIRE(x && w_packed && y, "locov_conv3x3_nhwc_bf16: null pointer");
    LOCOV_REQUIRE(Cin % 64 == 0, "locov_conv3x3_nhwc_bf16: Cin must be a multiple of 64 (got %d)", Cin);
    LOCOV_REQUIRE((uintptr_t)x % 16 == 0 && (uintptr_t)w_packed % 16 == 0, "locov_conv3x3_nhwc_bf16: misaligned pointer");
    LOCOV_REQUIRE(R <= 0x7fffffffLL / (H * W), "locov_conv3x3_nhwc_bf16: R too large");
    Epilogue epi{scale, shift, residual, flags};
    ConvGeom cg{H, W, Cin, pos_major ? (int)R : 0, 0};
    return launch_gemm_nt<__bf16, float>(reinterpret_cast<const __bf16 *>(x), (int64_t)Cin,
                                         reinterpret_cast<const __bf16 *>(w_packed), (int64_t)9 * Cin, y, (int64_t)N,
                                         R * H * W, N, 9 * Cin, epi, as_stream(stream), "locov_conv3x3_nhwc_bf16", cg);
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
