"""DIAGNOSTIC builds of gemm_split.hip (tools/liblocov_splitv<N>.so, never the product; results are WRONG on purpose):
ablations that remove one cost at a time from the K-loop, to see what bounds the kernel.  Run with
LOCOV_HIP_LIB=tools/liblocov_splitv<N>.so python tools/bench_split.py speed
  1: no fp32 -> (hi, lo) conversion (raw bits stored)     2: no A refill loads in the K-loop
  3: no staging at all (no loads, no LDS writes, no DMA)  4: no fragment reads either (MFMA-only loop)
  5: as 3 but the W DMA stays                              6: as 3 but the A LDS stores stay
  7: A taken as ALREADY split and staged by LDS DMA like W (what the kernel would do if the producers wrote the
     activations in split format): timing only
  8: no epilogue traffic (no residual loads, no output stores)
  9 / 10: variant 4 / the full kernel with 2 x v_mfma_f32_16x16x32_f16 in place of each 32x32x16 (timing only)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = open(os.path.join(ROOT, "locov_amd/csrc/gemm_split.hip")).read().replace('#include "gemm_nt.h"', '#include "%s/locov_amd/csrc/gemm_nt.h"' % ROOT)
def variant(n):
    s = SRC
    def rep(a, b):
        nonlocal s
        assert s.count(a) == 1, a
        s = s.replace(a, b)
    if n == 1:
        rep("        split4(ra[i], a_scale, hi, lo);", "        hi = u32x2{__builtin_bit_cast(unsigned, ra[i][0]), __builtin_bit_cast(unsigned, ra[i][1])}; lo = u32x2{__builtin_bit_cast(unsigned, ra[i][2]), __builtin_bit_cast(unsigned, ra[i][3])};")
    if n in (9, 10):  # 9: variant 4 (MFMA-only loop), 10: the full kernel -- with each 32x32x16 MFMA replaced by two 16x16x32 ones
        rep("            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[q][i][w == 2 ? 1 : 0], fb[q][j][w == 1 ? 1 : 0], acc[i][j], 0, 0, 0);",
            """            {
                f32x4 c0 = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]}, c1 = {acc[i][j][4], acc[i][j][5], acc[i][j][6], acc[i][j][7]};
                c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[q][i][w == 2 ? 1 : 0], fb[q][j][w == 1 ? 1 : 0], c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[q][i][w == 2 ? 1 : 0], fb[q][j][w == 1 ? 1 : 0], c1, 0, 0, 0);
                acc[i][j][0] = c0[0]; acc[i][j][1] = c0[1]; acc[i][j][2] = c0[2]; acc[i][j][3] = c0[3];
                acc[i][j][4] = c1[0]; acc[i][j][5] = c1[1]; acc[i][j][6] = c1[2]; acc[i][j][7] = c1[3];
            }""")
        if n == 10:
            return s
        n = 4
    if n == 8:       # no epilogue traffic: neither residual loads nor output stores (K-loop + LDS re-layout only)
        rep("        if (epi.residual && n_ok) {\n#pragma unroll\n            for (int it = 0; it < NIT; it++) {\n                if (it < NPRE", "        if (false) {\n#pragma unroll\n            for (int it = 0; it < NIT; it++) {\n                if (it < NPRE")
        rep("        if (NPRE == 0 || !epi.residual || !n_ok) return;", "        return;")
        rep("                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r_out,", "                if (v[0] == 123.456f) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r_out,")
        rep("                if (epi.residual) v += res[it];\n", "")
        return s
    if n == 7:
        rep("    f32x4 ra[CH];\n", """    f32x4 ra[CH];
    unsigned a_voff[CH];
#pragma unroll
    for (int i = 0; i < CH; i++) {
        const int row = (wave * CH + i) * 8 + (lane >> 3);
        const int64_t gm = m0 + row;
        a_voff[i] = (unsigned)((((gm < M ? gm : M - 1) - m0) * lda * 4) + (((lane & 7) ^ ((row >> 1) & 7)) * 16));
    }
    const char *a_dbase = reinterpret_cast<const char *>(A + m0 * lda);
    auto dma_a = [&](int stage) {
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(a_dbase), 0, 0xffffffff, 0x00020000);
#pragma unroll
        for (int i = 0; i < CH; i++)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(
                r, (__attribute__((address_space(3))) void *)(ldsb + stage * STAGEB + (wave * CH + i) * 8 * WROWB), 16, a_voff[i], 0, 0, 0);
    };
""")
        rep("            fa[q][i][0] = *reinterpret_cast<const f16x8 *>(As + (wm + i * 32 + frow) * ROWB);\n            fa[q][i][1] = *reinterpret_cast<const f16x8 *>(As + (wm + i * 32 + frow) * ROWB + 16);",
            "            fa[q][i][0] = *reinterpret_cast<const f16x8 *>(ldsb + stage * STAGEB + (wm + i * 32) * WROWB + bfo[q][0]);\n            fa[q][i][1] = *reinterpret_cast<const f16x8 *>(ldsb + stage * STAGEB + (wm + i * 32) * WROWB + bfo[q][1]);")
        rep("#pragma unroll\n    for (int i = 0; i < CH; i++) ra[i] = ld_a(i);\n#pragma unroll\n    for (int i = 0; i < CH; i++) st_a(i, 0);\n", "    dma_a(0);\n    a_dbase += BK * 4;\n")
        rep("#pragma unroll\n    for (int i = 0; i < CH; i++) ra[i] = ld_a(i);\n    __builtin_amdgcn_s_waitcnt(0x0F70 | CH);", "    __builtin_amdgcn_s_waitcnt(0x0F70);")
        rep("        dma_b(s ^ 1);\n        b_base += BK * 4;\n", "        dma_b(s ^ 1);\n        b_base += BK * 4;\n        dma_a(s ^ 1);\n        a_dbase += BK * 4;\n")
        rep("            st_a(g, s ^ 1);\n            ra[g] = ld_a(g);\n", "")
        rep("        __builtin_amdgcn_s_waitcnt(0x0F70 | CH);            // vmcnt(CH): the DMA is older than the CH A loads", "        __builtin_amdgcn_s_waitcnt(0x0F70);")
        return s
    if n >= 2:
        rep("            ra[g] = ld_a(g);\n", "")
    if n >= 3 and n != 6:
        rep("            st_a(g, s ^ 1);\n", "")
    if n >= 3 and n != 5:
        rep("        dma_b(s ^ 1);\n", "")
    if n == 4:
        rep("        read_frags(s, 1);\n        __builtin_amdgcn_sched_barrier(0);\n", "        __builtin_amdgcn_sched_barrier(0);\n")
        rep("        __syncthreads();\n        read_frags(s ^ 1, 0);\n        __builtin_amdgcn_sched_barrier(0);\n        mma_range(1, NMFMA / 2, NMFMA);", "        __syncthreads();\n        __builtin_amdgcn_sched_barrier(0);\n        mma_range(1, NMFMA / 2, NMFMA);")
        rep("    read_frags(0, 0);\n    __builtin_amdgcn_s_setprio(0);", "    read_frags(0, 0); read_frags(0, 1);\n    __builtin_amdgcn_s_setprio(0);")
    return s
cs = os.path.join(ROOT, "locov_amd/csrc")
others = [os.path.join(cs, "build", f) for f in sorted(os.listdir(os.path.join(cs, "build"))) if f.endswith(".o") and f != "gemm_split.o"]
for n in [int(a) for a in sys.argv[1:]] or [1, 2, 3, 4]:
    open("/tmp/gemm_splitv.hip", "w").write(variant(n))
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fno-gpu-rdc", "-w", "-I" + os.path.join(ROOT, "include"), "-I" + cs, "-c", "/tmp/gemm_splitv.hip", "-o", "/tmp/gemm_splitv.o"])
    out = os.path.join(ROOT, "tools/liblocov_splitv%d.so" % n)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "--offload-arch=gfx950", "-fno-gpu-rdc", "-o", out, "/tmp/gemm_splitv.o"] + others)
    print("built", out)
