"""DIAGNOSTIC builds of gemm_split.hip (tools/liblocov_splitv<N>.so, never the product; results are WRONG on purpose):
ablations that remove one cost at a time from the K-loop, to see what bounds the kernel.  Run with
LOCOV_HIP_LIB=tools/liblocov_splitv<N>.so python tools/attic/bench_split.py speed
  1: no fp32 -> (hi, lo) conversion (raw bits stored)     2: also no A refill loads in the K-loop
  3: no staging at all (no loads, no LDS writes, no DMA)  4: no fragment reads either (MFMA-only loop)
  7: A taken as ALREADY split and staged by LDS DMA like W (what the kernel would do if the producers wrote the
     activations in split format): timing only
 11: conversion executed but raw bits stored (variant 1's data with variant 0's instruction stream)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SRC = open(os.path.join(ROOT, "locov_amd/csrc/gemm_split.hip")).read().replace('#include "gemm_nt.h"', '#include "%s/locov_amd/csrc/gemm_nt.h"' % ROOT)
def variant(n):
    s = SRC
    def rep(a, b):
        nonlocal s
        assert s.count(a) == 1, a
        s = s.replace(a, b)
    if n == 1:
        rep("        split4(ra[i], a_scale, hi, lo);", "        hi = u32x2{__builtin_bit_cast(unsigned, ra[i][0]), __builtin_bit_cast(unsigned, ra[i][1])}; lo = u32x2{__builtin_bit_cast(unsigned, ra[i][2]), __builtin_bit_cast(unsigned, ra[i][3])};")
    if n == 11:      # the conversion is executed, but raw bits are stored (as variant 1): separates the conversion's issue cost
                     # from the effect the (garbage) operand data of variant 1 has on the matrix pipe's power / clock
        rep("        split4(ra[i], a_scale, hi, lo);", "        split4(ra[i], a_scale, hi, lo);\n        asm volatile(\"\" :: \"v\"(hi[0]), \"v\"(hi[1]), \"v\"(lo[0]), \"v\"(lo[1]));\n        hi = u32x2{__builtin_bit_cast(unsigned, ra[i][0]), __builtin_bit_cast(unsigned, ra[i][1])}; lo = u32x2{__builtin_bit_cast(unsigned, ra[i][2]), __builtin_bit_cast(unsigned, ra[i][3])};")
        return s
    if n == 7:
        rep("    f32x4 ra[CH];\n", """    f32x4 ra[CH];
    unsigned a_voff[CH];
#pragma unroll
    for (int i = 0; i < CH; i++) {
        const int row = (wave * CH + i) * 8 + (lane >> 3);
        const int64_t gm = m0 + row;
        a_voff[i] = (unsigned)((((gm < M ? gm : M - 1) - m0) * lda * 4) + (((lane & 7) ^ wswz(row)) * 16));
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
        rep("        const char *As = ldsb + stage * STAGEB + (wm + ga * 32) * ROWB + afo;", "        const char *As = ldsb + stage * STAGEB + (wm + ga * 32) * WROWB;")
        rep("            fa[2 * ga + i][0] = *reinterpret_cast<const f16x8 *>(As + i * 16 * ROWB);\n            fa[2 * ga + i][1] = *reinterpret_cast<const f16x8 *>(As + i * 16 * ROWB + 32);",
            "            fa[2 * ga + i][0] = *reinterpret_cast<const f16x8 *>(As + i * 16 * WROWB + bfo[0]);\n            fa[2 * ga + i][1] = *reinterpret_cast<const f16x8 *>(As + i * 16 * WROWB + bfo[1]);")
        rep("#pragma unroll\n    for (int i = 0; i < CH; i++) ra[i] = ld_a(i);\n#pragma unroll\n    for (int i = 0; i < CH; i++) st_a(i, 0);\n", "    dma_a(0);\n    a_dbase += BK * 4;\n")
        rep("#pragma unroll\n    for (int i = 0; i < CH; i++) ra[i] = ld_a(i);\n    __builtin_amdgcn_s_waitcnt(0x0F70 | CH);", "    __builtin_amdgcn_s_waitcnt(0x0F70);")
        rep("        dma_b(s ^ 1);\n        b_base += BK * 4;\n", "        dma_b(s ^ 1);\n        b_base += BK * 4;\n        dma_a(s ^ 1);\n        a_dbase += BK * 4;\n")
        rep("            st_a(g, s ^ 1);\n            ra[g] = ld_a(g);\n", "")
        rep("        __builtin_amdgcn_s_waitcnt(0x0F70 | CH);            // vmcnt(CH): the DMA is older than the CH A loads", "        __builtin_amdgcn_s_waitcnt(0x0F70);")
        return s
    if n >= 2:
        rep("            ra[g] = ld_a(g);\n", "")
    if n >= 3:
        rep("            st_a(g, s ^ 1);\n", "")
        rep("        dma_b(s ^ 1);\n", "")
    if n == 4:
        rep("        rd_b(s, y);\n        rd_a(s, 1);\n        __builtin_amdgcn_sched_barrier(0);\n", "        __builtin_amdgcn_sched_barrier(0);\n")
        rep("        rd_a(s ^ 1, 0);\n        __builtin_amdgcn_sched_barrier(0);\n        quarter(1, y, 0, NQM);\n        __builtin_amdgcn_sched_barrier(0);\n        rd_b(s ^ 1, y);\n", "        quarter(1, y, 0, NQM);\n        __builtin_amdgcn_sched_barrier(0);\n")
        rep("    rd_a(0, 0);\n    rd_b(0, 0);\n    __builtin_amdgcn_s_setprio(0);", "    rd_a(0, 0); rd_a(0, 1); rd_b(0, 0); rd_b(0, 1);\n    __builtin_amdgcn_s_setprio(0);")
    return s
cs = os.path.join(ROOT, "locov_amd/csrc")
others = [os.path.join(cs, "build", f) for f in sorted(os.listdir(os.path.join(cs, "build"))) if f.endswith(".o") and f != "gemm_split.o"]
for n in [int(a) for a in sys.argv[1:]] or [1, 2, 3, 4]:
    open("/tmp/gemm_splitv.hip", "w").write(variant(n))
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fno-gpu-rdc", "-w", "-I" + os.path.join(ROOT, "include"), "-I" + cs, "-c", "/tmp/gemm_splitv.hip", "-o", "/tmp/gemm_splitv.o"])
    out = os.path.join(ROOT, "tools/liblocov_splitv%d.so" % n)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "--offload-arch=gfx950", "-fno-gpu-rdc", "-o", out, "/tmp/gemm_splitv.o"] + others)
    print("built", out)
