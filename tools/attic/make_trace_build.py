"""DIAGNOSTIC build (tools/liblocov_trace.so, never the product): every workgroup of the GEMM records
(s_memtime at start, after its K-loop, at the end of its epilogue, HW_ID, XCC_ID) so that the overlap of the
workgroups sharing a CU can be read off (tools/attic/trace_gemm.py).  Enabled per launch by epilogue flag 0x2000 with the
record buffer passed in the `scale` slot."""
import os, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
s = open(os.path.join(ROOT, "locov_amd/csrc/gemm_nt.hip")).read().replace('#include "gemm_nt.h"', '#include "%s/locov_amd/csrc/gemm_nt.h"' % ROOT)
def rep(a, b):
    global s
    assert s.count(a) == 1, a
    s = s.replace(a, b)
rep('    const int tiles_n = (N + BN - 1) / BN;\n    const int nwg = gridDim.x;',
    '    unsigned long long t0_, t1_, t2_; unsigned hw_, xcc_;\n'
    '    asm volatile("s_memtime %0\\n\\ts_getreg_b32 %1, hwreg(HW_REG_HW_ID)\\n\\ts_getreg_b32 %2, hwreg(HW_REG_XCC_ID)\\n\\ts_waitcnt lgkmcnt(0)" : "=s"(t0_), "=s"(hw_), "=s"(xcc_) :: "memory");\n'
    '    unsigned long long *trc_ = (epi.flags & 0x2000u) ? (unsigned long long *)epi.scale : nullptr;\n'
    '    if (epi.flags & 0x2000u) epi.scale = nullptr;\n'
    '    const int tiles_n = (N + BN - 1) / BN;\n    const int nwg = gridDim.x;')
rep('    // Epilogue.  C/D layout of the 32x32 MFMA', '    asm volatile("s_memtime %0\\n\\ts_waitcnt lgkmcnt(0)" : "=s"(t1_) :: "memory");\n    // Epilogue.  C/D layout of the 32x32 MFMA')
rep('        return;\n    }\n\n    // General path',
    '        if (trc_ && threadIdx.x == 0) { asm volatile("s_waitcnt vmcnt(0)\\n\\ts_memtime %0\\n\\ts_waitcnt lgkmcnt(0)" : "=s"(t2_) :: "memory");\n'
    '            unsigned long long *r = trc_ + (size_t)blockIdx.x * 4; r[0] = t0_; r[1] = t1_; r[2] = t2_; r[3] = ((unsigned long long)xcc_ << 32) | hw_; }\n'
    '        return;\n    }\n\n    // General path')
# finer epilogue stamps (slots 4..6): after the first barrier, after the LDS re-layout + second barrier, after the store loop issued
STAMP = 'asm volatile("s_memtime %%0\\n\\ts_waitcnt lgkmcnt(0)" : "=s"(%s) :: "memory");'
rep('            __syncthreads();                              // every wave is done reading the last stage\n',
    '            __syncthreads();                              // every wave is done reading the last stage\n            unsigned long long ta_, tb_, tc_; ' + STAMP % 'ta_' + '\n')
rep('            __syncthreads();\n            if (n_ok) {\n                f32x4 sc',
    '            __syncthreads();\n            ' + STAMP % 'tb_' + '\n            if (n_ok) {\n                f32x4 sc')
rep('                                                           FULL ? it * vstep : 0u, 0);\n                }\n            }\n        };',
    '                                                           FULL ? it * vstep : 0u, 0);\n                }\n            }\n            ' + STAMP % 'tc_' + '\n            if (trc_ && threadIdx.x == 0) { unsigned long long *r = trc_ + (size_t)blockIdx.x * 8; r[4] = ta_; r[5] = tb_; r[6] = tc_; }\n        };')
s = s.replace('(size_t)blockIdx.x * 4; r[0] = t0_', '(size_t)blockIdx.x * 8; r[0] = t0_')
open('/tmp/gemm_trace.hip', 'w').write(s)
cs = os.path.join(ROOT, "locov_amd/csrc")
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fno-gpu-rdc", "-w", "-I" + os.path.join(ROOT, "include"), "-c", "/tmp/gemm_trace.hip", "-o", "/tmp/gemm_trace.o"])
subprocess.check_call(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "--offload-arch=gfx950", "-fno-gpu-rdc", "-o", os.path.join(ROOT, "tools/liblocov_trace.so"), "/tmp/gemm_trace.o"] + [os.path.join(cs, "build", f) for f in ("common.o", "head.o", "roi_align.o", "roi_align_nhwc.o")])
print("built tools/liblocov_trace.so")
