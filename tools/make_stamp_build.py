"""Generates and builds a DIAGNOSTIC copy of csrc/gemm_nt.hip with s_memtime stamps around the K-loop
phases (tools/liblocov_stamp.so; never the product library).  flag 0x4000: every workgroup reads the
same A rows (everything L2-resident) to separate memory latency from issue effects."""
import os, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = open(os.path.join(ROOT, "locov_amd/csrc/gemm_nt.hip")).read()
s = src.replace('#include "gemm_nt.h"', '#include "%s/locov_amd/csrc/gemm_nt.h"' % ROOT)
stamp = '''
#define LOCOV_STAMP(i) do { __builtin_amdgcn_sched_barrier(0); unsigned long long t_; \\
    asm volatile("s_memtime %0\\n\\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \\
    __builtin_amdgcn_sched_barrier(0); if ((i) >= 0) dsum[(i) < 0 ? 0 : (i)] += t_ - tprev; tprev = t_; } while (0)
'''
def rep(a, b, cnt=1):
    global s
    assert s.count(a) >= 1, a
    s = s.replace(a, b, cnt)
rep('namespace locov {\n', 'namespace locov {\n' + stamp)
rep('    for (int k0 = 0; k0 < k_last; k0 += BK) {               // tile at k0 has a successor',
    '    unsigned long long dsum[8] = {0,0,0,0,0,0,0,0}, tprev = 0;\n    for (int k0 = 0; k0 < k_last; k0 += BK) {\n        LOCOV_STAMP(-1);')
# generic sub-step loop: stamp index = q for q < NQ-1 (NQ = 4: A,B,C), then D1 / barrier / D2
rep('                __builtin_amdgcn_sched_barrier(0);\n            } else {\n                // last sub-step', '                __builtin_amdgcn_sched_barrier(0);\n                LOCOV_STAMP(q);\n            } else {\n                // last sub-step')
rep('                __syncthreads();\n                read_frags(s ^ 1, 0, nxt);', '                LOCOV_STAMP(3);\n                __syncthreads();\n                LOCOV_STAMP(4);\n                read_frags(s ^ 1, 0, nxt);')
rep('        s ^= 1;\n    }\n    // last tile', '        LOCOV_STAMP(5);\n        s ^= 1;\n    }\n    // last tile')
rep('    // Epilogue.  C/D layout', '''    if (blockIdx.x == 300 && threadIdx.x == 0 && (epi.flags & 0x2000u)) {
        unsigned long long *dbg = (unsigned long long *)epi.scale;
        for (int i = 0; i < 8; i++) dbg[i] = dsum[i];
    }
    if (epi.flags & 0x2000u) epi.scale = nullptr;
    // Epilogue.  C/D layout''')
rep('        const int64_t gm = m0 + row;\n        a_ok[i] = gm < M;', '        const int64_t gm = ((epi.flags & 0x4000u) ? 0 : m0) + row;\n        a_ok[i] = gm < M;')
rep('    const int tiles_n = (N + BN - 1) / BN;', '    unsigned long long tb_, rb_, tl0_, tl1_;\n    asm volatile("s_memtime %0\\n\\ts_memrealtime %1\\n\\ts_waitcnt lgkmcnt(0)" : "=s"(tb_), "=s"(rb_) :: "memory");\n    const int tiles_n = (N + BN - 1) / BN;')
rep('    unsigned long long dsum[8]', '    asm volatile("s_memtime %0\\n\\ts_waitcnt lgkmcnt(0)" : "=s"(tl0_) :: "memory");\n    unsigned long long dsum[8]')
rep('    if (blockIdx.x == 300 && threadIdx.x == 0 && (epi.flags & 0x2000u)) {', '    asm volatile("s_memtime %0\\n\\ts_waitcnt lgkmcnt(0)" : "=s"(tl1_) :: "memory");\n    if (blockIdx.x == 300 && threadIdx.x == 0 && (epi.flags & 0x2000u)) {\n        ((unsigned long long *)epi.scale)[8] = tb_; ((unsigned long long *)epi.scale)[9] = rb_; ((unsigned long long *)epi.scale)[10] = tl0_; ((unsigned long long *)epi.scale)[11] = tl1_;')
# end-of-kernel stamp for the vec epilogue path: before its return
rep('        return;\n    }\n\n    // General path', '        if (blockIdx.x == 300 && threadIdx.x == 0 && dbgp_) { unsigned long long te_, re_; asm volatile("s_memtime %0\\n\\ts_memrealtime %1\\n\\ts_waitcnt lgkmcnt(0)" : "=s"(te_), "=s"(re_) :: "memory"); dbgp_[12] = te_; dbgp_[13] = re_; }\n        return;\n    }\n\n    // General path')
rep('    if (epi.flags & 0x2000u) epi.scale = nullptr;', '    unsigned long long *dbgp_ = (epi.flags & 0x2000u) ? (unsigned long long *)epi.scale : nullptr;\n    if (epi.flags & 0x2000u) epi.scale = nullptr;')
# ablation switches for the staging phase (timing only -- results are wrong when set)
rep('                As[(idx / BK16) * LDS16 + idx % BK16] = ra[g];\n                a_ptr[g] += da;\n                ra[g] = *reinterpret_cast<const frag_t *>(a_ptr[g]);',
    '                if (!(epi.flags & 0x10000u)) As[(idx / BK16) * LDS16 + idx % BK16] = ra[g];\n                a_ptr[g] += da;\n                if (!(epi.flags & 0x20000u)) ra[g] = *reinterpret_cast<const frag_t *>(a_ptr[g]);')
rep('                Bs[(idx / BK16) * LDS16 + idx % BK16] = rb[h];\n                b_ptr[h] += db;\n                rb[h] = *reinterpret_cast<const frag_t *>(b_ptr[h]);',
    '                if (!(epi.flags & 0x10000u)) Bs[(idx / BK16) * LDS16 + idx % BK16] = rb[h];\n                b_ptr[h] += db;\n                if (!(epi.flags & 0x20000u)) rb[h] = *reinterpret_cast<const frag_t *>(b_ptr[h]);')
open('/tmp/gemm_stamp.hip', 'w').write(s)
cs = os.path.join(ROOT, "locov_amd/csrc")
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fno-gpu-rdc", "-w", "-c", "/tmp/gemm_stamp.hip", "-o", "/tmp/gemm_stamp.o"])
subprocess.check_call(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "--offload-arch=gfx950", "-fno-gpu-rdc", "-o", os.path.join(ROOT, "tools/liblocov_stamp.so"), "/tmp/gemm_stamp.o"] + [os.path.join(cs, "build", f) for f in ("common.o", "head.o", "roi_align.o", "roi_align_nhwc.o")])
print("built tools/liblocov_stamp.so")
