"""Developer aid: builds tools/liblocov_<tag>.so = the product library with ONE csrc file recompiled under extra -D flags
(A/B experiments in one gpurun call: LOCOV_HIP_LIB=tools/liblocov_<tag>.so python tools/...).  Never the product.
usage: python tools/make_variant.py <tag> <file.hip> -DX=1 [-DY=2 ...]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from locov_amd import build
tag, fname, flags = sys.argv[1], sys.argv[2], sys.argv[3:]
build.build_extension()
src = os.path.join(build.CSRC, fname)
obj = f"/tmp/locov_{tag}_{fname[:-4]}.o"
subprocess.check_call([build._hipcc()] + build.CXXFLAGS + build.FILE_FLAGS.get(fname, []) + flags + ["-c", src, "-o", obj])
objs = [os.path.join(build.OBJ_DIR, os.path.basename(s)[:-4] + ".o") for s in build.sources() if os.path.basename(s) != fname] + [obj]
out = os.path.join(ROOT, "tools", f"liblocov_{tag}.so")
subprocess.check_call([build._hipcc(), "-shared", "-fPIC", f"--offload-arch={build.ARCH}", "-fno-gpu-rdc", "-o", out] + objs)
print(out)
