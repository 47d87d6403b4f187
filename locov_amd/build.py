"""Builds locov_amd/liblocov_hip.so in-tree with hipcc for gfx950 (no JIT cache).

    python -m locov_amd.build [--force]

Each csrc/*.hip is compiled to an object (in parallel) and linked into one shared library
that exports the C ABI of include/locov_hip.h.  hipcc cross-compiles without a GPU.
"""
from __future__ import annotations

import concurrent.futures
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ_DIR = os.path.join(CSRC, "build")
LIB_PATH = os.path.join(HERE, "liblocov_hip.so")
ARCH = "gfx950"

CXXFLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-fno-gpu-rdc",
            "-Wall", "-Wno-unused-function"]


# ROIAlign coordinates/weights must be un-fused fp32 (bit-exact with the oracle).  HIP's
# default -ffp-contract=fast lets the backend fuse globally, which a source pragma cannot
# veto, so these files are compiled with contraction off (they call fmaf() where they want it).
FILE_FLAGS = {
    "roi_align.hip": ["-ffp-contract=off"],
    "roi_align_nhwc.hip": ["-ffp-contract=off"],
    "roi_align_tiles.hip": ["-ffp-contract=off"],
    "label.hip": ["-ffp-contract=off"],          # (IoU values must be the torch ops', rounded step by step)
    "losses.hip": ["-ffp-contract=off"],         # (box deltas / log-softmax pieces as the torch ops form them)
    "detect.hip": ["-ffp-contract=off"],         # (box decoding / shifted IoU as the torch ops round them)
}


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm; set HIPCC=/path/to/hipcc)")


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _headers():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs.append(os.path.join(os.path.dirname(HERE), "include", "locov_hip.h"))
    return hs


def source_fingerprint() -> str:
    """sha256[:16] over the kernel sources and headers the library is built from (what a recorded PMC pass is valid for)."""
    import hashlib
    h = hashlib.sha256()
    for path in sorted(sources() + _headers()):
        h.update(os.path.basename(path).encode())
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src: str, force: bool) -> str:
    obj = os.path.join(OBJ_DIR, os.path.basename(src)[:-4] + ".o")
    if force or _stale(obj, [src] + _headers()):
        cmd = [_hipcc()] + CXXFLAGS + FILE_FLAGS.get(os.path.basename(src), []) + ["-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
        if r.stderr.strip():
            sys.stderr.write(r.stderr)
    return obj


def build_extension(force: bool = False, jobs: int = 4) -> str:
    os.makedirs(OBJ_DIR, exist_ok=True)
    srcs = sources()
    with concurrent.futures.ThreadPoolExecutor(max_workers=jobs) as ex:
        objs = list(ex.map(lambda s: _compile(s, force), srcs))
    if force or _stale(LIB_PATH, objs):
        cmd = [_hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", "-fno-gpu-rdc", "-o", LIB_PATH] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
    return LIB_PATH


if __name__ == "__main__":
    print(build_extension(force="--force" in sys.argv))
