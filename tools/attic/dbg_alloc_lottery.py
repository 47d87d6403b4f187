"""Developer experiment: the same GEMM launch writing into DIFFERENT freshly allocated output / residual buffers (held alive, so
each is a new hipMalloc): is the launch time a property of the buffer (physical placement / page fragments)?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from locov_amd import ops, _lib
from locov_amd.ops import _ptr, _stream, _overflow_word, check
lib = _lib.load()
g = torch.Generator().manual_seed(0)
M, N, K = 196000, 2048, 512
x = torch.relu(torch.randn(M, K, generator=g)).cuda()
w = (torch.randn(N, K, generator=g) * 0.05).cuda()
wp, xs = ops.split_pack(w), ops.split_pack(x, 16.0)
keep = []
def measure(yb, rb):
    def f():
        check(lib.locov_gemm_nt_f32_split(_ptr(xs.data), K, _ptr(wp.data), None, None, _ptr(rb), _ptr(yb), N, M, N, K,
                                          _lib.EPI_RELU | _lib.GEMM_A_SPLIT, 16.0, wp.scale, _ptr(_overflow_word(x)), _stream(x)), "gemm")
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / 5 * 1e3
for i in range(10):
    yb = torch.empty((M, N), device="cuda"); rb = torch.randn((M, N), device="cuda")
    keep += [yb, rb]
    t1 = measure(yb, rb); t2 = measure(yb, rb)
    print(f"buffers {i}: y @ {yb.data_ptr():#x} (mod 2 MB {yb.data_ptr() % (2 << 20):#x})  r @ {rb.data_ptr():#x}   {t1:.3f} / {t2:.3f} ms", flush=True)
# cross: y of pair a with r of pair b
for a, b in ((0, 1), (1, 0), (2, 5), (5, 2)):
    print(f"y{a} + r{b}: {measure(keep[2 * a], keep[2 * b + 1]):.3f} ms")

# one output buffer rewritten by every launch vs two used alternately (what a caller that allocates its result per call sees)
def measure2(ys, rb, n=6):
    def f(i):
        check(lib.locov_gemm_nt_f32_split(_ptr(xs.data), K, _ptr(wp.data), None, None, _ptr(rb), _ptr(ys[i % len(ys)]), N, M, N, K,
                                          _lib.EPI_RELU | _lib.GEMM_A_SPLIT, 16.0, wp.scale, _ptr(_overflow_word(x)), _stream(x)), "gemm")
    f(0); f(1); torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n): f(i)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for rep in range(3):
    print(f"one output buffer {measure2([keep[2]], keep[3]):.3f} ms   two alternating {measure2([keep[2], keep[4]], keep[3]):.3f} ms   "
          f"fresh torch.empty per launch {measure2([torch.empty((M, N), device='cuda') for _ in range(6)], keep[3]):.3f} ms", flush=True)
