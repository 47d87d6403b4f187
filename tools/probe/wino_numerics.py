"""Developer probe (numpy, CPU): rounding error of alternative Winograd segmentations of the 7-wide, 3-tap, pad-1 correlation against the shipped
F(4,3)|F(3,3) form, with the kernels' arithmetic (fp32 transforms, operands cut to the 22 bits of the split layout, fp32 accumulate) on post-ReLU
activations x N(0, 0.02) weights, 512 input channels.  DESIGN.md section 8 / docs/experiments.md cite its output:
    python tools/probe/wino_numerics.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
from fractions import Fraction as Fr
from gen_winograd_tables import toom_cook

def build(segs):
    nf = sum(s["m"] + 2 for s in segs)
    BT = [[Fr(0)] * 7 for _ in range(nf)]; G = [[Fr(0)] * 3 for _ in range(nf)]; AT = [[Fr(0)] * nf for _ in range(7)]
    f0 = 0
    for s in segs:
        bt, g, at = toom_cook(s["m"], 3, s["pts"])
        n = s["m"] + 2
        for i in range(n):
            for j in range(n):
                x = s["in0"] + j
                if 0 <= x < 7: BT[f0 + i][x] = bt[i][j]
            G[f0 + i] = g[i]
            for k in range(s["m"]): AT[s["out0"] + k][f0 + i] = at[k][i]
        f0 += n
    f = lambda m: np.array([[float(v) for v in r] for r in m])
    return f(BT), f(G), f(AT)

def q22(x):
    # hi = fp16(x), lo = fp16(x - hi) on a tensor scaled into [2^12, 2^13) by its amax (the split layout)
    s = 2.0 ** (12 - np.floor(np.log2(np.abs(x).max())))
    xs = (x * s).astype(np.float32)
    hi = xs.astype(np.float16).astype(np.float32)
    lo = (xs - hi).astype(np.float16).astype(np.float32)
    return ((hi + lo) / np.float32(s)).astype(np.float32)

def run(name, segs, R=48, C=512, N=32, seed=0, rowscale=None):
    bt, g, at = build(segs)
    nf = bt.shape[0]
    rng = np.random.default_rng(seed)
    x = np.maximum(rng.standard_normal((R, C, 7, 7)), 0) * rng.lognormal(0, 1, (1, C, 1, 1))   # post-ReLU activations, channel scales
    w = rng.standard_normal((N, C, 3, 3)) * 0.02
    xp = np.pad(x, ((0, 0), (0, 0), (1, 1), (1, 1)))
    want = np.zeros((R, N, 7, 7))
    for dy in range(3):
        for dx in range(3):
            want += np.einsum('rcyx,nc->rnyx', xp[:, :, dy:dy + 7, dx:dx + 7], w[:, :, dy, dx])
    bt32 = bt.astype(np.float32); at32 = at.astype(np.float32)
    x32 = x.astype(np.float32)
    # V in fp32 (two passes, as the kernel), then the 22-bit split
    t = np.einsum('fy,rcyx->rcfx', bt32, x32).astype(np.float32)
    V = np.einsum('gx,rcfx->rcfg', bt32, t).astype(np.float32)
    V = q22(V)
    U = np.einsum('fy,ncyx,gx->ncfg', g, w, g)
    U = q22(U.astype(np.float32))
    M = np.einsum('rcfg,ncfg->rnfg', V.astype(np.float64), U.astype(np.float64)).astype(np.float32)  # fp32 accumulate ~ exact products
    t = np.einsum('yf,rnfg->rnyg', at32, M).astype(np.float32)
    Y = np.einsum('xg,rnyg->rnyx', at32, t).astype(np.float32)
    err = np.abs(Y - want)
    rms = np.sqrt((want ** 2).mean())
    print('%-34s NF=%2d  max err / rms %.2e   rms err / rms %.2e   max|BT| %.1f max|AT| %.1f' % (name, nf, err.max() / rms, np.sqrt((err**2).mean()) / rms, np.abs(bt).max(), np.abs(at).max()))

run('F(4,3)|F(3,3) (shipped)', [dict(m=4, pts=[0, 1, -1, 2, -2], in0=-1, out0=0), dict(m=3, pts=[0, 1, -1, 2], in0=3, out0=4)])
h = Fr(1, 2); q = Fr(1, 4)
for pts in ([0, 1, -1, 2, -2, h, -h, 4], [0, 1, -1, 2, -2, h, -h, -q], [0, 1, -1, 2, -2, h, -h, Fr(3,2)], [0,1,-1,h,-h,2,-2,Fr(-3,2)], [0,1,-1,h,-h,Fr(3,2),-Fr(3,2),2],
            [0,1,-1,h,-h,2,-2,Fr(3,4)], [0,1,-1,h,-h,2,-2,3]):
    run('F(7,3) ' + ','.join(str(p) for p in pts), [dict(m=7, pts=pts, in0=-1, out0=0)])
run('F(5,3)|F(2,3)', [dict(m=5, pts=[0, 1, -1, 2, -2, h], in0=-1, out0=0), dict(m=2, pts=[0, 1, -1], in0=4, out0=5)])
run('F(6,3) + F(1,3)', [dict(m=6, pts=[0, 1, -1, 2, -2, h, -h], in0=-1, out0=0), dict(m=1, pts=[0, 1], in0=5, out0=6)])


# one axis F(7,3), the other the shipped form: 99 points
def run2(name, segs_y, segs_x, R=48, C=512, N=32, seed=0):
    bty, gy, aty = build(segs_y); btx, gx, atx = build(segs_x)
    rng = np.random.default_rng(seed)
    x = np.maximum(rng.standard_normal((R, C, 7, 7)), 0) * rng.lognormal(0, 1, (1, C, 1, 1))
    w = rng.standard_normal((N, C, 3, 3)) * 0.02
    xp = np.pad(x, ((0, 0), (0, 0), (1, 1), (1, 1)))
    want = np.zeros((R, N, 7, 7))
    for dy in range(3):
        for dx in range(3):
            want += np.einsum('rcyx,nc->rnyx', xp[:, :, dy:dy + 7, dx:dx + 7], w[:, :, dy, dx])
    f32=np.float32
    t = np.einsum('fy,rcyx->rcfx', bty.astype(f32), x.astype(f32)).astype(f32)
    V = q22(np.einsum('gx,rcfx->rcfg', btx.astype(f32), t).astype(f32))
    U = q22(np.einsum('fy,ncyx,gx->ncfg', gy, w, gx).astype(f32))
    M = np.einsum('rcfg,ncfg->rnfg', V.astype(np.float64), U.astype(np.float64)).astype(f32)
    t = np.einsum('yf,rnfg->rnyg', aty.astype(f32), M).astype(f32)
    Y = np.einsum('xg,rnyg->rnyx', atx.astype(f32), t).astype(f32)
    err = np.abs(Y - want); rms = np.sqrt((want ** 2).mean())
    print('%-50s points %3d  max err/rms %.2e  rms err/rms %.2e' % (name, bty.shape[0]*btx.shape[0], err.max()/rms, np.sqrt((err**2).mean())/rms))
h=Fr(1,2)
hyb=[dict(m=4, pts=[0, 1, -1, 2, -2], in0=-1, out0=0), dict(m=3, pts=[0, 1, -1, 2], in0=3, out0=4)]
run2('hybrid x hybrid (shipped)', hyb, hyb)
for pts in ([0,1,-1,h,-h,2,-2,Fr(3,4)], [0,1,-1,h,-h,2,-2,-Fr(1,4)], [0,1,-1,h,-h,Fr(3,2),-Fr(3,2),2]):
    f7=[dict(m=7, pts=pts, in0=-1, out0=0)]
    run2('F(7,3)%s x hybrid'%str([str(p) for p in pts]), f7, hyb)
