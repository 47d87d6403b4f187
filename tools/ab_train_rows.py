"""Developer aid: what a 64-row remainder tile could buy the training step's Winograd-domain launches.  Those GEMMs
run per transform point over R = 4 x 200 = 800 rows = 6.25 row tiles of 128: the seventh tile is three-quarters empty.  If the step's
time is a staircase in R with steps at multiples of 128, a finer tile would flatten it -- so: the LSM step at 768 / 800 / 832 / 896
sampled proposals (192 / 200 / 208 / 224 per image), median of per-step HIP-event times.
usage: python3 tools/ab_train_rows.py  ->  profiles/rNN_train_rows_ab.txt"""
import gc, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
res = {}
for rounds in range(2):
    for per in (192, 200, 208, 224):
        args = bench.parse(["--train-samples", str(per)])
        tw = bench.TrainWorkload(args, dev, "hip", 1, config="lsm")
        for _ in range(10):
            tw.step()
        gc.collect(); gc.freeze()
        ts = []
        for _ in range(40):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); tw.step(); e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        gc.unfreeze()
        ts.sort()
        res.setdefault(per, []).append(ts[len(ts) // 2])
        del tw
        torch.cuda.empty_cache()
for per, v in res.items():
    print(f"{4 * per:4d} sampled proposals ({per}/img, {4 * per / 128:.2f} row tiles of 128 per transform point): LSM step medians {' / '.join(f'{x:.2f}' for x in v)} ms"
          f" = {min(v) / (4 * per) * 1e3:.2f} us per proposal")
a, b, c = min(res[192]), min(res[200]), min(res[224])
print(f"slope between full tiles: {(c - a) / 128 * 1e3:.2f} us per proposal; 800 on that line: {a + (c - a) * 32 / 128:.2f} ms, measured {b:.2f} ms "
      f"-> the remainder tile's whole budget at 800 rows is {b - (a + (c - a) * 32 / 128):+.2f} ms of the step")
