"""Developer aid: N consecutive training steps of bench.py's LSM / STT workloads (SGD as configured): the loss must stay finite
and fall, no range guard may trip (RuntimeWarning), step time and reserved memory must stay flat.

    python3 tools/train_soak.py [steps]                    the fixed synthetic batch (1333 x 800, 1000 proposals, 7 GT boxes)
    python3 tools/train_soak.py [steps] --multiscale       the reference's real training shapes (bench.TrainWorkload.make_pool): a
                                                           different batch every step out of a pool of 64 -- map sizes from
                                                           MIN_SIZE_TRAIN (640 ... 800) x COCO-like aspect ratios, 2 000 proposals per
                                                           image, 0-15 GT boxes (images without any included), every 8th batch with an
                                                           image that cannot fill the sampling budget
-> profiles/rNN_train_soak.txt / rNN_train_soak_multiscale.txt"""
import gc, os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

argv = [a for a in sys.argv[1:] if not a.startswith("--")]
N = int(argv[0]) if argv else 300
MULTI = "--multiscale" in sys.argv
POOL = 64
args = bench.parse([])
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
for cfg in ("lsm", "stt"):
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        tw = bench.TrainWorkload(args, dev, "hip", 1, config=cfg)
        step = tw.step
        if MULTI:
            pool = tw.make_pool(POOL, seed=11, n_props=2000, underfill_every=8)
            maps = sorted({b["map"] for b in pool})
            step = tw.step_pool
        for _ in range(POOL if MULTI else 5): step()
        tw.heads.stats.clear()
        gc.collect(); gc.freeze()
        losses, times, mem = [], [], []
        for i in range(N):
            if i % 50 == 0:
                torch.cuda.synchronize(); t0 = time.time()
            if MULTI:
                tw.use_batch(tw.pool[tw.pool_at % POOL]); tw.pool_at += 1
            tw.opt.zero_grad(set_to_none=True)
            loss, _ = tw.forward_backward()
            tw.opt.step()
            if i % 50 == 49:
                torch.cuda.synchronize(); times.append((time.time() - t0) / 50 * 1e3)
                losses.append(float(loss))
                mem.append((torch.cuda.memory_allocated(dev) >> 20, torch.cuda.memory_reserved(dev) >> 20))
        gw = [str(w.message)[:80] for w in caught if issubclass(w.category, RuntimeWarning)]
    what = (f"multi-scale pool of {POOL} batches, {len(maps)} res4 map sizes from {maps[0][0]}x{maps[0][1]} to {maps[-1][0]}x{maps[-1][1]}, "
            f"2000 proposals/img, 0-15 GT, every 8th batch under-fills") if MULTI else "fixed batch"
    print(f"{cfg} ({what}): {N} steps; loss every 50 steps: {' '.join(f'{l:.4f}' for l in losses)}; ms/step per 50: {' '.join(f'{t:.2f}' for t in times)}; "
          f"finite: {all(l == l and abs(l) < 1e30 for l in losses)}; res5_dtype at the end: {tw.heads.res5_dtype}; RuntimeWarnings: {gw or 'none'}; "
          f"retry stats: {dict(tw.heads.stats)}; "
          f"device memory allocated / reserved (MiB) per checkpoint: {' '.join(f'{a}/{r}' for a, r in mem)}", flush=True)
    del tw
    if MULTI:
        del pool
    gc.unfreeze(); torch.cuda.empty_cache()
