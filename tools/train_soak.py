"""Developer aid: N consecutive training steps of bench.py's LSM / STT workloads (fixed synthetic batch, SGD as configured): the loss must stay finite
and fall, no range guard may trip (RuntimeWarning), and the step time must stay flat.  usage: python3 tools/train_soak.py [steps]"""
import gc, os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
args = bench.parse([])
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
for cfg in ("lsm", "stt"):
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        tw = bench.TrainWorkload(args, dev, "hip", 1, config=cfg)
        for _ in range(5): tw.step()
        gc.collect(); gc.freeze()
        losses, times, mem = [], [], []
        for i in range(N):
            if i % 50 == 0:
                torch.cuda.synchronize(); t0 = time.time()
            tw.opt.zero_grad(set_to_none=True)
            loss, _ = tw.forward_backward()
            tw.opt.step()
            if i % 50 == 49:
                torch.cuda.synchronize(); times.append((time.time() - t0) / 50 * 1e3)
                losses.append(float(loss))
                mem.append((torch.cuda.memory_allocated(dev) >> 20, torch.cuda.memory_reserved(dev) >> 20))
        gw = [str(w.message)[:80] for w in caught if issubclass(w.category, RuntimeWarning)]
    print(f"{cfg}: {N} steps; loss every 50 steps: {' '.join(f'{l:.4f}' for l in losses)}; ms/step per 50: {' '.join(f'{t:.2f}' for t in times)}; "
          f"finite: {all(l == l and abs(l) < 1e30 for l in losses)}; res5_dtype at the end: {tw.heads.res5_dtype}; RuntimeWarnings: {gw or 'none'}; "
          f"device memory allocated / reserved (MiB) at the first and last checkpoint: {mem[0]} -> {mem[-1]}", flush=True)
    del tw
    gc.unfreeze(); torch.cuda.empty_cache()
