#!/usr/bin/env python3
"""bench_train.py -- BASELINE.json configs[1]: CAE training (Conv2D enc/dec, MSE loss) on 50k
synthetic 64x64 crops, fp32, one MI355X.  Secondary benchmark (the headline metric is
bench.py's screening rate).  One "step" = one fit() batch of 32 crops: forward with
BatchNormalization in training mode, MSE loss, backward, Adam -- CAE_improved_modeltrain.py:286-293.
Crops are resident in HBM; the 80/20 split gives 1250 steps per epoch as in the reference.

    python bench_train.py [--steps K] [--warmup W] [--batch 32] [--gpus N via torchrun]
With N > 1 ranks the step is the one cellscreen/training.py ships (ImprovedAnomalyDetectionTraining(data_parallel=True)): ONE
global batch of --batch cells split over the ranks, BatchNormalization statistics over the whole batch (all-gather of the
per-rank partials, cs_train_set_sync_bn), the 337 KB gradient averaged with one RCCL all-reduce, no host synchronisation
beyond what those exchanges need (the wrappers order the library's stream and torch's).  --per-gpu-batch keeps --batch cells
PER rank instead (global batch N x 32: not the reference's step; labelled so in the line)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, "cell-image-analysis_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

FLOP_PER_CELL_STEP = 3 * 100_270_080     # forward + backward-data + backward-weight (conv MACs x2 each)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1250)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--cells", type=int, default=50_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--per-gpu-batch", action="store_true", help="N > 1: --batch cells per rank instead of one global batch split over the ranks")
    ap.add_argument("--no-sync-bn", action="store_true", help="N > 1: per-rank BatchNormalization statistics")
    ap.add_argument("--variant", action="store_true",
                    help="BASELINE.json configs[4]'s architecture instead: 128x128 crops, filters 32-64-128 | 128-64-32-1 (generic trainer)")
    args = ap.parse_args()
    import numpy as np
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1")); rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        sys.exit("bench_train.py needs an MI355X")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    from cellscreen import dist as csdist
    from cellscreen import synth
    from cellscreen.trainer import Trainer

    n_train = int(args.cells * 0.8)
    hw, ch, n_enc = ((128, 128), (32, 64, 128, 128, 64, 32, 1), 3) if args.variant else ((64, 64), (32, 64, 32, 32, 64, 32, 1), 3)
    flop_step = 3 * 2 * 349.18e6 if args.variant else FLOP_PER_CELL_STEP
    if args.variant:
        n_train = min(n_train, 4096)
    X = torch.from_numpy(synth.blob_crops(42 + rank, min(n_train, 4096), hw=hw)).to(dev)   # structured crops; reused cyclically
    reps = (n_train + len(X) - 1) // len(X)
    X = X.repeat(reps, 1, 1)[:n_train].contiguous()
    tr = Trainer(synth.random_cae(seed=42, hw=hw, channels=ch, n_enc=n_enc, trivial_bn=True), device_id=local_rank)
    g = torch.zeros(tr.n_trainable, dtype=torch.float32, device=dev)
    local_b = args.batch
    if world > 1:
        tr.use_grad_tensor(g)
        if not args.per_gpu_batch:
            if args.batch % world:
                sys.exit("--batch %d does not split over %d ranks" % (args.batch, world))
            local_b = args.batch // world
        if not args.no_sync_bn and not args.variant:
            tr.enable_sync_bn(dist, rank, world)
    gen = torch.Generator(device=dev); gen.manual_seed(1234 + rank)

    def step():
        idx = torch.randint(0, n_train, (local_b,), device=dev, generator=gen)
        xb = X[idx].contiguous()
        if world > 1:
            l, _ = tr.forward_backward(xb, xb)    # the wrapper orders the library's stream after torch's (the gather above)
            csdist.allreduce_mean_(g)
            tr.apply(1e-3)                        # ... and the update after the all-reduce (Trainer.apply): no host sync of ours
        else:
            l, _ = tr.step(xb, xb, 1e-3)
        return l

    first = None
    for _ in range(args.warmup):
        l = step(); first = l if first is None else first
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        last = step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([el], dtype=torch.float64, device=dev); dist.all_reduce(t, op=dist.ReduceOp.MAX); el = float(t.item())
    if rank == 0:
        cells = args.steps * local_b * world
        line = {"metric": "cells/sec trained (CAE fwd+bwd+Adam, batch 32, fp32)", "value": round(cells / el, 1), "unit": "cells/s",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(el / args.steps * 1e3, 4),
                "higher_is_better": True, "scaling": "weak" if (world == 1 or args.per_gpu_batch) else "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "config": {"workload": ("BASELINE.json configs[4] architecture (128x128, filters 32-64-128 | 128-64-32-1): generic trainer, batch %d per GPU" % args.batch)
                                       if args.variant else
                                       ("BASELINE.json configs[1]: CAE training on 50k synthetic 64x64 crops (40k train), " +
                                        ("batch %d on one GPU" % args.batch if world == 1 else
                                         ("global batch %d = %d per GPU x %d (NOT the reference's batch of 32), " % (local_b * world, local_b, world) if args.per_gpu_batch else
                                          "ONE batch of %d split over %d GPUs (%d cells each, the step training.py ships), " % (args.batch, world, local_b)) +
                                         ("per-rank BatchNormalization statistics" if args.no_sync_bn or args.variant else "BatchNormalization over the whole batch (all-gathered partials)") +
                                         ", gradient all-reduce over RCCL")),
                           "batch_per_gpu": local_b, "global_batch": local_b * world, "steps_per_epoch": n_train // (local_b * world), "parallelism": "dp%d" % world},
                "tflops_algorithmic": round(cells / el * flop_step / 1e12 / world, 3),
                "epoch_seconds_at_1250_steps": round(el / args.steps * 1250, 3),
                "loss_first_last": [round(first, 6), round(last, 6)]}
        if world == 1 and not args.no_cpu_baseline and not args.variant:
            from oracle import train_oracle as T
            st = T.TrainState(synth.random_cae(seed=42, trivial_bn=True), dtype=np.float32)
            xb = X[:args.batch].cpu().numpy()
            T.train_step(st, xb, xb)
            t0 = time.perf_counter(); nrep = 5
            for _ in range(nrep):
                T.train_step(st, xb, xb)
            dt = (time.perf_counter() - t0) / nrep
            line["cpu_baseline"] = {"value": round(args.batch / dt, 1), "unit": "cells/s", "cores": os.cpu_count(), "kind": "port",
                                    "sample": "%d steps of batch %d, oracle/train_oracle.py (numpy float32, BLAS threads)" % (nrep, args.batch)}
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier(); dist.destroy_process_group()
    tr.close()


if __name__ == "__main__":
    main()
