#!/usr/bin/env python3
"""bench.py -- cells/sec screened on MI355X (BASELINE.json metric), one process per GPU.

    python bench.py --gpus N --steps K --warmup W          (N = 1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path -- compute_anomaly_scores, improved_detection.py:117-153:
CAE forward + per-cell MSE/MAE + encoder features + RobustScaler + PCA + 2x one-class-SVM
score -- over this rank's batch of synthetic 64x64 crops that are already resident in HBM,
ending with the per-cell results on the host of rank 0 (for N > 1: after the RCCL gather).
N = 1 runs BASELINE.json configs[2] (1M crops on one GPU); N > 1 shards N x 1M crops by
contiguous index ranges (weak scaling, no data-path collective before the final gather).

Rank 0 prints ONE JSON line; besides the contract keys it carries
  roofline      the dominant kernel priced against the exact-fp32 MFMA peak, from HIP events
                recorded on the library's stream inside the timed steps
  cpu_baseline  the CPU oracle (oracle/cae_oracle.c, a port of the same path) timed on this
                host's cores on a bounded sample of the same workload (rank 0, N = 1 only)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, "cell-image-analysis_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

FP32_MFMA_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md, "Peak FP32 (matrix)"
HBM_PEAK_GBS = 8000.0              # spec
FLOP_PER_CELL = 100_270_080        # ALGORITHMIC: 2 x 50,135,040 conv MACs of the reference graph, SURVEY.md section 8d
# EXECUTED multiply-adds as a fraction of the algorithmic ones, per kernel (all exact algebra): conv2 and
# conv3 are Winograd F(2x2,3x3) convolutions (4/9); the convs behind an UpSampling2D are four 2x2-tap phase
# convs with pre-summed weights (4/9: conv7), conv5 and conv6 additionally as Winograd F(2x2,2x2) per phase
# (9/16 of 4/9 = 1/4); conv1 pads K = 9 to 12 for the 16x16x4 MFMA.
EXEC_FRACTION = {"conv1_relu_bn_pool": 12.0 / 9.0,
                 "conv5_up_relu_bn": 4.0 / 9.0 if (os.environ.get("CS_NO_WINOGRAD") or os.environ.get("CS_NO_WINO6") or os.environ.get("CS_NO_WINO5")) else 0.25,
                 "conv6_up_relu_bn": 4.0 / 9.0 if (os.environ.get("CS_NO_WINOGRAD") or os.environ.get("CS_NO_WINO6")) else 0.25,
                 "conv7_up_sigmoid_err": 4.0 / 9.0,
                 # conv6 (1/4 of 18,874,368 MACs) and conv7 (4/9 of 1,179,648: channel contraction T = a6 W_eff, then a gather) in one kernel
                 "conv6_conv7_fused_err": (0.25 * 18874368 + 4.0 / 9.0 * 1179648) / (18874368 + 1179648),
                 "conv2_relu_bn_pool": 1.0 if os.environ.get("CS_NO_WINOGRAD") else 4.0 / 9.0,
                 "conv3_relu_bn_pool": 1.0 if (os.environ.get("CS_NO_WINOGRAD") or os.environ.get("CS_NO_WINO3")) else 4.0 / 9.0}
BYTES_PER_CELL = 16_384 + 18       # algorithmic: read one crop, write 4 fp32/fp64-as-results + 2 int8


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--cells", type=int, default=1_000_000, help="crops per GPU per step")
    ap.add_argument("--chunk", type=int, default=65536, help="cells per internal pass (workspace ~0.4 MB per cell)")
    ap.add_argument("--train-cells", type=int, default=5000, help="synthetic crops the detector is fit on")
    ap.add_argument("--cpu-sample", type=int, default=0, help="cells for the CPU baseline (0 = auto, ~15 s)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--seed", type=int, default=42)
    return ap.parse_args()


def cpu_baseline(weights, det, seed, sample):
    """Times the CPU oracle on a bounded sample of the same synthetic workload."""
    import numpy as np
    from oracle import oracle
    # use the cores this process may actually run on (the GPU box gives a 1-GPU job a CPU share)
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    threads = max(1, min(oracle.num_threads(), avail))
    oracle.set_num_threads(threads)
    probe = oracle.synth_crops(seed, 0, 8 * threads)
    oracle.screen(weights, None, det, probe[:threads])           # warm up threads / page in
    t0 = time.perf_counter()
    oracle.screen(weights, None, det, probe)
    rate = len(probe) / (time.perf_counter() - t0)
    n = sample if sample > 0 else int(min(65536, max(64, rate * 15)))
    x = oracle.synth_crops(seed, 0, n)
    t0 = time.perf_counter()
    r = oracle.screen(weights, None, det, x)
    dt = time.perf_counter() - t0
    return dict(value=n / dt, unit="cells/s", cores=threads, kind="port", host_cpus_visible=avail,
                sample=f"{n} crops of the same synthetic workload (seed {seed}, cells 0..{n - 1}), "
                       f"oracle/cae_oracle.c fp32 + fp64 SVM, OpenMP {threads} threads, {dt:.1f} s"), r, x


def main():
    args = parse()
    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from cellscreen import dist as csdist
    from cellscreen import synth
    from cellscreen.detector_fit import fit_detector
    from cellscreen.engine import Engine

    # ---- model: random-init weights of the reference architecture, detector fit by sklearn
    weights = synth.random_cae(seed=args.seed)
    enc = Engine.from_weights(weights, device_id=local_rank)
    xt = torch.empty((args.train_cells, 64, 64), dtype=torch.float32, device=dev)
    enc.synth_crops(args.seed, 10_000_000_000, xt)               # disjoint from the screened cells
    feats = enc.encode(xt, which=0).cpu().numpy()
    enc.close()
    del xt
    det, _ = fit_detector(feats, pca_random_state=0)             # same on every rank (same inputs)
    eng = Engine.from_weights(weights, None, det, device_id=local_rank)
    eng.set_chunk(args.chunk)

    # ---- this rank's shard of the global synthetic batch, generated in HBM
    n_total = args.cells * world
    lo, hi = csdist.shard_range(n_total, rank, world)
    n_local = hi - lo
    x = torch.empty((n_local, 64, 64), dtype=torch.float32, device=dev)
    eng.synth_crops(args.seed, lo, x)
    out = dict(mse=torch.empty(n_local, dtype=torch.float32, device=dev), mae=torch.empty(n_local, dtype=torch.float32, device=dev),
               cons_score=torch.empty(n_local, dtype=torch.float64, device=dev), mod_score=torch.empty(n_local, dtype=torch.float64, device=dev),
               cons_pred=torch.empty(n_local, dtype=torch.int8, device=dev), mod_pred=torch.empty(n_local, dtype=torch.int8, device=dev))

    # the 18 B/cell of results end on rank 0's host: pinned destination buffers, one async copy per field
    host = {k: torch.empty(n_total, dtype=v.dtype, pin_memory=True) for k, v in out.items()} if rank == 0 else None

    def step():
        eng.screen(x, out=out, out_device=True)
        g = csdist.gather_results(out, n_total, dst=0) if world > 1 else out
        if rank != 0:
            return None
        for k, v in g.items():
            host[k].copy_(v, non_blocking=True)
        torch.cuda.current_stream().synchronize()                   # the step ends when the results are on the host
        return host

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    eng.profile_enable(True)
    eng.profile_reset()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    fence()
    elapsed = time.perf_counter() - t0
    eng.profile_enable(False)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    prof = eng.profile()
    if rank == 0:
        value = n_total * args.steps / elapsed
        # ---- roofline: dominant kernel by device time, algorithmic FLOPs / measured duration
        kern = {k: v for k, v in prof.items() if v["launches"] > 0}
        total_ms = sum(v["ms"] for v in kern.values())
        dom = max((k for k in kern if kern[k]["flops"] > 0), key=lambda k: kern[k]["ms"])
        d = kern[dom]
        avg_ms = d["ms"] / d["launches"]
        ach = d["flops"] / d["launches"] / (avg_ms * 1e-3) / 1e12
        # HBM traffic of that kernel: measured with rocprofv3 PMC passes (tools/pmc_traffic.py; a profiler
        # cannot run inside this process), bytes per cell x the cells this launch processed
        traffic, traffic_src = None, None
        try:
            pmc = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_pmc_traffic.json"))
            if pmc:
                tj = json.load(open(os.path.join(ROOT, "profiles", pmc[-1])))
                if dom in tj["kernels"]:
                    traffic = round(tj["kernels"][dom]["hbm_bytes_per_cell"] * (d["cells"] / d["launches"]))
                    traffic_src = "profiles/" + pmc[-1]
        except Exception:
            pass
        roofline = dict(bound="mfma", kernel=dom, achieved=round(ach, 3), peak=FP32_MFMA_PEAK_TFLOPS, unit="TFLOP/s",
                        frac=round(ach / FP32_MFMA_PEAK_TFLOPS, 4),
                        achieved_executed=round(ach * EXEC_FRACTION.get(dom, 1.0), 3),
                        frac_executed=round(ach * EXEC_FRACTION.get(dom, 1.0) / FP32_MFMA_PEAK_TFLOPS, 4),
                        note=("achieved counts the reference graph's algorithmic FLOPs of this layer; the kernel executes "
                              "%.3f of them (Winograd / folded upsample), which is why frac can exceed 1" % EXEC_FRACTION.get(dom, 1.0))
                        if EXEC_FRACTION.get(dom, 1.0) != 1.0 else None,
                        traffic=traffic, traffic_unit="HBM bytes per launch",
                        traffic_source=traffic_src,
                        algorithmic_bytes_per_launch=int((131072 + 65536) * (d["cells"] / d["launches"])) if dom.startswith("conv2") else None,
                        avg_launch_ms=round(avg_ms, 4), cells_per_launch=d["cells"] // d["launches"],
                        share_of_device_time=round(d["ms"] / total_ms, 4))
        kernels = {k: dict(ms=round(v["ms"], 3), launches=v["launches"], share=round(v["ms"] / total_ms, 4),
                           tflops=round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2) if v["flops"] > 0 else None,
                           tflops_executed=round(v["flops"] * EXEC_FRACTION.get(k, 1.0) / (v["ms"] * 1e-3) / 1e12, 2) if v["flops"] > 0 else None)
                   for k, v in kern.items()}
        enc_ms = sum(kern[k]["ms"] for k in ("conv1_relu_bn_pool", "conv2_relu_bn_pool", "conv3_relu_bn_pool") if k in kern)
        enc_cells = kern["conv2_relu_bn_pool"]["cells"]
        enc_hbm = {"algorithmic_bytes_per_cell": 16384 + 8192,
                   "algorithmic_gbs_at_encoder_rate": round(enc_cells * 24576 / (enc_ms * 1e-3) / 1e9, 1),
                   "frac_hbm_peak_algorithmic": round(enc_cells * 24576 / (enc_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                   "measured_bytes_per_cell": None, "frac_hbm_peak_measured": None, "encoder_ms_per_step": round(enc_ms / args.steps, 3)}
        try:
            if traffic_src:
                tk = json.load(open(os.path.join(ROOT, traffic_src)))["kernels"]
                mb = sum(tk[k]["hbm_bytes_per_cell"] for k in ("conv1_relu_bn_pool", "conv2_relu_bn_pool", "conv3_relu_bn_pool"))
                enc_hbm["measured_bytes_per_cell"] = round(mb)
                enc_hbm["frac_hbm_peak_measured"] = round(enc_cells * mb / (enc_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                enc_hbm["measured_source"] = traffic_src
        except Exception:
            pass
        exec_flop_per_cell = sum(kern[k]["flops"] * EXEC_FRACTION.get(k, 1.0) for k in kern if k.startswith("conv")) / max(
            1, kern["conv2_relu_bn_pool"]["cells"])
        line = {
            "metric": "cells/sec screened (CAE fwd + recon-MSE + SVM score), 64x64",
            "value": round(value, 1), "unit": "cells/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": ("BASELINE.json configs[2]: screening inference, %d synthetic 64x64 crops per GPU resident in HBM, "
                                    "CAE fwd + recon MSE/MAE + RobustScaler + PCA(100) + 2x OCSVM score on device" % args.cells)
                                   + ("" if world == 1 else "; configs[3] layout: contiguous shards, RCCL gather of per-cell scores"),
                       "cells_per_gpu": args.cells, "global_cells": n_total, "chunk_cells": args.chunk,
                       "n_sv": [int(det.conservative.n_sv), int(det.moderate.n_sv)], "detector_train_cells": args.train_cells,
                       "weights": "random init (Glorot, non-trivial BN), seed %d" % args.seed,
                       "parallelism": "dp%d" % world},
            "whole_path": {"tflops_algorithmic": round(value * FLOP_PER_CELL / 1e12 / world, 3),
                           "frac_fp32_mfma_peak": round(value * FLOP_PER_CELL / 1e12 / world / FP32_MFMA_PEAK_TFLOPS, 4),
                           "tflops_executed": round(value * exec_flop_per_cell / 1e12 / world, 3),
                           "frac_fp32_mfma_peak_executed": round(value * exec_flop_per_cell / 1e12 / world / FP32_MFMA_PEAK_TFLOPS, 4),
                           "note": "algorithmic = the reference graph's 100.27 MFLOP/cell; executed counts conv2/conv3 (Winograd F(2,3)) and conv7 (folded upsample) at 4/9, conv5/conv6 (folded + Winograd F(2,2)) at 1/4, conv1 at 12/9",
                           "hbm_gbs_algorithmic": round(value * BYTES_PER_CELL / 1e9 / world, 2),
                           "frac_hbm_peak": round(value * BYTES_PER_CELL / 1e9 / world / HBM_PEAK_GBS, 5),
                           "device_ms_per_step": round(total_ms / args.steps, 3),
                           # BASELINE.json asks for the fraction of the HBM roofline on the conv encoder (conv1-3): algorithmic
                           # bytes = one crop in + 8 KB of features out; measured = PMC traffic of the three kernels (p1 and p2 do
                           # go through HBM).  Exact-fp32 convs are MFMA-bound, so neither can approach 1.
                           "conv_encoder_hbm": enc_hbm},
            "roofline": roofline,
            "kernels": kernels,
        }
        if world == 1 and not args.no_cpu_baseline:
            cb, ref, xs = cpu_baseline(weights, det, args.seed, args.cpu_sample)
            line["cpu_baseline"] = cb
            # the bounded sample doubles as a live parity check of the benchmarked run
            n = len(xs)
            mse = res["mse"][:n].numpy()
            rel = float(np.max(np.abs(mse - ref["mse"]) / ref["mse"]))
            tol = 1e-4 * float(np.abs(det.conservative.dual_coef).sum())
            derr = float(np.max(np.abs(res["cons_score"][:n].numpy() - ref["cons_score"])))
            line["parity_on_cpu_sample"] = {"cells": n, "mse_max_rel": rel, "cons_score_max_abs": derr,
                                            "ok": bool(rel <= 1e-5 and derr <= tol)}
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
