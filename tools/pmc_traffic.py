#!/usr/bin/env python3
"""Turns two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE -- they do not fit one pass on gfx950:
MI355X_MICROARCH.md "rocprofv3 PMC slots") into per-kernel HBM traffic per cell.

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --steps 1 --warmup 0 --cells 65536 --chunk 16384 --no-cpu-baseline --no-pmc --no-extra-legs
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 bench.py --steps 1 --warmup 0 --cells 65536 --chunk 16384 --no-cpu-baseline --no-pmc --no-extra-legs
    python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write 16384 > profiles/rNN_pmc_traffic.json

(--no-pmc --no-extra-legs: bench.py would otherwise start its own rocprofv3 child passes inside the profiled process and run
its training / raw-crop legs under the profiler.)

Corrections, as the guide prescribes: FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
reports exactly half the bytes of a wide coalesced streaming read, so it is doubled; WRITE_SIZE
is exact for 16-B-per-lane stores and uncalibrated for our 4-B-per-lane epilogue stores (it
comes out within 2 % of the algorithmic byte count, so it is taken as is).  Only full-chunk
launches are used (the detector-fit encode pass and the tail chunk are filtered by duration)."""
import collections
import csv
import glob
import os
import json
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cell-image-analysis_amd"))

KERNELS = [("conv12_fused_kernel", "conv1_conv2_fused"), ("ConvCfg<64, 64, 1, 32, 1", "conv1_relu_bn_pool"), ("ConvCfg<32, 32, 32, 64, 1", "conv2_relu_bn_pool"),
           ("conv2_wino_ring_kernel", "conv2_relu_bn_pool"), ("WinoCfg<32, 32, 32, 64>", "conv2_relu_bn_pool"), ("WinoCfg<16, 16, 64, 32>", "conv3_relu_bn_pool"),
           ("ConvCfg<16, 16, 64, 32, 1", "conv3_relu_bn_pool"), ("ConvCfg<8, 8, 32, 32, 0", "conv4_relu_bn"),
           ("ConvCfg<16, 16, 32, 64, 0", "conv5_up_relu_bn"), ("ConvCfg<32, 32, 64, 32, 0", "conv6_up_relu_bn"),
           ("WUCfg<8, 8, 32, 64>", "conv5_up_relu_bn"), ("WUCfg<16, 16, 64, 32>", "conv6_up_relu_bn"), ("conv67_fused_kernel", "conv6_conv7_fused_err"), ("conv67_x3_kernel", "conv6_conv7_fused_err"),
           ("conv67_h2_kernel", "conv6_conv7_fused_err"), ("conv45_h2_kernel", "conv5_up_relu_bn"), ("conv3_wino_h2_kernel", "conv3_relu_bn_pool"), ("conv4_h2_kernel", "conv4_relu_bn"), ("conv5_h2_kernel", "conv5_up_relu_bn"),
           ("conv3_wino_x3_kernel", "conv3_relu_bn_pool"), ("conv4_bf16x3_kernel", "conv4_relu_bn"), ("conv5_bf16x3_kernel", "conv5_up_relu_bn"), ("conv7_err_kernel", "conv7_up_sigmoid_err"), ("scaler_pca_kernel", "scaler_pca"), ("scaler_pca_x3_kernel", "scaler_pca"), ("ocsvm_mfma_kernel", "ocsvm_decision"),
           ("preprocess_kernel", "preprocess")]


def per_kernel(d, counter, chunk):
    # the newest pass if the directory was reused
    rows = list(csv.DictReader(open(max(glob.glob(d + "/*/*_counter_collection.csv"), key=os.path.getmtime))))
    by = collections.defaultdict(list)
    for r in rows:
        if r["Counter_Name"] == counter:
            by[r["Kernel_Name"]].append((float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    out = {}
    for pat, name in KERNELS:
        for k, v in by.items():
            if pat in k:
                longest = max(x[1] for x in v)
                full = [x for x in v if x[1] > 0.7 * longest]          # full-chunk launches only
                out[name] = dict(kib_per_launch=sum(x[0] for x in full) / len(full), launches=len(full),
                                 avg_ms=sum(x[1] for x in full) / len(full) / 1e6)
    return out


def reduce_passes(fetch_dir, write_dir, chunk):
    """The two PMC passes -> {"kernels": {family: bytes per cell}}; stamped with the hash of the kernel sources so that
    bench.py can tell a table of this tree from a stale one."""
    from build import source_hash
    f = per_kernel(fetch_dir, "FETCH_SIZE", chunk)
    w = per_kernel(write_dir, "WRITE_SIZE", chunk)
    res = {"cells_per_launch": chunk, "source_hash": source_hash(),
           "note": "HBM bytes per cell = (2 x FETCH_SIZE + WRITE_SIZE) KiB x 1024 / cells_per_launch", "kernels": {}}
    for _, name in KERNELS:
        if name in f and name in w:
            rd = 2.0 * f[name]["kib_per_launch"] * 1024 / chunk
            wr = w[name]["kib_per_launch"] * 1024 / chunk
            res["kernels"][name] = dict(read_bytes_per_cell=round(rd, 1), write_bytes_per_cell=round(wr, 1),
                                        hbm_bytes_per_cell=round(rd + wr, 1), avg_launch_ms_under_pmc=round(f[name]["avg_ms"], 4))
    return res


def main():
    print(json.dumps(reduce_passes(sys.argv[1], sys.argv[2], int(sys.argv[3])), indent=1))


if __name__ == "__main__":
    main()
