#!/usr/bin/env python3
"""Per-kernel SQ instruction counters from rocprofv3 PMC passes of bench.py (one directory per pass; counters that do
not fit one pass go into separate passes, --kernel-trace only):

    rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA \\
        --output-format csv -d gpurun_out/pmc_sq1 -- python3 bench.py --steps 1 --warmup 0 --cells 131072 --no-cpu-baseline --no-pmc --no-extra-legs
    python tools/pmc_sq.py gpurun_out/pmc_sq1 [more pass dirs ...] > profiles/rNN_sq_counters.json

Values are per full-chunk launch, summed over the device.  SQ_INSTS_VALU counts MFMAs too; the derived field
simd_cycles_mfma_plus_valu = 32 x MFMA + 4 x (VALU - MFMA) is the issue time of the launch if MFMA and VALU
instructions never overlap (SQ_VALU_MFMA_COEXEC_CYCLES says they do not), and dividing it by the 1,024 SIMDs and the
launch time gives the clock the chip would need to be fully busy.
--no-pmc keeps bench.py from starting its own rocprofv3 child passes inside the profiled process."""
import collections
import csv
import glob
import os
import json
import sys

from pmc_traffic import KERNELS

N_SIMD = 256 * 4


def main():
    out = collections.defaultdict(dict)
    x3 = set()      # families whose main contraction ran as v_mfma_f32_16x16x32_{bf16,f16} (16 cycles each, VALU may issue beside them)
    all16 = set()
    for d in sys.argv[1:]:
        # the newest pass if the directory was reused
        rows = list(csv.DictReader(open(max(glob.glob(d + "/*/*_counter_collection.csv"), key=os.path.getmtime))))
        by = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in rows:
            by[r["Kernel_Name"]][r["Counter_Name"]].append((float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
        for pat, name in KERNELS:
            for k, counters in by.items():
                if pat not in k:
                    continue
                if "x3_kernel" in pat or "_h2_kernel" in pat or ("conv12_fused_kernel" in pat and ", true" in k):
                    x3.add(name)      # 16-bit MFMAs (bf16 or fp16 split); conv12_fused_kernel<DIAG, C1X3 = true, ...>: conv1 on them
                if "conv12_fused_kernel" in pat and "true, true, true>" in k:
                    all16.add(name)   # <DIAG, C1X3, C2H, C1H>: conv2 as an fp16 split too -- no fp32 MFMA left in the kernel
                for c, v in counters.items():
                    longest = max(x[1] for x in v)
                    full = [x for x in v if x[1] > 0.7 * longest]
                    out[name][c] = round(sum(x[0] for x in full) / len(full))
                    out[name]["launch_ms_under_pmc"] = round(sum(x[1] for x in full) / len(full) / 1e6, 3)
    for name, c in out.items():
        if "SQ_INSTS_MFMA" in c and "SQ_INSTS_VALU" in c:
            if name in x3:
                # bf16 MFMAs: 16 cycles each; the two fused kernels also issue fp32 MFMAs (32 cycles): conv7's contraction / conv2
                per_cell_f32 = 0 if name in all16 else {"conv6_conv7_fused_err": 512, "conv1_conv2_fused": 4608}.get(name, 0)
                f32 = per_cell_f32 * 65536
                c["mfma_kind"] = "v_mfma_f32_16x16x32_{bf16,f16}" + (" + %d v_mfma_f32_16x16x4_f32 per cell" % per_cell_f32 if f32 else "")
                c["simd_cycles_mfma"] = 16 * (c["SQ_INSTS_MFMA"] - f32) + 32 * f32
                c["mfma_util_at_2p1_ghz"] = round(c["simd_cycles_mfma"] / N_SIMD / (c["launch_ms_under_pmc"] * 1e-3 * 2.1e9), 3)
                continue
            cyc = 32 * c["SQ_INSTS_MFMA"] + 4 * (c["SQ_INSTS_VALU"] - c["SQ_INSTS_MFMA"])
            c["simd_cycles_mfma_plus_valu"] = cyc
            c["implied_clock_ghz_if_fully_busy"] = round(cyc / N_SIMD / (c["launch_ms_under_pmc"] * 1e-3) / 1e9, 3)
            # matrix-pipe share of the issue cycles, and matrix-pipe utilisation of the launch at the ~2.1 GHz the chip sustains
            c["mfma_share_of_issue_cycles"] = round(32 * c["SQ_INSTS_MFMA"] / cyc, 3)
            c["mfma_util_at_2p1_ghz"] = round(32 * c["SQ_INSTS_MFMA"] / N_SIMD / (c["launch_ms_under_pmc"] * 1e-3 * 2.1e9), 3)
    from build import source_hash
    print(json.dumps({"note": "see tools/pmc_sq.py", "source_hash": source_hash(), "cells_per_launch": 65536, "kernels": out}, indent=1))


if __name__ == "__main__":
    main()
