"""Runs the diagnostic build of the column-split Winograd kernels (CS_WINO_DIAG=1: s_memtime stamps summed
per wave) and prints the phase shares of the last conv2 / conv3 launch.  Diagnostic only; run on an MI355X
from the repo root."""
import os, sys, ctypes as C
os.environ["CS_WINO_DIAG"] = "1"
sys.path.insert(0, "cell-image-analysis_amd")
import numpy as np, torch
from cellscreen import synth, _lib
from cellscreen.engine import Engine
N = 65536
e = Engine.from_weights(synth.random_cae(42))
x = torch.empty((N, 64, 64), dtype=torch.float32, device="cuda")
e.synth_crops(42, 0, x)
e.set_chunk(N)
f = e.encode(x, which=0)          # warm-up
e.profile_enable(True)
e.profile_reset()
for _ in range(8):                # sustained load, so the clock is the one the bench sees
    f = e.encode(x, which=0)
prof = e.profile()
lib = _lib.load_library()
names = ["loads+T+MFMA+fold", "strip writes", "barrier A", "finalize", "barrier B", "(load issue only)"]
for layer, groups in ((1, 16), (2, 4)):
    out = (C.c_double * 6)()
    rc = lib.cs_debug_wino_cs_diag(layer, out)
    v = list(out)
    tot = sum(v[:5])
    items = N * groups / 512
    key = [k for k in prof if k.startswith(f"conv{layer + 1}_")][0]
    if rc != 0 or not prof[key]["launches"]:
        print(f"conv{layer + 1}: its stand-alone Winograd kernel did not run (rc {rc}; conv1 + conv2 run fused unless the handle was created with CS_DEBUG_NO_FUSE12)")
        continue
    ms = prof[key]["ms"] / prof[key]["launches"]
    print(f"conv{layer + 1}: rc {rc}; groups per WG {items:.0f}; cycles per group per wave {tot / items:.0f} (MFMA issue alone 4096); "
          f"launch {ms:.3f} ms -> s_memtime rate {tot / (ms * 1e-3) / 1e9:.3f} GHz")
    for n_, a in zip(names, v):
        print(f"    {n_:18s} {a / items:8.0f} cycles/group  {100 * a / tot:5.1f} %")

# the F(2x2,2x2) phase kernels of conv5 / conv6 (one 512-thread workgroup per CU)
f = e.reconstruct(x, want_recon=True)     # with the reconstruction asked for, conv6 and conv7 run as separate kernels
del f
prof = e.profile()
names5 = ["load issue", "transform+MFMA", "epilogue+stores", "strip writes", "barrier"]
for layer, groups in ((5, 4), (4, 1)):
    out = (C.c_double * 5)()
    rc = lib.cs_debug_wino_up_diag(layer, out)
    v = list(out)
    tot = sum(v)
    items = N * groups / 256
    if rc != 0 or tot == 0:
        print(f"conv{layer + 1}: its Winograd phase kernel did not run (rc {rc}; conv5 runs on the fp16-split kernel unless precision is fp32_exact)")
        continue
    print(f"conv{layer + 1}: rc {rc}; groups per WG {items:.0f}; cycles per group per wave {tot / items:.0f} (MFMA issue alone 4608)")
    for n_, a in zip(names5, v):
        print(f"    {n_:18s} {a / items:8.0f} cycles/group  {100 * a / tot:5.1f} %")

# the fused conv6 + conv7 + error kernel (slot 3), one workgroup per CU, 4 groups per cell
f = e.reconstruct(x, want_recon=False)
names67 = ["load issue", "(transform+)MFMA", "(out transform+)a6->LDS+barrier", "T+strip writes+barrier", "gather+sigmoid+err"]
out = (C.c_double * 5)()
rc = lib.cs_debug_wino_up_diag(3, out)
v = list(out)
tot = sum(v)
items = N * 4 / 256
print(f"conv6+7 fused: rc {rc}; groups per WG {items:.0f}; cycles per group per wave {tot / items:.0f} (MFMA issue per wave pair: Winograd fp32 form 4608 + 512, split-bf16 form 6144 + 512)")
for n_, a in zip(names67, v):
    print(f"    {n_:30s} {a / items:8.0f} cycles/group  {100 * a / tot:5.1f} %")
