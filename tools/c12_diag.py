"""Phase shares of the fused conv1 + conv2 kernel (CS_C12_DIAG=1: s_memtime stamps summed per wave).
Diagnostic only; run on an MI355X from the repo root:  python tools/c12_diag.py [split16|fp32_exact]"""
import os, sys, ctypes as C
os.environ["CS_C12_DIAG"] = "1"
sys.path.insert(0, "cell-image-analysis_amd")
import torch
from cellscreen import synth, _lib
if os.environ.get("CS_X_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["CS_X_LIB"])      # development: a variant build
from cellscreen.engine import Engine
N = 65536
PREC = sys.argv[1] if len(sys.argv) > 1 else "split16"
e = Engine.from_weights(synth.random_cae(42), precision=PREC)
x = torch.empty((N, 64, 64), dtype=torch.float32, device="cuda")
e.synth_crops(42, 0, x)
e.set_chunk(N)
f = e.layer_output(x, 1)
e.profile_enable(True)
e.profile_reset()
for _ in range(8):
    f = e.layer_output(x, 1)
prof = e.profile()
lib = _lib.load_library()
out = (C.c_double * 8)()
rc = lib.cs_debug_conv12_diag(out)
v = list(out)
tot = sum(v)
groups = N * 4 / 256
ms = prof["conv1_conv2_fused"]["ms"] / prof["conv1_conv2_fused"]["launches"]
print(f"conv1+conv2 fused ({PREC}): rc {rc}; groups per WG {groups:.0f}; cycles per group per wave {tot / groups:.0f} (16-bit MFMA issue alone: 174 x 16 = 2784 SIMD-cycles per group); "
      f"launch {ms:.3f} ms -> s_memtime rate {tot / (ms * 1e-3) / 1e9:.3f} GHz")
for n_, a in zip(["P1 conv1", "barrier 1", "P2 transform", "barrier 2", "P3 MFMA+row fold", "barrier 3", "P4 fold+store", "barrier 4"], v):
    print(f"    {n_:24s} {a / groups:8.0f} cycles/group  {100 * a / tot:5.1f} %")
