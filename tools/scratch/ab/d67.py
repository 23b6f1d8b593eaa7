import os, sys, ctypes as C
os.environ["CS_WINO_DIAG"] = "1"
sys.path.insert(0, "cell-image-analysis_amd")
import torch
from cellscreen import synth, _lib
if os.environ.get("CS_X_LIB"): _lib.LIB_PATH = os.path.abspath(os.environ["CS_X_LIB"])
from cellscreen.engine import Engine
N = 65536
e = Engine.from_weights(synth.random_cae(42))
x = torch.empty((N, 64, 64), dtype=torch.float32, device="cuda"); e.synth_crops(42, 0, x); e.set_chunk(N)
f = e.reconstruct(x, want_recon=False)
e.profile_enable(True); e.profile_reset()
for _ in range(6): f = e.reconstruct(x, want_recon=False)
p = e.profile()["conv6_conv7_fused_err"]
lib = _lib.load_library()
out = (C.c_double * 5)(); rc = lib.cs_debug_wino_up_diag(3, out); v = list(out); tot = sum(v); items = N * 4 / 256
print("conv67: %.2f ms per 1M; cycles/group/wave %.0f:" % (p["ms"] / p["launches"] / N * 1e6, tot / items), " ".join("%s %.0f" % (n, a / items) for n, a in zip((["loads", "MFMA", "a6+bar", "T+strip+bar", "gather"] if True else []), v)))
