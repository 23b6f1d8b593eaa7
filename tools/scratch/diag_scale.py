import os, sys, subprocess, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "cell-image-analysis_amd")); sys.path.insert(0, ROOT)
if len(sys.argv) > 1:
    from cellscreen import synth
    from cellscreen.engine import Engine
    from oracle import oracle
    w = synth.random_cae(seed=42)
    for scale in (1.0, 255.0):
        x = oracle.synth_crops(7, 100, 12) * np.float32(scale)
        x[3] *= np.float32(1e-9); x[5] = 0.0; x[7, :32] = 0.0
        ref = oracle.cae_forward(w, x, acc64=True, want=("features",), layers=True)["layers"]
        e = Engine.from_weights(w)
        for layer in (2, 3, 4):
            got = e.layer_output(x, layer)
            errs = [float(np.abs(got[c] - ref[layer][c]).max() / max(np.abs(ref[layer][c]).max(), 1e-30)) for c in range(12)]
            c = int(np.argmax(errs))
            i = np.unravel_index(np.argmax(np.abs(got[c] - ref[layer][c])), got[c].shape)
            print(sys.argv[1], "scale", scale, "layer", layer, "worst cell", c, "err", "%.3e" % errs[c], "at", i, "got", got[c][i], "ref", ref[layer][c][i],
                  "finite", bool(np.isfinite(got).all()), "max|ref|", float(np.abs(ref[layer][c]).max()), flush=True)
        e.close()
else:
    for name, env in (("fp16x2", {}), ("fp16x2-noconv2", {"CS_NO_FP16X2_CONV2": "1"}), ("bf16x3", {"CS_NO_FP16X2": "1"}), ("fp32", {"CS_NO_BF16X3": "1"})):
        subprocess.run([sys.executable, __file__, name], env=dict(os.environ, **env))
