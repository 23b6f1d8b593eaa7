import sys
import numpy as np
a, b = np.load(sys.argv[1]), np.load(sys.argv[2])
bad = 0
for k in a.files:
    same = np.array_equal(a[k], b[k], equal_nan=True)
    if not same:
        d = np.abs(a[k].astype(np.float64) - b[k].astype(np.float64))
        print(k, "DIFFERS: max abs", d.max(), "rel to range", d.max() / max(1e-30, np.abs(a[k]).max()), "count", int((a[k] != b[k]).sum()), "of", a[k].size)
        bad += 1
    else:
        print(k, "bit-identical")
sys.exit(1 if bad else 0)
