"""Builds libcellscreen.so (HIP, gfx950) in-tree: one hipcc -c per translation unit, then a
shared link.  hipcc cross-compiles without a GPU, so this also runs in the CPU-only build
container; the resulting .so travels to the GPU box with the working tree.
Usage: python build.py [--force] [--keep-temps]"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libcellscreen.so")
BUILD = os.path.join(HERE, "build")
SOURCES = ["api.hip", "conv_mfma.hip", "conv_out.hip", "detector.hip", "train.hip", "train_api.hip", "train_generic.hip", "preprocess.hip",
           "conv_wino_cs.hip", "conv12_fused.hip", "conv45_h2.hip", "conv_generic.hip", "conv_generic_x3.hip", "conv_wino_up.hip", "fit.hip"]
HEADERS = ["common.hpp", "api_internal.hpp", "train_internal.hpp", "tensor_archive.hpp", os.path.join("..", "..", "include", "cellscreen.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function",
         "-fno-gpu-rdc"]


def source_hash():
    """sha256 over the kernel sources (csrc/*.hip, *.hpp and the C-ABI header): names a tree's kernels without git (the
    GPU box gets a snapshot without .git).  Measurement files under profiles/ carry it; bench.py refuses a PMC table
    whose hash is not this tree's."""
    import hashlib
    h = hashlib.sha256()
    for name in sorted(SOURCES) + sorted(HEADERS):
        with open(os.path.join(CSRC, name), "rb") as f:
            h.update(name.encode() + b"\0" + f.read() + b"\0")
    return h.hexdigest()


def _newer(target, deps):
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(d) <= t for d in deps)


def build(force=False, keep_temps=False, verbose=True):
    os.makedirs(BUILD, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    jobs = []
    objs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(BUILD, src.replace(".hip", ".o"))
        objs.append(o)
        if force or not _newer(o, [s] + hdrs):
            cmd = [HIPCC] + FLAGS + ["-c", s, "-o", o]
            if keep_temps:
                cmd += ["-save-temps=obj", "-Rpass-analysis=kernel-resource-usage"]
            jobs.append(cmd)

    def run(cmd):
        r = subprocess.run(cmd, capture_output=True, text=True, cwd=BUILD)
        return cmd, r

    if jobs:
        with ThreadPoolExecutor(max_workers=4) as ex:
            for cmd, r in ex.map(run, jobs):
                if verbose and (r.stdout.strip() or r.stderr.strip()):
                    sys.stderr.write(r.stdout + r.stderr)
                if r.returncode != 0:
                    raise RuntimeError("hipcc failed: " + " ".join(cmd) + "\n" + r.stderr)
    if force or jobs or not _newer(OUT, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed: " + r.stderr)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, keep_temps="--keep-temps" in sys.argv))
