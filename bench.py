#!/usr/bin/env python3
"""bench.py -- cells/sec screened on MI355X (BASELINE.json metric), one process per GPU.

    python bench.py --gpus N --steps K --warmup W          (N = 1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path -- compute_anomaly_scores, improved_detection.py:117-153:
CAE forward + per-cell MSE/MAE + encoder features + RobustScaler + PCA + 2x one-class-SVM
score -- over this rank's batch of synthetic 64x64 crops that are already resident in HBM,
ending with the per-cell results on the host of rank 0 (for N > 1: after the RCCL gather).
N = 1 runs BASELINE.json configs[2] (1M crops on one GPU); N > 1 runs configs[3]'s layout at its per-GPU size: 1.25 M crops
per GPU (10 M global at N = 8), contiguous index ranges, no data-path collective before the final RCCL gather (weak scaling).

Rank 0 prints ONE JSON line; besides the contract keys it carries
  roofline      the dominant kernel priced against the matrix pipes it runs on.  `achieved` counts the matrix-pipe work
                the kernel EXECUTES (its fp32 MFMAs x 2,048 FLOP + its bf16 MFMAs x 16,384 FLOP -- the library reports
                both counts, their sum is what SQ_INSTS_MFMA shows) over its launch time from HIP events recorded on
                the library's stream inside the timed steps; `peak` is the rate that same instruction mix reaches with
                both pipes at their peaks (157.3 TFLOP/s for a pure fp32-MFMA kernel, 2,500 for a pure bf16 one), so
                `frac` = achieved / peak is the fraction of the kernel's time its matrix work needs at peak -- never
                above 1; the reference graph's algorithmic FLOPs of the same layers are `achieved_algorithmic`
                (Winograd and the folded upsample execute fewer multiply-adds).
                `traffic` = HBM bytes per launch from two rocprofv3 PMC passes (FETCH_SIZE x 2 on
                gfx950, WRITE_SIZE) of THIS tree: collected by this run in two child processes before
                the parent touches the GPU, or taken from profiles/*_pmc_traffic.json only when that
                file carries the hash of the kernel sources of this tree (else null).
  cpu_baseline  the reference's CPU sequence (improved_detection.py:122-142) on this host's cores:
                Keras/TF if importable, else its counterpart -- the same graph in torch-CPU at Keras's
                predict batch of 32, autoencoder pass + separate encoder pass, NumPy MSE/MAE, the real
                scikit-learn transform / predict / decision_function calls (4 SVM passes) -- on
                configs[0]'s 128 crops and on 4,096, each at the best of a sweep over the host's thread counts (the sweep is in
                the line); plus the OpenMP oracle port (rank 0, N = 1 only)
  exact_fp32    the same step -- same --steps / --warmup, the same 18 B/cell copy to the host and synchronisation -- on a second engine
                created with precision="fp32_exact" (cs_model_options: every contraction on the fp32 matrix instructions), with its
                own parity_on_cpu_sample: the headline runs the fp32 contractions as split 16-bit products (dtype says so), this is
                the reference's fp32 arithmetic next to it
  large_variant BASELINE.json configs[4]'s single-GPU half: 8,192 crops 128x128 through the 32-64-128 | 128-64-32-1 instance of the
                layer grammar (forward + MSE/MAE: cells/s, per-kernel fractions of the 16-bit matrix peak) and 200 batch-32 training
                steps of the run-time-shaped trainer (ms/step, algorithmic TFLOP/s)
  e2e_raw       the production path end to end: 1 M RAW uint16 bounding-box crops (sides U[32,100]) in pinned host memory ->
                H2D -> cs_preprocess (CLAHE + anti-aliased resize, improved_detection.py:98-99) -> cs_screen -> 18 B/cell on the
                host, the copy of chunk i + 1 under the kernels of chunk i
  train_leg     BASELINE.json configs[1] as written: 50,000 synthetic crops -> 40,000 / 10,000, one epoch = 1,250 augmented
                batch-32 steps + the validation pass
  small_n       single-call latency of the hot path from host buffers at N = 128 / 1,024 / 10,240
"""
import argparse
import json
import os
import subprocess
import sys
import time

# HIP deals its streams round-robin onto GPU_MAX_HW_QUEUES hardware queues (default 4), and two streams that share one run
# their work in submission order.  This process has more than four (torch's, each handle's compute and copy streams, the
# e2e leg's upload stream): measured on the e2e_raw leg, the upload of chunk i + 1 and the preprocess kernel of chunk i
# landed on one queue and ran one after the other (2.1 M cells/s at 4 queues; 3.0 M at 5, 6, 7 or 8).  More is not better:
# from 7 queues on, the training step that runs later in the same process -- ~75 small kernels on two streams with 14
# cross-stream events -- takes 1.05 ms instead of 0.67 (tools/scratch/train_probe3.py: the command processor then
# time-slices its queues and every cross-queue dependency waits for a slice).  Must be set before HIP initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "6")

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, "cell-image-analysis_amd"), os.path.join(ROOT, "tools"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

FP32_MFMA_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md, "Peak FP32 (matrix)"
HBM_PEAK_GBS = 8000.0              # spec
FLOP_PER_CELL = 100_270_080        # ALGORITHMIC: 2 x 50,135,040 conv MACs of the reference graph, SURVEY.md section 8d
FLOP_PER_MFMA = 2048               # v_mfma_f32_16x16x4_f32: 16 x 16 x 4 multiply-adds
FLOP_PER_BF16_MFMA = 16384         # v_mfma_f32_16x16x32_bf16 (kernels that carry the fp32 contraction as six bf16 products)
BF16_MFMA_PEAK_TFLOPS = 2500.0     # dense bf16 matrix peak (MI355X_MICROARCH.md)
BYTES_PER_CELL = 16_384 + 18       # algorithmic: read one crop, write 4 fp32/fp64-as-results + 2 int8
# algorithmic HBM bytes per cell of each kernel family (what it must read + write if nothing is re-read)
ALG_BYTES = {"conv1_relu_bn_pool": 16384 + 131072, "conv2_relu_bn_pool": 131072 + 65536, "conv3_relu_bn_pool": 65536 + 8192,
             "conv4_relu_bn": 8192 + 8192, "conv5_up_relu_bn": 8192 + 65536, "conv6_up_relu_bn": 65536 + 131072,
             "conv7_up_sigmoid_err": 131072 + 16384, "conv6_conv7_fused_err": 65536 + 16384 + 64,
             "conv1_conv2_fused": 16384 + 65536, "scaler_pca": 8192 + 400}
ENCODER_KERNELS = ("conv1_relu_bn_pool", "conv2_relu_bn_pool", "conv1_conv2_fused", "conv3_relu_bn_pool")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--cells", type=int, default=0, help="crops per GPU per step (0 = BASELINE.json: 1,000,000 at N = 1 [configs[2]], "
                                                         "1,250,000 at N > 1 [configs[3]: 10 M over 8 GPUs])")
    ap.add_argument("--chunk", type=int, default=65536, help="cells per internal pass (workspace ~0.3 MB per cell)")
    ap.add_argument("--train-cells", type=int, default=5000, help="synthetic crops the detector is fit on")
    ap.add_argument("--cpu-sample", type=int, default=0, help="cells for the CPU port baseline (0 = auto, ~10 s)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="do not collect HBM traffic with rocprofv3 child passes")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the exact-fp32, raw-crop, training, 128x128-variant and small-N legs")
    ap.add_argument("--train-steps", type=int, default=0, help="timed training steps (0 = one configs[1] epoch: 1,250)")
    ap.add_argument("--raw-crops", type=int, default=1_000_000, help="raw crops of the e2e_raw leg")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--seed", type=int, default=42)
    return ap.parse_args()


# ------------------------------------------------------------------------------------------- HBM traffic (PMC)
def collect_pmc_traffic(args):
    """Two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950) of a 2-launch run of this
    script, as child processes started BEFORE this process initialises the GPU.  Returns the tools/pmc_traffic.py
    table or None (rocprofv3 absent / failed: the caller falls back to a hash-matched file in profiles/)."""
    import shutil
    import tempfile
    from pmc_traffic import reduce_passes
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    base = tempfile.mkdtemp(prefix="cs_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    chunk = 65536
    child = [sys.executable, os.path.abspath(__file__), "--pmc-child", "--steps", "1", "--warmup", "0", "--cells", str(2 * chunk),
             "--chunk", str(chunk), "--train-cells", "1000", "--seed", str(args.seed)]
    try:
        for name in ("FETCH_SIZE", "WRITE_SIZE"):
            cmd = [exe, "--kernel-trace", "--pmc", name, "--output-format", "csv", "-d", os.path.join(base, name), "--"] + child
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=420)
            if r.returncode != 0:
                return None, "rocprofv3 --pmc %s failed (rc %d): %s" % (name, r.returncode, r.stderr.decode(errors="replace")[-300:])
        return reduce_passes(os.path.join(base, "FETCH_SIZE"), os.path.join(base, "WRITE_SIZE"), chunk), "this run (rocprofv3 child passes)"
    except Exception as e:  # noqa: BLE001 - any profiler trouble only costs the traffic figure
        return None, "PMC passes failed: %r" % (e,)
    finally:
        shutil.rmtree(base, ignore_errors=True)


def committed_pmc_traffic():
    """profiles/*_pmc_traffic.json of THIS tree only: the file must carry the hash of the kernel sources."""
    from build import source_hash
    want = source_hash()
    pdir = os.path.join(ROOT, "profiles")
    for f in sorted(os.listdir(pdir), reverse=True) if os.path.isdir(pdir) else []:
        if f.endswith("_pmc_traffic.json"):
            try:
                tj = json.load(open(os.path.join(pdir, f)))
            except Exception:  # noqa: BLE001
                continue
            if tj.get("source_hash") == want:
                return tj, "profiles/" + f + " (source_hash matches this tree)"
    return None, "no profiles/*_pmc_traffic.json carries this tree's source_hash " + want[:12]


# ------------------------------------------------------------------------------------------- CPU baselines
def cpu_port(weights, det, seed, sample):
    """The OpenMP oracle (a port of the same path: shared encoder, one SVM pass) on a bounded sample, at the best thread count of a
    short sweep inside what the host gives this job (affinity mask cut by the cgroup quota: a 1-GPU job gets a share of the box)."""
    import math
    from oracle import oracle
    avail, quota = _host_cpu_budget()
    budget = max(1, min(avail, int(math.ceil(quota)) if quota else avail))
    cands = sorted({max(1, budget // 2), budget, min(avail, 2 * budget), min(avail, oracle.num_threads())})
    probe = oracle.synth_crops(seed, 0, 256)
    oracle.screen(weights, None, det, probe[:32])                # page in
    sweep = {}
    for th in cands:
        oracle.set_num_threads(th)
        best = 0.0
        for _ in range(2):
            t0 = time.perf_counter()
            oracle.screen(weights, None, det, probe)
            best = max(best, len(probe) / (time.perf_counter() - t0))
        sweep[str(th)] = round(best, 1)
    threads = int(max(sweep, key=lambda k: sweep[k]))
    oracle.set_num_threads(threads)
    n = sample if sample > 0 else int(min(65536, max(64, sweep[str(threads)] * 10)))
    x = oracle.synth_crops(seed, 0, n)
    t0 = time.perf_counter()
    r = oracle.screen(weights, None, det, x)
    dt = time.perf_counter() - t0
    return dict(value=round(n / dt, 1), unit="cells/s", cores=threads, kind="port", host_cpus_visible=avail, cgroup_cpu_quota=quota,
                cells_per_s_by_threads=sweep,
                sample=f"{n} crops of the same synthetic workload (seed {seed}, cells 0..{n - 1}), "
                       f"oracle/cae_oracle.c fp32 + fp64 SVM, OpenMP {threads} threads (best of the sweep), {dt:.1f} s"), r, x


def _torch_graph(weights):
    """The reference graph (CAE_improved_modeltrain.py:191-216) as torch-CPU functional ops, inference mode."""
    import torch
    import torch.nn.functional as F
    ks = [torch.from_numpy(k.transpose(3, 2, 0, 1).copy()) for k in weights.kernels]        # HWIO -> OIHW
    bs = [torch.from_numpy(b.copy()) for b in weights.biases]
    bn = [tuple(torch.from_numpy(a.copy()) for a in (weights.bn_mean[l], weights.bn_var[l], weights.bn_gamma[l], weights.bn_beta[l]))
          for l in range(len(weights.bn_gamma))]
    n_enc, n_conv, eps = weights.n_enc, weights.n_conv, float(weights.bn_eps)

    def block(h, l):
        h = F.relu(F.conv2d(h, ks[l], bs[l], padding=1))
        m, v, g, b = bn[l]
        return F.batch_norm(h, m, v, g, b, training=False, eps=eps)

    def encoder(x):
        h = x
        for l in range(n_enc):
            h = F.max_pool2d(block(h, l), 2)
        return h

    def autoencoder(x):
        h = block(encoder(x), n_enc)
        for l in range(n_enc + 1, n_conv - 1):
            h = block(F.interpolate(h, scale_factor=2, mode="nearest"), l)
        h = F.interpolate(h, scale_factor=2, mode="nearest")
        return torch.sigmoid(F.conv2d(h, ks[-1], bs[-1], padding=1))

    def predict(fn, X):          # Keras Model.predict: batches of 32, outputs concatenated
        with torch.no_grad():
            return torch.cat([fn(X[i:i + 32]) for i in range(0, len(X), 32)]).numpy()
    return autoencoder, encoder, predict


def _keras_models(weights):
    """Only if TensorFlow is importable (not expected in this image): the identical Keras graph with these weights."""
    import numpy as np
    import tensorflow as tf  # noqa: F401
    from tensorflow.keras import layers, models
    inp = layers.Input(shape=(64, 64, 1))
    h = inp
    convs, bns = [], []
    ch = weights.channels
    for l in range(weights.n_enc):
        c = layers.Conv2D(ch[l], (3, 3), activation="relu", padding="same"); h = c(h); convs.append(c)
        b = layers.BatchNormalization(); h = b(h); bns.append(b)
        h = layers.MaxPooling2D((2, 2), padding="same")(h)
    enc_out = h
    c = layers.Conv2D(ch[weights.n_enc], (3, 3), activation="relu", padding="same"); h = c(h); convs.append(c)
    b = layers.BatchNormalization(); h = b(h); bns.append(b)
    for l in range(weights.n_enc + 1, weights.n_conv - 1):
        h = layers.UpSampling2D((2, 2))(h)
        c = layers.Conv2D(ch[l], (3, 3), activation="relu", padding="same"); h = c(h); convs.append(c)
        b = layers.BatchNormalization(); h = b(h); bns.append(b)
    h = layers.UpSampling2D((2, 2))(h)
    c = layers.Conv2D(1, (3, 3), activation="sigmoid", padding="same"); h = c(h); convs.append(c)
    ae, en = models.Model(inp, h), models.Model(inp, enc_out)
    for l, c in enumerate(convs):
        c.set_weights([weights.kernels[l], weights.biases[l]])
    for l, b in enumerate(bns):
        b.set_weights([weights.bn_gamma[l], weights.bn_beta[l], weights.bn_mean[l], weights.bn_var[l]])
    return (lambda X: ae.predict(X, verbose=0)), (lambda X: en.predict(X, verbose=0)), np


def _host_cpu_budget():
    """CPUs this process may actually use: the affinity mask, cut by the cgroup's CPU quota when there is one (a GPU box gives a
    1-GPU job a share of a 256-CPU host: 128 torch threads on a 16-CPU quota is oversubscription, not a baseline)."""
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        pass
    return avail, quota


def cpu_reference_sequence(weights, sk, seed, sizes=(128, 4096), threads=(4, 8, 16, 32, 64, 128), sweep_crops=512):
    """improved_detection.py:122-142 literally, on the host: X -> autoencoder.predict -> MSE/MAE -> encoder.predict ->
    flatten -> scaler.transform -> pca.transform -> 2x predict + 2x decision_function.  sk = the fitted scikit-learn
    objects.  Keras/TF when importable ("reference"), else torch-CPU for the two predict calls ("counterpart").
    The intra-op thread count is SWEPT (a 512-crop run per candidate, seconds each), each size is then timed at the best
    count, and the sweep is reported beside the value: at Keras's predict batch of 32 more threads than the host gives this job
    only adds contention."""
    import numpy as np
    import torch
    from cellscreen import synth
    kind, label = "counterpart", "torch-CPU + sklearn counterpart of the reference CPU path"
    try:
        ae_predict, en_predict, _ = _keras_models(weights)
        kind, label = "reference", "Keras/TF-CPU, reference sequence"
        to_in = lambda X: X                                                              # noqa: E731
    except Exception:  # noqa: BLE001 - TensorFlow absent (the expected case) or unusable
        ae, en, predict = _torch_graph(weights)
        ae_predict = lambda X: predict(ae, X).transpose(0, 2, 3, 1)                      # noqa: E731
        en_predict = lambda X: predict(en, X).transpose(0, 2, 3, 1)                      # noqa: E731
        to_in = lambda X: torch.from_numpy(np.ascontiguousarray(X.transpose(0, 3, 1, 2)))  # noqa: E731
    scaler, pca, dets = sk["scaler"], sk["pca"], sk["detectors"]

    def run(cells):
        t0 = time.perf_counter()
        X = np.expand_dims(np.array(cells), axis=-1).astype("float32")                  # :122
        Xin = to_in(X)
        rec = ae_predict(Xin)                                                          # :125
        mse = np.mean(np.square(X - rec), axis=(1, 2, 3))                               # :126
        mae = np.mean(np.abs(X - rec), axis=(1, 2, 3))                                  # :127
        t1 = time.perf_counter()
        enc = en_predict(Xin)                                                          # :130
        flat = np.ascontiguousarray(enc).reshape(len(enc), -1)                          # :131
        t2 = time.perf_counter()
        red = pca.transform(scaler.transform(flat))                                     # :134-135
        t3 = time.perf_counter()
        cp = dets["Conservative"].predict(red); mp = dets["Moderate"].predict(red)      # :138-139
        cs = dets["Conservative"].decision_function(red); ms = dets["Moderate"].decision_function(red)   # :141-142
        t4 = time.perf_counter()
        n = len(cells)
        rec_ = dict(cells_per_s=round(n / (t4 - t0), 1), wall_ms=round((t4 - t0) * 1e3, 2),
                    split_ms=dict(autoencoder_predict_and_errors=round((t1 - t0) * 1e3, 2), encoder_predict=round((t2 - t1) * 1e3, 2),
                                  scaler_pca=round((t3 - t2) * 1e3, 2), svm_4_calls=round((t4 - t3) * 1e3, 2)))
        return rec_, dict(mse=mse, mae=mae, cons_score=-cs, mod_score=-ms, cons_pred=cp, mod_pred=mp)

    avail, quota = _host_cpu_budget()
    default_threads = torch.get_num_threads()
    cand = sorted({t for t in threads if t <= avail} | {min(avail, default_threads)})
    sweep = {}
    out = {}
    last = None
    try:
        for n in sizes:
            cells_sw = list(synth.synth_crops(seed, 0, min(n, sweep_crops)))
            sw = {}
            for t in cand:
                torch.set_num_threads(t)
                run(cells_sw[:64])                                     # thread pool of this size up, pages in
                sw[str(t)] = max(run(cells_sw)[0]["cells_per_s"] for _ in range(2))
            best_t = int(max(sw, key=lambda k: sw[k]))
            sweep[str(n)] = dict(crops=len(cells_sw), cells_per_s_by_threads=sw, best_threads=best_t)
            torch.set_num_threads(best_t)
            cells = list(synth.synth_crops(seed, 0, n))
            best = None
            for rep in range(3):                                      # best of three: the host is shared (a 1-GPU job's CPU quota on a 256-CPU box)
                rec_, res = run(cells)
                if best is None or rec_["wall_ms"] < best["wall_ms"]:
                    best = rec_
                last = res
            best["torch_threads"] = best_t
            out[str(n)] = best
    finally:
        torch.set_num_threads(default_threads)
    big = out[str(sizes[-1])]
    return dict(value=big["cells_per_s"], unit="cells/s", cores=big["torch_threads"], kind=kind, label=label,
                sample="%d crops (seed %d) at the best intra-op thread count of the sweep; also BASELINE.json configs[0]'s 128 crops: %.1f cells/s at %d threads"
                       % (sizes[-1], seed, out[str(sizes[0])]["cells_per_s"], out[str(sizes[0])]["torch_threads"]),
                sizes=out, thread_sweep=sweep, host_cpu_count=os.cpu_count(), host_cpus_in_affinity_mask=avail, cgroup_cpu_quota=quota,
                torch_threads=big["torch_threads"], torch_threads_default=default_threads,
                omp_num_threads=os.environ.get("OMP_NUM_THREADS"), cpu_model=_cpu_model()), last


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return None


# ------------------------------------------------------------------------------------------- extra legs
def train_leg(steps, local_rank, seed):
    """BASELINE.json configs[1] as written (CAE_improved_modeltrain.py:240,246-254,286-293): 50,000 synthetic crops split
    40,000 / 10,000, resident in HBM; one epoch = 1,250 fit() batches of 32 -- shuffled gather, the reference's
    ImageDataGenerator augmentation on the INPUT only, forward + backward + Adam, each batch ONE library call (cs_train_fit_step: the
    transforms are drawn in C from a keyed generator, the batch is gathered by the resampling kernel) -- then the validation pass
    over the 10,000 held-out crops.  Nothing synchronises the host inside the epoch: the per-step loss / MAE stay on the device
    and are read once at the end (cs_train_read_metrics)."""
    import numpy as np
    import torch
    from cellscreen import synth
    from cellscreen.augment import ImageDataGenerator
    from cellscreen.trainer import Trainer
    dev = torch.device("cuda", local_rank)
    n_all, n_val = 50_000, 10_000
    X = torch.from_numpy(synth.blob_crops(seed, n_all)).to(dev)
    Xtr, Xva = X[:n_all - n_val], X[n_all - n_val:]
    full_epoch = steps <= 0
    steps = (n_all - n_val) // 32 if full_epoch else steps
    tr = Trainer(synth.random_cae(seed=seed, trivial_bn=True), device_id=local_rank)
    gen = ImageDataGenerator(rotation_range=2, width_shift_range=0.02, height_shift_range=0.02, zoom_range=0.02,
                             horizontal_flip=True, vertical_flip=True, fill_mode="nearest")        # CAE...:246-254
    rng = np.random.default_rng(1234)
    warm = 20
    tg = torch.Generator(device=dev); tg.manual_seed(1234)
    try:
        idx = torch.randint(0, len(Xtr), (warm, 32), device=dev, generator=tg)
        first = None
        for i in range(warm):
            yb = Xtr[idx[i]].contiguous()
            l, _ = tr.step(tr.augment(yb, gen.random_transforms(32, (64, 64), rng)), yb, 1e-3)
            first = l if first is None else first
        torch.cuda.synchronize()
        Xtr = Xtr.contiguous()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        perm = torch.randperm(len(Xtr), device=dev, generator=tg)[:steps * 32].view(steps, 32)       # one shuffled pass
        tr.reset_metrics()
        idx = perm.cpu().numpy().astype(np.int32)          # the epoch's shuffled order, on the host (32 x 4 bytes per step)
        cfg = gen.config()
        for i in range(steps):                             # ONE library call per fit() batch: gather, keyed augmentation draws, step
            tr.fit_step(Xtr, idx[i], cfg, seed=seed, step=i, lr=1e-3)
        t_enq = time.perf_counter() - t0                   # when the host has enqueued the epoch (it runs ahead of the device, or bounds it)
        loss_epoch, mae_epoch, _ = tr.read_metrics()       # one host round trip for the epoch (Keras's running means)
        torch.cuda.synchronize()
        t_train = time.perf_counter() - t0
        val_loss, val_mae = tr.evaluate(Xva, Xva) if full_epoch else (None, None)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        # what a step costs the HOST: a burst shorter than the ring of in-flight steps (16), so that nothing waits for the device
        # (inside the epoch the host runs 16 steps ahead and then waits for slots: its loop time there is the device's)
        nb = 12
        tb = time.perf_counter()
        for i in range(nb):
            tr.fit_step(Xtr, idx[i % len(idx)], cfg, seed=seed + 1, step=i, lr=1e-3)
        t_host = (time.perf_counter() - tb) / nb
        torch.cuda.synchronize()
    finally:
        tr.close()
    return dict(workload="BASELINE.json configs[1]: CAE training (fwd + bwd + Adam, BN batch statistics, on-device augmentation of the input), "
                         "50,000 synthetic crops -> 40,000 / 10,000, batch 32, fp32, 1 GPU" + ("" if full_epoch else " (--train-steps: partial epoch, no validation pass)"),
                steps=steps, ms_per_step=round(t_train / steps * 1e3, 4), host_enqueue_ms_per_step=round(t_host * 1e3, 4),
                host_loop_ms_per_step_in_epoch=round(t_enq / steps * 1e3, 4), host_calls_per_step=1, cells_per_s=round(steps * 32 / t_train, 1),
                tflops_algorithmic=round(steps * 32 / t_train * 3 * FLOP_PER_CELL / 1e12, 3),
                epoch_s=round(el, 3) if full_epoch else None, validation_s=round(el - t_train, 3) if full_epoch else None,
                loss_first_step=round(first, 6), loss_epoch_mean=round(loss_epoch, 6), mae_epoch_mean=round(mae_epoch, 6),
                val_loss=None if val_loss is None else round(val_loss, 6))


def exact_fp32_leg(weights, det, x, out, host, args, local_rank):
    """The headline step on a second engine created with precision="fp32_exact" (cs_model_options: every contraction on
    v_mfma_f32_16x16x4_f32 -- the reference's float32 arithmetic, improved_detection.py:122,125,130): the same --steps / --warmup,
    the same 18 B/cell copy into pinned host memory and the same synchronisation as step().  Returns the leg and the results'
    first 65,536 cells (for its own parity check against the CPU oracle)."""
    import torch
    from cellscreen.engine import Engine
    e = Engine.from_weights(weights, None, det, device_id=local_rank, precision="fp32_exact")
    assert e.precision == "fp32_exact"

    def step():
        e.screen(x, out=out, out_device=True)
        for k, v in out.items():
            host[k].copy_(v, non_blocking=True)
        torch.cuda.current_stream().synchronize()

    try:
        e.set_chunk(args.chunk)
        for _ in range(max(1, args.warmup)):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
        head = {k: v[:65536].clone().numpy() for k, v in host.items()}
    finally:
        e.close()
    return dict(value=round(len(x) / dt, 1), unit="cells/s", ms_per_step=round(dt * 1e3, 3), steps=args.steps, warmup=max(1, args.warmup),
                precision="fp32_exact", dtype="f32 (v_mfma_f32_16x16x4_f32 for every contraction; SVM fp64)",
                note="same step as the headline: results end in pinned host memory, stream synchronised"), head


def large_variant_leg(local_rank, seed, n=8192, train_steps=200):
    """BASELINE.json configs[4], the half one GPU can run: 128x128 crops through the 32-64-128 | 128-64-32-1 instance of the layer
    grammar (CAE_improved_modeltrain.py:184: input_shape is an argument) -- autoencoder forward + per-cell MSE / MAE on crops resident
    in HBM, and batch-32 training steps (forward with batch statistics, backward, Adam) of the run-time-shaped trainer."""
    import torch
    from cellscreen import synth
    from cellscreen.engine import Engine
    from cellscreen.trainer import Trainer
    hw, ch = (128, 128), (32, 64, 128, 128, 64, 32, 1)
    macs = 349.18e6
    dev = torch.device("cuda", local_rank)
    w = synth.random_cae(seed=seed + 3, hw=hw, channels=ch, n_enc=3)
    e = Engine.from_weights(w, device_id=local_rank)
    try:
        x = torch.rand((n,) + hw, dtype=torch.float32, device=dev)
        e.set_chunk(4096)
        e.reconstruct(x, want_recon=False)
        torch.cuda.synchronize()
        e.profile_enable(True); e.profile_reset()
        steps = 3
        t0 = time.perf_counter()
        for _ in range(steps):
            e.reconstruct(x, want_recon=False)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        e.profile_enable(False)
        kern = {}
        for k, v in e.profile().items():
            if not v["launches"]:
                continue
            r = dict(ms_per_step=round(v["ms"] / steps, 3))
            if v.get("bf16_mfma_per_cell", 0) > 0:
                tf = v["bf16_mfma_per_cell"] * FLOP_PER_BF16_MFMA * v["cells"] / (v["ms"] * 1e-3) / 1e12
                r.update(mfma16_per_cell=v["bf16_mfma_per_cell"], tflops_executed_16bit=round(tf, 1), frac_bf16_mfma_peak=round(tf / BF16_MFMA_PEAK_TFLOPS, 4))
            elif v["mfma_per_cell"] > 0:
                tf = v["mfma_per_cell"] * FLOP_PER_MFMA * v["cells"] / (v["ms"] * 1e-3) / 1e12
                r.update(mfma_per_cell=v["mfma_per_cell"], tflops_executed=round(tf, 2), frac_fp32_mfma_peak=round(tf / FP32_MFMA_PEAK_TFLOPS, 4))
            kern[k] = r
        del x
    finally:
        e.close()
    torch.cuda.empty_cache()
    fwd = dict(crops=n, cells_per_s=round(n / dt, 1), ms_per_step=round(dt * 1e3, 3), steps=steps, precision="split16",
               tflops_algorithmic=round(2 * macs * n / dt / 1e12, 2), kernels=kern)
    X = torch.from_numpy(synth.blob_crops(seed, 1024, hw=hw)).to(dev)
    tr = Trainer(synth.random_cae(seed=seed, hw=hw, channels=ch, n_enc=3, trivial_bn=True), device_id=local_rank)
    gen = torch.Generator(device=dev); gen.manual_seed(99)
    try:
        idx = torch.randint(0, len(X), (train_steps + 20, 32), device=dev, generator=gen)
        first = None
        for i in range(20):
            xb = X[idx[i]].contiguous()
            l, _ = tr.step(xb, xb, 1e-3)
            first = l if first is None else first
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tr.reset_metrics()
        for i in range(20, 20 + train_steps):
            xb = X[idx[i]].contiguous()
            tr.step_async(xb, xb, 1e-3)
        loss_mean, _, _ = tr.read_metrics()
        torch.cuda.synchronize()
        dtt = (time.perf_counter() - t0) / train_steps
    finally:
        tr.close()
    train = dict(steps=train_steps, batch=32, ms_per_step=round(dtt * 1e3, 4), cells_per_s=round(32 / dtt, 1),
                 tflops_algorithmic=round(32 / dtt * 3 * 2 * macs / 1e12, 2), frac_fp32_mfma_peak_algorithmic=round(32 / dtt * 3 * 2 * macs / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4),
                 loss_first_step=round(first, 6), loss_mean=round(loss_mean, 6))
    return dict(workload="BASELINE.json configs[4] on one GPU: 128x128 crops, deeper CAE (filters 32-64-128 | 128-64-32-1, 128-channel bottleneck, "
                         "349.18 M MAC/cell), run-time-shaped MFMA implicit-GEMM kernels; the 8-GPU gradient all-reduce half is bench_train.py --variant under torchrun",
                forward=fwd, train=train)


def e2e_raw_leg(weights, n, seed, local_rank, chunk=65536):
    """improved_detection.py:98-99 -> :199 as one pipeline: raw uint16 bounding-box crops (ragged, sides U[32,100]) in PINNED host
    memory -> H2D (copy stream, chunk i + 1 under the kernels of chunk i) -> cs_preprocess (device -> device) -> cs_screen
    -> 18 B/cell back on the host."""
    import numpy as np
    import torch
    from cellscreen import preprocess as pp
    from cellscreen import synth
    dev = torch.device("cuda", local_rank)
    base = synth.raw_crops(seed, 4096, np.uint16, 32, 100)
    bpix, boff, bhs, bws = pp.pack_crops(base)
    reps = (n + len(base) - 1) // len(base)
    hs, ws = np.tile(bhs, reps)[:n], np.tile(bws, reps)[:n]
    sizes = hs.astype(np.int64) * ws.astype(np.int64)
    off = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64)
    total = int(sizes.sum())
    host = torch.empty(total, dtype=torch.int16, pin_memory=True)          # uint16 pixels (torch has no pinned uint16 before 2.3-style views)
    hv = host.numpy().view(np.uint16)
    for r in range(reps):                                                  # the packed base, repeated
        lo = r * len(bpix)
        m = min(len(bpix), total - lo)
        if m > 0:
            hv[lo:lo + m] = bpix[:m]
    # this leg's detector is fit on what it screens: encoder features of PREPROCESSED raw crops (2,048 of another seed, the device
    # fit of csrc/fit.hip), so its anomaly rates mean something -- about nu = 0.05 / 0.10 for crops from the training distribution
    from cellscreen.detector_fit import fit_detector_device
    from cellscreen.engine import Engine
    tpix, toff, ths, tws = pp.pack_crops(synth.raw_crops(seed + 17, 2048, np.uint16, 32, 100))
    tcrops = torch.empty((len(ths), 64, 64), dtype=torch.float32, device=dev)
    tproc = pp.Preprocessor(local_rank)
    tproc.run_packed(torch.from_numpy(tpix.view(np.int16)).to(dev), toff, ths, tws, out=tcrops)
    tproc.close()
    enc = Engine.from_weights(weights, device_id=local_rank)
    tfeat = enc.encode(tcrops, which=0).cpu().numpy()
    enc.close()
    det_raw, _ = fit_detector_device(tfeat, device_id=local_rank)
    del tcrops
    eng = Engine.from_weights(weights, None, det_raw, device_id=local_rank)
    eng.set_chunk(chunk)
    proc = pp.Preprocessor(local_rank)
    # HIP deals streams round-robin onto its hardware queues (GPU_MAX_HW_QUEUES = 6 here), and two streams on one queue run their
    # work in submission order: an upload stream that lands on the queue of the preprocess or the screening stream serialises the
    # pipeline (2.1-2.3 M instead of 3.2 M cells/s: DESIGN.md section 6).  Which queue a stream gets depends on how many were
    # created before it, so the upload stream is CHOSEN: six candidates (consecutive creations: one per queue), a four-chunk
    # trial of the pipeline on each, the fastest kept.  A deployment does the same once at start-up.
    main_stream = torch.cuda.Stream(device=dev)          # NOT torch's default stream: see below
    copy_candidates = [torch.cuda.Stream(device=dev) for _ in range(6)]
    copy_stream = copy_candidates[0]
    bounds = [(i, min(i + chunk, n)) for i in range(0, n, chunk)]
    span = max(int(off[b - 1] + sizes[b - 1] - off[a]) for a, b in bounds)
    d_pix = [torch.empty(span, dtype=torch.int16, device=dev) for _ in range(2)]
    d_crops = torch.empty((chunk, 64, 64), dtype=torch.float32, device=dev)
    out = dict(mse=torch.empty(n, dtype=torch.float32, device=dev), mae=torch.empty(n, dtype=torch.float32, device=dev),
               cons_score=torch.empty(n, dtype=torch.float64, device=dev), mod_score=torch.empty(n, dtype=torch.float64, device=dev),
               cons_pred=torch.empty(n, dtype=torch.int8, device=dev), mod_pred=torch.empty(n, dtype=torch.int8, device=dev))
    res = {k: torch.empty(n, dtype=v.dtype, pin_memory=True) for k, v in out.items()}
    evs = [torch.cuda.Event() for _ in range(2)]

    def copy_in(ci):
        a, b = bounds[ci]
        lo, hi = int(off[a]), int(off[b - 1] + sizes[b - 1])
        with torch.cuda.stream(copy_stream):
            d_pix[ci & 1][:hi - lo].copy_(host[lo:hi], non_blocking=True)
            evs[ci & 1].record(copy_stream)

    # NOT torch's default stream: that is the legacy null stream, and every operation on it (the event the wrappers record to
    # order the library after torch, the wait below) is a barrier against all blocking streams -- the upload of chunk i + 1 on
    # the copy stream would be waited for before chunk i's kernels start (measured: 0.47 s = upload + preprocess + screen)

    def run(nchunks=None):
        nb = len(bounds) if nchunks is None else min(nchunks, len(bounds))
        with torch.cuda.stream(main_stream):
            copy_in(0)
            for ci, (a, b) in enumerate(bounds[:nb]):
                if ci + 1 < nb:
                    copy_in(ci + 1)            # the other buffer: its last reader (chunk ci - 1's preprocess) has returned
                main_stream.wait_event(evs[ci & 1])
                proc.run_packed(d_pix[ci & 1], off[a:b] - off[a], hs[a:b], ws[a:b], out=d_crops[:b - a])
                eng.screen(d_crops[:b - a], out={k: v[a:b] for k, v in out.items()}, out_device=True)
            for k in out:
                res[k].copy_(out[k], non_blocking=True)
        torch.cuda.synchronize()

    trial_ms = []
    try:
        run()                              # warm-up (allocations, first-touch of the pinned pages)
        for cand in copy_candidates:
            copy_stream = cand
            run(4)
            t0 = time.perf_counter()
            run(4)
            trial_ms.append(round((time.perf_counter() - t0) * 1e3, 2))
        copy_stream = copy_candidates[int(np.argmin(trial_ms))]
        t0 = time.perf_counter()
        run()
        dt = time.perf_counter() - t0
    finally:
        proc.close()
        eng.close()
    return dict(value=round(n / dt, 1), unit="cells/s", wall_s=round(dt, 4), crops=n, chunk_crops=chunk,
                h2d_bytes_per_cell=round(2.0 * total / n, 1), d2h_bytes_per_cell=18,
                h2d_gbs=round(2.0 * total / dt / 1e9, 2), gpu_max_hw_queues=os.environ.get("GPU_MAX_HW_QUEUES"),
                upload_stream_trials_ms=trial_ms, upload_stream_chosen=int(np.argmin(trial_ms)),
                workload="%d raw uint16 crops, sides U[32,100] (4,096 distinct, repeated), pinned host -> cs_preprocess -> cs_screen -> host" % n,
                detector="fit on the encoder features of 2,048 preprocessed raw crops of another seed (device fit: nu = 0.05 / 0.10)",
                n_sv=[int(det_raw.conservative.n_sv), int(det_raw.moderate.n_sv)],
                anomaly_rate_conservative=round(float((res["cons_pred"] == -1).float().mean()), 4),
                anomaly_rate_moderate=round(float((res["mod_pred"] == -1).float().mean()), 4))


def small_n_leg(eng, seed):
    """The reference calls the hot path once per sample with 1e2..1e4 cells in host memory (improved_detection.py:199):
    single-call wall time of cs_screen from numpy crops to numpy results."""
    import numpy as np
    from cellscreen import synth
    out = {}
    for n in (128, 1024, 10240):
        x = synth.synth_crops(seed, 5_000_000, n)
        eng.screen(x)
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            eng.screen(x)
            ts.append(time.perf_counter() - t0)
        out[str(n)] = dict(ms=round(float(np.median(ts)) * 1e3, 3), cells_per_s=round(n / float(np.median(ts)), 1))
    return out


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    # ---- HBM traffic of this tree: PMC child passes first, while this process has not touched the GPU yet
    pmc, pmc_src = None, None
    under_profiler = any(k.startswith(("ROCPROFILER_", "ROCP_")) for k in os.environ) or "rocprofiler" in os.environ.get("LD_PRELOAD", "")
    if world == 1 and not args.pmc_child and not args.no_pmc and not under_profiler:      # never nest profiler runs
        pmc, pmc_src = collect_pmc_traffic(args)

    import numpy as np
    import torch
    import torch.distributed as dist

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
    torch.cuda.current_stream().synchronize()
    feats = enc.encode(xt, which=0).cpu().numpy()
    enc.close()
    del xt
    det, sk = fit_detector(feats, pca_random_state=0)            # same on every rank (same inputs)
    eng = Engine.from_weights(weights, None, det, device_id=local_rank)
    eng.set_chunk(args.chunk)

    # ---- this rank's shard of the global synthetic batch, generated in HBM
    if args.cells <= 0:
        args.cells = 1_000_000 if world == 1 else 1_250_000          # BASELINE.json configs[2] | configs[3] (10 M over 8 GPUs)
    n_total = args.cells * world
    lo, hi = csdist.shard_range(n_total, rank, world)
    n_local = hi - lo
    x = torch.empty((n_local, 64, 64), dtype=torch.float32, device=dev)
    eng.synth_crops(args.seed, lo, x)
    out = dict(mse=torch.empty(n_local, dtype=torch.float32, device=dev), mae=torch.empty(n_local, dtype=torch.float32, device=dev),
               cons_score=torch.empty(n_local, dtype=torch.float64, device=dev), mod_score=torch.empty(n_local, dtype=torch.float64, device=dev),
               cons_pred=torch.empty(n_local, dtype=torch.int8, device=dev), mod_pred=torch.empty(n_local, dtype=torch.int8, device=dev))
    torch.cuda.synchronize()

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
        kern = {k: v for k, v in prof.items() if v["launches"] > 0}
        total_ms = sum(v["ms"] for v in kern.values())
        ex_tf = lambda v: v["mfma_per_cell"] * FLOP_PER_MFMA * v["cells"] / (v["ms"] * 1e-3) / 1e12      # noqa: E731
        bf_tf = lambda v: v["bf16_mfma_per_cell"] * FLOP_PER_BF16_MFMA * v["cells"] / (v["ms"] * 1e-3) / 1e12      # noqa: E731
        # ---- roofline: dominant kernel by device time, EXECUTED matrix-pipe FLOPs / measured duration
        # a kernel's matrix work at the peaks of the pipes it runs on: seconds at peak, executed FLOP (fp32-MFMA + bf16-MFMA)
        def at_peak(v):
            f32 = v["mfma_per_cell"] * FLOP_PER_MFMA * v["cells"]
            b16 = v.get("bf16_mfma_per_cell", 0) * FLOP_PER_BF16_MFMA * v["cells"]
            return f32 / (FP32_MFMA_PEAK_TFLOPS * 1e12) + b16 / (BF16_MFMA_PEAK_TFLOPS * 1e12), f32 + b16, b16
        dom = max((k for k in kern if at_peak(kern[k])[1] > 0), key=lambda k: kern[k]["ms"])
        d = kern[dom]
        avg_ms = d["ms"] / d["launches"]
        cpl = d["cells"] / d["launches"]
        t_pk, fl_ex, fl_b16 = at_peak(d)
        ach = fl_ex / (d["ms"] * 1e-3) / 1e12
        # the peak the kernel's own mix of instructions could reach: all of it fp32-MFMA -> 157.3; with bf16 MFMAs in the mix, higher
        peak_mix = fl_ex / t_pk / 1e12
        alg = d["flops"] / (d["ms"] * 1e-3) / 1e12
        if pmc is None and not args.pmc_child:
            pmc, pmc_src = committed_pmc_traffic()
        tk = pmc["kernels"] if pmc else {}
        traffic = round(tk[dom]["hbm_bytes_per_cell"] * cpl) if dom in tk else None
        # cross-check of the MFMA count against SQ_INSTS_MFMA when a counters file of this tree is committed
        sq_check = None
        try:
            from build import source_hash
            for f in sorted(os.listdir(os.path.join(ROOT, "profiles")), reverse=True):
                if f.endswith("_sq_counters.json"):
                    sj = json.load(open(os.path.join(ROOT, "profiles", f)))
                    if sj.get("source_hash") == source_hash() and dom in sj["kernels"] and "SQ_INSTS_MFMA" in sj["kernels"][dom]:
                        per_cell = sj["kernels"][dom]["SQ_INSTS_MFMA"] / sj.get("cells_per_launch", 65536)
                        sq_check = dict(file="profiles/" + f, sq_insts_mfma_per_cell=round(per_cell, 2),
                                        library_mfma_per_cell=d["mfma_per_cell"] + d.get("bf16_mfma_per_cell", 0),
                                        equal=bool(abs(per_cell - d["mfma_per_cell"] - d.get("bf16_mfma_per_cell", 0)) < 0.5))
                    break
        except Exception:  # noqa: BLE001
            pass
        roofline = dict(bound="mfma", kernel=dom, achieved=round(ach, 3), peak=round(peak_mix, 1), unit="TFLOP/s",
                        frac=round(ach / peak_mix, 4),
                        executed_mfma_per_cell=d["mfma_per_cell"], executed_bf16_mfma_per_cell=d.get("bf16_mfma_per_cell", 0),
                        executed_flop_per_launch=int(fl_ex / d["launches"]),
                        peak_note=("peak = executed FLOP / (fp32-MFMA FLOP / 157.3 TFLOP/s + bf16-MFMA FLOP / 2,500 TFLOP/s): the rate this kernel's "
                                   "own instruction mix would reach with both matrix pipes at their peaks; 157.3 for a pure fp32-MFMA kernel"),
                        achieved_algorithmic=round(alg, 3), algorithmic_speedup=round(alg / ach, 4) if ach > 0 else None,
                        note="achieved/frac price the multiply-adds the kernel executes (fp32 MFMAs x 2,048 FLOP + bf16 MFMAs x 16,384 FLOP); achieved_algorithmic "
                             "counts the reference graph's FLOPs of the same layers, of which Winograd / folded-upsample kernels execute a fraction",
                        sq_insts_mfma_check=sq_check,
                        traffic=traffic, traffic_unit="HBM bytes per launch", traffic_source=pmc_src,
                        algorithmic_bytes_per_launch=int(ALG_BYTES[dom] * cpl) if dom in ALG_BYTES else None,
                        avg_launch_ms=round(avg_ms, 4), cells_per_launch=int(cpl),
                        share_of_device_time=round(d["ms"] / total_ms, 4))
        kernels = {k: dict(ms=round(v["ms"], 3), launches=v["launches"], share=round(v["ms"] / total_ms, 4),
                           # fp32-MFMA pricing; a kernel whose main contraction runs on bf16 MFMAs is priced by the bf16 keys below
                           tflops_executed=round(ex_tf(v), 2) if v["mfma_per_cell"] > 0 and not v.get("bf16_mfma_per_cell", 0) else None,
                           frac_executed=round(ex_tf(v) / FP32_MFMA_PEAK_TFLOPS, 4) if v["mfma_per_cell"] > 0 and not v.get("bf16_mfma_per_cell", 0) else None,
                           tflops_algorithmic=round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2) if v["flops"] > 0 else None,
                           # seconds the kernel's executed matrix work takes with both pipes at their peaks / its device time
                           frac_matrix_peaks=round(at_peak(v)[0] / (v["ms"] * 1e-3), 4) if at_peak(v)[1] > 0 else None,
                           **({"bf16_mfma_per_cell": v["bf16_mfma_per_cell"],
                               "tflops_executed_bf16": round(bf_tf(v), 1), "frac_bf16_mfma_peak": round(bf_tf(v) / BF16_MFMA_PEAK_TFLOPS, 4)}
                              if v.get("bf16_mfma_per_cell", 0) > 0 else {}),
                           hbm_bytes_per_cell=tk[k]["hbm_bytes_per_cell"] if k in tk else None,
                           algorithmic_bytes_per_cell=ALG_BYTES.get(k))
                   for k, v in kern.items()}
        enc_k = [k for k in ENCODER_KERNELS if k in kern]
        enc_ms = sum(kern[k]["ms"] for k in enc_k)
        enc_cells = kern[enc_k[0]]["cells"]
        enc_hbm = {"kernels": enc_k, "algorithmic_bytes_per_cell": 16384 + 8192,
                   "algorithmic_gbs_at_encoder_rate": round(enc_cells * 24576 / (enc_ms * 1e-3) / 1e9, 1),
                   "frac_hbm_peak_algorithmic": round(enc_cells * 24576 / (enc_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                   "measured_bytes_per_cell": None, "frac_hbm_peak_measured": None, "encoder_ms_per_step": round(enc_ms / args.steps, 3)}
        if tk and all(k in tk for k in enc_k):
            mb = sum(tk[k]["hbm_bytes_per_cell"] for k in enc_k)
            enc_hbm["measured_bytes_per_cell"] = round(mb)
            enc_hbm["frac_hbm_peak_measured"] = round(enc_cells * mb / (enc_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
            enc_hbm["measured_source"] = pmc_src
        conv_cells = max(kern[k]["cells"] for k in kern if k.startswith("conv"))
        exec_flop_per_cell = sum(kern[k]["mfma_per_cell"] * FLOP_PER_MFMA * kern[k]["cells"] for k in kern if k.startswith("conv")) / max(1, conv_cells)
        exec_bf16_flop_per_cell = sum(kern[k].get("bf16_mfma_per_cell", 0) * FLOP_PER_BF16_MFMA * kern[k]["cells"] for k in kern if k.startswith("conv")) / max(1, conv_cells)
        whole_traffic = round(sum(tk[k]["hbm_bytes_per_cell"] for k in kern if k in tk)) if tk else None
        line = {
            "metric": "cells/sec screened (CAE fwd + recon-MSE + SVM score), 64x64",
            "value": round(value, 1), "unit": "cells/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32 (results in the fp32 error class; the contractions run as split 16-bit products on the matrix pipe, fp32 accumulate: "
                     "conv1..conv6 as 2 fp16 terms x 3 products with exact power-of-two scales from each cell's own data, the PCA GEMM as 3 bf16 terms x 6 products, "
                     "conv7's 32 -> 16 channel contraction on fp32 MFMAs; "
                     "SVM fp64; exact_fp32 = the fp32-MFMA form)",
            "data": "synthetic",
            "config": {"workload": (("BASELINE.json configs[2]: screening inference, %d synthetic 64x64 crops on one GPU, resident in HBM, " % args.cells) if world == 1 else
                                    ("BASELINE.json configs[3]: %d-GPU data-parallel screening, %d synthetic 64x64 crops per GPU (%d global; 10 M at 8 GPUs), "
                                     "contiguous shards resident in HBM, RCCL gather of per-cell scores, " % (world, args.cells, n_total)))
                                   + "CAE fwd + recon MSE/MAE + RobustScaler + PCA(100) + 2x OCSVM score on device",
                       "cells_per_gpu": args.cells, "global_cells": n_total, "chunk_cells": args.chunk,
                       "n_sv": [int(det.conservative.n_sv), int(det.moderate.n_sv)], "detector_train_cells": args.train_cells,
                       "weights": "random init (Glorot, non-trivial BN), seed %d" % args.seed,
                       "parallelism": "dp%d" % world},
            "whole_path": {"tflops_executed_fp32_mfma": round(value * exec_flop_per_cell / 1e12 / world, 3),
                           "tflops_executed_bf16_mfma": round(value * exec_bf16_flop_per_cell / 1e12 / world, 3),
                           # device time the executed matrix work would take at the two pipes' peaks / the time it takes
                           "frac_matrix_peaks": round(value / world * (exec_flop_per_cell / (FP32_MFMA_PEAK_TFLOPS * 1e12)
                                                                        + exec_bf16_flop_per_cell / (BF16_MFMA_PEAK_TFLOPS * 1e12)), 4),
                           "executed_fp32_mfma_flop_per_cell": int(exec_flop_per_cell),
                           "executed_bf16_mfma_flop_per_cell": int(exec_bf16_flop_per_cell),
                           "tflops_algorithmic": round(value * FLOP_PER_CELL / 1e12 / world, 3),
                           "note": "executed = the conv kernels' MFMA counts per cell: v_mfma_f32_16x16x4_f32 x 2,048 FLOP (conv1+conv2, conv3, conv7's contraction) "
                                   "and the 16-bit forms v_mfma_f32_16x16x32_{bf16,f16} x 16,384 FLOP (the fp32 contraction as six bf16 or three fp16 products); "
                                   "algorithmic = the reference graph's 100.27 MFLOP/cell",
                           "hbm_gbs_algorithmic": round(value * BYTES_PER_CELL / 1e9 / world, 2),
                           "frac_hbm_peak": round(value * BYTES_PER_CELL / 1e9 / world / HBM_PEAK_GBS, 5),
                           "hbm_bytes_per_cell_measured": whole_traffic,
                           "hbm_gbs_measured": round(value * whole_traffic / 1e9 / world, 1) if whole_traffic else None,
                           "device_ms_per_step": round(total_ms / args.steps, 3),
                           # BASELINE.json asks for the fraction of the HBM roofline on the conv encoder (conv1-3): algorithmic
                           # bytes = one crop in + 8 KB of features out; measured = PMC traffic of its kernels.  Exact-fp32
                           # convs are MFMA-bound, so neither can approach 1.
                           "conv_encoder_hbm": enc_hbm},
            "roofline": roofline,
            "kernels": kernels,
        }
        oracle_ref = None
        if world == 1 and not args.no_cpu_baseline and not args.pmc_child:
            cb, ref_seq = cpu_reference_sequence(weights, sk, args.seed)
            port, ref, xs = cpu_port(weights, det, args.seed, args.cpu_sample)
            oracle_ref = ref
            # the contract's object is the oracle port ("port"); the reference's own call sequence on torch-CPU + scikit-learn rides along
            port["reference_sequence"] = cb
            line["cpu_baseline"] = port
            # the bounded samples double as live parity checks of the benchmarked run
            n = len(xs)
            mse = res["mse"][:n].numpy()
            rel = float(np.max(np.abs(mse - ref["mse"]) / ref["mse"]))
            tol = 1e-4 * float(np.abs(det.conservative.dual_coef).sum())
            derr = float(np.max(np.abs(res["cons_score"][:n].numpy() - ref["cons_score"])))
            line["parity_on_cpu_sample"] = {"cells": n, "mse_max_rel": rel, "cons_score_max_abs": derr,
                                            "ok": bool(rel <= 1e-5 and derr <= tol)}
            m = len(ref_seq["mse"])       # the reference sequence itself (fp32 library arithmetic): looser by its own rounding
            rel2 = float(np.max(np.abs(res["mse"][:m].numpy() - ref_seq["mse"]) / ref_seq["mse"]))
            derr2 = float(np.max(np.abs(res["cons_score"][:m].numpy() - ref_seq["cons_score"])))
            sure = np.abs(ref_seq["cons_score"]) > tol
            line["parity_vs_reference_sequence"] = {"cells": m, "mse_max_rel": rel2, "cons_score_max_abs": derr2, "score_tol": tol,
                                                    "labels_equal_away_from_0": bool(np.array_equal(res["cons_pred"][:m].numpy()[sure], ref_seq["cons_pred"][sure]))}
        if world == 1 and not args.no_extra_legs and not args.pmc_child:
            try:
                line["small_n"] = small_n_leg(eng, args.seed)
            except Exception as e:  # noqa: BLE001 - an extra leg never costs the headline line
                line["small_n"] = {"error": repr(e)}
            try:
                line["exact_fp32"], head = exact_fp32_leg(weights, det, x, out, host, args, local_rank)
                if oracle_ref is not None:          # the exact mode's own live parity check, same sample and bars as the headline's
                    n = len(oracle_ref["mse"])
                    rel = float(np.max(np.abs(head["mse"][:n] - oracle_ref["mse"]) / oracle_ref["mse"]))
                    derr = float(np.max(np.abs(head["cons_score"][:n] - oracle_ref["cons_score"])))
                    tol = 1e-4 * float(np.abs(det.conservative.dual_coef).sum())
                    line["exact_fp32"]["parity_on_cpu_sample"] = {"cells": n, "mse_max_rel": rel, "cons_score_max_abs": derr, "ok": bool(rel <= 1e-5 and derr <= tol)}
            except Exception as e:  # noqa: BLE001
                line["exact_fp32"] = {"error": repr(e)}
            del x
            torch.cuda.empty_cache()
            try:
                line["e2e_raw"] = e2e_raw_leg(weights, args.raw_crops, args.seed, local_rank)
            except Exception as e:  # noqa: BLE001
                line["e2e_raw"] = {"error": repr(e)}
            torch.cuda.empty_cache()
            try:
                line["train_leg"] = train_leg(args.train_steps, local_rank, args.seed)
            except Exception as e:  # noqa: BLE001
                line["train_leg"] = {"error": repr(e)}
            torch.cuda.empty_cache()
            try:
                line["large_variant"] = large_variant_leg(local_rank, args.seed)
            except Exception as e:  # noqa: BLE001
                line["large_variant"] = {"error": repr(e)}
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
