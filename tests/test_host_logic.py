"""CPU tests of everything around the kernels: the C-ABI library loads and exports every
symbol the header declares, fails loudly without a GPU, model files round-trip, sharding
and CSV bookkeeping are bit-exact restatements of improved_detection.py:155-255."""
import ctypes as C
import io
import os
import re

import numpy as np
import pytest

import helpers as H
from cellscreen import _lib as L
from cellscreen import dist, model_io, spec, synth


def _header_symbols():
    src = open(os.path.join(H.ROOT, "include", "cellscreen.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(cs_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_header_symbol():
    lib = L.load_library()
    names = _header_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"libcellscreen.so does not export {n}"
    assert sorted(L.SIGNATURES) == names, "ctypes SIGNATURES out of sync with include/cellscreen.h"
    assert lib.cs_abi_version() == 2
    assert lib.cs_profile_kernel_count() == 13 and lib.cs_profile_kernel_name(12) == b"conv1_conv2_fused"
    assert lib.cs_profile_kernel_name(1) == b"conv2_relu_bn_pool"
    assert lib.cs_status_string(-4) == b"no usable gfx950 device"


def test_model_options_are_validated_before_any_device_work():
    """cs_model_options (ABI version 2): precision is an argument of the handle, not an environment variable.  Bad options are
    CS_ERR_INVALID before the device is looked at; the struct mirrors include/cellscreen.h."""
    import ctypes as C
    from cellscreen.engine import _fill_cae
    lib = L.load_library()
    assert C.sizeof(L.CSModelOptions) == 32 and L.PRECISION_SPLIT16 == 0 and L.PRECISION_FP32_EXACT == 1
    keep, h = [], C.c_void_p()
    w = _fill_cae(synth.random_cae(), keep)
    for bad in (dict(precision=2), dict(precision=-1), dict(debug_flags=0x10), dict(struct_size=4), dict(reserved0=1)):
        o = L.model_options()
        for k, v in bad.items():
            if k == "reserved0":
                o.reserved[0] = v
            else:
                setattr(o, k, v)
        assert lib.cs_model_from_arrays(C.byref(w), None, None, 0, C.byref(o), C.byref(h)) == -1, bad
        assert b"cs_model_options" in lib.cs_last_error()
    with pytest.raises(ValueError):
        L.model_options("bf16")
    for name, val in (("split16", 0), ("fp32_exact", 1), ("exact", 1)):
        assert L.model_options(name).precision == val
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "cellscreen.h")).read()
    assert "CS_PRECISION_SPLIT16 = 0" in hdr and "CS_PRECISION_FP32_EXACT = 1" in hdr
    # nothing but the two debug overrides reads the environment to choose arithmetic
    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cell-image-analysis_amd", "csrc")
    envs = set()
    for f in os.listdir(csrc):
        envs |= set(re.findall(r'getenv\("([A-Z0-9_]+)"\)', open(os.path.join(csrc, f)).read()))
    assert envs <= {"CS_DEBUG_PRECISION", "CS_DEBUG_FLAGS", "CS_C12_DIAG", "CS_WINO_DIAG"}, envs


def test_no_cpu_fallback_without_gpu():
    """The product path must fail loudly when there is no GPU (no oracle, no CPU route)."""
    lib = L.load_library()
    if lib.cs_device_count() > 0:
        pytest.skip("a GPU is visible")
    from cellscreen.engine import Engine
    with pytest.raises(L.CellScreenError) as e:
        Engine.from_weights(synth.random_cae())
    assert e.value.status == -4


def test_product_package_never_touches_the_oracle():
    pkg = os.path.join(H.ROOT, "cell-image-analysis_amd")
    for dirpath, _, files in os.walk(pkg):
        if "build" in dirpath.split(os.sep):
            continue
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f), errors="replace").read()
                assert "liboracle" not in txt and "import oracle" not in txt and "from oracle" not in txt, f


def test_bad_arguments_are_rejected_before_any_device_work():
    lib = L.load_library()
    h = C.c_void_p()
    assert lib.cs_model_load(b"/nonexistent/dir", 0, None, C.byref(h)) == -2          # CS_ERR_IO
    assert b"cannot open" in lib.cs_last_error()
    w = synth.random_cae()
    w.kernels[1] = np.zeros((3, 3, 32, 48), np.float32)                          # not the reference graph
    from cellscreen.engine import _fill_cae
    keep = []
    s = _fill_cae(w, keep)
    assert lib.cs_model_from_arrays(C.byref(s), None, None, 0, None, C.byref(h)) == -6  # CS_ERR_UNSUPPORTED
    assert lib.cs_model_from_arrays(None, None, None, 0, None, C.byref(h)) == -1       # CS_ERR_INVALID
    # the architecture is judged before the device is touched: a non-reference instance of the layer grammar
    # that the generic kernels cover gets as far as "no device" here; one outside the grammar is refused
    big = synth.random_cae(seed=5, hw=(128, 128), channels=(32, 64, 128, 128, 64, 32, 1), n_enc=3)
    rc = lib.cs_model_from_arrays(C.byref(_fill_cae(big, keep)), None, None, 0, None, C.byref(h))
    assert rc == (-4 if lib.cs_device_count() <= 0 else 0)
    if rc == 0:
        lib.cs_model_free(h)
    odd = synth.random_cae(seed=5, hw=(64, 64), channels=(32, 64, 32, 32, 1), n_enc=3)           # n_conv != 2 n_enc + 1
    assert lib.cs_model_from_arrays(C.byref(_fill_cae(odd, keep)), None, None, 0, None, C.byref(h)) == -6
    assert b"grammar" in lib.cs_last_error()


def test_model_dir_round_trip(tmp_path, golden_det):
    ae = synth.random_cae(3)
    enc = synth.perturbed_encoder(ae)
    det = H.det_from_golden(golden_det)
    d = str(tmp_path / "m")
    model_io.save_model_dir(d, ae, enc, det)
    assert sorted(os.listdir(d)) == ["cae.bin", "detector.bin", "manifest.json"]
    ae2, enc2, det2 = model_io.load_model_dir(d)
    for a, b in zip(ae.kernels + ae.bn_var, ae2.kernels + ae2.bn_var):
        assert np.array_equal(a, b)
    assert enc2 is not None and np.array_equal(enc.kernels[2], enc2.kernels[2])
    assert np.array_equal(det.scaler_scale, det2.scaler_scale) and det2.scaler_scale.dtype == np.float64
    assert np.array_equal(det.moderate.support_vectors, det2.moderate.support_vectors)
    assert det2.conservative.rho == det.conservative.rho
    # malformed file: the C reader must say so, not crash
    with open(os.path.join(d, "cae.bin"), "r+b") as f:
        f.seek(0); f.write(b"XXXX")
    h = C.c_void_p()
    assert L.load_library().cs_model_load(d.encode(), 0, None, C.byref(h)) == -3       # CS_ERR_FORMAT


def test_reference_pickles_convert(tmp_path, golden_det):
    """scaler.pkl / pca.pkl / detector_*.pkl as CAE_improved_modeltrain.py:437-444 writes them."""
    pytest.importorskip("sklearn")
    from cellscreen.detector_fit import fit_detector
    rng = np.random.default_rng(1)
    feats = rng.standard_normal((150, 2048)).astype(np.float32)
    params, _ = fit_detector(feats, output_dir=str(tmp_path), pca_random_state=0)
    assert sorted(p for p in os.listdir(tmp_path)) == ["detector_conservative.pkl", "detector_moderate.pkl", "pca.pkl", "scaler.pkl"]
    p2 = model_io.detector_from_reference_pickles(str(tmp_path))
    assert np.array_equal(p2.pca_components, params.pca_components)
    assert p2.moderate.rho == params.moderate.rho and p2.n_components == 100


def test_shard_ranges_partition_exactly():
    for n in (0, 1, 7, 8, 1000003, 10_000_000):
        for world in (1, 2, 3, 8):
            r = [dist.shard_range(n, k, world) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[i][1] == r[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in r]
            assert max(sizes) - min(sizes) <= 1 and sizes == dist.shard_counts(n, world)
    with pytest.raises(ValueError):
        dist.shard_range(10, 2, 2)


class _FakeEngine:
    """Deterministic stand-in for the GPU engine: bookkeeping tests only."""
    class info:
        has_detector = 1
        height, width = 64, 64

    def screen(self, X):
        n = len(X)
        m = X.reshape(n, -1).mean(axis=1).astype(np.float32)
        dec = (m.astype(np.float64) - 0.5) * 10
        return dict(mse=m * m, mae=m, cons_score=-dec, mod_score=-(dec - 0.01),
                    cons_pred=np.where(dec > 0, 1, -1).astype(np.int8),
                    mod_pred=np.where(dec - 0.01 > 0, 1, -1).astype(np.int8))


def _screener(tmp_path, **kw):
    from cellscreen.screening import ProductionMutantScreening
    s = ProductionMutantScreening.__new__(ProductionMutantScreening)
    s.model_dir, s.device_id, s.cell_extractor, s.file_pattern = str(tmp_path), 0, kw.get("extractor"), kw.get("pattern", "*.npy")
    s.engine = _FakeEngine()
    return s


def test_compute_anomaly_scores_contract(tmp_path):
    s = _screener(tmp_path)
    assert s.compute_anomaly_scores([]) == {}                                    # improved_detection.py:119-120
    cells = list(synth.synth_crops(1, 0, 5).astype(np.float64))                  # list of (64,64) float64, as :101
    r = s.compute_anomaly_scores(cells)
    assert tuple(r.keys()) == spec.SCORE_KEYS
    assert r["reconstruction_mse"].dtype == np.float32 and r["conservative_scores"].dtype == np.float64
    assert r["conservative_predictions"].dtype == np.int64
    assert set(np.unique(r["conservative_predictions"])) <= {-1, 1}
    assert r["conservative_anomaly_rate"] == np.sum(r["conservative_predictions"] == -1) / 5


def test_screening_driver_skip_rules_and_csv(tmp_path):
    import pandas as pd
    a = tmp_path / "strainA"; b = tmp_path / "strainB"; c = tmp_path / "empty"; d = tmp_path / "zero"
    for p in (a, b, c, d):
        p.mkdir()
    np.save(a / "img2.npy", synth.synth_crops(1, 0, 3))
    np.save(a / "img1.npy", synth.synth_crops(1, 10, 2))
    np.save(b / "x.npy", synth.blob_crops(2, 4))
    np.save(d / "bad.npy", np.zeros((0, 64, 64), np.float32))
    s = _screener(tmp_path)
    out = tmp_path / "out"
    folders = {"B": str(b), "none": str(c), "A": str(a), "Z": str(d)}             # insertion order kept (:164)
    results, detailed = s.screen_mutant_samples(folders, str(out))
    assert list(results) == ["B", "A"]                                            # :168-170, :194-196 skips
    assert results["A"]["total_cells"] == 5 and results["A"]["files_processed"] == 2
    assert [r["cell_id"] for r in detailed] == [0, 1, 2, 3, 0, 1, 2, 3, 4]        # restarts per sample (:217)
    # files are processed in sorted order (:167): img1 then img2
    x = np.concatenate([synth.synth_crops(1, 10, 2), synth.synth_crops(1, 0, 3)])
    ref = _FakeEngine().screen(x)
    assert np.array_equal([r["mse"] for r in detailed[4:]], ref["mse"])
    assert results["A"]["std_mse"] == np.std(ref["mse"])                           # ddof = 0 (:209)
    summ = pd.read_csv(out / "screening_summary.csv", index_col=0)
    assert tuple(summ.columns) == spec.SUMMARY_COLUMNS and list(summ.index) == ["B", "A"]
    det = open(out / "detailed_cell_results.csv").read().splitlines()
    assert det[0] == ",".join(spec.DETAIL_COLUMNS) and len(det) == 10
    assert re.search(r",(True|False),(True|False),", det[1])                       # numpy bools print as True/False
    # same frame built the way the reference builds it gives byte-identical text
    exp = io.StringIO()
    pd.DataFrame(detailed).to_csv(exp, index=False)
    assert exp.getvalue() == open(out / "detailed_cell_results.csv").read()


def test_extractor_errors_are_swallowed_like_the_reference(tmp_path):
    def boom(path):
        raise RuntimeError("bad tiff")
    s = _screener(tmp_path, extractor=boom)
    assert s.extract_quality_cells("whatever.tif") == ([], [])                    # :113-115


def _build_c_demo():
    import subprocess
    exe = os.path.join(H.ROOT, "examples", "screen_demo")
    cmd = ["gcc", "-O2", "-Wall", "-Werror", "-I" + os.path.join(H.ROOT, "include"), os.path.join(H.ROOT, "examples", "screen_demo.c"),
           "-o", exe, "-L" + os.path.join(H.ROOT, "cell-image-analysis_amd"), "-lcellscreen",
           "-Wl,-rpath," + os.path.join(H.ROOT, "cell-image-analysis_amd"), "-lm"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def test_c_abi_is_usable_from_plain_c():
    """include/cellscreen.h compiles as C and links against libcellscreen.so with gcc alone; without a GPU the
    program stops at the no-device check (no CPU path)."""
    import subprocess
    L.load_library()                                   # the .so must exist
    exe = _build_c_demo()
    r = subprocess.run([exe, "/nonexistent", "4"], capture_output=True, text=True)
    if L.load_library().cs_device_count() <= 0:
        assert r.returncode == 3 and "no CPU path" in r.stderr
    else:
        assert r.returncode == 1 and "cs_model_load failed" in r.stderr


def test_sklearn_objects_rebuilt_from_fit_results_answer_like_fitted_ones(tmp_path):
    """The host half of the device detector fit (cellscreen/detector_fit.py): estimators rebuilt from plain arrays
    must behave like scikit-learn-fitted ones, since the reference unpickles and calls them
    (improved_detection.py:32-41, 134-142)."""
    import pickle
    from sklearn.decomposition import PCA
    from sklearn.preprocessing import RobustScaler
    from sklearn.svm import OneClassSVM
    from cellscreen import detector_fit as df
    rng = np.random.default_rng(0)
    x = (rng.normal(size=(300, 6)) @ rng.normal(size=(6, 64)) + 0.1 * rng.normal(size=(300, 64))).astype(np.float32)
    sc = RobustScaler().fit(x)
    sc2 = pickle.loads(pickle.dumps(df._sklearn_scaler(sc.center_, sc.scale_)))
    xs = sc.transform(x)
    assert np.array_equal(sc2.transform(x), xs)
    # principal axes from the scatter matrix == PCA(svd_solver='full'), order and signs included
    full = PCA(n_components=6, svd_solver="full").fit(xs)
    xc = (xs - xs.mean(axis=0)).astype(np.float64)
    comps, ev, total = df.principal_axes(xc.T @ xc, len(x), 6)
    assert np.abs(comps - full.components_).max() < 1e-4
    assert np.allclose(ev, full.explained_variance_, rtol=1e-4)
    p2 = pickle.loads(pickle.dumps(df._sklearn_pca(comps.astype(np.float32), xs.mean(axis=0), ev, total, len(x))))
    assert np.abs(p2.transform(xs) - full.transform(xs)).max() < 1e-3
    assert np.allclose(p2.explained_variance_ratio_, full.explained_variance_ratio_, rtol=1e-3)
    # a OneClassSVM carrying somebody else's solve
    red = full.transform(xs).astype(np.float64)
    det = OneClassSVM(kernel="rbf", gamma="scale", nu=0.1).fit(red)
    alpha = np.zeros(len(red))
    alpha[det.support_] = det.dual_coef_.ravel()
    d2 = pickle.loads(pickle.dumps(df.sklearn_ocsvm(red, alpha, -det.intercept_[0], det._gamma, 0.1, det.n_iter_)))
    y = rng.normal(size=(40, 6)) * 3
    assert np.array_equal(d2.decision_function(y), det.decision_function(y))
    assert np.array_equal(d2.predict(y), det.predict(y)) and np.array_equal(d2.score_samples(y), det.score_samples(y))
    assert np.array_equal(d2.support_, det.support_) and d2.offset_ == det.offset_


def test_training_mirror_validates_detector_fit_choice(tmp_path):
    from cellscreen.training import ImprovedAnomalyDetectionTraining
    assert ImprovedAnomalyDetectionTraining(str(tmp_path)).detector_fit == "device"
    assert ImprovedAnomalyDetectionTraining(str(tmp_path), detector_fit="sklearn").detector_fit == "sklearn"
    with pytest.raises(ValueError):
        ImprovedAnomalyDetectionTraining(str(tmp_path), detector_fit="cpu")
