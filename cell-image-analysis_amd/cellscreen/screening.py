"""Host-side mirror of the reference's screening class (improved_detection.py:18-261):
same class name, method names, argument meaning, result keys, skip rules and CSV schemas,
with the arithmetic of compute_anomaly_scores executed by libcellscreen on the GPU.

Out of scope here: StarDist cell extraction (:48-115) -- supply a
`cell_extractor(image_path) -> (list_of_64x64_arrays, list_of_stat_dicts)` or pre-extracted
`.npy` crop files; plots and the text report (:263-403)."""
from __future__ import annotations

import os
from glob import glob
from typing import Callable, Dict, List, Optional, Tuple

import numpy as np

from . import spec
from .engine import Engine


class ProductionMutantScreening:
    def __init__(self, model_dir: str, device_id: int = 0,
                 cell_extractor: Optional[Callable[[str], Tuple[list, list]]] = None,
                 file_pattern: str = "*.tif", precision: str = "split16"):
        """precision: "split16" (default) or "fp32_exact" -- how the fp32 contractions run (Engine.from_weights; the reference
        predicts in Keras's float32, improved_detection.py:122,125,130: "fp32_exact" is that arithmetic up to summation order,
        "split16" the same results inside every stated fp32 tolerance at 1.7x the rate)."""
        self.model_dir = model_dir                     # improved_detection.py:20
        self.device_id = device_id
        self.precision = precision
        self.cell_extractor = cell_extractor
        self.file_pattern = file_pattern
        self.load_trained_models()                     # :21

    def load_trained_models(self):
        """:23-46.  Reads the native file set (cae.bin, detector.bin); any failure raises, as
        the reference's uncaught load errors do."""
        print("Loading trained models...")
        self.engine = Engine.from_model_dir(self.model_dir, self.device_id, precision=self.precision)
        if not self.engine.info.has_detector:
            raise FileNotFoundError(f"{self.model_dir}: detector.bin missing (scaler/pca/detector_* of :32-41)")
        print("All models loaded successfully!")

    # ---- cell extraction hook (the reference's :48-115 is out of scope) ----------------
    def extract_quality_cells(self, image_path: str):
        try:
            if self.cell_extractor is not None:
                return self.cell_extractor(image_path)
            if image_path.endswith(".npy"):
                cells = np.load(image_path)
                hw = (self.engine.info.height, self.engine.info.width)      # 64 x 64 for the reference graph
                if cells.ndim != 3 or cells.shape[1:] != hw:
                    raise ValueError(f"expected (N,{hw[0]},{hw[1]}) crops, got {cells.shape}")
                stats = [{"mean_intensity": float(np.mean(c)), "std_intensity": float(np.std(c))} for c in cells]
                return list(cells), stats
            raise NotImplementedError("StarDist cell extraction is out of scope; pass cell_extractor= or use .npy crop files")
        except Exception as e:                          # :113-115
            print(f"Error processing {image_path}: {e}")
            return [], []

    def preprocess_crops(self, raw_crops) -> np.ndarray:
        """:98-99 for a list of raw bounding-box crops (uint8 / uint16, ragged):
        equalize_adapthist(clip_limit=0.02) + resize((64, 64), anti_aliasing=True) on the GPU.
        A `cell_extractor` that segments images can call this instead of scikit-image."""
        if getattr(self, "_preproc", None) is None:
            from .preprocess import Preprocessor
            self._preproc = Preprocessor(self.device_id)
        return self._preproc(raw_crops)

    # ---- the hot path -------------------------------------------------------------------
    def compute_anomaly_scores(self, cell_images) -> Dict:
        """:117-153.  Same keys, dtypes and conventions as the reference's dict."""
        if len(cell_images) == 0:                       # :119-120
            return {}
        X = np.asarray(cell_images).astype("float32")   # :122 (the channel axis of 1 is implicit)
        r = self.engine.screen(X)
        cons_pred = r["cons_pred"].astype(np.int64)     # sklearn predict returns int64
        mod_pred = r["mod_pred"].astype(np.int64)
        return {
            "reconstruction_mse": r["mse"],
            "reconstruction_mae": r["mae"],
            "conservative_predictions": cons_pred,
            "moderate_predictions": mod_pred,
            "conservative_scores": r["cons_score"],      # already -decision_function (:149)
            "moderate_scores": r["mod_score"],
            "conservative_anomaly_rate": np.sum(cons_pred == -1) / len(cons_pred),   # :151
            "moderate_anomaly_rate": np.sum(mod_pred == -1) / len(mod_pred),         # :152
        }

    # ---- per-sample driver ----------------------------------------------------------------
    @staticmethod
    def _sample_rows(sample_name: str, n_files: int, n_cells: int, s: Dict):
        sample_result = {                                # :202-212
            "sample_name": sample_name,
            "total_cells": n_cells,
            "files_processed": n_files,
            "conservative_anomaly_rate": s["conservative_anomaly_rate"],
            "moderate_anomaly_rate": s["moderate_anomaly_rate"],
            "mean_mse": np.mean(s["reconstruction_mse"]),
            "std_mse": np.std(s["reconstruction_mse"]),
            "mean_mae": np.mean(s["reconstruction_mae"]),
            "std_mae": np.std(s["reconstruction_mae"]),
        }
        detailed = []
        for i, (mse, mae, cp, mp, cs, ms) in enumerate(zip(                 # :217-234
                s["reconstruction_mse"], s["reconstruction_mae"], s["conservative_predictions"],
                s["moderate_predictions"], s["conservative_scores"], s["moderate_scores"])):
            detailed.append({"sample_name": sample_name, "cell_id": i, "mse": mse, "mae": mae,
                             "conservative_anomaly": cp == -1, "moderate_anomaly": mp == -1,
                             "conservative_score": cs, "moderate_score": ms})
        return sample_result, detailed

    def screen_mutant_samples(self, test_folders_dict: Dict[str, str], output_dir: str):
        """:155-244.  Samples with no matching files or zero extracted cells are skipped."""
        os.makedirs(output_dir, exist_ok=True)
        print("=== Starting Mutant Screening with Improved Model ===")
        results, detailed_results = {}, []
        for sample_name, folder_path in test_folders_dict.items():
            print(f"\nProcessing {sample_name}...")
            files = sorted(glob(os.path.join(folder_path, self.file_pattern)))     # :167
            if not files:                                                           # :168-170
                print(f"  No {self.file_pattern} files found in {folder_path}")
                continue
            sample_cells: List = []
            for file_path in files:                                                 # :177-190
                cells, _stats = self.extract_quality_cells(file_path)
                sample_cells.extend(cells)
                print(f"  {os.path.basename(file_path)}: {len(cells)} cells")
            print(f"  Total {sample_name} cells: {len(sample_cells)}")
            if len(sample_cells) == 0:                                              # :194-196
                print(f"  No quality cells extracted from {sample_name}")
                continue
            scores = self.compute_anomaly_scores(sample_cells)                      # :199
            sample_result, detailed = self._sample_rows(sample_name, len(files), len(sample_cells), scores)
            results[sample_name] = sample_result
            detailed_results.extend(detailed)
            print(f"    Conservative anomaly rate: {sample_result['conservative_anomaly_rate']*100:.2f}%")
            print(f"    Moderate anomaly rate: {sample_result['moderate_anomaly_rate']*100:.2f}%")
            print(f"    Mean MSE: {sample_result['mean_mse']:.6f}")
        self.save_and_visualize_results(results, detailed_results, output_dir)      # :242
        return results, detailed_results

    def screen_cell_arrays(self, samples: Dict[str, np.ndarray], output_dir: str, files_processed: int = 1):
        """Same outputs as screen_mutant_samples for already-extracted crops per sample."""
        os.makedirs(output_dir, exist_ok=True)
        results, detailed_results = {}, []
        for sample_name, cells in samples.items():
            if len(cells) == 0:
                continue
            scores = self.compute_anomaly_scores(cells)
            sample_result, detailed = self._sample_rows(sample_name, files_processed, len(cells), scores)
            results[sample_name] = sample_result
            detailed_results.extend(detailed)
        self.save_and_visualize_results(results, detailed_results, output_dir)
        return results, detailed_results

    def save_and_visualize_results(self, results, detailed_results, output_dir):
        """:246-261 -- the two CSVs, written by the same pandas calls.  Plots (:258) and the
        text report (:261) are out of scope."""
        write_screening_csvs(results, detailed_results, output_dir)


def write_screening_csvs(results: Dict, detailed_results: List[dict], output_dir: str):
    import pandas as pd
    results_df = pd.DataFrame.from_dict(results, orient="index")                       # :250
    results_df.to_csv(os.path.join(output_dir, "screening_summary.csv"))              # :251
    detailed_df = pd.DataFrame(detailed_results)                                       # :254
    detailed_df.to_csv(os.path.join(output_dir, "detailed_cell_results.csv"), index=False)   # :255
    return results_df, detailed_df
