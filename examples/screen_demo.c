/* screen_demo.c -- the C ABI of libcellscreen.so used from plain C (no Python, no torch):
 *
 *     gcc -O2 -Iinclude examples/screen_demo.c -o examples/screen_demo \
 *         -Lcell-image-analysis_amd -lcellscreen -Wl,-rpath,'$ORIGIN/../cell-image-analysis_amd' -lm
 *     examples/screen_demo <model_dir> [n_cells]
 *
 * Loads a native model directory (cae.bin + detector.bin, what load_trained_models reads in the reference,
 * improved_detection.py:23-46), makes n_cells deterministic synthetic 64x64 crops on the host, screens them
 * (compute_anomaly_scores, :117-153) and prints the sample-level numbers screen_mutant_samples reports (:196-212).
 * Exit code 0 on success; every failure prints cs_last_error(). */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "cellscreen.h"

static int die(const char *what, int rc)
{
    fprintf(stderr, "%s failed: status %d (%s): %s\n", what, rc, cs_status_string(rc), cs_last_error());
    return 1;
}

int main(int argc, char **argv)
{
    if (argc < 2) {
        fprintf(stderr, "usage: %s <model_dir> [n_cells]\n", argv[0]);
        return 2;
    }
    const int64_t n = argc > 2 ? atoll(argv[2]) : 1000;
    if (cs_abi_version() != CS_ABI_VERSION) return die("cs_abi_version", -1);
    if (cs_device_count() <= 0) {
        fprintf(stderr, "no gfx950 device visible: libcellscreen has no CPU path\n");
        return 3;
    }
    cs_model *m = NULL;
    int rc = cs_model_load(argv[1], 0, NULL /* default options: CS_PRECISION_SPLIT16 */, &m);
    if (rc) return die("cs_model_load", rc);
    cs_model_info info;
    if ((rc = cs_model_get_info(m, &info))) return die("cs_model_get_info", rc);
    printf("model: %dx%d, %d convs (%d encoder), %d features -> %d components, %d + %d support vectors, %s kernels\n",
           info.height, info.width, info.n_conv, info.n_enc, info.feature_dim, info.n_components, info.n_sv_conservative,
           info.n_sv_moderate, info.reference_arch ? "reference-graph" : "generic-shape");

    const size_t npix = (size_t)info.height * info.width;
    float *crops = malloc((size_t)n * npix * sizeof(float));
    float *mse = malloc(n * sizeof(float)), *mae = malloc(n * sizeof(float));
    double *sc = malloc(n * sizeof(double)), *sm = malloc(n * sizeof(double));
    int8_t *pc = malloc(n), *pm = malloc(n);
    if (!crops || !mse || !mae || !sc || !sm || !pc || !pm) return die("malloc", -7);
    for (int64_t i = 0; i < n; ++i)                       /* a blob + ripple per cell, values in [0,1] */
        for (size_t p = 0; p < npix; ++p) {
            const double y = (double)(p / info.width) / info.height - 0.5, x = (double)(p % info.width) / info.width - 0.5;
            const double r2 = x * x + y * y, ph = 0.37 * (double)(i % 97);
            crops[i * npix + p] = (float)(0.1 + 0.8 * exp(-r2 * (6.0 + (double)(i % 7))) * (0.75 + 0.25 * sin(24.0 * x + ph)));
        }

    rc = cs_screen(m, crops, n, CS_MEM_HOST, mse, mae, sc, sm, pc, pm, CS_MEM_HOST);
    if (rc) return die("cs_screen", rc);

    double mean_mse = 0.0, mean_mae = 0.0;
    int64_t an_c = 0, an_m = 0;
    for (int64_t i = 0; i < n; ++i) {
        mean_mse += mse[i]; mean_mae += mae[i];
        an_c += pc[i] == -1; an_m += pm[i] == -1;
        if ((pc[i] != 1 && pc[i] != -1) || (pc[i] == 1) != (-sc[i] > 0.0)) {   /* label = sign rule on the returned score */
            fprintf(stderr, "cell %lld: inconsistent label %d for score %g\n", (long long)i, pc[i], sc[i]);
            return 4;
        }
    }
    printf("total_cells %lld  conservative_anomaly_rate %.4f  moderate_anomaly_rate %.4f  mean_mse %.6f  mean_mae %.6f\n",
           (long long)n, (double)an_c / n, (double)an_m / n, mean_mse / n, mean_mae / n);
    /* empty input is valid and touches nothing (the reference returns {} at :119-120) */
    if ((rc = cs_screen(m, crops, 0, CS_MEM_HOST, mse, mae, sc, sm, pc, pm, CS_MEM_HOST))) return die("cs_screen(n=0)", rc);
    cs_model_free(m);
    free(crops); free(mse); free(mae); free(sc); free(sm); free(pc); free(pm);
    return 0;
}
