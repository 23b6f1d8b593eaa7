// pk_fma_rate.hip -- issue rate of v_pk_fma_f32 against v_fma_f32 in a pure-VALU phase (no MFMA beside it), at one and two
// waves per SIMD.  hipcc --offload-arch=gfx950 -O3 -o pk_fma_rate pk_fma_rate.hip ; ./pk_fma_rate -> one JSON line
// (cycles per instruction per wave from s_memtime over the loop; 100 MHz counter scaled by the measured kernel time).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, int iters, float a, float b, unsigned long long* cyc)
{
    const unsigned long long t0 = __builtin_readcyclecounter();
    float x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = threadIdx.x * 0.001f + i;
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x[i]) : "v"(a), "v"(b));
        } else {
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                f32x2 p = {x[i], x[i + 1]};
                const f32x2 av = {a, a}, bv = {b, b};
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p) : "v"(av), "v"(bv));
                x[i] = p[0]; x[i + 1] = p[1];
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
static double run(int threads, int iters, float* d, unsigned long long* dc, double* cycles)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) k<MODE><<<256, threads>>>(d, iters, 1.0f, 0.0f, dc);
    hipEventRecord(e0);
    k<MODE><<<256, threads>>>(d, iters, 1.0f, 0.0f, dc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c = 0;
    hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
    *cycles = (double)c;
    return ms;
}

int main()
{
    float* d;
    hipMalloc(&d, 256 * 512 * 4);
    const int iters = 200000;
    unsigned long long* dc;
    hipMalloc(&dc, 8);
    // per wave and iteration: MODE 0 issues 16 v_fma_f32 (16 values), MODE 1 issues 8 v_pk_fma_f32 (the same 16 values)
    printf("{");
    for (int wps = 1; wps <= 2; ++wps) {
        const int threads = 256 * wps;
        double c0 = 0, c1 = 0;
        const double m0 = run<0>(threads, iters, d, dc, &c0), m1 = run<1>(threads, iters, d, dc, &c1);
        printf("\"waves_per_simd_%d\": {\"ms_16_fma\": %.3f, \"ms_8_pk_fma\": %.3f, \"counter_ticks_per_fma_per_simd\": %.3f, \"counter_ticks_per_pk_fma_per_simd\": %.3f, \"counter_ghz\": %.3f}%s",
               wps, m0, m1, c0 / ((double)iters * 16 * wps), c1 / ((double)iters * 8 * wps), c0 / (m0 * 1e6), wps == 1 ? ", " : "");
    }
    printf("}\n");
    return 0;
}
