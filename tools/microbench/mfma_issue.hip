// Microbenchmark: issue rate of v_mfma_f32_16x16x4_f32 / 32x32x2 under the operand patterns of
// conv_mfma_kernel (B operand from a different VGPR every instruction, 2 accumulators), to see
// whether the ~88 % matrix-pipe plateau of the conv kernels is an operand-fetch effect.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_issue mfma_issue.hip ; run on an MI355X.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NB, int NACC, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k16(const float* __restrict__ src, float* __restrict__ dst, int iters, long long* cyc)
{
    float B[NB];
#pragma unroll
    for (int s = 0; s < NB; ++s) B[s] = src[s * 64 + (threadIdx.x & 63)];
    f32x4 a0 = *(const f32x4*)(src + threadIdx.x * 4), a1 = *(const f32x4*)(src + 4096 + threadIdx.x * 4);
    f32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < NB; ++s) {
            acc[(2 * s) % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[s & 3], B[s], acc[(2 * s) % NACC], 0, 0, 0);
            acc[(2 * s + 1) % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[s & 3], B[s], acc[(2 * s + 1) % NACC], 0, 0, 0);
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    f32x4 r = acc[0];
#pragma unroll
    for (int i = 1; i < NACC; ++i) r += acc[i];
    dst[blockIdx.x * blockDim.x + threadIdx.x] = r[0] + r[1] + r[2] + r[3];
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NB, int NACC, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k32(const float* __restrict__ src, float* __restrict__ dst, int iters, long long* cyc)
{
    float B[NB];
#pragma unroll
    for (int s = 0; s < NB; ++s) B[s] = src[s * 64 + (threadIdx.x & 63)];
    f32x4 a0 = *(const f32x4*)(src + threadIdx.x * 4), a1 = *(const f32x4*)(src + 4096 + threadIdx.x * 4);
    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.0f;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < NB; ++s) {
            acc[(2 * s) % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[s & 3], B[s], acc[(2 * s) % NACC], 0, 0, 0);
            acc[(2 * s + 1) % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[s & 3], B[s], acc[(2 * s + 1) % NACC], 0, 0, 0);
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float r = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) r += acc[i][j];
    dst[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <class K>
void run(const char* name, K kern, int waves, int nb, int cyc_per_mfma, float* src, float* dst, long long* cyc)
{
    const int iters = 2000, blocks = 256 * (waves >= 4 ? 1 : 4 / waves);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64 * waves), 0, 0, src, dst, 10, cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64 * waves), 0, 0, src, dst, iters, cyc);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(blocks);
    hipMemcpy(h.data(), cyc, blocks * sizeof(long long), hipMemcpyDeviceToHost);
    const double mfma_per_wave = 2.0 * nb * iters;
    const double ticks = (double)h[blocks / 2];                // s_memtime: 100 MHz constant clock
    const double flops = mfma_per_wave * (cyc_per_mfma == 32 ? 2048.0 : 4096.0) * waves * blocks;
    printf("%-44s %8.3f ms  %7.1f TFLOP/s  (%.1f%% of 157.3)  memtime ticks/MFMA %.3f\n", name, ms, flops / ms / 1e9, flops / ms / 1e9 / 157.3 * 100,
           ticks / mfma_per_wave);
}

int main()
{
    float *src, *dst; long long* cyc;
    hipMalloc(&src, 1 << 20); hipMalloc(&dst, 1 << 24); hipMalloc(&cyc, 1 << 16);
    std::vector<float> h(1 << 18);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) >> 8) / 16777216.0f - 0.5f;
    hipMemcpy(src, h.data(), 1 << 20, hipMemcpyHostToDevice);
    // waves per workgroup = waves per CU (1 block per CU): 4 -> one wave per SIMD, 8 -> two, 12 -> three
    run("16x16x4  B[72]  2 acc  1 wave/SIMD", k16<72, 2, 4>, 4, 72, 32, src, dst, cyc);
    run("16x16x4  B[72]  4 acc  1 wave/SIMD", k16<72, 4, 4>, 4, 72, 32, src, dst, cyc);
    run("16x16x4  B[72]  2 acc  2 waves/SIMD", k16<72, 2, 8>, 8, 72, 32, src, dst, cyc);
    run("16x16x4  B[72]  2 acc  3 waves/SIMD", k16<72, 2, 12>, 12, 72, 32, src, dst, cyc);
    run("16x16x4  B[144] 2 acc  2 waves/SIMD", k16<144, 2, 8>, 8, 144, 32, src, dst, cyc);
    run("16x16x4  B[4]   4 acc  1 wave/SIMD", k16<4, 4, 4>, 4, 4, 32, src, dst, cyc);
    run("32x32x2  B[72]  2 acc  1 wave/SIMD", k32<72, 2, 4>, 4, 72, 64, src, dst, cyc);
    run("32x32x2  B[72]  2 acc  2 waves/SIMD", k32<72, 2, 8>, 8, 72, 64, src, dst, cyc);
    run("32x32x2  B[144] 2 acc  1 wave/SIMD", k32<144, 2, 4>, 4, 144, 64, src, dst, cyc);
    return 0;
}
