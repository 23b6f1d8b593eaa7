// fp16_mfma_probe.hip -- what v_mfma_f32_16x16x32_f16 does with the operands a two-term fp16 split would feed it
// (the arithmetic is studied on the CPU by tests/study_split_fp16.py).  One wave, one MFMA per
// experiment, D = A B with C = 0 unless stated; A[i][k] = lane (i = l & 15, k = 8 (l >> 4) + j), B[k][n] likewise with n = l & 15.
//   1  subnormal inputs: A = 2^-20 (fp16 subnormal), B = 2^10: exact result 2^-10 per product -- 0 if inputs are flushed
//   2  subnormal x subnormal: A = B = 2^-20: product 2^-40 (exact in fp32) -- 0 if flushed
//   3  in-instruction sum of unlike magnitudes: k = 0 carries 2^12 * 2^12 = 2^24, k = 1..31 carry 1 * 1: exact sum 2^24 + 31
//   4  the same with the large term 2^30 (A = B = 2^15): 2^30 + 31 is not a float32; what comes back tells how wide the adder is
//   5  many small below one large: 2^24 (k = 0) + 31 * 0.5: exact 2^24 + 15.5 -> RNE 2^24 + 16; a truncating adder gives 2^24 + 15 or 2^24
//   6  accumulate into a large C: C = 2^24, sum of products = 31 * 0.03125: must round once (2^24 + 1 after RNE of 0.96875)
//   7  product exactness: A = 1 + 2^-10 (all bits of fp16), B = 1 + 2^-10, one k only: (1 + 2^-10)^2 = 1 + 2^-9 + 2^-20, exact in fp32
//   8  random fp16 operands in [-2^15, 2^15] scaled as the kernels would: max relative error of D against float64 (one fp32 rounding = 6e-8)
//   9  the same with the hi/lo pair of a split value: hi b + lo b summed in ONE instruction (lo = 2^-11 hi scale) vs in two instructions
// Prints one JSON line.  Build: hipcc --offload-arch=gfx950 -O2 -o fp16_mfma_probe fp16_mfma_probe.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void probe(const _Float16* __restrict__ A, const _Float16* __restrict__ B, const float* __restrict__ C, float* __restrict__ D, int nexp)
{
    const int lane = threadIdx.x;
    for (int e = 0; e < nexp; ++e) {
        // A[e][i][k], B[e][k][n] row-major 16x32 / 32x16; C, D [e][16][16]
        f16x8 a, b;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            a[j] = A[(e * 16 + (lane & 15)) * 32 + 8 * (lane >> 4) + j];
            b[j] = B[(e * 32 + 8 * (lane >> 4) + j) * 16 + (lane & 15)];
        }
        f32x4 c;
#pragma unroll
        for (int r = 0; r < 4; ++r) c[r] = C[(e * 16 + 4 * (lane >> 4) + r) * 16 + (lane & 15)];
        const f32x4 d = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) D[(e * 16 + 4 * (lane >> 4) + r) * 16 + (lane & 15)] = d[r];
    }
}

// f32 -> f16 conversions the split would use: cast (v_cvt_f16_f32, RNE), and what subnormal results it produces
__global__ void cvt_probe(const float* __restrict__ x, _Float16* __restrict__ h, float* __restrict__ back, int n)
{
    const int i = threadIdx.x + blockIdx.x * blockDim.x;
    if (i < n) {
        const _Float16 v = (_Float16)x[i];
        h[i] = v;
        back[i] = x[i] - (float)v;
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

int main()
{
    const int NE = 16;
    std::vector<_Float16> A(NE * 16 * 32, (_Float16)0.0f), B(NE * 32 * 16, (_Float16)0.0f);
    std::vector<float> C(NE * 256, 0.0f), D(NE * 256, 0.0f);
    auto a = [&](int e, int i, int k) -> _Float16& { return A[(e * 16 + i) * 32 + k]; };
    auto b = [&](int e, int k, int n) -> _Float16& { return B[(e * 32 + k) * 16 + n]; };
    for (int i = 0; i < 16; ++i)
        for (int n = 0; n < 16; ++n)
            for (int k = 0; k < 32; ++k) {
                a(1, i, k) = (_Float16)ldexpf(1.0f, -20); b(1, k, n) = (_Float16)1024.0f;
                a(2, i, k) = (_Float16)ldexpf(1.0f, -20); b(2, k, n) = (_Float16)ldexpf(1.0f, -20);
                a(3, i, k) = (_Float16)(k == 0 ? 4096.0f : 1.0f); b(3, k, n) = (_Float16)(k == 0 ? 4096.0f : 1.0f);
                a(4, i, k) = (_Float16)(k == 0 ? 32768.0f : 1.0f); b(4, k, n) = (_Float16)(k == 0 ? 32768.0f : 1.0f);
                a(5, i, k) = (_Float16)(k == 0 ? 4096.0f : 0.5f); b(5, k, n) = (_Float16)(k == 0 ? 4096.0f : 1.0f);
                a(6, i, k) = (_Float16)(k == 0 ? 0.0f : 0.03125f); b(6, k, n) = (_Float16)1.0f;
                a(7, i, k) = (_Float16)(k == 0 ? 1.0f + 1.0f / 1024.0f : 0.0f); b(7, k, n) = (_Float16)(1.0f + 1.0f / 1024.0f);
            }
    for (int i = 0; i < 256; ++i) C[6 * 256 + i] = 16777216.0f;
    // 8: random operands; 9 / 10: a split pair in one instruction against two
    srand(7);
    auto rnd = []() { return (float)rand() / (float)RAND_MAX; };
    std::vector<float> xa(16 * 32), xb(32 * 16);
    for (auto& v : xa) v = (rnd() * 2.0f - 1.0f) * 30000.0f;
    for (auto& v : xb) v = (rnd() * 2.0f - 1.0f) * 30000.0f;
    for (int i = 0; i < 16; ++i)
        for (int k = 0; k < 32; ++k) {
            a(8, i, k) = (_Float16)xa[i * 32 + k];
            // 9: even k = hi of value k/2, odd k = lo (unscaled residual): one instruction sums hi b + lo b
            const float v = xa[i * 32 + k / 2];
            const _Float16 hi = (_Float16)v, lo = (_Float16)(v - (float)hi);
            a(9, i, k) = (k & 1) ? lo : hi;
            // 10 / 11: the same products in two instructions
            a(10, i, k) = k < 16 ? (_Float16)xa[i * 32 + k] : (_Float16)0.0f;
            a(11, i, k) = k < 16 ? (_Float16)(xa[i * 32 + k] - (float)(_Float16)xa[i * 32 + k]) : (_Float16)0.0f;
        }
    for (int k = 0; k < 32; ++k)
        for (int n = 0; n < 16; ++n) {
            b(8, k, n) = (_Float16)xb[k * 16 + n];
            b(9, k, n) = (_Float16)xb[(k / 2) * 16 + n];
            b(10, k, n) = (_Float16)xb[k * 16 + n];
            b(11, k, n) = (_Float16)xb[k * 16 + n];
        }
    _Float16 *dA, *dB; float *dC, *dD;
    CK(hipMalloc(&dA, A.size() * 2)); CK(hipMalloc(&dB, B.size() * 2)); CK(hipMalloc(&dC, C.size() * 4)); CK(hipMalloc(&dD, D.size() * 4));
    CK(hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dC, C.data(), C.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD, NE);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost));
    auto d = [&](int e) { return (double)D[e * 256]; };
    double e8 = 0, e9 = 0, e10 = 0, m8 = 0;
    for (int i = 0; i < 16; ++i)
        for (int n = 0; n < 16; ++n) {
            double r8 = 0, r9 = 0;
            for (int k = 0; k < 32; ++k) {
                r8 += (double)(float)A[(8 * 16 + i) * 32 + k] * (double)(float)B[(8 * 32 + k) * 16 + n];
                r9 += (double)(float)A[(9 * 16 + i) * 32 + k] * (double)(float)B[(9 * 32 + k) * 16 + n];
            }
            m8 = fmax(m8, fabs(r8));
            e8 = fmax(e8, fabs((double)D[(8 * 16 + i) * 16 + n] - r8));
            e9 = fmax(e9, fabs((double)D[(9 * 16 + i) * 16 + n] - r9));
        }
    // 10 + 11 in two instructions cover k < 16 of experiment 8's values: compare with the one-instruction sum of the same 16 values' hi + lo
    for (int i = 0; i < 16; ++i)
        for (int n = 0; n < 16; ++n) {
            double r = 0;
            for (int k = 0; k < 16; ++k)
                r += ((double)(float)A[(10 * 16 + i) * 32 + k] + (double)(float)A[(11 * 16 + i) * 32 + k]) * (double)(float)B[(10 * 32 + k) * 16 + n];
            const double got = (double)D[(10 * 16 + i) * 16 + n] + (double)D[(11 * 16 + i) * 16 + n];
            e10 = fmax(e10, fabs(got - r));
        }
    // conversions
    const int NC = 8;
    const float xs[NC] = {ldexpf(1.0f, -15), ldexpf(1.5f, -20), ldexpf(1.0f, -24), ldexpf(1.0f, -25) * 1.5f, 65504.0f, 65519.0f, 65520.0f, 1.0f + ldexpf(1.0f, -11)};
    float *dx, *dback; _Float16* dh;
    CK(hipMalloc(&dx, NC * 4)); CK(hipMalloc(&dback, NC * 4)); CK(hipMalloc(&dh, NC * 2));
    CK(hipMemcpy(dx, xs, NC * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(cvt_probe, dim3(1), dim3(64), 0, 0, dx, dh, dback, NC);
    CK(hipDeviceSynchronize());
    _Float16 hs[NC]; float backs[NC];
    CK(hipMemcpy(hs, dh, NC * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(backs, dback, NC * 4, hipMemcpyDeviceToHost));
    printf("{\"subnormal_a_times_normal_b\": %.10g, \"expect\": %.10g, \"subnormal_times_subnormal\": %.10g, \"expect2\": %.10g, "
           "\"sum_2p24_plus_31\": %.1f, \"sum_2p30_plus_31_minus_2p30\": %.1f, \"sum_2p24_plus_31_halves_minus_2p24\": %.1f, "
           "\"c_2p24_plus_0p96875_minus_2p24\": %.1f, \"product_1p0009765625_squared_minus_1\": %.12g, \"expect7\": %.12g, "
           "\"random_k32_max_abs_err\": %.6g, \"random_k32_max_abs\": %.6g, \"random_rel\": %.3g, "
           "\"split_pair_one_instruction_max_abs_err\": %.6g, \"split_pair_two_instructions_max_abs_err\": %.6g",
           d(1), 32.0 * ldexp(1.0, -10), d(2), 32.0 * ldexp(1.0, -40), d(3), d(4) - 1073741824.0, d(5) - 16777216.0, d(6) - 16777216.0,
           d(7) - 1.0, ldexp(1.0, -9) + ldexp(1.0, -20), e8, m8, e8 / m8, e9, e10);
    printf(", \"cvt\": [");
    // non-finite values as strings: JSON has no inf
    auto num = [](double v, char* buf) { if (std::isfinite(v)) snprintf(buf, 40, "%.10g", v); else snprintf(buf, 40, "\"%s\"", v > 0 ? "inf" : (v < 0 ? "-inf" : "nan")); return buf; };
    char b1[40], b2[40];
    for (int i = 0; i < NC; ++i) printf("%s{\"x\": %.10g, \"f16\": %s, \"residual\": %s}", i ? ", " : "", xs[i], num((double)(float)hs[i], b1), num(backs[i], b2));
    printf("]}\n");
    return 0;
}
