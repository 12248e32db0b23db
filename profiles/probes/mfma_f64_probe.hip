// probe: issue rate of v_mfma_f64_16x16x4_f64 on gfx950 (cycles per MFMA per SIMD), NACC independent
// accumulators per wave, W waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 -o mfma_f64_probe mfma_f64_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ void __launch_bounds__(256) probe(double* out, int iters, double a0, double b0) {
    d4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
    double a = a0 + threadIdx.x, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
void run(int wgs_per_cu, double* out) {
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    probe<NACC><<<256 * wgs_per_cu, 256>>>(out, 10, 1.0, 1.0);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<NACC><<<256 * wgs_per_cu, 256>>>(out, iters, 1.0, 1.0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfma_per_simd = double(iters) * 8 * NACC * wgs_per_cu;   // one wave of each WG per SIMD
    const double tf = mfma_per_simd * 1024 * 2048 / (ms * 1e-3) * 1e-12;
    printf("NACC=%d waves/SIMD=%d: %.3f ms, %.1f ns per MFMA per SIMD (%.0f cycles at 2.4 GHz), %.1f TF\n", NACC, wgs_per_cu, ms,
           ms * 1e6 / mfma_per_simd, ms * 1e6 / mfma_per_simd * 2.4, tf);
}
int main() {
    double* out; hipMalloc(&out, sizeof(double) * 256 * 256 * 8);
    run<1>(1, out); run<2>(1, out); run<3>(1, out); run<4>(1, out); run<1>(2, out); run<1>(3, out); run<4>(2, out);
    return 0;
}
