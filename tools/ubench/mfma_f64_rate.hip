// Micro-benchmark: v_mfma_f64_16x16x4_f64 issue rate on gfx950 (cycles per MFMA per SIMD, TFLOP/s over the chip).
// The guide (MI355X_MICROARCH.md) has no float64 MFMA row; this measures the peak the fit-side kernels are held to.
//   hipcc -O3 --offload-arch=gfx950 mfma_f64_rate.hip -o mfma_f64_rate && ./mfma_f64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k(double* out, int iters) {
    d4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = d4{0.0, 0.0, 0.0, 0.0};
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-6;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC>
void run(int waves_per_simd) {
    const int iters = 4000;
    const int blocks = 256 * waves_per_simd;
    double* out;
    if (hipMalloc(&out, sizeof(double) * blocks * 256) != hipSuccess) { printf("hipMalloc failed\n"); return; }
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<NACC><<<blocks, 256>>>(out, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<NACC><<<blocks, 256>>>(out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double n_mfma = double(blocks) * 4 * iters * NACC;            // wave-level MFMAs
    const double flops = n_mfma * 2.0 * 16 * 16 * 4;
    const double per_simd = double(iters) * NACC * waves_per_simd;      // MFMAs issued by one SIMD
    printf("independent accumulators %2d, waves/SIMD %d: %.3f ms  %.1f TFLOP/s  %.1f ns per MFMA per SIMD (x 2.4 GHz = %.1f cycles)\n", NACC,
           waves_per_simd, ms, flops / ms / 1e9, ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.4);
    hipFree(out);
}

int main() {
    run<1>(1); run<4>(1); run<8>(1); run<16>(1); run<8>(2); run<16>(2);
    return 0;
}
