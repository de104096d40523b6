// Micro-benchmark (gfx950): do MFMA bursts of one wave and VALU work of ANOTHER wave of the same SIMD overlap?
//   hipcc -O3 --offload-arch=gfx950 mfma_valu_overlap.hip -o bin/mfma_valu_overlap && bin/mfma_valu_overlap
// One 512-thread workgroup per CU = 2 waves per SIMD (waves w and w + 4 share SIMD w & 3).  Roles per launch:
//   MODE 0: both waves of a SIMD issue MFMAs            MODE 1: both issue VALU
//   MODE 2: wave w MFMA, wave w + 4 VALU (the ping-pong the streaming screen wants)
//   MODE 3: every wave alternates 16 MFMAs and NV VALU itself (what one wave of the screen does)
// Reports cycles per "unit" = 16 MFMA 32x32x16 f16 (512 pipe cycles) and / or NV VALU instructions.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));

template <int NV>
__device__ __forceinline__ void valu_block(float (&f)[8], float c) {
#pragma unroll
    for (int i = 0; i < NV / 8; ++i) {
        asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                     "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8"
                     : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]) : "v"(c));
    }
}

__device__ __forceinline__ void mfma_block(float16v (&acc)[2], half8 a, half8 b) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[1], 0, 0, 0);
    }
}

template <int MODE, int NV>
__global__ __launch_bounds__(512) void k(float* out, const float* in, int iters, unsigned long long* stamps) {
    const int wave = threadIdx.x >> 6;
    float f[8];
    for (int i = 0; i < 8; ++i) f[i] = in[(threadIdx.x + i) & 255];
    const float c = in[0];
    half8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = _Float16(in[i]); b[i] = _Float16(in[8 + i]); }
    float16v acc[2];
    for (int i = 0; i < 16; ++i) { acc[0][i] = 0.f; acc[1][i] = 1.f; }
    //   MODE 4: wave w MFMA, wave w + 4 exits      MODE 5: wave w exits, wave w + 4 VALU      (one wave per SIMD: the single-wave rates)
    //   MODE 6: every wave interleaves: one MFMA, then NV / 16 VALU, 16 times (VALU in the shadow of the wave's own MFMAs)
    const bool do_mfma = MODE == 0 || MODE == 3 || ((MODE == 2 || MODE == 4) && wave < 4);
    const bool do_valu = MODE == 1 || MODE == 3 || ((MODE == 2 || MODE == 5) && wave >= 4);
    unsigned long long t0 = 0, t1 = 0, r0 = 0, r1 = 0;
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    r0 = __builtin_amdgcn_s_memrealtime();
    if (MODE == 6) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[0], 0, 0, 0);
                valu_block<(NV / 16 + 7) / 8 * 8>(f, c);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[1], 0, 0, 0);
                valu_block<(NV / 16 + 7) / 8 * 8>(f, c);
            }
        }
    } else {
        for (int it = 0; it < iters; ++it) {
            if (do_mfma) mfma_block(acc, a, b);
            if (do_valu) valu_block<NV>(f, c);
        }
    }
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    r1 = __builtin_amdgcn_s_memrealtime();
    if (blockIdx.x == 7 && (threadIdx.x & 63) == 0 && (do_mfma || do_valu || MODE == 6)) { stamps[2 * wave] = t1 - t0; stamps[2 * wave + 1] = r1 - r0; }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += f[i];
    for (int i = 0; i < 16; ++i) s += acc[0][i] + acc[1][i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int MODE, int NV>
void run(const char* name) {
    const int iters = 4000, blocks = 256;
    float *out, *in;
    hipMalloc(&out, sizeof(float) * blocks * 512);
    hipMalloc(&in, sizeof(float) * 256);
    hipMemset(in, 0, sizeof(float) * 256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    unsigned long long* stamps; hipMalloc(&stamps, 16 * 8); hipMemset(stamps, 0, 16 * 8);
    k<MODE, NV><<<blocks, 512>>>(out, in, 50, stamps);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE, NV><<<blocks, 512>>>(out, in, iters, stamps);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[16]; hipMemcpy(h, stamps, sizeof(h), hipMemcpyDeviceToHost);
    // in-kernel clocks of two waves of one SIMD (w = 0 and w = 4): s_memtime ticks and 100 MHz real-time ticks
    double cyc[2], us[2];
    for (int j = 0; j < 2; ++j) { cyc[j] = double(h[2 * (4 * j)]) / iters; us[j] = double(h[2 * (4 * j) + 1]) / 100.0; }
    printf("%-52s NV=%3d  %.3f ms | wave 0: %5.0f ticks/iter (%.0f MHz)  wave 4: %5.0f ticks/iter (%.0f MHz)\n", name, NV, ms,
           cyc[0], us[0] > 0 ? cyc[0] * iters / us[0] : 0.0, cyc[1], us[1] > 0 ? cyc[1] * iters / us[1] : 0.0);
    hipFree(out); hipFree(in); hipFree(stamps);
}

int main() {
    run<0, 128>("both waves: 16 MFMA each");
    run<1, 128>("both waves: NV VALU each");
    run<2, 128>("wave w: 16 MFMA | wave w+4: NV VALU");
    run<3, 128>("both waves: 16 MFMA then NV VALU (the screen's shape)");
    run<1, 256>("both waves: NV VALU each");
    run<2, 256>("wave w: 16 MFMA | wave w+4: NV VALU");
    run<3, 256>("both waves: 16 MFMA then NV VALU");
    run<3, 64>("both waves: 16 MFMA then NV VALU");
    run<4, 128>("one wave per SIMD: 16 MFMA");
    run<5, 128>("one wave per SIMD: NV VALU");
    run<5, 256>("one wave per SIMD: NV VALU");
    run<6, 128>("both waves: MFMA, NV/16 VALU, MFMA, ... (interleaved)");
    run<6, 256>("both waves: MFMA, NV/16 VALU, MFMA, ... (interleaved)");
    return 0;
}
