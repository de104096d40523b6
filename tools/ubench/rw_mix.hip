// What can the memory system do on the prologue's traffic MIX?  T float32 tables read as 16-byte chunks, one float32 row written (16 B per lane), optionally a
// float16 copy written as 8-byte (half4) or 16-byte (half8: two rows' chunks per lane) stores - no arithmetic beyond a sum, grid-stride, 256-thread blocks.
//   hipcc -O3 --offload-arch=gfx950 -o bin/rw_mix rw_mix.hip && ./bin/rw_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));

template <int T, int MODE>   // MODE 0: read only (sum kept alive), 1: + float4 store, 2: + half4 store, 3: + half8 store (lane handles 8 columns)
__global__ __launch_bounds__(256) void mix_kernel(const float* const* tabs, float* out, void* out16, int64_t n4, float* sink) {
    float4 acc_keep = make_float4(0.f, 0.f, 0.f, 0.f);
    if (MODE == 3) {
        for (int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x; 2 * i + 1 < n4; i += int64_t(gridDim.x) * 256) {
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const float4 v = reinterpret_cast<const float4*>(tabs[t])[2 * i], w = reinterpret_cast<const float4*>(tabs[t])[2 * i + 1];
                a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; b.x += w.x; b.y += w.y; b.z += w.z; b.w += w.w;
            }
            reinterpret_cast<float4*>(out)[2 * i] = a;
            reinterpret_cast<float4*>(out)[2 * i + 1] = b;
            reinterpret_cast<half8_t*>(out16)[i] = half8_t{(_Float16)a.x, (_Float16)a.y, (_Float16)a.z, (_Float16)a.w, (_Float16)b.x, (_Float16)b.y, (_Float16)b.z, (_Float16)b.w};
        }
        return;
    }
    for (int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x; i < n4; i += int64_t(gridDim.x) * 256) {
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const float4 v = reinterpret_cast<const float4*>(tabs[t])[i];
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
        if (MODE >= 1) reinterpret_cast<float4*>(out)[i] = a;
        if (MODE == 2) reinterpret_cast<half4_t*>(out16)[i] = half4_t{(_Float16)a.x, (_Float16)a.y, (_Float16)a.z, (_Float16)a.w};
        if (MODE == 0) { acc_keep.x += a.x; acc_keep.y += a.y; acc_keep.z += a.z; acc_keep.w += a.w; }
    }
    if (MODE == 0 && acc_keep.x + acc_keep.y + acc_keep.z + acc_keep.w == 12345.678f) sink[0] = 1.f;
}

template <int T, int MODE>
static void run(const char* name, const float* const* d_tabs, float* out, void* out16, int64_t n4, float* sink, int grid) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((mix_kernel<T, MODE>), dim3(grid), dim3(256), 0, 0, d_tabs, out, out16, n4, sink);
    hipEventRecord(e0);
    const int reps = 10;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((mix_kernel<T, MODE>), dim3(grid), dim3(256), 0, 0, d_tabs, out, out16, n4, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    const double bytes = double(n4) * 16.0 * T + (MODE >= 1 ? double(n4) * 16.0 : 0.0) + (MODE >= 2 ? double(n4) * 8.0 : 0.0);
    printf("%-46s grid %5d  %.3f ms  %.2f TB/s (%.2f GB)\n", name, grid, ms, bytes / ms / 1e9, bytes / 1e9);
}

int main() {
    const int64_t rows = 1254096, d = 128, n4 = rows * d / 4;   // the c4 shape: 642 MB per table
    std::vector<float*> tabs(4);
    for (auto& p : tabs) { hipMalloc(&p, n4 * 16); hipMemset(p, 0, n4 * 16); }
    float *out, *sink; void* out16; const float** d_tabs;
    hipMalloc(&out, n4 * 16); hipMalloc(&out16, n4 * 8); hipMalloc(&sink, 4); hipMalloc(&d_tabs, 4 * sizeof(float*));
    hipMemcpy(d_tabs, tabs.data(), 4 * sizeof(float*), hipMemcpyHostToDevice);
    for (int grid : {1024, 2048, 4096, 16384}) {
        run<2, 0>("T=2 read only", d_tabs, out, out16, n4, sink, grid);
        run<2, 1>("T=2 + float4 row store", d_tabs, out, out16, n4, sink, grid);
        run<2, 2>("T=2 + float4 row + half4 (8 B) copy", d_tabs, out, out16, n4, sink, grid);
        run<2, 3>("T=2 + float4 row + half8 (16 B) copy", d_tabs, out, out16, n4, sink, grid);
        run<4, 1>("T=4 + float4 row store", d_tabs, out, out16, n4, sink, grid);
        run<4, 2>("T=4 + float4 row + half4 (8 B) copy", d_tabs, out, out16, n4, sink, grid);
        run<1, 1>("T=1 copy (float4)", d_tabs, out, out16, n4, sink, grid);
    }
    return 0;
}
