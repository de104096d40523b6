// Check of the DPP wave scan used by the select kernels' histogram rank search (csrc/hwer_common.h: wave_incl_scan_i32) against a host prefix sum.
//   hipcc -O3 --offload-arch=gfx950 -I../../hybrid-weighted-embedding-recommender_amd/csrc -I../../include scan_test.hip -o bin/scan_test && ./bin/scan_test
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "hwer_common.h"

__global__ void k(const int* in, int* out) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    out[i] = hwer::wave_incl_scan_i32(in[i]);
}

int main() {
    const int W = 1024;
    std::vector<int> h(W * 64), r(W * 64);
    srand(7);
    for (auto& x : h) x = rand() % 1000 - 100;
    int *din, *dout;
    hipMalloc(&din, h.size() * 4);
    hipMalloc(&dout, h.size() * 4);
    hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    k<<<W, 64>>>(din, dout);
    hipMemcpy(r.data(), dout, r.size() * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int w = 0; w < W; ++w) {
        int s = 0;
        for (int l = 0; l < 64; ++l) {
            s += h[w * 64 + l];
            if (r[w * 64 + l] != s && bad++ < 8) printf("wave %d lane %d: got %d want %d\n", w, l, r[w * 64 + l], s);
        }
    }
    printf("wave_incl_scan_i32: %s (%d mismatches)\n", bad ? "FAILED" : "ok", bad);
    return bad ? 1 : 0;
}
