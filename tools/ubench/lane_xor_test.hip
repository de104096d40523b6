// Checks the DPP / permlane-swap lane exchanges of csrc/hwer_common.h (lane_xor32<LM>) against __shfl_xor on the GPU:
//   hipcc --offload-arch=gfx950 -O3 -I hybrid-weighted-embedding-recommender_amd/csrc -I include tools/ubench/lane_xor_test.hip -o tools/ubench/bin/lane_xor_test
#include <hip/hip_runtime.h>
#include <cstdio>
#include "hwer_common.h"

__global__ void k(uint32_t* out) {
    const uint32_t v = 1000u + threadIdx.x * 7u;
    out[0 * 64 + threadIdx.x] = hwer::lane_xor32<1>(v);
    out[1 * 64 + threadIdx.x] = hwer::lane_xor32<2>(v);
    out[2 * 64 + threadIdx.x] = hwer::lane_xor32<4>(v);
    out[3 * 64 + threadIdx.x] = hwer::lane_xor32<8>(v);
    out[4 * 64 + threadIdx.x] = hwer::lane_xor32<16>(v);
    out[5 * 64 + threadIdx.x] = hwer::lane_xor32<32>(v);
    uint64_t key[4];
    for (int j = 0; j < 4; ++j) key[j] = (uint64_t((threadIdx.x * 2654435761u + j * 40503u) >> 7) << 20) | (threadIdx.x * 4 + j);
    hwer::wave_sort_u64<4>(key);
    for (int j = 0; j < 4; ++j) reinterpret_cast<uint64_t*>(out + 6 * 64)[threadIdx.x * 4 + j] = key[j];
    out[6 * 64 + 512 + threadIdx.x] = uint32_t(hwer::wave_sum_i32(int(threadIdx.x)));
}

int main() {
    uint32_t* d;
    const size_t n = 6 * 64 + 512 + 64;
    hipMalloc(&d, n * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    uint32_t h[6 * 64 + 512 + 64];
    hipMemcpy(h, d, n * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    const int lm[6] = {1, 2, 4, 8, 16, 32};
    for (int m = 0; m < 6; ++m)
        for (int l = 0; l < 64; ++l)
            if (h[m * 64 + l] != 1000u + uint32_t(l ^ lm[m]) * 7u) { if (bad < 10) printf("xor %d lane %d: got %u\n", lm[m], l, h[m * 64 + l]); ++bad; }
    const uint64_t* s = reinterpret_cast<const uint64_t*>(h + 6 * 64);
    for (int p = 1; p < 256; ++p) if (s[p - 1] > s[p]) { if (bad < 20) printf("sort: position %d out of order\n", p); ++bad; }
    uint64_t chk = 0, ref = 0;
    for (int p = 0; p < 256; ++p) chk += s[p] & 0xFFFFF;
    for (int p = 0; p < 256; ++p) ref += p;
    if (chk != ref) { printf("sort: payload checksum %llu != %llu\n", (unsigned long long)chk, (unsigned long long)ref); ++bad; }
    for (int l = 0; l < 64; ++l) if (h[6 * 64 + 512 + l] != 2016u) { printf("wave_sum lane %d: %u\n", l, h[6 * 64 + 512 + l]); ++bad; }
    printf(bad ? "FAILED (%d)\n" : "lane_xor32 / wave_sort_u64 / wave_sum_i32 OK\n", bad);
    return bad != 0;
}
