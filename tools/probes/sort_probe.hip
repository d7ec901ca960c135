// timing probe: the two in-LDS sorts of common.h on 8192 / 2048 keys (one workgroup of 1024 threads)
#include "../../cet_pick_amd/csrc/common.h"
#include <cstdio>
#include <vector>
#include <algorithm>
template <int MODE, int P>
__global__ __launch_bounds__(1024) void k(unsigned long long* d) {
    __shared__ unsigned long long keys[P];
    for (int i = threadIdx.x; i < P; i += 1024) keys[i] = d[blockIdx.x * P + i];
    __syncthreads();
    if (MODE == 0) block_bitonic_sort_desc(keys, P, threadIdx.x, 1024);
    else block_sort_desc_fast(keys, P, threadIdx.x, 1024);
    for (int i = threadIdx.x; i < P; i += 1024) d[blockIdx.x * P + i] = keys[i];
}
template <int MODE, int P>
void run(const char* name) {
    std::vector<unsigned long long> h(2 * P);
    unsigned long long s = 88172645463325252ull;
    for (auto& v : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = s >> 1; }
    unsigned long long* d;
    hipMalloc(&d, sizeof(unsigned long long) * 2 * P);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int it = 0; it < 6; ++it) {
        hipMemcpy(d, h.data(), sizeof(unsigned long long) * 2 * P, hipMemcpyHostToDevice);
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<MODE, P>), dim3(2), dim3(1024), 0, 0, d);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); best = std::min(best, ms);
    }
    std::vector<unsigned long long> o(2 * P);
    hipMemcpy(o.data(), d, sizeof(unsigned long long) * 2 * P, hipMemcpyDeviceToHost);
    bool ok = true;
    for (int b = 0; b < 2; ++b) {
        std::vector<unsigned long long> ref(h.begin() + b * P, h.begin() + (b + 1) * P);
        std::sort(ref.begin(), ref.end(), std::greater<unsigned long long>());
        for (int i = 0; i < P; ++i) ok &= ref[i] == o[b * P + i];
    }
    printf("%-28s P=%5d  %.1f us  %s\n", name, P, best * 1e3, ok ? "sorted" : "WRONG");
    hipFree(d);
}
int main() {
    run<0, 8192>("block_bitonic_sort_desc"); run<1, 8192>("block_sort_desc_fast");
    run<0, 2048>("block_bitonic_sort_desc"); run<1, 2048>("block_sort_desc_fast");
    run<0, 1024>("block_bitonic_sort_desc"); run<1, 1024>("block_sort_desc_fast");
    return 0;
}
