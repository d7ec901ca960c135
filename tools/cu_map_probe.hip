// Which CU does workgroup L of a 1-D launch land on?  (256 threads, 32 KiB LDS like conv_igemm_kernel)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
__global__ __launch_bounds__(256) void probe(unsigned* out, int spin) {
    __shared__ float lds[8192];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);     // HW_REG_HW_ID
        unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);   // HW_REG_XCC_ID
        out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc;
    }
    float a = lds[threadIdx.x];
    for (int i = 0; i < spin; ++i) a = a * 1.0001f + 0.5f;      // stay resident so that all blocks co-reside
    if (a == 12345.f) out[0] = 0;
}
int main(int argc, char** argv) {
    int blocks = argc > 1 ? atoi(argv[1]) : 512;
    unsigned* d; hipMalloc(&d, 8 * blocks);
    hipLaunchKernelGGL(probe, dim3(blocks), dim3(256), 0, 0, d, 200000);
    hipDeviceSynchronize();
    std::vector<unsigned> h(2 * blocks);
    hipMemcpy(h.data(), d, 8 * blocks, hipMemcpyDeviceToHost);
    std::map<unsigned, std::vector<int>> cu;
    for (int b = 0; b < blocks; ++b) {
        unsigned hw = h[2 * b], xcc = h[2 * b + 1] & 0xf;
        unsigned cu_id = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        cu[(xcc << 12) | (se << 8) | (sh << 4) | cu_id].push_back(b);
    }
    printf("distinct CUs: %zu\n", cu.size());
    int shown = 0;
    for (auto& kv : cu) { if (shown++ < 24) { printf("cu %05x:", kv.first); for (int b : kv.second) printf(" %d", b); printf("\n"); } }
    std::map<int, int> hist, diff;
    for (auto& kv : cu) { hist[(int)kv.second.size()]++; if (kv.second.size() == 2) diff[kv.second[1] - kv.second[0]]++; }
    for (auto& kv : hist) printf("blocks/CU %d: %d CUs\n", kv.first, kv.second);
    for (auto& kv : diff) printf("pair distance %d: %d\n", kv.first, kv.second);
    return 0;
}
