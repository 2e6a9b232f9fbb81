// placeprobe.hip - where do the single-wavefront workgroups of a persistent grid land?
// hipcc --offload-arch=gfx950 -O3 -o gpurun_exp/placeprobe tools/exp/placeprobe.hip && gpurun_exp/placeprobe
// 1536 workgroups of 64 threads with the tile kernel's LDS footprint (26 784 B: six per CU) record
// HW_REG_HW_ID (wave slot, SIMD, CU, SH, SE) and HW_REG_XCC_ID; five launches in a row.
// Questions: is blockIdx -> SIMD the same in every launch?  Which workgroups share a SIMD?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>

__global__ void __launch_bounds__(64) probe(unsigned *out, int spin)
{
    extern __shared__ float lds[];
    const unsigned hw = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 4);     // HW_REG_HW_ID
    const unsigned xcc = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20) & 0xfu;
    float x = threadIdx.x;
    for (int i = 0; i < spin; i++) x = x * 1.0001f + 0.5f;     // stay resident until everybody has started
    lds[threadIdx.x] = x;
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc | (lds[0] > 1e30f ? 16u : 0u); }
}

int main()
{
    const int nb = 1536, lds = 26784;
    unsigned *d; hipMalloc(&d, 2 * nb * sizeof(unsigned));
    hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    std::vector<std::vector<unsigned>> runs;
    for (int r = 0; r < 5; r++) {
        hipLaunchKernelGGL(probe, dim3(nb), dim3(64), lds, 0, d, 20000);
        std::vector<unsigned> h(2 * nb);
        hipMemcpy(h.data(), d, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost);
        runs.push_back(h);
    }
    auto simd = [](unsigned hw) { return (hw >> 4) & 3; };
    auto cuid = [](unsigned hw, unsigned xcc) { return (xcc & 15) << 16 | ((hw >> 8) & 0xff) << 4 | 0; };  // xcc, (cu, sh, se bits 8..15)
    for (int r = 0; r < 5; r++) {
        std::map<unsigned, std::vector<int>> per_cu;   // cu -> waves per simd
        int same_simd = 0, same_cu = 0, xcc_mod = 0;
        for (int b = 0; b < nb; b++) {
            const unsigned hw = runs[r][2 * b], xcc = runs[r][2 * b + 1] & 15;
            auto &v = per_cu[cuid(hw, xcc)]; v.resize(4); v[simd(hw)]++;
            same_simd += simd(hw) == simd(runs[0][2 * b]);
            same_cu += cuid(hw, xcc) == cuid(runs[0][2 * b], runs[0][2 * b + 1] & 15);
            xcc_mod += xcc == (runs[r][1] & 15) + 0 ? 0 : 0;
        }
        std::map<std::vector<int>, int> shapes;
        for (auto &kv : per_cu) shapes[kv.second]++;
        printf("launch %d: %zu CUs used; same SIMD as launch 0: %d / %d, same CU: %d; waves per SIMD shapes:", r, per_cu.size(), same_simd, nb, same_cu);
        for (auto &kv : shapes) printf(" [%d %d %d %d]x%d", kv.first[0], kv.first[1], kv.first[2], kv.first[3], kv.second);
        printf("\n");
    }
    // blockIdx -> (xcc, simd, wave slot) for the first 48 blocks of launch 0 and the rule b -> xcc = b % 8
    int xcc_rule = 0;
    for (int b = 0; b < nb; b++) xcc_rule += ((runs[0][2 * b + 1] & 15) == ((runs[0][1] & 15) + b) % 8);
    printf("xcc == (xcc(0) + b) %% 8 for %d / %d blocks\n", xcc_rule, nb);
    for (int b = 0; b < 64; b++) {
        const unsigned hw = runs[0][2 * b];
        printf("b %3d xcc %u se %u sh %u cu %2u simd %u slot %u%s", b, runs[0][2 * b + 1] & 15, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15, simd(hw), hw & 15, b % 4 == 3 ? "\n" : "   ");
    }
    // per block index within its XCD (j = b / 8): simd as a function of j
    printf("simd by j = b / 8 (xcc of block 0), launch 0:\n");
    for (int j = 0; j < 192; j++) printf("%u%s", simd(runs[0][2 * (8 * j)]), j % 32 == 31 ? "\n" : "");
    return 0;
}
