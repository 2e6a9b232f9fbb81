// What v_permlane16_swap / v_permlane32_swap / row_ror:8 do on gfx950, lane by lane, and a check of
// the "value of lane - 8" sequence built from them.  hipcc --offload-arch=gfx950 -O3 -o permprobe permprobe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ int shift8(int x)        // x of lane - 8 (lanes 0..7: unspecified)
{
    const int lane = threadIdx.x & 63;
    const int b = __builtin_amdgcn_update_dpp(0, x, 0x128, 0xf, 0xf, false);       // row_ror:8
    auto p = __builtin_amdgcn_permlane16_swap(b, b, false, false);                   // p[0] = [B0,B0,B2,B2]? p[1] = [B1,B1,B3,B3]?
    auto q = __builtin_amdgcn_permlane32_swap(p[0], p[1], false, false);
    // candidates for "B of the previous row": rows 1, 3 from p[0]; row 2 from q[0] or q[1]
    const int row = lane >> 4;
    int w = row == 2 ? q[0] : p[0];
    return (lane & 15) >= 8 ? b : w;
}
__global__ void k(int *out)
{
    const int lane = threadIdx.x;
    const int x = lane;
    const int b = __builtin_amdgcn_update_dpp(-1, x, 0x128, 0xf, 0xf, false);
    auto p = __builtin_amdgcn_permlane16_swap(b, b, false, false);
    auto q = __builtin_amdgcn_permlane32_swap(p[0], p[1], false, false);
    out[lane] = b; out[64 + lane] = p[0]; out[128 + lane] = p[1]; out[192 + lane] = q[0]; out[256 + lane] = q[1];
    out[320 + lane] = shift8(x);
    out[384 + lane] = __builtin_amdgcn_update_dpp(-1, x, 0x111, 0xf, 0xf, false);   // row_shr:1
}
int main()
{
    int *d, h[448];
    hipMalloc(&d, sizeof h);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    const char *names[] = {"ror8", "p16[0]", "p16[1]", "q32[0]", "q32[1]", "shift8", "shr1"};
    for (int a = 0; a < 7; a++) {
        printf("%-7s", names[a]);
        for (int l = 0; l < 64; l++) printf(" %2d", h[64 * a + l]);
        printf("\n");
    }
    int bad = 0;
    for (int l = 8; l < 64; l++) bad += h[320 + l] != l - 8;
    printf("shift8 wrong lanes: %d\n", bad);
    return 0;
}
