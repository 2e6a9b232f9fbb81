// valu_rate2.hip - which VALU ops run at the full (32 lanes/clk) rate on gfx950?
#include <hip/hip_runtime.h>
#include <cstdio>
#define S1(OPSTR, R) OPSTR " %" #R ", %" #R ", %8\n"
#define BODY2(OP) S1(OP,0) S1(OP,1) S1(OP,2) S1(OP,3) S1(OP,4) S1(OP,5) S1(OP,6) S1(OP,7)
#define T1(OPSTR, R) OPSTR " %" #R ", %" #R ", %8, %9\n"
#define BODY3(OP) T1(OP,0) T1(OP,1) T1(OP,2) T1(OP,3) T1(OP,4) T1(OP,5) T1(OP,6) T1(OP,7)
#define REP8(X) X X X X X X X X
#define KERNEL(NAME, BODY)                                                                        \
__global__ void __launch_bounds__(256) NAME(float *out, int iters)                                \
{                                                                                                 \
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5,      \
          a6 = a0 + 6, a7 = a0 + 7, b = 1.0001f, c = 0.5f;                                        \
    for (int i = 0; i < iters; i++) {                                                             \
        REP8(asm volatile(BODY : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5),      \
                          "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");)                          \
    }                                                                                             \
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                  \
}
KERNEL(k_add, BODY2("v_add_f32"))
KERNEL(k_sub, BODY2("v_sub_f32"))
KERNEL(k_fma, BODY3("v_fma_f32"))
KERNEL(k_max, BODY2("v_max_f32"))
KERNEL(k_min, BODY2("v_min_f32"))
KERNEL(k_minu, BODY2("v_min_u32"))
KERNEL(k_mini, BODY2("v_min_i32"))
KERNEL(k_min3u, BODY3("v_min3_u32"))
KERNEL(k_med3, BODY3("v_med3_f32"))
KERNEL(k_and, BODY2("v_and_b32"))
KERNEL(k_addu, BODY2("v_add_u32"))
KERNEL(k_minimum3, BODY3("v_minimum3_f32"))
KERNEL(k_cndmask, BODY2("v_cndmask_b32"))
KERNEL(k_addlit, "v_add_f32 %0, 0x40490fdb, %0\n v_add_f32 %1, 0x40490fdb, %1\n v_add_f32 %2, 0x40490fdb, %2\n v_add_f32 %3, 0x40490fdb, %3\n v_add_f32 %4, 0x40490fdb, %4\n v_add_f32 %5, 0x40490fdb, %5\n v_add_f32 %6, 0x40490fdb, %6\n v_add_f32 %7, 0x40490fdb, %7\n")
KERNEL(k_addinl, "v_add_f32 %0, 1.0, %0\n v_add_f32 %1, 1.0, %1\n v_add_f32 %2, 1.0, %2\n v_add_f32 %3, 1.0, %3\n v_add_f32 %4, 1.0, %4\n v_add_f32 %5, 1.0, %5\n v_add_f32 %6, 1.0, %6\n v_add_f32 %7, 1.0, %7\n")
KERNEL(k_mix, "v_add_f32 %0, %0, %8\n v_min_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_min_f32 %3, %3, %8\n v_add_f32 %4, %4, %8\n v_min_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_min_f32 %7, %7, %8\n")
KERNEL(k_relax, "v_add_f32 %0, %1, %8\n v_mul_f32 %0, %0, %9\n v_add_f32 %0, %0, %2\n v_add_f32 %3, %4, %8\n v_mul_f32 %3, %3, %9\n v_add_f32 %3, %3, %5\n v_min3_f32 %6, %6, %0, %3\n v_add_f32 %7, %7, %8\n")

template <typename K> void run(const char *name, K kern)
{
    float *out; hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    printf("%-14s", name);
    for (int wps : {1, 2, 4, 8}) {
        int blocks = 256 * wps;
        kern<<<blocks, 256>>>(out, 10); hipDeviceSynchronize();
        hipEventRecord(e0); kern<<<blocks, 256>>>(out, iters); hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double winstr = (double)blocks * 4 * iters * 64;        // wave-instructions
        printf("  w/SIMD %d: %6.1f Tlane/s (%.2f clk@2.4GHz per wave-instr per SIMD)", wps,
               winstr * 64 / (ms * 1e-3) / 1e12, (ms * 1e-3) * 2.4e9 / (winstr / 1024.0));
    }
    printf("\n"); hipFree(out);
}
int main()
{
    run("v_add_f32", k_add); run("v_sub_f32", k_sub); run("v_fma_f32", k_fma); run("v_max_f32", k_max);
    run("v_min_f32", k_min); run("v_min_u32", k_minu); run("v_min_i32", k_mini); run("v_min3_u32", k_min3u);
    run("v_med3_f32", k_med3); run("v_and_b32", k_and); run("v_add_u32", k_addu); run("v_minimum3_f32", k_minimum3);
    run("v_cndmask", k_cndmask); run("v_add literal", k_addlit); run("v_add inline", k_addinl);
    run("add/min mix", k_mix); run("relax x2+min3", k_relax);
    return 0;
}
