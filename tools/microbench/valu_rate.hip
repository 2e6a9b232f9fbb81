// valu_rate.hip - VALU issue-rate probe for gfx950 (design input for the sweep kernel).
// Measures wave-instructions per cycle per SIMD for the f32 ops the relaxation uses,
// plain and packed, at 1/2/4/8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));

#define REP8(X) X X X X X X X X
template <int OP>
__global__ void __launch_bounds__(256) probe(float *out, int iters, unsigned long long *cyc)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b = 1.0001f, c = 0.5f;
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
    f2 pb = {b, b};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
        if (OP == 0) { REP8(asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
        if (OP == 1) { REP8(asm volatile("v_min_f32 %0, %0, %8\n v_min_f32 %1, %1, %8\n v_min_f32 %2, %2, %8\n v_min_f32 %3, %3, %8\n v_min_f32 %4, %4, %8\n v_min_f32 %5, %5, %8\n v_min_f32 %6, %6, %8\n v_min_f32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
        if (OP == 2) { REP8(asm volatile("v_min3_f32 %0, %0, %8, %9\n v_min3_f32 %1, %1, %8, %9\n v_min3_f32 %2, %2, %8, %9\n v_min3_f32 %3, %3, %8, %9\n v_min3_f32 %4, %4, %8, %9\n v_min3_f32 %5, %5, %8, %9\n v_min3_f32 %6, %6, %8, %9\n v_min3_f32 %7, %7, %8, %9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));) }
        if (OP == 3) { REP8(asm volatile("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pb));) }
        if (OP == 4) { REP8(asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pb));) }
        if (OP == 5) { REP8(asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
        if (OP == 6) { REP8(asm volatile("v_add_f32 %0, s4, %0\n v_add_f32 %1, s4, %1\n v_add_f32 %2, s4, %2\n v_add_f32 %3, s4, %3\n v_add_f32 %4, s4, %4\n v_add_f32 %5, s4, %5\n v_add_f32 %6, s4, %6\n v_add_f32 %7, s4, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "s4");) }
        if (OP == 7) { REP8(asm volatile("v_add_f32_dpp %0, %8, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %8, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %2, %8, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %8, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %4, %8, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %5, %8, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %6, %8, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %7, %8, %7 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + p4.x + p5.y + p6.x + p7.y;
    out[blockIdx.x * 256 + threadIdx.x] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int OP>
void run(const char *name, int lanes_per_instr)
{
    float *out; unsigned long long *cyc, hc;
    hipMalloc(&out, 256 * 8 * 256 * 4 * sizeof(float)); hipMalloc(&cyc, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    for (int wps : {1, 2, 4, 8}) {      // waves per SIMD = blocks of 256 threads per CU
        int blocks = 256 * wps;
        probe<OP><<<blocks, 256>>>(out, 10, cyc);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        probe<OP><<<blocks, 256>>>(out, iters, cyc);
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
        double instr_per_wave = (double)iters * 64;
        double cyc_per_instr_per_simd = (double)hc / (instr_per_wave * wps);   // wave-instr issue interval seen by one SIMD
        double tlane = (double)blocks * 4 * instr_per_wave * lanes_per_instr / (ms * 1e-3) / 1e12;
        printf("%-12s waves/SIMD %d: %.2f cycles per wave-instr per SIMD (memtime), %.3f ms, %.1f Tlane-ops/s, eff clock %.2f GHz\n",
               name, wps, cyc_per_instr_per_simd, ms, tlane, (double)hc / (ms * 1e-3) / 1e9);
    }
}

int main()
{
    run<0>("v_add_f32", 64);
    run<5>("v_mul_f32", 64);
    run<1>("v_min_f32", 64);
    run<2>("v_min3_f32", 64);
    run<3>("v_pk_add_f32", 128);
    run<4>("v_pk_mul_f32", 128);
    run<6>("v_add sgpr", 64);
    run<7>("v_add dpp", 64);
    return 0;
}
