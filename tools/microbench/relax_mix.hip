// relax_mix.hip - achievable rate of the relaxation instruction mix on gfx950 for several
// code shapes (serial chain per cell, grouped, packed, min3) at 1..4 waves per SIMD.
// Pure register kernel: no LDS / memory in the loop.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int K = 16;

template <int SHAPE>
__global__ void __launch_bounds__(256) kern(float *out, int iters, float hh)
{
    float acc[K], vc[K], vN[K + 16], tN[K + 16];
    for (int q = 0; q < K; q++) { acc[q] = 1e30f; vc[q] = 0.1f * (threadIdx.x + q); }
    for (int w = 0; w < K + 16; w++) { vN[w] = 0.01f * (threadIdx.x + w); tN[w] = 3.0f * w + threadIdx.x; }
    float h = hh;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int t = 1; t < 16; t += 1) {
            if (SHAPE == 0) {           // serial chain per cell
#pragma unroll
                for (int q = 0; q < K; q++) {
                    float x = vc[q] + vN[q + t]; x = h * x; x = x + tN[q + t];
                    acc[q] = fminf(acc[q], x);
                }
            } else if (SHAPE == 1) {    // groups of 8
#pragma unroll
                for (int q0 = 0; q0 < K; q0 += 8) {
                    float x[8];
#pragma unroll
                    for (int i = 0; i < 8; i++) x[i] = vc[q0 + i] + vN[q0 + i + t];
#pragma unroll
                    for (int i = 0; i < 8; i++) x[i] = h * x[i];
#pragma unroll
                    for (int i = 0; i < 8; i++) x[i] = x[i] + tN[q0 + i + t];
#pragma unroll
                    for (int i = 0; i < 8; i++) acc[q0 + i] = fminf(acc[q0 + i], x[i]);
                }
            } else if (SHAPE == 2) {    // min3: two offsets per min
                if (t + 1 < 16 && (t & 1)) {
#pragma unroll
                    for (int q = 0; q < K; q++) {
                        float x = vc[q] + vN[q + t]; x = h * x; x = x + tN[q + t];
                        float y = vc[q] + vN[q + t + 1]; y = h * y; y = y + tN[q + t + 1];
                        acc[q] = fminf(acc[q], fminf(x, y));
                    }
                } else if (t == 15) {
#pragma unroll
                    for (int q = 0; q < K; q++) {
                        float x = vc[q] + vN[q + t]; x = h * x; x = x + tN[q + t];
                        acc[q] = fminf(acc[q], x);
                    }
                }
            }
            // keep the loop body from being hoisted: perturb an input
            h += 1e-9f;
        }
        vN[it & 15] += 1e-7f;
    }
    float r = 0;
    for (int q = 0; q < K; q++) r += acc[q];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int SHAPE>
__global__ void __launch_bounds__(256) kern_pk(float *out, int iters, float hh)
{
    float acc[K]; f32x2 vce[K / 2], vN2[(K + 16) / 2], tN2[(K + 16) / 2];
    for (int q = 0; q < K; q++) acc[q] = 1e30f;
    for (int p = 0; p < K / 2; p++) vce[p] = f32x2{0.1f * (threadIdx.x + p), 0.2f * p};
    for (int w = 0; w < (K + 16) / 2; w++) { vN2[w] = f32x2{0.01f * (threadIdx.x + w), 0.02f * w}; tN2[w] = f32x2{3.0f * w + threadIdx.x, 5.0f * w}; }
    float h = hh;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int t = 0; t < 16; t += 2) {      // 8 distinct "even" (aligned) offsets
            const f32x2 h2 = {h, h};
            const int tt = t;
#pragma unroll
            for (int p = 0; p < K / 2; p++) {
                f32x2 x = vce[p] + vN2[p + tt / 2];
                x = h2 * x;
                x = x + tN2[p + tt / 2];
                if (SHAPE == 0) { acc[2 * p] = fminf(acc[2 * p], x.x); acc[2 * p + 1] = fminf(acc[2 * p + 1], x.y); }
                else { acc[2 * p] = fminf(acc[2 * p], fminf(x.x, x.y)); }      // half the mins
            }
            h += 1e-9f;
        }
        vN2[it & 7] += 1e-7f;
    }
    float r = 0;
    for (int q = 0; q < K; q++) r += acc[q];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <typename F> void run(const char *name, F kernel, int blocks_per_iter = 15)
{
    float *out; hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 400;
    printf("%-22s", name);
    for (int wps : {1, 2, 3, 4}) {
        int blocks = 256 * wps;
        kernel<<<blocks, 256>>>(out, 4, 0.5f); hipDeviceSynchronize();
        hipEventRecord(e0); kernel<<<blocks, 256>>>(out, iters, 0.5f); hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double relax = (double)blocks * 256 * iters * blocks_per_iter * K;       // lane-relaxations
        printf("  w/SIMD %d: %6.2f T relax/s", wps, relax / (ms * 1e-3) / 1e12);
    }
    printf("\n"); hipFree(out);
}
int main()
{
    run("serial chain", kern<0>); run("grouped x8", kern<1>); run("min3 pairs", kern<2>);
    run("packed", kern_pk<0>, 8); run("packed, half mins", kern_pk<1>, 8);
    return 0;
}
