// relax_static.hip - ceiling of a STRAIGHT-LINE (compile-time offset set) packed relaxation
// block sequence with odd/even pair shifting and min3 pairing, registers only.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int K = 16, CF = 8, W = K + 2 * CF;

template <int T> __device__ __forceinline__ void cand(float hv, const f32x2 (&vce)[K / 2], const f32x2 (&vco)[K / 2 - 1],
                                                      const f32x2 (&vN2)[W / 2], const f32x2 (&tN2)[W / 2], float (&x)[K])
{
    const f32x2 h2 = {hv, hv};
    if ((T & 1) == 0) {
#pragma unroll
        for (int p = 0; p < K / 2; p++) { f32x2 y = vce[p] + vN2[p + T / 2]; y = h2 * y; y = y + tN2[p + T / 2]; x[2 * p] = y.x; x[2 * p + 1] = y.y; }
    } else {
#pragma unroll
        for (int p = 0; p < K / 2 - 1; p++) { f32x2 y = vco[p] + vN2[p + (T + 1) / 2]; y = h2 * y; y = y + tN2[p + (T + 1) / 2]; x[2 * p + 1] = y.x; x[2 * p + 2] = y.y; }
        { float y = vce[0].x + vN2[(T - 1) / 2].y; y = hv * y; x[0] = y + tN2[(T - 1) / 2].y; }
        { float y = vce[K / 2 - 1].y + vN2[(K - 1 + T) / 2].x; y = hv * y; x[K - 1] = y + tN2[(K - 1 + T) / 2].x; }
    }
}

template <int MODE>
__global__ void __launch_bounds__(256, 2) kern(float *out, const float *hin, int iters)
{
    const int lane = threadIdx.x & 63;
    float acc[K]; f32x2 vce[K / 2], vco[K / 2 - 1], vA[W / 2], tA[W / 2];
    for (int q = 0; q < K; q++) acc[q] = 1e30f;
    for (int p = 0; p < K / 2; p++) vce[p] = f32x2{0.1f * (lane + p), 0.2f * p + 1};
    for (int p = 0; p < K / 2 - 1; p++) vco[p] = f32x2{vce[p].y, vce[p + 1].x};
    for (int w = 0; w < W / 2; w++) { vA[w] = f32x2{0.01f * (lane + w), 0.02f * w}; tA[w] = f32x2{3.0f * w + lane, 5.0f * w}; }
    for (int it = 0; it < iters; it++) {
        float h[16];
#pragma unroll
        for (int t = 0; t < 16; t++) h[t] = hin[(it * 16 + t) & 1023];      // runtime lengths (uniform loads)
        // offsets t = 1..14 (14 blocks = 7 min3 pairs)
#define PAIR(TA, TB) { float xa[K], xb[K]; cand<TA>(h[TA], vce, vco, vA, tA, xa); cand<TB>(h[TB], vce, vco, vA, tA, xb); \
        _Pragma("unroll") for (int q = 0; q < K; q++) { if (MODE == 0) acc[q] = fminf(acc[q], fminf(xa[q], xb[q])); else { acc[q] = fminf(acc[q], xa[q]); acc[q] = fminf(acc[q], xb[q]); } } }
        PAIR(1, 2) PAIR(3, 4) PAIR(5, 6) PAIR(7, 8) PAIR(9, 10) PAIR(11, 12) PAIR(13, 14)
        vA[it & 7] += 1e-7f;
    }
    float r = 0; for (int q = 0; q < K; q++) r += acc[q];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

int main()
{
    float *out, *hin; hipMalloc(&out, 256 * 8 * 256 * 4); hipMalloc(&hin, 4096);
    float hh[1024]; for (int i = 0; i < 1024; i++) hh[i] = 5.0f + 0.01f * i; hipMemcpy(hin, hh, 4096, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 2; mode++) for (int wps : {1, 2, 3}) {
        int blocks = 256 * wps, iters = 300;
        auto k = mode ? kern<1> : kern<0>;
        k<<<blocks, 256>>>(out, hin, 2); hipDeviceSynchronize();
        hipEventRecord(e0); k<<<blocks, 256>>>(out, hin, iters); hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("static packed, %s, waves/SIMD %d: %.2f T relax/s\n", mode ? "plain min" : "min3 pairs", wps, (double)blocks * 256 * iters * 14 * K / (ms * 1e-3) / 1e12);
    }
    return 0;
}
