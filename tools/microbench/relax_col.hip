// relax_col.hip - the STRIP kernel's relax_column() in isolation (registers only),
// with runtime dc masks, to separate code-shape cost from memory-system cost.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int K = 16, CF = 8, W = K + 2 * CF;


// 4 independent packed chains, stage-major, so that no instruction depends on one of the
// previous three (the compiler serialises the chains on one temporary pair + s_nop).
#define PK4(x0, x1, x2, x3, a0, a1, a2, a3, b0, b1, b2, b3, c0, c1, c2, c3, h2)                     \
    asm volatile("v_pk_add_f32 %0, %4, %8\n\tv_pk_add_f32 %1, %5, %9\n\tv_pk_add_f32 %2, %6, %10\n\t"   \
                 "v_pk_add_f32 %3, %7, %11\n\tv_pk_mul_f32 %0, %0, %16\n\tv_pk_mul_f32 %1, %1, %16\n\t"  \
                 "v_pk_mul_f32 %2, %2, %16\n\tv_pk_mul_f32 %3, %3, %16\n\tv_pk_add_f32 %0, %0, %12\n\t"  \
                 "v_pk_add_f32 %1, %1, %13\n\tv_pk_add_f32 %2, %2, %14\n\tv_pk_add_f32 %3, %3, %15"        \
                 : "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3)                                          \
                 : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(b0), "v"(b1), "v"(b2), "v"(b3), "v"(c0),    \
                   "v"(c1), "v"(c2), "v"(c3), "v"(h2))

template <int MODE>
__device__ __forceinline__ void relax_column(unsigned mask, const float *hs_arr, unsigned desc,
                                             const f32x2 (&vc2e)[K / 2], const f32x2 (&vc2o)[K / 2 - 1],
                                             const f32x2 (&vN2)[W / 2], const f32x2 (&tN2)[W / 2], float (&acc)[K])
{
#pragma unroll
    for (int t = 1; t < 2 * CF; t++) {
        if (mask & (1u << t)) {
            float hv;
            if (MODE == 0) { const float hs = __int_as_float(__builtin_amdgcn_readlane((int)desc, 4 + t)); asm volatile("v_mov_b32 %0, %1" : "=v"(hv) : "s"(hs)); }
            else hv = hs_arr[t];
            const f32x2 h2 = {hv, hv};
            if ((t & 1) == 0 && MODE == 2) {
                f32x2 x[K / 2];
                PK4(x[0], x[1], x[2], x[3], vc2e[0], vc2e[1], vc2e[2], vc2e[3], vN2[t / 2], vN2[1 + t / 2], vN2[2 + t / 2], vN2[3 + t / 2], tN2[t / 2], tN2[1 + t / 2], tN2[2 + t / 2], tN2[3 + t / 2], h2);
                PK4(x[4], x[5], x[6], x[7], vc2e[4], vc2e[5], vc2e[6], vc2e[7], vN2[4 + t / 2], vN2[5 + t / 2], vN2[6 + t / 2], vN2[7 + t / 2], tN2[4 + t / 2], tN2[5 + t / 2], tN2[6 + t / 2], tN2[7 + t / 2], h2);
#pragma unroll
                for (int p = 0; p < K / 2; p++) {
                    acc[2 * p] = fminf(acc[2 * p], x[p].x);
                    acc[2 * p + 1] = fminf(acc[2 * p + 1], x[p].y);
                }
            } else if ((t & 1) == 0) {
                f32x2 x[K / 2];
#pragma unroll
                for (int p = 0; p < K / 2; p++) x[p] = vc2e[p] + vN2[p + t / 2];
#pragma unroll
                for (int p = 0; p < K / 2; p++) x[p] = h2 * x[p];
#pragma unroll
                for (int p = 0; p < K / 2; p++) x[p] = x[p] + tN2[p + t / 2];
#pragma unroll
                for (int p = 0; p < K / 2; p++) {
                    acc[2 * p] = fminf(acc[2 * p], x[p].x);
                    acc[2 * p + 1] = fminf(acc[2 * p + 1], x[p].y);
                }
            } else {
                f32x2 x[K / 2 - 1];
                // cell 0: window element t (odd: high half of pair (t-1)/2);
                // cell K-1: window element K-1+t (even: low half)
                float y0 = vc2e[0].x + vN2[(t - 1) / 2].y;
                float y1 = vc2e[K / 2 - 1].y + vN2[(K - 1 + t) / 2].x;
#pragma unroll
                for (int p = 0; p < K / 2 - 1; p++) x[p] = vc2o[p] + vN2[p + (t + 1) / 2];
                y0 = hv * y0;
                y1 = hv * y1;
#pragma unroll
                for (int p = 0; p < K / 2 - 1; p++) x[p] = h2 * x[p];
                y0 = y0 + tN2[(t - 1) / 2].y;
                y1 = y1 + tN2[(K - 1 + t) / 2].x;
#pragma unroll
                for (int p = 0; p < K / 2 - 1; p++) x[p] = x[p] + tN2[p + (t + 1) / 2];
                acc[0] = fminf(acc[0], y0);
                acc[K - 1] = fminf(acc[K - 1], y1);
#pragma unroll
                for (int p = 0; p < K / 2 - 1; p++) {
                    acc[2 * p + 1] = fminf(acc[2 * p + 1], x[p].x);
                    acc[2 * p + 2] = fminf(acc[2 * p + 2], x[p].y);
                }
            }
        }
    }
}

template <int MODE>
__global__ void __launch_bounds__(256, 2) kern(float *out, const unsigned *cols, int ncols, int iters)
{
    const int lane = threadIdx.x & 63;
    float acc[K]; f32x2 vc2e[K / 2], vc2o[K / 2 - 1], vA[W / 2], tA[W / 2];
    for (int q = 0; q < K; q++) acc[q] = 1e30f;
    for (int p = 0; p < K / 2; p++) vc2e[p] = f32x2{0.1f * (lane + p), 0.2f * p + 1};
    for (int p = 0; p < K / 2 - 1; p++) vc2o[p] = f32x2{vc2e[p].y, vc2e[p + 1].x};
    for (int w = 0; w < W / 2; w++) { vA[w] = f32x2{0.01f * (lane + w), 0.02f * w}; tA[w] = f32x2{3.0f * w + lane, 5.0f * w}; }
    float hs_arr[16]; for (int t = 0; t < 16; t++) hs_arr[t] = 0.5f + t;
    for (int it = 0; it < iters; it++) {
        for (int c = 0; c < ncols; c++) {
            const unsigned desc = cols[c * 20 + min(lane, 19)];
            const unsigned mask = (unsigned)__builtin_amdgcn_readlane((int)desc, 1);
            relax_column<MODE>(mask, hs_arr, desc, vc2e, vc2o, vA, tA, acc);
            vA[c & 7] += 1e-7f;
        }
    }
    float r = 0; for (int q = 0; q < K; q++) r += acc[q];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

int main()
{
    // 153 columns with the dc-set sizes of the 818 star: histogram {2:29,3:8,4:36,5:8,6:32,7:8,9:28,13:4}
    std::vector<unsigned> cols; int sizes[] = {2,3,4,5,6,7,9,13}, counts[] = {29,8,36,8,32,8,28,4}; long nrel = 0;
    for (int g = 0; g < 8; g++) for (int n = 0; n < counts[g]; n++) {
        unsigned mask = 0; int placed = 0; for (int t = 8 - sizes[g] / 2; placed < sizes[g]; t += (sizes[g] > 7 ? 1 : 2), placed++) mask |= 1u << (((t - 1) % 15) + 1);
        unsigned d[20] = {0, mask, 0xff, 0}; for (int t = 0; t < 16; t++) { float h = 5.0f + t; d[4 + t] = *(unsigned *)&h; }
        for (int i = 0; i < 20; i++) cols.push_back(d[i]); nrel += __builtin_popcount(mask);
    }
    int ncols = cols.size() / 20; unsigned *dc; hipMalloc(&dc, cols.size() * 4); hipMemcpy(dc, cols.data(), cols.size() * 4, hipMemcpyHostToDevice);
    float *out; hipMalloc(&out, 256 * 8 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    if (getenv("MASK")) { unsigned m = strtoul(getenv("MASK"), 0, 16); nrel = 0; for (int c = 0; c < ncols; c++) { cols[c * 20 + 1] = m; nrel += __builtin_popcount(m); } hipMemcpy(dc, cols.data(), cols.size() * 4, hipMemcpyHostToDevice); }
    printf("columns %d, offsets %ld\n", ncols, nrel);
    for (int mode = 1; mode < 3; mode++) for (int wps : {1, 2}) {
        int blocks = 256 * wps, iters = 20;
        auto k = mode == 2 ? kern<2> : kern<1>;
        k<<<blocks, 256>>>(out, dc, ncols, 2); hipDeviceSynchronize();
        hipEventRecord(e0); k<<<blocks, 256>>>(out, dc, ncols, iters); hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("mode %d (h via %s) waves/SIMD %d: %.2f T relax/s\n", mode, mode ? "VGPR array" : "readlane+v_mov", wps, (double)blocks * 256 * iters * nrel * K / (ms * 1e-3) / 1e12);
    }
    return 0;
}
