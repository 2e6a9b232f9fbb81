// Microbenchmark: cycles per ds_read_b64 wave-instruction for the pair sweep's address patterns
// (lane (i', j') reads the float2 at row * 32 + 2 m, row = 10 (ci + 1) + (cj + 1) - 1, m = d - i' - j')
// against a linear pattern.  hipcc --offload-arch=gfx950 -O3 -o ldsprobe ldsprobe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ void probe(int pattern, long long *out, float *sink)
{
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = (float)i;
    __syncthreads();
    const int ip = lane >> 3, jp = lane & 7;
    int idx;
    switch (pattern) {
    case 0: idx = 2 * lane; break;                                   // linear float2
    case 1: idx = (10 * (ip + 1) + jp) * 32 + 2 * (14 - ip - jp); break;      // (+,+) ordering
    case 2: idx = (10 * (8 - ip) + jp) * 32 + 2 * (14 - ip - jp); break;      // (-,+)
    case 3: idx = (10 * (ip + 1) + (7 - jp)) * 32 + 2 * (14 - ip - jp); break;  // (+,-)
    default: idx = (10 * (ip + 1) + jp) * 40 + 4 + 2 * (14 - ip - jp); break;  // the 40-float pitch of the general kernel
    }
    f2 acc = {0, 0};
    const long long t0 = clock64();
#pragma unroll 8
    for (int it = 0; it < 4096; it++) {
        acc += *reinterpret_cast<const f2 *>(lds + idx + 2 * (it & 7));
    }
    const long long t1 = clock64();
    if (lane == 0) out[blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = t1 - t0;
    if (acc.x == 12345.f) sink[0] = acc.y;
}
int main()
{
    long long *out; float *sink;
    hipMalloc(&out, 4096 * sizeof(long long)); hipMalloc(&sink, 4);
    for (int waves : {1, 6}) for (int pattern = 0; pattern < 5; pattern++) {
        hipLaunchKernelGGL(probe, dim3(256), dim3(64 * waves), 40000, 0, pattern, out, sink);
        long long h[4096];
        hipMemcpy(h, out, 256 * waves * sizeof(long long), hipMemcpyDeviceToHost);
        double s = 0; for (int i = 0; i < 256 * waves; i++) s += h[i];
        printf("waves per CU %d pattern %d: %.1f clock64 ticks per ds_read_b64 wave-instruction (x%d waves)\n", waves, pattern, s / (256 * waves) / 4096, waves);
    }
    return 0;
}
