// Microbenchmark: how fast does one MI355X fetch "tiles" made of scattered 128-byte rows?
// Single-wavefront workgroups (W per CU, set by the dynamic LDS size) each fetch NT tiles of
// NI wave-instructions x 64 lanes x 16 B.  Patterns (what the 8 lanes of a row group and the
// 8 row groups of an instruction point at):
//   0 stream   the tile is 32 KiB of contiguous memory
//   1 rows     256 rows of 128 B, 2 KiB apart (one plane of a 512-float-pitch volume)
//   2 planes   16 planes (2 MiB apart) x 16 rows (2 KiB apart) x 128 B     <- the TILE kernel's shape
//   3 planes, rows 160 B starting 16 B before a line (3 lines per row, 10 lanes per row)
//   4 planes, but the 16 tiles that share a 2 KiB line in pattern 2 are moved to 16 different
//     plane groups: every 2 KiB line of memory is touched by ONE tile (128 B of it) - what a
//     launch over a tile hyperplane I+J+K = D does
//   5 planes, and consecutive tiles (= the tiles in flight at one time) each in a plane group of
//     their own: as many 2 MiB regions live at a time as tiles in flight x 16
//   6 as 5 with the planes of a tile 128 KiB apart instead of 2 MiB (a tile spans 2 MiB)
// mode 0: loads to registers; mode 1: LDS-DMA.
// build: hipcc --offload-arch=gfx950 -O3 -o rowprobe rowprobe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ rsrc_t make_rsrc(const float *p)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p), 0, 0xffffffff, 0x00020000);
}

template <int NI>
__device__ __forceinline__ unsigned lane_offset(int pattern, int i, int lane)
{
    switch (pattern) {
    case 0: return (unsigned)(i * 1024 + lane * 16);
    case 1: { const int r = i * 8 + (lane >> 3); return (unsigned)(r * 2048 + (lane & 7) * 16); }
    case 2: case 4: case 5: { const int r = i * 8 + (lane >> 3);
              return (unsigned)((r >> 4) * (2u << 20) + (r & 15) * 2048 + (lane & 7) * 16); }
    case 6: { const int r = i * 8 + (lane >> 3);
              return (unsigned)((r >> 4) * (128u << 10) + (r & 15) * 2048 + (lane & 7) * 16); }
    default: { const int q = i * 64 + lane, r = q / 10, c = q % 10;          // 160-byte rows
               return (unsigned)((r >> 4) * (2u << 20) + (r & 15) * 2048 + 112 + c * 16); }
    }
}

template <int NI, int MODE>
__global__ __launch_bounds__(64) void probe(const float *buf, long long ntiles, int pattern,
                                            long long tiles_per_row, float *out)
{
    extern __shared__ float lds[];
    const int lane = threadIdx.x;
    f4 acc = {0, 0, 0, 0};
    for (long long t = blockIdx.x; t < ntiles; t += gridDim.x) {
        // tile base: pattern 0 packs tiles; the others lay `tiles_per_row` tiles side by side
        // along the row (128 B each: 16 for a 2 KiB pitch), then step over the rows they cover
        size_t base;
        if (pattern == 0) base = (size_t)t * (NI * 1024);
        else if (pattern == 1) base = (size_t)(t % tiles_per_row) * 128 + (size_t)(t / tiles_per_row) * (NI * 8) * 2048;
        else if (pattern >= 5) {    // group c = t % ngroups first, then (a, b) inside the group
            const size_t pstride = pattern == 5 ? (2u << 20) : (128u << 10);
            const long long per_group = tiles_per_row * (long long)(pstride / (16 * 2048));
            const long long ngroups = ntiles / per_group;
            const long long c = t % ngroups, ab = t / ngroups, a = ab % tiles_per_row, b = ab / tiles_per_row;
            base = (size_t)a * 128 + (size_t)b * 16 * 2048 + (size_t)c * ((NI * 8 + 15) / 16) * pstride;
        }
        else {  // 16 rows per plane per tile; 64 such tile rows fit a 2 MiB plane (2 KiB pitch, 1024 rows)
            const long long a = t % tiles_per_row, b = (t / tiles_per_row) % 64, c = t / (tiles_per_row * 64);
            base = (size_t)a * 128 + (size_t)b * 16 * 2048
                 + (size_t)(pattern == 4 ? c * tiles_per_row + a : c) * ((NI * 8 + 15) / 16) * (2u << 20);
        }
        const rsrc_t r = make_rsrc(buf + base / 4);
        if (MODE == 0) {
            f4 v[NI];
#pragma unroll
            for (int i = 0; i < NI; i++) {
                const unsigned off = lane_offset<NI>(pattern, i, lane);
                v[i] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
            }
#pragma unroll
            for (int i = 0; i < NI; i++) acc += v[i];
        } else {
#pragma unroll
            for (int i = 0; i < NI; i++) {
                const unsigned off = lane_offset<NI>(pattern, i, lane);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void *)(lds + (i & 7) * 256),
                                                         16, (int)off, 0, 0, 0);
            }
            __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
            acc += *(f4 *)(lds + lane * 4);
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[blockIdx.x] = acc.x;
}

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int NI, int MODE>
static void run(const float *buf, size_t bytes, int pattern, int waves_per_cu, float *out)
{
    const int lds = 160 * 1024 / waves_per_cu - 512;       // exactly `waves_per_cu` workgroups fit a CU
    CHK(hipFuncSetAttribute((const void *)probe<NI, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    const long long tiles_per_row = 16;
    long long ntiles;
    if (pattern == 0) ntiles = bytes / (NI * 1024);
    else if (pattern == 1) ntiles = (long long)(bytes / ((size_t)NI * 8 * 2048)) * tiles_per_row;
    else ntiles = (long long)(bytes / ((size_t)((NI * 8 + 15) / 16 + 1) * (2u << 20))) * tiles_per_row * 64;
    if (pattern == 6) ntiles = (long long)(bytes / ((size_t)((NI * 8 + 15) / 16) * (128u << 10))) * tiles_per_row * 4;
    if (pattern == 4) ntiles = ntiles / (tiles_per_row * tiles_per_row * 64) * (tiles_per_row * 64);
    const int grid = 256 * waves_per_cu;
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
        CHK(hipEventRecord(e0));
        hipLaunchKernelGGL((probe<NI, MODE>), dim3(grid), dim3(64), lds, 0, buf, ntiles, pattern, tiles_per_row, out);
        CHK(hipEventRecord(e1));
        CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const double inst = (double)ntiles * NI;
    const char *names[] = {"stream", "rows", "planes", "planes160", "diag", "spread", "spread128k"};
    printf("%-9s mode %d NI %2d waves/CU %d: %8.3f ms  %6.0f GB/s requested  %5.1f cycles per wave-instruction per CU\n",
           names[pattern], MODE, NI, waves_per_cu, best, inst * 1024 / best * 1e-6,
           best * 1e-3 * 2.4e9 / (inst / 256));
    fflush(stdout);
}

int main()
{
    const size_t bytes = 24ull << 30;
    float *buf, *out;
    CHK(hipMalloc(&buf, bytes + (64u << 20)));
    CHK(hipMalloc(&out, 1 << 20));
    CHK(hipMemset(buf, 0, bytes + (64u << 20)));
    for (int pattern : {2, 5, 6}) {
        for (int w : {2, 5, 10}) {
            run<32, 0>(buf, bytes, pattern, w, out);
            run<32, 1>(buf, bytes, pattern, w, out);
        }
        run<8, 0>(buf, bytes, pattern, 5, out);
        run<8, 1>(buf, bytes, pattern, 5, out);
    }
    return 0;
}
