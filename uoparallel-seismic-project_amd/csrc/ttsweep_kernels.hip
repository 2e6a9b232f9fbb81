// ttsweep_kernels.hip - gfx950 (MI355X) kernels of the travel-time sweep.
//
// Arithmetic contract (bit parity with serial_new/sweep-tt-multistart.c:216,
// :229-246): every candidate is
//     cand = fl( fl( h * fl(v[c] + v[o]) ) + T[o] ),   h = fs.d / 2
// with three separately rounded float operations.  The file is compiled with
// -ffp-contract=off so the multiply and the add are never fused into an FMA.
// Halving is exact, so folding "/ 2.0" into h on the host leaves every bit
// unchanged (fl(d*s)/2 == fl((d/2)*s) for normal floats).
#include "ttsweep_kernels.h"

namespace ttsweep {

// ===========================================================================
// layout conversion / initialisation
// ===========================================================================

__device__ __forceinline__ void padded_coords(const DevLayout &L, long long idx,
                                              int &a, int &b, int &c)
{
    long long pa = idx / L.s0;
    long long rem = idx - pa * L.s0;
    int pb = (int)(rem / L.s1);
    int pc = (int)(rem - (long long)pb * L.s1);
    a = (int)pa - L.lo[0];
    b = pb - L.lo[1];
    c = pc - L.lo[2];
}

__device__ __forceinline__ bool interior(const DevLayout &L, int a, int b, int c)
{
    return (unsigned)a < (unsigned)L.n[0] && (unsigned)b < (unsigned)L.n[1]
        && (unsigned)c < (unsigned)L.n[2];
}

// user flat index (include/floatbox.h:127-129,160: x*ny*nz + y*nz + z) of the
// device-axis cell (a,b,c)
__device__ __forceinline__ long long user_index(const DevLayout &L, int a, int b, int c)
{
    int u[3];
    u[L.perm[0]] = a;
    u[L.perm[1]] = b;
    u[L.perm[2]] = c;
    return ((long long)u[0] * L.un[1] + u[1]) * L.un[2] + u[2];
}

__global__ void __launch_bounds__(256)
pack_kernel(DevLayout L, const float *__restrict__ user, float *__restrict__ padded,
            float halo_value)
{
    long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= L.cells) return;
    int a, b, c;
    padded_coords(L, idx, a, b, c);
    padded[idx] = interior(L, a, b, c) ? user[user_index(L, a, b, c)] : halo_value;
}

__global__ void __launch_bounds__(256)
unpack_kernel(DevLayout L, const float *__restrict__ padded, float *__restrict__ user)
{
    long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= L.cells) return;
    int a, b, c;
    padded_coords(L, idx, a, b, c);
    if (interior(L, a, b, c)) user[user_index(L, a, b, c)] = padded[idx];
}

__global__ void __launch_bounds__(256)
init_tt_kernel(long long cells, float *__restrict__ padded, long long sidx)
{
    long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= cells) return;
    padded[idx] = (idx == sidx) ? 0.0f : __builtin_inff();
}

static inline unsigned blocks_for(long long n, int per) { return (unsigned)((n + per - 1) / per); }

hipError_t launch_pack(const DevLayout &L, const float *user, float *padded,
                       float halo_value, hipStream_t st)
{
    hipLaunchKernelGGL(pack_kernel, dim3(blocks_for(L.cells, 256)), dim3(256), 0, st,
                       L, user, padded, halo_value);
    return hipGetLastError();
}

hipError_t launch_unpack(const DevLayout &L, const float *padded, float *user, hipStream_t st)
{
    hipLaunchKernelGGL(unpack_kernel, dim3(blocks_for(L.cells, 256)), dim3(256), 0, st,
                       L, padded, user);
    return hipGetLastError();
}

hipError_t launch_init_tt(const DevLayout &L, float *padded, long long sidx, hipStream_t st)
{
    hipLaunchKernelGGL(init_tt_kernel, dim3(blocks_for(L.cells, 256)), dim3(256), 0, st,
                       L.cells, padded, sidx);
    return hipGetLastError();
}

// ===========================================================================
// sweep, variant CELL: one thread per cell, pull form, in place
// ===========================================================================
//
// Each thread owns one interior cell c and is the only writer of T[c] during
// the launch.  It reads its neighbours' travel times straight from the volume
// other threads are updating: a read returns either the old or the new value
// of a neighbour, both of which are lengths of real paths, and values only
// ever decrease, so the iteration converges to the same least fixed point as
// the serial sweep whatever the interleaving.
// A pass in which no thread stores leaves `changed` at 0: every read of that
// pass then saw the final values, so the state is the fixed point.
//
// Liveness (serial_new/...:160,:206 exclusive star bound; :219-221 start skip):
// neighbour o = c + e is used iff
//     (flags & PULL_FWD and c != start) or (flags & PULL_REV and o != start).

constexpr int CELL_BX = 64;     // lanes along the stride-1 axis c
constexpr int CELL_BY = 4;      // rows of b per block

__global__ void __launch_bounds__(CELL_BX *CELL_BY)
sweep_cell_kernel(DevLayout L, const float *__restrict__ v,
                  const StartDesc *__restrict__ starts, const int *__restrict__ active,
                  int *__restrict__ changed, const CellEntry *__restrict__ entries,
                  int nentries, int cblocks, int bblocks)
{
    // blockIdx.x -> (active start, a, b-block, c-block), c-block fastest
    unsigned bid = blockIdx.x;
    const int cb = bid % cblocks; bid /= cblocks;
    const int bb = bid % bblocks; bid /= bblocks;
    const int a = bid % L.n[0];   bid /= L.n[0];
    const int s = active[bid];

    const int c = cb * CELL_BX + threadIdx.x;
    const int b = bb * CELL_BY + threadIdx.y;
    const bool inside = (c < L.n[2]) && (b < L.n[1]);

    const StartDesc sd = starts[s];
    float *__restrict__ T = sd.T;

    bool improved = false;
    if (inside) {
        const long long ci = dev_index(L, a, b, c);
        const bool c_is_start = (ci == sd.sidx);
        const float vc = v[ci];
        const float told = T[ci];
        float best = told;
        for (int e = 0; e < nentries; e++) {
            const CellEntry en = entries[e];
            const long long oi = ci + en.delta;
            const bool live = ((en.flags & PULL_FWD) && !c_is_start)
                           || ((en.flags & PULL_REV) && oi != sd.sidx);
            const float sum = vc + v[oi];
            const float delay = en.h * sum;
            const float cand = delay + T[oi];
            if (live && cand < best) best = cand;
        }
        if (best < told) {
            T[ci] = best;
            improved = true;
        }
    }
    // one atomic per wavefront that improved anything (64-lane ballot)
    if (__ballot(improved) != 0ull && (threadIdx.x & 63) == 0) atomicOr(&changed[s], 1);
}

hipError_t launch_sweep_cell(const DevLayout &L, const float *v, const StartDesc *starts,
                             const int *active, int nactive, int *changed,
                             const CellEntry *entries, int nentries, hipStream_t st)
{
    const int cblocks = (L.n[2] + CELL_BX - 1) / CELL_BX;
    const int bblocks = (L.n[1] + CELL_BY - 1) / CELL_BY;
    const long long nblocks = (long long)nactive * L.n[0] * bblocks * cblocks;
    if (nblocks <= 0) return hipSuccess;
    if (nblocks > 0x7fffffffLL) return hipErrorInvalidValue;
    hipLaunchKernelGGL(sweep_cell_kernel, dim3((unsigned)nblocks), dim3(CELL_BX, CELL_BY), 0, st,
                       L, v, starts, active, changed, entries, nentries, cblocks, bblocks);
    return hipGetLastError();
}

} // namespace ttsweep
