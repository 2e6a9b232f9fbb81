// ttsweep_kernels.hip - gfx950 (MI355X) kernels of the travel-time sweep.
//
// Arithmetic contract (bit parity with serial_new/sweep-tt-multistart.c:216,
// :229-246): every candidate is
//     cand = fl( fl( h * fl(v[c] + v[o]) ) + T[o] ),   h = fs.d / 2
// with three separately rounded float operations.  The file is compiled with
// -ffp-contract=off so the multiply and the add are never fused into an FMA.
// Halving is exact, so folding "/ 2.0" into h on the host leaves every bit
// unchanged (fl(d*s)/2 == fl((d/2)*s)) - for NORMAL products d*s: a denormal product
// is rounded at denormal precision first and halving it rounds again.  The boundary
// therefore refuses velocity volumes in which a product could be denormal
// (ttsweep_set_velocity*, count_bad_velocity_kernel).
#include "ttsweep_kernels.h"
#include <cstdio>

#include <algorithm>
#include <cstdlib>
#include <cstring>

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

// The same copies, four cells per thread, for layouts whose stride-1 axis is the user's z
// and whose rows are 16-byte aligned on both sides (the usual case): whole float4 of a padded
// row lie either inside or outside the interior.
__global__ void __launch_bounds__(256)
pack_rows_kernel(DevLayout L, const float *__restrict__ user, float *__restrict__ padded, float halo_value)
{
    const unsigned idx = blockIdx.x * 256u + threadIdx.x;       // float4 of the padded volume
    const unsigned qrow = (unsigned)(L.s1 >> 2), qplane = (unsigned)(L.s0 >> 2);
    if (idx >= (unsigned)(L.cells >> 2)) return;
    const unsigned pa = idx / qplane, rem = idx - pa * qplane, pb = rem / qrow, pq = rem - pb * qrow;
    const int a = (int)pa - L.lo[0], b = (int)pb - L.lo[1], c = (int)(4 * pq) - L.lo[2];
    float4 val = make_float4(halo_value, halo_value, halo_value, halo_value);
    if (interior(L, a, b, c)) val = *reinterpret_cast<const float4 *>(user + user_index(L, a, b, c));
    reinterpret_cast<float4 *>(padded)[idx] = val;
}

__global__ void __launch_bounds__(256)
unpack_rows_kernel(DevLayout L, const float *__restrict__ padded, float *__restrict__ user)
{
    const unsigned idx = blockIdx.x * 256u + threadIdx.x;       // float4 of the interior
    const unsigned nq = (unsigned)L.n[2] >> 2;
    if (idx >= (unsigned)L.n[0] * (unsigned)L.n[1] * nq) return;
    const unsigned row = idx / nq, q = idx - row * nq;
    const int a = (int)(row / (unsigned)L.n[1]), b = (int)(row - (unsigned)a * (unsigned)L.n[1]), c = (int)(4 * q);
    *reinterpret_cast<float4 *>(user + user_index(L, a, b, c)) =
        *reinterpret_cast<const float4 *>(padded + dev_index(L, a, b, c));
}

// The same for every start of a solve in ONE launch (blockIdx.y = start): a launch per start and kernel
// was 1.2 ms of small, serialised kernels around the 24-start solve.
__global__ void __launch_bounds__(256)
init_tt_batch_kernel(long long cells, float *__restrict__ T0, const StartDesc *__restrict__ starts)
{
    const long long idx4 = (long long)blockIdx.x * 256 + threadIdx.x;     // float4 of a padded volume
    if (4 * idx4 >= cells) return;
    const long long sidx = starts[blockIdx.y].sidx;
    const float inf = __builtin_inff();
    float4 val = make_float4(inf, inf, inf, inf);
    if ((sidx >> 2) == idx4) reinterpret_cast<float *>(&val)[sidx & 3] = 0.0f;
    reinterpret_cast<float4 *>(T0 + (long long)blockIdx.y * cells)[idx4] = val;
}

__global__ void __launch_bounds__(256)
unpack_batch_kernel(DevLayout L, const float *__restrict__ padded0, float *const *__restrict__ users)
{
    long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= L.cells) return;
    int a, b, c;
    padded_coords(L, idx, a, b, c);
    if (interior(L, a, b, c)) users[blockIdx.y][user_index(L, a, b, c)] = padded0[(long long)blockIdx.y * L.cells + idx];
}

static inline unsigned blocks_for(long long n, int per) { return (unsigned)((n + per - 1) / per); }

hipError_t launch_init_tt_batch(const DevLayout &L, float *T0, const StartDesc *starts, int nstart, hipStream_t st)
{
    if (nstart <= 0 || L.cells % 4 != 0 || nstart > 65535) return hipErrorInvalidValue;
    hipLaunchKernelGGL(init_tt_batch_kernel, dim3(blocks_for(L.cells / 4, 256), nstart), dim3(256), 0, st, L.cells, T0, starts);
    return hipGetLastError();
}

hipError_t launch_unpack_batch(const DevLayout &L, const float *padded0, float *const *users, int nstart, hipStream_t st)
{
    if (nstart <= 0 || nstart > 65535) return hipErrorInvalidValue;
    hipLaunchKernelGGL(unpack_batch_kernel, dim3(blocks_for(L.cells, 256), nstart), dim3(256), 0, st, L, padded0, users);
    return hipGetLastError();
}

static bool rows_vectorisable(const DevLayout &L, const void *user, const void *padded)
{
    return L.perm[2] == 2 && L.n[2] % 4 == 0 && L.lo[2] % 4 == 0 && L.s0 % 4 == 0 && L.s1 % 4 == 0
        && L.cells % 4 == 0 && L.cells / 4 < 0x7fffffffLL
        && (reinterpret_cast<uintptr_t>(user) & 15u) == 0 && (reinterpret_cast<uintptr_t>(padded) & 15u) == 0;
}

hipError_t launch_pack(const DevLayout &L, const float *user, float *padded,
                       float halo_value, hipStream_t st)
{
    if (rows_vectorisable(L, user, padded))
        hipLaunchKernelGGL(pack_rows_kernel, dim3(blocks_for(L.cells / 4, 256)), dim3(256), 0, st,
                           L, user, padded, halo_value);
    else
        hipLaunchKernelGGL(pack_kernel, dim3(blocks_for(L.cells, 256)), dim3(256), 0, st,
                           L, user, padded, halo_value);
    return hipGetLastError();
}

hipError_t launch_unpack(const DevLayout &L, const float *padded, float *user, hipStream_t st)
{
    if (rows_vectorisable(L, user, padded))
        hipLaunchKernelGGL(unpack_rows_kernel, dim3(blocks_for((long long)L.n[0] * L.n[1] * (L.n[2] / 4), 256)),
                           dim3(256), 0, st, L, padded, user);
    else
        hipLaunchKernelGGL(unpack_kernel, dim3(blocks_for(L.cells, 256)), dim3(256), 0, st,
                           L, padded, user);
    return hipGetLastError();
}

// ... four cells per thread
__global__ void __launch_bounds__(256)
init_tt4_kernel(long long cells, float *__restrict__ padded, long long sidx)
{
    const long long idx4 = (long long)blockIdx.x * 256 + threadIdx.x;
    if (4 * idx4 >= cells) return;
    const float inf = __builtin_inff();
    float4 val = make_float4(inf, inf, inf, inf);
    if ((sidx >> 2) == idx4) reinterpret_cast<float *>(&val)[sidx & 3] = 0.0f;
    reinterpret_cast<float4 *>(padded)[idx4] = val;
}

hipError_t launch_init_tt(const DevLayout &L, float *padded, long long sidx, hipStream_t st)
{
    if (L.cells % 4 == 0 && (reinterpret_cast<uintptr_t>(padded) & 15u) == 0)
        hipLaunchKernelGGL(init_tt4_kernel, dim3(blocks_for(L.cells / 4, 256)), dim3(256), 0, st, L.cells, padded, sidx);
    else
        hipLaunchKernelGGL(init_tt_kernel, dim3(blocks_for(L.cells, 256)), dim3(256), 0, st, L.cells, padded, sidx);
    return hipGetLastError();
}

// ===========================================================================
// XCD census
// ===========================================================================
// Which XCD (accelerator complex die, own L2) a wave runs on: HW_REG_XCC_ID (id 20), bits 3:0.
__device__ __forceinline__ unsigned xcc_id()
{
    return __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20) & 0xfu;
}

// Every workgroup reports the XCD it landed on; the host counts the distinct ids (the
// number of per-XCD unit queues follows the device instead of a hard-coded 8).
__global__ void __launch_bounds__(64) xcc_census_kernel(unsigned *__restrict__ seen)
{
    if (threadIdx.x == 0) atomicOr(seen, 1u << xcc_id());
}

hipError_t launch_xcc_census(unsigned *seen, int nblocks, hipStream_t st)
{
    hipLaunchKernelGGL(xcc_census_kernel, dim3(nblocks), dim3(64), 0, st, seen);
    return hipGetLastError();
}

// Cells of the caller's velocity volume the solver refuses: negative, infinite or NaN (the
// relaxation needs delays >= 0; a NaN would also defeat the kernels'
// NaN-free arithmetic mode), and positive values below `tiny_bits` (as a bit pattern): so small
// that a delay fl(d * (v[c] + v[o])) could fall into the denormal range, where halving is no
// longer exact - the reference halves the rounded product (serial_new/...:216: "/ 2.0"), the
// kernels multiply by d/2, and the two agree bit for bit only while the product is a normal
// number (or zero).  Zero - either sign - is accepted as the reference accepts it.  Integer
// tests on the bit pattern: this file is compiled with -fno-honor-nans.
__global__ void __launch_bounds__(256)
count_bad_velocity_kernel(const float *__restrict__ v, long long n, unsigned tiny_bits, unsigned long long *__restrict__ bad)
{
    unsigned mine = 0, mine_small = 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const unsigned bits = __float_as_uint(v[i]);
        const bool zero = bits == 0u || bits == 0x80000000u;
        const bool small = !zero && bits < tiny_bits;               // (positive: the sign bit makes a number large)
        const bool ok = zero || bits < 0x7f800000u;
        mine += !ok;
        mine_small += small;
    }
#pragma unroll
    for (int w = 32; w >= 1; w >>= 1) { mine += __shfl_xor(mine, w); mine_small += __shfl_xor(mine_small, w); }
    if ((threadIdx.x & 63) == 0 && mine) atomicAdd(bad, (unsigned long long)mine);
    if ((threadIdx.x & 63) == 0 && mine_small) atomicAdd(bad + 1, (unsigned long long)mine_small);
}

hipError_t launch_count_bad_velocity(const float *v, long long n, float tiny, unsigned long long *bad, hipStream_t st)
{
    const unsigned nblocks = (unsigned)std::min<long long>((n + 255) / 256, 4096);
    unsigned tiny_bits = 1u;            // (every positive number)
    if (tiny > 0.0f) memcpy(&tiny_bits, &tiny, sizeof tiny_bits);
    hipLaunchKernelGGL(count_bad_velocity_kernel, dim3(nblocks), dim3(256), 0, st, v, n, tiny_bits, bad);
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

// EXACT: the reference's own rounding of a delay - the product d (v[c] + v[o]) rounded, THEN halved
// (serial_new/sweep-tt-multistart.c:216) - for velocity volumes with values so small that the product can be a
// denormal number, where d / 2 times the sum rounds differently (h + h = d exactly).
template <bool EXACT>
__device__ __forceinline__ float edge_delay(float h, float sum)
{
    if (EXACT) {
        float p = (h + h) * sum;
        asm volatile("" : "+v"(p));     // (the product is rounded before it is halved)
        return p * 0.5f;
    }
    return h * sum;
}

template <bool EXACT>
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
            const float delay = edge_delay<EXACT>(en.h, sum);
            const float cand = delay + T[oi];
            if (live && cand < best) best = cand;
        }
        if (best < told) {
            T[ci] = best;
            improved = true;
        }
    }
    // one atomic per wavefront that improved anything (64-lane ballot)
    if (__ballot(improved) != 0ull && (threadIdx.x & 63) == 0) atomicOr(&changed[s], CHANGED_IMPROVED);
}

hipError_t launch_sweep_cell(const DevLayout &L, const float *v, const StartDesc *starts,
                             const int *active, int nactive, int *changed,
                             const CellEntry *entries, int nentries, bool exact, hipStream_t st)
{
    const int cblocks = (L.n[2] + CELL_BX - 1) / CELL_BX;
    const int bblocks = (L.n[1] + CELL_BY - 1) / CELL_BY;
    const long long nblocks = (long long)nactive * L.n[0] * bblocks * cblocks;
    if (nblocks <= 0) return hipSuccess;
    if (nblocks > 0x7fffffffLL) return hipErrorInvalidValue;
    hipLaunchKernelGGL(exact ? sweep_cell_kernel<true> : sweep_cell_kernel<false>, dim3((unsigned)nblocks),
                       dim3(CELL_BX, CELL_BY), 0, st, L, v, starts, active, changed, entries, nentries, cblocks, bblocks);
    return hipGetLastError();
}


// ===========================================================================
// validator: the reference's store conditions, read-only
// ===========================================================================
// For every interior centre cell c != start and every forward entry (offset e, l in
// [starstart, starstop)) with c + e inside the grid: would serial_new/...:225-249 store?
//   exactly one of T[c], T[o] infinite, or delay + T[o] < T[c], or delay + T[c] < T[o].

template <bool EXACT>
__global__ void __launch_bounds__(CELL_BX *CELL_BY)
validate_kernel(DevLayout L, const float *__restrict__ v, const float *__restrict__ T,
                long long sidx, const FwdEntry *__restrict__ entries, int nentries,
                unsigned long long *__restrict__ counts, int cblocks, int bblocks)
{
    unsigned bid = blockIdx.x;
    const int cb = bid % cblocks; bid /= cblocks;
    const int bb = bid % bblocks; bid /= bblocks;
    const int a = bid;
    const int c = cb * CELL_BX + threadIdx.x;
    const int b = bb * CELL_BY + threadIdx.y;
    unsigned open = 0, inf = 0;
    if (c < L.n[2] && b < L.n[1]) {
        const long long ci = dev_index(L, a, b, c);
        const float vc = v[ci], tc = T[ci];
        inf = (tc == __builtin_inff());
        if (ci != sidx) {
            for (int e = 0; e < nentries; e++) {
                const FwdEntry en = entries[e];
                const int oa = a + en.da, ob = b + en.db, oc = c + en.dc;
                if ((unsigned)oa >= (unsigned)L.n[0] || (unsigned)ob >= (unsigned)L.n[1]
                    || (unsigned)oc >= (unsigned)L.n[2])
                    continue;
                const long long oi = dev_index(L, oa, ob, oc);
                const float to = T[oi];
                const float sum = vc + v[oi];
                const float delay = edge_delay<EXACT>(en.h, sum);
                const bool tinf = tc == __builtin_inff(), oinf = to == __builtin_inff();
                if (tinf && oinf) continue;
                if (tinf != oinf || delay + to < tc || delay + tc < to) open++;
            }
        }
    }
    // wave-level sums, one atomic pair per wave
#pragma unroll
    for (int w = 32; w >= 1; w >>= 1) {
        open += __shfl_xor(open, w);
        inf += __shfl_xor(inf, w);
    }
    if ((threadIdx.x & 63) == 0) {
        if (open) atomicAdd(&counts[0], (unsigned long long)open);
        if (inf) atomicAdd(&counts[1], (unsigned long long)inf);
    }
}

// counts[2] += cells (other than the start) whose finite travel time is SMALLER than every
// candidate their live edges offer: no edge can have produced it.  Together with
// counts[0] == 0 (no edge can still improve anything) this pins T to the one fixed point of
// the relaxation, i.e. to the reference's converged result.  One thread per cell, the whole
// pull star with the liveness rule of sweep_cell_kernel.
template <bool EXACT>
__global__ void __launch_bounds__(CELL_BX *CELL_BY)
support_kernel(DevLayout L, const float *__restrict__ v, const float *__restrict__ T,
               long long sidx, const CellEntry *__restrict__ entries, int nentries,
               unsigned long long *__restrict__ counts, int cblocks, int bblocks)
{
    unsigned bid = blockIdx.x;
    const int cb = bid % cblocks; bid /= cblocks;
    const int bb = bid % bblocks; bid /= bblocks;
    const int a = bid;
    const int c = cb * CELL_BX + threadIdx.x;
    const int b = bb * CELL_BY + threadIdx.y;
    unsigned unsupported = 0;
    if (c < L.n[2] && b < L.n[1]) {
        const long long ci = dev_index(L, a, b, c);
        const float vc = v[ci], tc = T[ci];
        if (ci != sidx && tc < __builtin_inff()) {
            float best = __builtin_inff();
            for (int e = 0; e < nentries; e++) {
                const CellEntry en = entries[e];
                const long long oi = ci + en.delta;
                // (ci is not the start, so PULL_FWD entries are live)
                const bool live = (en.flags & PULL_FWD) || ((en.flags & PULL_REV) && oi != sidx);
                const float sum = vc + v[oi];
                const float delay = edge_delay<EXACT>(en.h, sum);
                const float cand = delay + T[oi];
                if (live && cand < best) best = cand;
            }
            unsupported = tc < best;
        }
    }
#pragma unroll
    for (int w = 32; w >= 1; w >>= 1) unsupported += __shfl_xor(unsupported, w);
    if ((threadIdx.x & 63) == 0 && unsupported) atomicAdd(&counts[2], (unsigned long long)unsupported);
}

hipError_t launch_validate(const DevLayout &L, const float *v, const float *T, long long sidx,
                           const FwdEntry *entries, int nentries, const CellEntry *cell_entries,
                           int ncell_entries, unsigned long long *counts, bool exact, hipStream_t st)
{
    const int cblocks = (L.n[2] + CELL_BX - 1) / CELL_BX;
    const int bblocks = (L.n[1] + CELL_BY - 1) / CELL_BY;
    const long long nblocks = (long long)L.n[0] * bblocks * cblocks;
    if (nblocks <= 0) return hipSuccess;
    if (nblocks > 0x7fffffffLL) return hipErrorInvalidValue;
    hipLaunchKernelGGL(exact ? validate_kernel<true> : validate_kernel<false>, dim3((unsigned)nblocks), dim3(CELL_BX, CELL_BY),
                       0, st, L, v, T, sidx, entries, nentries, counts, cblocks, bblocks);
    hipLaunchKernelGGL(exact ? support_kernel<true> : support_kernel<false>, dim3((unsigned)nblocks), dim3(CELL_BX, CELL_BY),
                       0, st, L, v, T, sidx, cell_entries, ncell_entries, counts, cblocks, bblocks);
    return hipGetLastError();
}

// ===========================================================================
// sweep, variant STRIP: LDS-staged neighbour planes, register strips
// ===========================================================================
//
// Work decomposition (device axes a, b, c; c is stride-1):
//   unit      = TWO neighbouring planes (2A, 2A+1) x 64 cells along b x one strip of STRIP_K
//               cells along c: what one workgroup relaxes at a time, and the granule of
//               activity tracking
//   lane      = one b row of the unit: K consecutive c cells of either plane, in registers
//   wave      = a share of the items (slab rows x offset sets) of every staged plane; the
//               four waves of the workgroup min-combine their partial results at the end
// Each lane keeps acc[j][K] (best travel time so far) and its own velocities (as register
// pairs) of both own planes in registers.  The workgroup stages every neighbour plane q
// the star reaches from either own plane - its (64 + 2 rb) x (K + 16) window of v and T -
// into LDS once.  For every item (a row offset db of the staged plane) a lane reads ONE
// register window of K + 16 neighbour values per array (8 ds_read_b128 each) and relaxes
// against it every offset dc of plane offset q - 2A (into the first own plane) and of plane
// offset q - 2A - 1 (into the second): a loaded value is reused for up to 30 relaxations,
// and slab traffic, barriers and per-unit overheads are shared by two planes of output.
//
// The relaxation is branch-free: there are no bounds tests (halo cells hold
// +INF / 0) and no liveness tests.  The few cells that own a dead edge (inside
// StartDesc::box) are computed but not stored; relax_special_cell owns them.

// The first words of the dynamic LDS region carry workgroup-wide scalars (no static
// __shared__ object: it would shift the 16-byte alignment of the dynamic base).
constexpr int strip_lds_head(int ns) { return 16 + 16 * ns + 32; }  // words reserved in front of the slabs: 16 scalars, the
                                            // waves' item ranges per staged plane (16 per wave), 32 spare
                                            // (sweep_units_kernel; a multiple of four: the slabs stay 16-byte aligned)
// slab geometry (bytes): up to 64 + 2*7 rows, rounded up to 8, of 128 B for v, then for T
constexpr int SLAB_MAX_ROWS8 = (STRIP_TB + 2 * STRIP_MAX_RA + 7) / 8 * 8;
constexpr int SLAB_T_BYTES = SLAB_MAX_ROWS8 * STRIP_W * 4;
constexpr int SLAB_BYTES = 2 * SLAB_T_BYTES;

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// Scalars an item's relaxation needs besides its window: the offset lengths of either own
// plane (one s_load_dwordx16 each) and the next item's header.  They are handed to the window
// load, whose "all loads issued" point (an empty asm statement) takes them as inputs: the scalar
// loads are then issued BEFORE that point and travel beside the window's LDS reads.  (LDS and
// scalar loads share the lgkmcnt counter and scalar loads return out of order, so the one that
// is issued second is waited for with lgkmcnt(0) anyway; left to the scheduler, the scalar loads
// came behind the window's wait and their latency was paid a second time, every item.)
struct ItemScalars {
    f32x16 h0, h1;
    int next_rowoff;
    unsigned next_m0, next_m1;
};

// Read-only tables written before the launch (items of the star, start descriptors, the
// queues the planner filled) are read through the scalar cache: uniform addresses in the
// constant address space become s_load instructions.
typedef const __attribute__((address_space(4))) StripItem *const_item_ptr;
typedef const __attribute__((address_space(4))) StartDesc *const_start_ptr;
struct QueueEntry { int s, unit, planes, pad; };        // (an int4 as the planner writes it)
typedef const __attribute__((address_space(4))) QueueEntry *const_entry_ptr;

// What is needed of an item before its window can be read.
struct ItemHdr { int rowoff; unsigned m0, m1; };
__device__ __forceinline__ ItemHdr load_hdr(const StripItem *items, int i)
{
    const const_item_ptr it = (const_item_ptr)(items + i);
    return ItemHdr{it->rowoff, it->mask[0], it->mask[1]};
}

// Activity flag word of a unit (tile_flags): where something improved, as seen by the
// neighbour that would read it.  Along the strip axis c and along the lane axis b a zone is
// ANY (anywhere), LO or HI (within reach of the unit's low / high border: the only cells a
// neighbouring strip or lane tile reads); bit 3 * zone_b + zone_c is set when a cell in that
// combination of zones improved.
enum : int { ZONE_ANY = 0, ZONE_LO = 1, ZONE_HI = 2, FLAG_ALL = 0x1ff };
__host__ __device__ constexpr int flag_bit(int zone_b, int zone_c) { return 1 << (3 * zone_b + zone_c); }

// Relax the offsets `mask` (bit t <-> dc = t - 8) of one item into acc: window element w of
// the neighbour row is pair w/2, half w&1 of vN2 / tN2 (registers).
//
// MASK != 0: the offset set is a compile-time constant -> straight-line code.
// MASK == 0: runtime set (`mask`), one scalar-branch-selected block per offset.
//
// Arithmetic is packed (v_pk_add_f32 / v_pk_mul_f32, two cells per instruction; each
// component rounds exactly like the scalar op).  Cell q needs window element q + t; the
// pair (q, q+1) of cells is chosen so that both elements sit in ONE aligned register
// pair: even t -> cells (0,1),(2,3)..; odd t -> cells (1,2),(3,4).., with cells 0 and
// K-1 done singly.  Measured (tools/microbench/relax_static.hip): straight-line packed
// blocks sustain ~15 T relaxations/s against ~11 T/s for any scalar form.
template <int K, unsigned MASK>
__device__ __forceinline__ void relax_window(unsigned rt_mask, const float (&h)[16],
                                             const f32x2 (&vN2)[(K + 2 * STRIP_CF) / 2],
                                             const f32x2 (&tN2)[(K + 2 * STRIP_CF) / 2],
                                             const f32x2 (&vce)[K / 2], const f32x2 (&vco)[K / 2 - 1],
                                             float (&acc)[K])
{
    const unsigned mask = MASK != 0u ? MASK : rt_mask;
#pragma unroll
    for (int t = 1; t < 2 * STRIP_CF; t++) {
        if (mask & (1u << t)) {
            const float hv = h[t];
            const f32x2 h2 = {hv, hv};
            if ((t & 1) == 0) {
                f32x2 x[K / 2];
#pragma unroll
                for (int p = 0; p < K / 2; p++) x[p] = vce[p] + vN2[p + t / 2];
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
                float y0 = vce[0].x + vN2[(t - 1) / 2].y;
                float y1 = vce[K / 2 - 1].y + vN2[(K - 1 + t) / 2].x;
#pragma unroll
                for (int p = 0; p < K / 2 - 1; p++) x[p] = vco[p] + vN2[p + (t + 1) / 2];
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

// The float4 chunks of a K + 16 wide window that the offsets `mask` read (elements t .. t+K-1).
template <int K>
__host__ __device__ constexpr unsigned window_chunks(unsigned mask)
{
    unsigned chunks = 0;
    for (int t = 1; t < 2 * STRIP_CF; t++)
        if (mask & (1u << t))
            for (int j = 0; j < (K + 2 * STRIP_CF) / 4; j++)
                if (4 * j + 3 >= t && 4 * j <= t + K - 1) chunks |= 1u << j;
    return chunks;
}

// The neighbour window of one slab row, both arrays, into registers: chunk j of the row is
// float4 j ^ swizzle (stage_slab); one v_xad_u32 per float4 pair, the T row sits at a
// compile-time distance behind the v row.  CHUNKS: the float4s that are needed (others stand
// in for a loaded one: no instructions).
template <int K, unsigned CHUNKS>
__device__ __forceinline__ void load_window(const char *prow, unsigned swb, const ItemScalars &sc,
                                            f32x2 (&vN2)[(K + 2 * STRIP_CF) / 2], f32x2 (&tN2)[(K + 2 * STRIP_CF) / 2])
{
    constexpr int W = K + 2 * STRIP_CF;
    static_assert(W / 4 == 8 && CHUNKS != 0u, "the window is 8 float4 wide");
    f32x4 xw[W / 4], yw[W / 4];
#pragma unroll
    for (int jj = 0; jj < W / 4; jj++) {
        if (CHUNKS & (1u << jj)) {
            const char *at = prow + (swb ^ (unsigned)(16 * jj));
            xw[jj] = *reinterpret_cast<const f32x4 *>(at);
            yw[jj] = *reinterpret_cast<const f32x4 *>(at + SLAB_T_BYTES);
        }
    }
    constexpr int JF = __builtin_ctz(CHUNKS);       // a chunk that is read
#pragma unroll
    for (int jj = 0; jj < W / 4; jj++)
        if (!(CHUNKS & (1u << jj))) { xw[jj] = xw[JF]; yw[jj] = yw[JF]; }
    // All loads are issued before the first value is used, and whole float4s are kept: partly
    // used chunks would otherwise be narrowed to ds_read2_b64 pairs (8 LDS cycles instead of 4).
#ifdef TTSWEEP_NO_SLOAD_PIN
    asm volatile("" :: "v"(xw[0]), "v"(xw[1]), "v"(xw[2]), "v"(xw[3]), "v"(xw[4]), "v"(xw[5]),
                 "v"(xw[6]), "v"(xw[7]), "v"(yw[0]), "v"(yw[1]), "v"(yw[2]), "v"(yw[3]), "v"(yw[4]),
                 "v"(yw[5]), "v"(yw[6]), "v"(yw[7]));
#else
    asm volatile("" :: "v"(xw[0]), "v"(xw[1]), "v"(xw[2]), "v"(xw[3]), "v"(xw[4]), "v"(xw[5]),
                 "v"(xw[6]), "v"(xw[7]), "v"(yw[0]), "v"(yw[1]), "v"(yw[2]), "v"(yw[3]), "v"(yw[4]),
                 "v"(yw[5]), "v"(yw[6]), "v"(yw[7]), "s"(sc.h0), "s"(sc.h1), "s"(sc.next_rowoff),
                 "s"(sc.next_m0), "s"(sc.next_m1));
#endif
#pragma unroll
    for (int jj = 0; jj < W / 4; jj++) {
        vN2[2 * jj] = f32x2{xw[jj].x, xw[jj].y}; vN2[2 * jj + 1] = f32x2{xw[jj].z, xw[jj].w};
        tN2[2 * jj] = f32x2{yw[jj].x, yw[jj].y}; tN2[2 * jj + 1] = f32x2{yw[jj].z, yw[jj].w};
    }
}

// Units of one plane: window load and relaxation in one routine per offset set, so that only
// the float4s the set reads are loaded.
template <int K>
__device__ __forceinline__ void relax_item_single(unsigned mask, const float (&h)[16], const char *prow, unsigned swb,
                                                  const ItemScalars &sc,
                                                  const f32x2 (&vce)[K / 2], const f32x2 (&vco)[K / 2 - 1],
                                                  float (&acc)[K])
{
    f32x2 vN2[(K + 2 * STRIP_CF) / 2], tN2[(K + 2 * STRIP_CF) / 2];
    switch (mask) {
    case 0u: break;
#define STRIP_MASK_CASE(m) \
    case m: load_window<K, window_chunks<K>(m)>(prow, swb, sc, vN2, tN2); relax_window<K, m>(mask, h, vN2, tN2, vce, vco, acc); break;
#include "strip_masks.inc"
#undef STRIP_MASK_CASE
    default: load_window<K, 0xffu>(prow, swb, sc, vN2, tN2); relax_window<K, 0u>(mask, h, vN2, tN2, vce, vco, acc); break;
    }
}

// The straight-line routine of the item's offset set, or the generic one.
template <int K>
__device__ __forceinline__ void relax_dispatch(unsigned mask, const float (&h)[16],
                                               const f32x2 (&vN2)[(K + 2 * STRIP_CF) / 2],
                                               const f32x2 (&tN2)[(K + 2 * STRIP_CF) / 2],
                                               const f32x2 (&vce)[K / 2], const f32x2 (&vco)[K / 2 - 1],
                                               float (&acc)[K])
{
    switch (mask) {
    case 0u: break;
#define STRIP_MASK_CASE(m) case m: relax_window<K, m>(mask, h, vN2, tN2, vce, vco, acc); break;
#include "strip_masks.inc"
#undef STRIP_MASK_CASE
    default: relax_window<K, 0u>(mask, h, vN2, tN2, vce, vco, acc); break;
    }
}

// Activity flags are kept per (plane, lane tile, strip) whatever the unit size: a unit of two
// planes sets and clears the words of its planes separately, so that a staged plane counts
// as changed only when that very plane improved.
__host__ __device__ inline int strip_flag_words(const DevLayout &L) { return L.n[0] * strip_btiles(L) * strip_cstrips(L); }

__global__ void __launch_bounds__(256)
init_tile_flags_kernel(int *__restrict__ flags, int nflag, int start_flag)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= nflag) return;
    flags[t] = 0;
    flags[nflag + t] = t == start_flag ? FLAG_ALL : 0;
    flags[2 * nflag + t] = 0;           // pend[] (per unit): nothing held back yet
    if (t == 0) flags[3 * nflag] = 1;   // one source: the distance gate applies
}

// Flags for a box that arrives with values in it: a (plane, lane tile, strip) counts as
// "changed" when it holds a finite travel time (only those can improve anything).  One wave
// per flag word.
__global__ void __launch_bounds__(64)
init_tile_flags_box_kernel(DevLayout L, const float *__restrict__ T, int *__restrict__ flags,
                           int nflag, int btiles, int cstrips)
{
    int u = blockIdx.x;
    const int cs = u % cstrips;  u /= cstrips;
    const int bt = u % btiles;   u /= btiles;
    const int a = u;
    const int lane = threadIdx.x;
    bool finite = false;
    if (bt * STRIP_TB + lane < L.n[1]) {
        const long long g = (long long)(a + L.lo[0]) * L.s0
                          + (long long)(bt * STRIP_TB + lane + L.lo[1]) * L.s1 + (cs * STRIP_K + L.lo[2]);
#pragma unroll
        for (int q = 0; q < STRIP_K; q++) finite |= T[g + q] < __builtin_inff();
    }
    const bool any = __ballot(finite) != 0ull;
    if (lane == 0) {
        flags[blockIdx.x] = 0;
        flags[nflag + blockIdx.x] = any ? FLAG_ALL : 0;
        flags[2 * nflag + blockIdx.x] = 0;
        if (any) atomicAdd(&flags[3 * nflag], 1);       // number of sources
    }
}

__global__ void seed_pend_kernel(DevLayout L, int *__restrict__ tile_flags, int nflag, int ra, int np, int btiles,
                                 int cstrips, int nunits, long long stride);      // (below, next to push_improved)

// init_tile_flags_kernel for every start of a solve (blockIdx.y = start; flags0 + start * stride)
__global__ void __launch_bounds__(256)
init_tile_flags_batch_kernel(int *__restrict__ flags0, long long stride, int nflag, const StartDesc *__restrict__ starts,
                             int btiles, int cstrips)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= nflag) return;
    int *const flags = flags0 + (long long)blockIdx.y * stride;
    const int sa = starts[blockIdx.y].sa, sb = starts[blockIdx.y].sb, sc = starts[blockIdx.y].sc;
    const int start_flag = (sa * btiles + sb / STRIP_TB) * cstrips + sc / STRIP_K;
    flags[t] = 0;
    flags[nflag + t] = t == start_flag ? FLAG_ALL : 0;
    flags[2 * nflag + t] = 0;
    if (t == 0) flags[3 * nflag] = 1;
}

hipError_t launch_init_tile_flags_batch(const DevLayout &L, int *flags0, long long stride, const StartDesc *starts,
                                        int nstart, int ra, int np, hipStream_t st)
{
    if (nstart <= 0 || nstart > 65535) return hipErrorInvalidValue;
    const int btiles = strip_btiles(L), cstrips = strip_cstrips(L);
    const int nflag = strip_flag_words(L);
    hipLaunchKernelGGL(init_tile_flags_batch_kernel, dim3((nflag + 255) / 256, nstart), dim3(256), 0, st,
                       flags0, stride, nflag, starts, btiles, cstrips);
    const int nunits = strip_units(L, np);
    hipLaunchKernelGGL(seed_pend_kernel, dim3((nunits + 255) / 256, nstart), dim3(256), 0, st, L, flags0, nflag, ra, np,
                       btiles, cstrips, nunits, stride);
    return hipGetLastError();
}

hipError_t launch_init_tile_flags(const DevLayout &L, const StartDesc &sd, bool from_box, int ra, int np, hipStream_t st)
{
    const int btiles = strip_btiles(L), cstrips = strip_cstrips(L);
    const int nflag = strip_flag_words(L);
    if (from_box) {
        hipError_t e = hipMemsetAsync(sd.tile_flags + 3 * (size_t)nflag, 0, sizeof(int), st);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(init_tile_flags_box_kernel, dim3(nflag), dim3(64), 0, st, L, sd.T,
                           sd.tile_flags, nflag, btiles, cstrips);
    } else {
        const int start_flag = (sd.sa * btiles + sd.sb / STRIP_TB) * cstrips + sd.sc / STRIP_K;
        hipLaunchKernelGGL(init_tile_flags_kernel, dim3((nflag + 255) / 256), dim3(256), 0, st,
                           sd.tile_flags, nflag, start_flag);
    }
    const int nunits = strip_units(L, np);
    hipLaunchKernelGGL(seed_pend_kernel, dim3((nunits + 255) / 256), dim3(256), 0, st, L, sd.tile_flags, nflag, ra, np,
                       btiles, cstrips, nunits, 0ll);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// a pass: unit queues drained by a persistent grid
// ---------------------------------------------------------------------------
//
// Activity tracking.  A unit's offsets against staged plane q have to be relaxed in a pass
// only if a patch they read from (plane q, +-1 lane tile, +-1 strip) improved since the unit
// last relaxed against it: everything else was already relaxed against unchanged values.
// Whoever improves a patch tells the units that stage it (push_improved: one bit per staged
// plane in the unit's pend word), by border zone (flag_bit), so that a neighbouring strip or
// lane tile reacts only to improvements within its reach.
//
// Distance gate.  Far from the start the first values to arrive (over long edges) are
// poor and get refined pass after pass; relaxing those units early is wasted work.  A unit
// is therefore held back until the gate radius (it grows by a fixed number of cells per
// pass) reaches it.  The plane bits it owes are remembered in pend[] (they accumulate
// across passes), so holding a unit back never loses an update, and a start with anything
// pending is not reported as converged.  The gate is for solves that grow from one source
// unit; a box that arrives with many finite units (e.g. an already converged one) is
// relaxed ungated.
//
// Only a shell of units is due in any pass, and most workgroups of a grid-per-unit launch
// would start, read their neighbours' flags and leave.  A pass therefore runs in two steps:
// plan_pass_kernel (one THREAD per unit) decides which units are due and writes them, in
// work-list order, into one queue per XCD; sweep_units_kernel, a grid of two workgroups per
// CU, drains the queues (own XCD's first, then the others').  A workgroup relaxes ONE unit
// at a time, its four waves splitting the items of every staged plane among themselves
// (nearly equal shares, StripPlan::wsplit) and min-combining their partial results through LDS.

// ---- who has to look again ----------------------------------------------------------------
// pend[unit] (third block of StartDesc::tile_flags): bit p set = staged plane p of the unit
// changed since the unit last relaxed against it (or the unit was held back by the gate with
// that bit set).  The bits are PUSHED by whoever improves a plane: the patch (plane a, lane
// tile bt, strip cs) that improved sets bit a - (np A' - ra) of every unit (A', bt', cs') that
// stages plane a from it - at most (2 ra + np) / np + 1 plane groups x 3 lane tiles x 3 strips,
// a lane each, atomicOr without return - where a neighbouring lane tile / strip counts only if
// the improvement lay within reach of the shared border (flag_bit).  The planner then reads
// ONE word per unit (round 2: it gathered 9 flag words per staged plane, 144 per unit of two
// planes, in every pass).
// Deferral (DeferRule).  Most improvements matter only to the units FARTHER from the start than
// the cells that improved: the units behind a front have their final values (or nearly), and
// relaxing them again for every improvement in front of them finds nothing (measured: 47 % of the
// unit relaxations of a solve improved no cell).  A bit for a unit whose centre is nearer to the
// start than the improved patch's by more than `margin` cells is therefore not put into the unit's
// pend word but into its `defer` word (first block of StartDesc::tile_flags), which nobody looks at
// until the start is otherwise at rest; then the deferred bits become pend bits (flush_deferred /
// the ring planner) and the solve goes on until both kinds are gone.  Every bit is honoured in
// the end - after the last change of the plane it stands for -, so the fixed point is the same;
// what is saved is the repetition: a deferred (unit, plane) pair is relaxed once, against final
// values, instead of once per pass in which the plane changed.  Where the geometry misleads
// (a fast path that runs back towards the start) the late relaxation improves cells and the solve
// simply continues from there.
struct DeferRule {
    int sa, sb, sc;         // the start (device axes)
    float margin;           // cells; < -1e30: nothing is deferred
};

// Hand-off (one launch per solve, AsyncSolve::handoff): what a worker needs to publish a unit into a ring itself.
struct HandOff {
    int on;                         // ASYNC_HANDOFF_* bits; 0: plane bits only, the planner does the rest
    float gate_r2;                  // squared gate radius of the ring's planner (cells): units beyond it are left to it
    unsigned long long *ht;         // the ring's head | tail << 32 | done << 63 word
    unsigned long long *ents;       // the ring's slots
    unsigned cap_mask;
    unsigned long long tag_bits;    // start << 36 | "tell every unit at once" << 63 of the entries to publish
    const unsigned *status;         // AsyncSolve::status
    long long deadline;             // wall clock after which every wait gives up
};

// a ring slot as it is in memory: a read-modify-write that changes nothing (-DTTSWEEP_SLOT_RMW; experiment)
__device__ __forceinline__ unsigned long long slot_load(const unsigned long long *p)
{
#ifdef TTSWEEP_SLOT_RMW
    return __hip_atomic_fetch_or(const_cast<unsigned long long *>(p), 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}

// Publishes (unit, planes) at the tail of the ring: one returning add reserves the position, the entry goes into its
// slot as soon as the slot is empty (it is, unless the ring has been lapped while a worker sat on the old entry).
// The unit's pend word holds ASYNC_BUSY already (the caller's compare-and-swap).
__device__ __forceinline__ void handoff_publish(const HandOff &ho, unsigned unit, unsigned planes)
{
    const unsigned long long old = atomicAdd(ho.ht, 1ull << 32);
    const unsigned pos = (unsigned)(old >> 32) & 0x7fffffffu;
#ifdef TTSWEEP_RING_DEBUG
    atomicAdd(ho.ht + 27, 1ull);
#endif
    unsigned long long *const slot = ho.ents + (pos & ho.cap_mask);
    for (unsigned spin = 0; slot_load(slot) != 0ull; spin++) {
        if ((spin & 63u) == 63u
            && (__hip_atomic_load(ho.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != ASYNC_OK || wall_clock64() > ho.deadline))
            return;         // (the solve has failed or is about to: the pass driver rebuilds the activity words)
        __builtin_amdgcn_s_sleep(2);
    }
    const unsigned long long e = (unsigned long long)planes | ((unsigned long long)unit << 16) | ho.tag_bits
                               | ((unsigned long long)(pos & ASYNC_TAG_MASK) << 44) | ASYNC_ENTRY_VALID;
    __hip_atomic_store(slot, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef TTSWEEP_RING_DEBUG
    atomicAdd(ho.ht + 25, 1ull);
#endif
}

// Takes an idle unit whose pend word holds plane bits: pend -> ASYNC_BUSY.  `seen`: the word as last read (bits, no
// busy bit).  Returns the plane bits taken, 0 when somebody else was faster.
__device__ __forceinline__ unsigned handoff_take(unsigned *w, unsigned seen)
{
    for (int tries = 0; tries < 4 && seen != 0u && !(seen & ASYNC_BUSY); tries++) {
        const unsigned prev = atomicCAS(w, seen, ASYNC_BUSY);
        if (prev == seen) return seen;
        seen = prev;        // (more bits have arrived, or the unit has been taken)
    }
    return 0u;
}

template <int NP, bool HO = false>
__device__ __forceinline__ void push_improved(const DevLayout &L, int ra, int btiles, int cstrips,
                                              unsigned *__restrict__ pend, int a, int bt, int cs, int improved, int lane,
                                              const DeferRule &rule, unsigned *__restrict__ defer,
                                              const HandOff &ho = HandOff{})
{
    const int lo_num = a - ra - NP + 1;                                 // A' >= ceil(lo_num / NP)
    const int Alo = max(NP == 1 ? lo_num : (lo_num + 1) >> 1, 0);
    const int Ahi = min((a + ra) / NP, strip_agroups(L, NP) - 1);
    const int n = (Ahi - Alo + 1) * 9;
    const float half_b = 0.5f * (float)min(STRIP_TB, L.n[1]);
    // distance of the improved patch from the start, less the margin: a target nearer than that is deferred
    const float sda = (float)(a - rule.sa), sdb = (float)(bt * STRIP_TB - rule.sb) + half_b,
                sdc = (float)(cs * STRIP_K - rule.sc) + 0.5f * STRIP_K;
    const float lim = sqrtf(sda * sda + sdb * sdb + sdc * sdc) - rule.margin;
    const float lim2 = lim > 0.f && rule.margin > -1.0e30f ? lim * lim : -1.f;
    for (int idx = lane; idx < n; idx += 64) {
        const int A = Alo + idx / 9, r = idx % 9;
        const int db = r / 3 - 1, dc = r % 3 - 1;
        const int nb = bt + db, nc = cs + dc;
        // the unit above / behind reads across our high border: our HI zone matters to it
        const int need = flag_bit(db == 1 ? ZONE_HI : db == -1 ? ZONE_LO : ZONE_ANY,
                                  dc == 1 ? ZONE_HI : dc == -1 ? ZONE_LO : ZONE_ANY);
        if (nb >= 0 && nb < btiles && nc >= 0 && nc < cstrips && (improved & need)) {
            const float tda = (float)(NP * A - rule.sa) + 0.5f * (NP - 1), tdb = (float)(nb * STRIP_TB - rule.sb) + half_b,
                        tdc = (float)(nc * STRIP_K - rule.sc) + 0.5f * STRIP_K;
            const bool later = tda * tda + tdb * tdb + tdc * tdc < lim2;
            const int unit = (A * btiles + nb) * cstrips + nc;
            const unsigned bit = 1u << (a - (NP * A - ra));
            if (HO && !later && (ho.on & ASYNC_HANDOFF_NEIGHBOURS)) {
                // the unit hears of it AND, when it is idle and inside the gate, goes into the ring at once
                const unsigned old = atomicOr(&pend[unit], bit);
                if (!(old & ASYNC_BUSY)) {
                    // (the planner's distance: from the start to the nearest cell of the unit)
                    const int a0 = NP * A, b0 = nb * STRIP_TB, c0 = nc * STRIP_K, tb_eff = min(STRIP_TB, L.n[1]);
                    const float ga = (float)max(max(a0 - rule.sa, rule.sa - (a0 + NP - 1)), 0);
                    const float gb = (float)max(max(b0 - rule.sb, rule.sb - (b0 + tb_eff - 1)), 0);
                    const float gc = (float)max(max(c0 - rule.sc, rule.sc - (c0 + STRIP_K - 1)), 0);
                    if (ga * ga + gb * gb + gc * gc <= ho.gate_r2) {
                        const unsigned planes = handoff_take(&pend[unit], old | bit);
                        if (planes) handoff_publish(ho, (unsigned)unit, planes);
                    }
                }
            } else {
                atomicOr(&(later ? defer : pend)[unit], bit);
            }
        }
    }
}

// pend |= defer, defer = 0 for every unit of the starts listed in `active` (one thread per unit and
// start); changed[s] |= CHANGED_PENDING where a bit moved.  The pass driver runs it when every
// start is at rest.
__global__ void __launch_bounds__(256)
flush_deferred_kernel(int *__restrict__ flags0, long long flags_stride, int nflag, int nunits,
                      const int *__restrict__ active, int *__restrict__ changed)
{
    const int unit = blockIdx.x * 256 + threadIdx.x;
    const int s = active[blockIdx.y];
    if (unit >= nunits) return;
    unsigned *const defer = reinterpret_cast<unsigned *>(flags0 + (long long)s * flags_stride);
    const unsigned d = defer[unit];
    if (d) {
        defer[unit] = 0;
        atomicOr(defer + 2 * nflag + unit, d);
        atomicOr(&changed[s], CHANGED_PENDING);
    }
}

hipError_t launch_flush_deferred(const DevLayout &L, int np, int *flags0, long long flags_stride,
                                 const int *active, int nactive, int *changed, hipStream_t st)
{
    if (nactive <= 0) return hipSuccess;
    const int nunits = strip_units(L, np);
    hipLaunchKernelGGL(flush_deferred_kernel, dim3((nunits + 255) / 256, nactive), dim3(256), 0, st,
                       flags0, flags_stride, strip_flag_words(L), nunits, active, changed);
    return hipGetLastError();
}

// First pend words of a start, from the patch flags the initialisation kernels leave in the
// second flag block (the start's patch, or every patch that holds a finite value): one thread
// per unit, the rule of push_improved read backwards.
__global__ void __launch_bounds__(256)
seed_pend_kernel(DevLayout L, int *__restrict__ tile_flags, int nflag, int ra, int np, int btiles, int cstrips, int nunits,
                 long long stride)
{
    const int unit = blockIdx.x * 256 + threadIdx.x;
    if (unit >= nunits) return;
    tile_flags += (long long)blockIdx.y * stride;       // (one launch for all starts of a solve: blockIdx.y = start)
    int u = unit;
    const int cs = u % cstrips;  u /= cstrips;
    const int bt = u % btiles;   u /= btiles;
    const int a0 = np * u;
    const int *__restrict__ flags = tile_flags + nflag;
    unsigned planes = 0;
    for (int p = 0; p < 2 * ra + np; p++) {
        const int q = a0 - ra + p;
        if (q < 0 || q >= L.n[0]) continue;
        for (int r = 0; r < 9; r++) {
            const int nb = bt + r / 3 - 1, nc = cs + r % 3 - 1;
            const int need = flag_bit(r / 3 == 0 ? ZONE_HI : r / 3 == 2 ? ZONE_LO : ZONE_ANY,
                                      r % 3 == 0 ? ZONE_HI : r % 3 == 2 ? ZONE_LO : ZONE_ANY);
            if (nb >= 0 && nb < btiles && nc >= 0 && nc < cstrips && (flags[(q * btiles + nb) * cstrips + nc] & need))
                planes |= 1u << p;
        }
    }
    reinterpret_cast<unsigned *>(tile_flags + 2 * nflag)[unit] = planes;
}

struct PlaneCounts { int n[STRIP_STAGED][STRIP_PLANES]; };     // StripPlan::nent (offsets per staged plane and own plane)

__global__ void __launch_bounds__(256)
plan_pass_kernel(DevLayout L, const StartDesc *__restrict__ starts, const int2 *__restrict__ work,
                 long long nwork, int *__restrict__ changed, int4 *__restrict__ lists, int list_cap,
                 int *__restrict__ ctrl, int nlists, int ra, int np, int btiles, int cstrips,
                 float gate_r2, PlaneCounts pc, int *__restrict__ flags0, long long flags_stride)
{
    // wave W handles 64 consecutive entries of ONE XCD's sub-list, so that a wave-level
    // compaction keeps the work-list order (nearest to the start first) inside a queue
    const int lane = threadIdx.x & 63;
    const long long W = ((long long)blockIdx.x * 256 + threadIdx.x) >> 6;
    const int x = (int)(W % nlists);
    const long long k = (W / nlists) * 64 + lane;
    const long long i = k * nlists + x;
    unsigned planes = 0;
    int s = 0, unit = -1;
    if (i < nwork) {
        const int2 item = work[i];
        s = item.x;
        unit = item.y;
    }
    if (unit >= 0) {
        // staged planes that changed since this unit last relaxed against them, and plane
        // bits it was held back with (push_improved, seed_pend_kernel); most units have none
        // and leave after this one word
        const int nflag = L.n[0] * btiles * cstrips;
        int *const tile_flags = flags0 + (long long)s * flags_stride;       // (= starts[s].tile_flags)
        unsigned *__restrict__ pend = reinterpret_cast<unsigned *>(tile_flags + 2 * nflag);
        planes = pend[unit];
    }
    if (planes != 0) {
        const StartDesc sd = starts[s];
        const int nflag = L.n[0] * btiles * cstrips;
        unsigned *__restrict__ pend = reinterpret_cast<unsigned *>(sd.tile_flags + 2 * nflag);
        int u = unit;
        const int cs = u % cstrips;  u /= cstrips;
        const int bt = u % btiles;   u /= btiles;
        const int a0 = np * u;                  // first own plane
        // distance gate
        const int b0 = bt * STRIP_TB, cb0 = cs * STRIP_K;
        const int tb_eff = min(STRIP_TB, L.n[1]);
        const float da_ = (float)max(max(a0 - sd.sa, sd.sa - (a0 + np - 1)), 0);
        const float db_ = (float)max(max(b0 - sd.sb, sd.sb - (b0 + tb_eff - 1)), 0);
        const float dc_ = (float)max(max(cb0 - sd.sc, sd.sc - (cb0 + STRIP_K - 1)), 0);
        const bool gated = sd.tile_flags[3 * nflag] == 1;
        const bool open = !gated || da_ * da_ + db_ * db_ + dc_ * dc_ <= gate_r2;
        if (!open) {
            if (planes) atomicOr(&changed[s], CHANGED_PENDING);     // work is waiting (the bits stay)
            planes = 0;
        } else if (planes) {
            pend[unit] = 0;         // (no unit kernel runs beside the planner: a plain store)
        }
    }
    const unsigned long long due_lanes = __ballot(planes != 0);
    if (due_lanes == 0ull) return;
    int base = 0;
    if (lane == 0) base = atomicAdd(&ctrl[x], __popcll(due_lanes));
    base = __shfl(base, 0);
    unsigned long long relax = 0;               // cells x offsets this unit will relax (statistics)
    if (planes != 0) {
        const int rank = __popcll(due_lanes & ((1ull << lane) - 1ull));
        lists[(size_t)x * list_cap + base + rank] = make_int4(s, unit, (int)planes, 0);
        int u = unit;
        const int cs = u % cstrips;  u /= cstrips;
        const int bt = u % btiles;   u /= btiles;
        const int wb = min(min(STRIP_TB, L.n[1]), L.n[1] - bt * STRIP_TB), wc = max(min(STRIP_K, L.n[2] - cs * STRIP_K), 0);
        const bool two = np > 1 && np * u + 1 < L.n[0];
        int nent = 0;
        for (int p = 0; p < 2 * ra + np; p++)
            if ((planes >> p) & 1u) nent += pc.n[p][0] + (two ? pc.n[p][1] : 0);
        relax = (unsigned long long)(wb * wc) * (unsigned long long)nent;
    }
    // The starts' work counters: one pair of atomics per wavefront and start here, not per unit
    // in the sweep kernel (where the few words of a small shard's starts were a hot spot that
    // every unit's first wave waited on: 3 starts 17.2 -> 16.6 ms).
    unsigned long long rest = due_lanes;
    while (rest) {
        const int first = __builtin_ctzll(rest);
        const int s0 = __shfl(s, first);
        const bool mine = planes != 0 && s == s0;
        const unsigned long long mm = __ballot(mine);
        unsigned long long sum = mine ? relax : 0ull;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
        if (lane == first) {
            atomicAdd(starts[s0].work, sum);
            atomicAdd(starts[s0].work + 2, (unsigned long long)__popcll(mm));
        }
        rest &= ~mm;
    }
}

// ---------------------------------------------------------------------------
// dead-edge cells: the cells inside StartDesc::box own an edge the reference never
// relaxes ({start, start - last offset}); one wave relaxes one such cell against the whole
// star with the full liveness rule (PULL_FWD entries are dead iff the cell is the start,
// PULL_REV entries iff the neighbour is).  Called by every wave of sweep_units_kernel
// before it turns to the unit queues.
// ---------------------------------------------------------------------------
template <bool ASYNC = false>
__device__ __forceinline__ void relax_special_cell(const DevLayout &L, const float *__restrict__ v,
                                                   const StartDesc &sd, int s, int cell,
                                                   int *__restrict__ changed,
                                                   const CellEntry *__restrict__ entries,
                                                   int nentries, int ra, int np, int lane, float defer_margin,
                                                   const HandOff &ho = HandOff{})
{
    const int ea = sd.box_hi[0] - sd.box_lo[0] + 1;
    const int eb = sd.box_hi[1] - sd.box_lo[1] + 1;
    const int ec = sd.box_hi[2] - sd.box_lo[2] + 1;
    if (ea <= 0 || eb <= 0 || ec <= 0 || cell >= ea * eb * ec) return;
    const int c = sd.box_lo[2] + cell % ec; cell /= ec;
    const int b = sd.box_lo[1] + cell % eb; cell /= eb;
    const int a = sd.box_lo[0] + cell;

    float *__restrict__ T = sd.T;
    const long long ci = dev_index(L, a, b, c);
    const bool c_is_start = (ci == sd.sidx);
    const float vc = v[ci];
    const float told = T[ci];
    float best = told;
    // (eight entries per lane at a time, all their loads in flight together: this runs at the
    // head of every pass)
    constexpr int U = 8;
    for (int e0 = 0; e0 < nentries; e0 += 64 * U) {
        CellEntry en[U];
        float vo[U], to[U];
#pragma unroll
        for (int u = 0; u < U; u++) en[u] = entries[min(e0 + u * 64 + lane, nentries - 1)];
#pragma unroll
        for (int u = 0; u < U; u++) {
            vo[u] = v[ci + en[u].delta];
            to[u] = T[ci + en[u].delta];
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const long long oi = ci + en[u].delta;
            const bool live = e0 + u * 64 + lane < nentries
                           && (((en[u].flags & PULL_FWD) && !c_is_start) || ((en[u].flags & PULL_REV) && oi != sd.sidx));
            const float sum = vc + vo[u];
            const float delay = en[u].h * sum;
            const float cand = delay + to[u];
            if (live && cand < best) best = cand;
        }
    }
#pragma unroll
    for (int w = 32; w >= 1; w >>= 1) best = fminf(best, __shfl_xor(best, w));
    const bool better = best < told;        // (wave-uniform: `best` is the wave's minimum)
    if (lane == 0 && better) {
        if (ASYNC) __hip_atomic_store(&T[ci], best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (write-through)
        else T[ci] = best;
        atomicOr(&changed[s], CHANGED_IMPROVED);
    }
    if (better) {
        // (one launch per solve: the store has arrived before anybody is told about it)
        if (ASYNC) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int btiles = strip_btiles(L), cstrips = strip_cstrips(L);
        unsigned *const pend = reinterpret_cast<unsigned *>(sd.tile_flags + 2 * strip_flag_words(L));
        const DeferRule rule{sd.sa, sd.sb, sd.sc, defer_margin};
        unsigned *const defer = reinterpret_cast<unsigned *>(sd.tile_flags);
        if (np == 1) push_improved<1, ASYNC>(L, ra, btiles, cstrips, pend, a, b / STRIP_TB, c / STRIP_K, FLAG_ALL, lane, rule, defer, ho);
        else push_improved<2, ASYNC>(L, ra, btiles, cstrips, pend, a, b / STRIP_TB, c / STRIP_K, FLAG_ALL, lane, rule, defer, ho);
    }
}

// Slab of one neighbour plane in LDS: `rows8` rows (rows rounded up to 8) of STRIP_W = 32
// floats for v and, SLAB_T_BYTES behind them, the same for T.  It is filled by LDS-DMA (global_load_lds_dwordx4: no
// VGPRs, asynchronous; one wave-instruction writes 1 KiB = 8 whole rows, lane l -> float4
// l % 8 of row l / 8), so the rows cannot be padded against bank conflicts; instead float4 j
// of row r is stored at position j ^ ((r >> 1) & 7): 16 consecutive lanes reading the same
// float4 of 16 consecutive rows then hit all 64 banks (the swizzle is applied to the SOURCE
// address here and to the read address in the window load).
__device__ __forceinline__ int slab_swizzle(int row) { return (row >> 1) & 7; }

//
// The loads are buffer loads whose resource starts at the slab's first element (a 64-bit
// scalar add per plane and array): the 8-row group goes into the scalar offset, the per-lane
// byte offset is the same few values for every plane and group, so an instruction costs no
// vector address arithmetic, and all offsets stay far below 2^31 whatever the volume size.
typedef __amdgpu_buffer_rsrc_t buf_rsrc;

__device__ __forceinline__ buf_rsrc make_rsrc(const float *base)
{
    // raw buffer, no stride, no range limit; word 3 as for gfx90a/gfx94x/gfx950 raw buffers
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base), 0, 0xffffffff, 0x00020000);
}

template <int NS>
__device__ __forceinline__ void stage_slab(const float *__restrict__ v_slab, const float *__restrict__ t_slab,
                                           unsigned s1_bytes, float *slab, int rows, int rows8,
                                           int wave, int lane)
{
    static_assert(NS % 2 == 0, "a wave's 8-row groups have one parity");
    const buf_rsrc rv = make_rsrc(v_slab), rt = make_rsrc(t_slab);
    const int ninstr = rows8 / 8;           // wave-instructions per array
    const int rloc = lane >> 3, p = lane & 7;
    // Wave w loads the 8-row groups kk = w, w + NS, ... of both arrays.  Row r = 8 kk + rloc
    // has swizzle ((r >> 1) & 7) = (rloc >> 1) ^ (4 (kk & 1)), and kk & 1 = w & 1 for all of
    // a wave's groups, so one per-lane byte offset serves all its loads (but the slab's last
    // group, whose rows past the slab re-read its last row).
    const unsigned lane_row = (unsigned)rloc * s1_bytes;
    const unsigned voff = lane_row + (unsigned)((p ^ (rloc >> 1) ^ ((wave & 1) << 2)) << 4);
    const unsigned voff_last = voff - (unsigned)max((ninstr - 1) * 8 + rloc - (rows - 1), 0) * s1_bytes;
    const unsigned group_bytes = 8 * s1_bytes;
    unsigned soff = (unsigned)wave * group_bytes;
    float *dst = slab + wave * (8 * STRIP_W);                               // wave-uniform
    for (int kk = wave; kk < ninstr; kk += NS) {
        const int vo = (int)(kk == ninstr - 1 ? voff_last : voff);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, (__attribute__((address_space(3))) void *)dst,
                                                 16, vo, (int)soff, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rt, (__attribute__((address_space(3))) void *)(dst + SLAB_T_BYTES / 4),
                                                 16, vo, (int)soff, 0, 0);
        soff += NS * group_bytes;
        dst += NS * 8 * STRIP_W;
    }
}

// -DTTSWEEP_PROFILE: cycle counts of wave 0 per phase, summed over all units (tuning aid)
#ifdef TTSWEEP_PROFILE
__device__ unsigned long long g_prof[10];
#define PROF_T(x) const long long x = clock64()
void prof_dump()
{
    unsigned long long h[10] = {};
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_prof), sizeof(h));
    const double n = (double)std::max<unsigned long long>(h[6], 1);
    fprintf(stderr, "prof (wave 0, cycles per unit, %llu units): fetch %.0f  prologue %.0f  wait %.0f (of it for the loads %.0f, %.2f group boundaries)  stage %.0f  "
            "compute %.0f  epilogue %.0f  | busy / resident cycles of the workgroups: %.3f\n",
            h[6], h[0] / n, h[1] / n, h[2] / n, h[8] / n, h[9] / n, h[3] / n, h[4] / n, h[5] / n,
            (double)(h[0] + h[1] + h[2] + h[3] + h[4] + h[5]) / (double)std::max<unsigned long long>(h[7], 1));
    unsigned long long z[10] = {};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_prof), z, sizeof(z));
}
#else
#define PROF_T(x)
#endif

#ifndef TTSWEEP_WGS_PER_CU
#define TTSWEEP_WGS_PER_CU 2        // persistent workgroups per CU (measured optimum)
#endif
#ifndef TTSWEEP_WGS_GRID
#define TTSWEEP_WGS_GRID TTSWEEP_WGS_PER_CU
#endif
int units_wgs_per_cu() { return TTSWEEP_WGS_GRID; }

// ---------------------------------------------------------------------------
// one launch per solve (AsyncSolve, ttsweep_dev.h): ring planner and ring claims
// ---------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long ald64(const unsigned long long *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned ald32(const unsigned *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void async_fail(const AsyncSolve &as, unsigned code)
{
    atomicCAS(as.status, (unsigned)ASYNC_OK, code);
}

// Planner of ring r: all 64 NS threads of one workgroup.  `lds`: scratch words of its own.
template <int NP, int NS>
__device__ __forceinline__ void async_planner(const DevLayout &L, const StartDesc *__restrict__ starts, const AsyncSolve &as,
                                           const int r, const int btiles, const int cstrips, const int ra,
                                           const PlaneCounts &pc, int *__restrict__ flags0, const long long flags_stride,
                                           int *lds)
{
    const int tid = threadIdx.y * STRIP_TB + threadIdx.x, lane = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.y);
    const int4 *__restrict__ list = as.list + as.ring_off[r];
    const int n = as.ring_len[r];
    unsigned long long *const ht = as.ctl + (size_t)r * ASYNC_CTL_STRIDE;
    unsigned *const gate_word = reinterpret_cast<unsigned *>(ht + 8);
    unsigned *const completed = reinterpret_cast<unsigned *>(ht + 16);
    unsigned long long *const ents = as.entries + (size_t)r * (size_t)(as.cap_mask + 1);
    const int ns = as.ring_start_off[r + 1] - as.ring_start_off[r];
    const int *__restrict__ rstarts = as.ring_starts + as.ring_start_off[r];
    const int nflag = L.n[0] * btiles * cstrips;
    const bool handoff = as.handoff != 0;       // workers publish too: positions are reserved with a returning add
    constexpr int KSCAN = 4;                    // list entries per thread and scan step
    constexpr int NT = STRIP_TB * NS;           // threads of the workgroup
    int *sh = lds;                              // [0] head, [1] completed, [2] stop, [3] dead-edge entries just published,
                                                // [4] first list position this round met with anything to do,
                                                // [5] first position of a reservation, [6] tail
    int *cnt = lds + 8;                         // [KSCAN][waves] due units found
    int *dirty = lds + 8 + KSCAN * NS;    // [ASYNC_RING_STARTS] units published since the start's last special
    int *minact = dirty + ASYNC_RING_STARTS;    // [ASYNC_RING_STARTS] smallest squared distance (float bits) met with work, this round
    float *gate2 = reinterpret_cast<float *>(minact + ASYNC_RING_STARTS);      // [ASYNC_RING_STARTS] squared gate radius per start
    // (the gate is for boxes that grow from ONE source patch - what the initialisation counted -: a box
    // that arrives with values all over it, e.g. a converged one, is relaxed ungated)
    int *gated = reinterpret_cast<int *>(gate2 + ASYNC_RING_STARTS);
    if (tid < ASYNC_RING_STARTS) {
        gated[tid] = tid < ns ? flags0[(long long)rstarts[tid] * flags_stride + 3 * nflag] == 1 : 0;
        dirty[tid] = tid < ns ? 1 : 0;          // (the first thing a ring does: its starts' dead-edge cells)
        minact[tid] = 0;                        // (the first round: the window around the start itself)
        if (tid == 0) sh[4] = 0x7fffffff;
        gate2[tid] = 3.0e38f;
    }
    __syncthreads();
    unsigned t = 0;                             // entries published so far (uniform; with hand-off: as last read)
    unsigned t_special = 0xffffffffu;           // (hand-off) the tail right after the dead-edge cells' last turn at rest
    const long long clock0 = wall_clock64();
    // state of the scan in progress (uniform but `activity`): where it continues, which round it belongs to,
    // whether the ring was at rest when it began, whether it met a word that was not zero
    int base = n, round = -1, activity = 0;
    bool quiet = false, from_zero = true;
    // bit 63 of the entries published from the first flush of deferred bits on: the workers then tell every unit
    // at once.  If the geometry misled (the late relaxations improved cells), what follows is an ordinary solve
    // from a good state, not a chain of deferrals and flushes.
    unsigned long long nodefer = 0ull;
    float gate_r2 = 3.0e38f, gate_r = as.gate_r0;

    // an entry goes into its slot when the slot is empty: it is, unless the ring has been lapped while a worker sat
    // on the entry that was there (ttsweep_dev.h, ring entry)
    auto store_entry = [&](unsigned long long *slot, unsigned long long e) {
        for (unsigned spin = 0; slot_load(slot) != 0ull; spin++) {
            if ((spin & 63u) == 63u && (ald32(as.status) != ASYNC_OK || wall_clock64() - clock0 > as.timeout_ticks)) break;
            __builtin_amdgcn_s_sleep(2);
        }
        __hip_atomic_store(slot, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef TTSWEEP_RING_DEBUG
        atomicAdd(ht + 24, 1ull);
#endif
    };
    // positions of `total` entries (uniform; every thread calls): without hand-off the planner is the ring's only
    // producer - the positions follow its own count and the tail moves when the entries are in memory (commit); with
    // hand-off they are reserved at once (a worker that claims one before its entry has arrived waits at the slot)
    auto reserve = [&](int total) -> unsigned {
        if (!handoff) return t;
        if (tid == 0) sh[5] = total ? (int)((unsigned)(atomicAdd(ht, (unsigned long long)total << 32) >> 32) & 0x7fffffffu) : (int)t;
#ifdef TTSWEEP_RING_DEBUG
        if (tid == 0 && total) atomicAdd(ht + 28, (unsigned long long)total);
        if (tid == 0 && total < 0) atomicAdd(ht + 29, 1ull);
#endif
        __syncthreads();
        const unsigned b = (unsigned)sh[5];
        __syncthreads();
        return b;
    };
    auto commit = [&](unsigned first, int total) {
        if (!handoff) {
            // the entries are in memory before the tail says so
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0 && total) atomicAdd(ht, (unsigned long long)total << 32);
        }
        t = first + (unsigned)total;
    };

    // publishes the dead-edge entries of the ring's starts whose counter has reached `threshold`
    auto publish_specials = [&](int threshold) -> int {
        // (the counters are what the scan's last step left: every wave has added its share - with hand-off nothing
        // else stands between those adds and this read; and who is due is decided ONCE: a counter that reached the
        // threshold between a count and a second look would get an entry at a position nobody reserved)
        __syncthreads();
        bool need = false;
        unsigned long long bal = 0ull;
        if (wave == 0) {
            need = lane < ns && dirty[lane] >= threshold;
            bal = __ballot(need);
            if (lane == 0) sh[3] = __popcll(bal);
        }
        __syncthreads();
        const int total = sh[3];
        __syncthreads();
        const unsigned first = reserve(total);
        if (wave == 0 && need) {
            const unsigned pos = first + (unsigned)__popcll(bal & ((1ull << lane) - 1ull));
            const unsigned long long e = 0xffffull | ((unsigned long long)ASYNC_UNIT_SPECIAL << 16)
                                       | ((unsigned long long)rstarts[lane] << 36)
                                       | ((unsigned long long)(pos & ASYNC_TAG_MASK) << 44) | ASYNC_ENTRY_VALID | nodefer;
            store_entry(ents + (pos & (unsigned)as.cap_mask), e);
            dirty[lane] = 0;
        }
        commit(first, total);
        __syncthreads();
        return total;
    };

    for (;;) {
        // ---- wait until the ring has room.  (A slot is never written before the entry it held has been read - see
        // store_entry -, so the bound on published - completed is a throttle, not what keeps the ring consistent.)
        unsigned h, c;
        for (;;) {
            if (tid == 0) {
                // completed first, then the tail: completed(then) == tail(now) means nothing is queued or running now
                // (with hand-off a running worker may publish; both counts only grow, completed never beyond the tail)
                sh[1] = (int)ald32(completed);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const unsigned long long x = ald64(ht);
                sh[0] = (int)(unsigned)x;
                sh[6] = (int)((unsigned)(x >> 32) & 0x7fffffffu);
                int stop = ald32(as.status) != ASYNC_OK;
                if (!stop && wall_clock64() - clock0 > as.timeout_ticks) { async_fail(as, ASYNC_ERR_TIMEOUT); stop = 1; }
                if (!stop && (long long)(handoff ? (unsigned)sh[6] : t) > as.max_entries) { async_fail(as, ASYNC_ERR_CAP); stop = 1; }
                sh[2] = stop;
            }
            __syncthreads();
            h = (unsigned)sh[0]; c = (unsigned)sh[1];
            if (handoff) t = (unsigned)sh[6];
            const int stop = sh[2];
            __syncthreads();
            if (stop) {         // give up: the workers see the status word and leave
                if (tid == 0) atomicOr(ht, 1ull << 63);
                return;
            }
            if ((int)(t - h) <= as.low && (int)(t - c) < as.cap_mask - 2048) break;
            __builtin_amdgcn_s_sleep(32);
        }
        if (as.policy == 0 || base >= n) {      // a scan begins
            if (base >= n) round++;
            // Where: at the first list position the previous round met with anything to do, less a
            // margin (what lies in front of it is at rest, but for a stray bit a neighbour may have
            // set since) - on long lists most of a scan would otherwise cross units that have long
            // converged; from the very beginning every 16th round, after a round that met nothing,
            // and therefore always before the ring is declared at rest.
            __syncthreads();
            const int fa = sh[4];
            __syncthreads();
            if (tid == 0) sh[4] = 0x7fffffff;
            base = 0;
            if (as.policy != 0 && fa < n && (round & 15) != 0) base = max(fa - as.scan_slack, 0) / (KSCAN * NT) * (KSCAN * NT);
            from_zero = base == 0;
            quiet = c == t;                     // nothing queued, nothing running: the scan sees a still picture
            activity = 0;
            // the gate opens by gate_speed cells per round while the workers have enough to do, by
            // gate_fast when they are running dry (everything published has been claimed): few starts
            // cannot fill the machine from behind a slow gate, many starts waste work behind a fast one
            if (round > 0) gate_r += (int)(t - h) <= 0 ? as.gate_fast : as.gate_speed;
            gate_r2 = as.policy == 1 && as.gate_speed > 0.f ? gate_r * gate_r : 3.0e38f;
            // (the workers that hand units on keep inside the same gate)
            if (handoff && tid == 0) __hip_atomic_store(gate_word, __float_as_uint(gate_r2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (as.policy == 2) {
                // window: a start's units are handed out up to `window` cells beyond the nearest unit
                // the previous round met with anything to do (everything, if it met none)
                __syncthreads();
                if (tid < ASYNC_RING_STARTS) {
                    const float m = __int_as_float(minact[tid]);
                    const float gw = sqrtf(m) + as.window;
                    gate2[tid] = m < 1.0e30f && as.window > 0.f ? gw * gw : 3.0e38f;
                    minact[tid] = 0x7f000000;
                }
                __syncthreads();
            }
        }
        for (; base < n; base += KSCAN * NT) {
            if ((int)(t - h) >= as.high) break;
            int ss[KSCAN], uu[KSCAN], dd[KSCAN];
            unsigned w[KSCAN], planes[KSCAN];
            unsigned *pw[KSCAN];
            int open_any = 0;
#pragma unroll
            for (int k = 0; k < KSCAN; k++) {
                const int idx = base + k * NT + tid;
                int4 it = make_int4(0, -1, 0, 0);
                if (idx < n) it = list[idx];
                if (gated[it.x >> 16] && __int_as_float(it.z) > (as.policy == 2 ? gate2[it.x >> 16] : gate_r2)) {   // behind the gate: not even looked at (the round ends
                    if (it.y >= 0) activity = 1;        // where a whole step lies behind it); may hold bits
                    it.y = -1;
                } else if (it.y >= 0) {
                    open_any = 1;
                }
                ss[k] = it.x; uu[k] = it.y; dd[k] = it.z;
                pw[k] = reinterpret_cast<unsigned *>(flags0 + (long long)(it.x & 0xffff) * flags_stride + 2 * nflag) + max(it.y, 0);
            }
            if (!__syncthreads_or(open_any)) {          // (the list is in order of distance, nearly: what is
                base = n;                               // missed now is met in a later round)
                break;
            }
#pragma unroll
            for (int k = 0; k < KSCAN; k++) w[k] = uu[k] >= 0 ? atomicOr(pw[k], 0u) : 0u;
            {
                bool anyw = false;
#pragma unroll
                for (int k = 0; k < KSCAN; k++) anyw |= w[k] != 0u;
                if (__ballot(anyw) != 0ull && lane == 0) atomicMin(&sh[4], base);
            }
#pragma unroll
            for (int k = 0; k < KSCAN; k++) {
                planes[k] = 0;
                if (w[k] != 0u) {
                    activity = 1;
                    if (as.policy == 2) atomicMin(&minact[ss[k] >> 16], dd[k]);
                    // without hand-off only the planner sets the busy bit: a word seen without it holds plane bits
                    // alone and stays that way until the exchange; with hand-off a worker may take the unit in
                    // between: compare-and-swap, the loser leaves the bits where they are
                    if (!(w[k] & ASYNC_BUSY))
                        planes[k] = handoff ? handoff_take(pw[k], w[k]) : atomicExch(pw[k], ASYNC_BUSY) & ~ASYNC_BUSY;
                }
            }
            int rank[KSCAN];
#pragma unroll
            for (int k = 0; k < KSCAN; k++) {
                const unsigned long long bal = __ballot(planes[k] != 0u);
                rank[k] = __popcll(bal & ((1ull << lane) - 1ull));
                if (lane == 0) cnt[k * NS + wave] = __popcll(bal);
            }
            __syncthreads();
            int total = 0, off[KSCAN];
#pragma unroll
            for (int k = 0; k < KSCAN; k++) {
#pragma unroll
                for (int ww = 0; ww < NS; ww++) {
                    if (ww == wave) off[k] = total;
                    total += cnt[k * NS + ww];
                }
            }
            const unsigned first = reserve(total);
#pragma unroll
            for (int k = 0; k < KSCAN; k++) {
                unsigned long long relax = 0;
                const int s = ss[k] & 0xffff;
                if (planes[k] != 0u) {
                    const unsigned pos = first + (unsigned)(off[k] + rank[k]);
                    const unsigned long long e = (unsigned long long)planes[k] | ((unsigned long long)(unsigned)uu[k] << 16)
                                               | ((unsigned long long)s << 36) | ((unsigned long long)(pos & ASYNC_TAG_MASK) << 44)
                                               | ASYNC_ENTRY_VALID | nodefer;
                    store_entry(ents + (pos & (unsigned)as.cap_mask), e);
                    atomicAdd(&dirty[ss[k] >> 16], 1);
                    if (!handoff) {
                        int u = uu[k];
                        const int cs = u % cstrips;  u /= cstrips;
                        const int bt = u % btiles;   u /= btiles;
                        const int wb = min(min(STRIP_TB, L.n[1]), L.n[1] - bt * STRIP_TB), wc = max(min(STRIP_K, L.n[2] - cs * STRIP_K), 0);
                        const bool two = NP > 1 && NP * u + 1 < L.n[0];
                        int nent = 0;
                        for (int p = 0; p < 2 * ra + NP; p++)
                            if ((planes[k] >> p) & 1u) nent += pc.n[p][0] + (two ? pc.n[p][1] : 0);
                        relax = (unsigned long long)(wb * wc) * (unsigned long long)nent;
                    }
                }
                // the starts' work counters: one pair of atomics per wavefront and start (with hand-off the workers
                // count what they relax themselves: the planner does not see every entry)
                unsigned long long rest = handoff ? 0ull : __ballot(planes[k] != 0u);
                while (rest) {
                    const int first_lane = __builtin_ctzll(rest);
                    const int s0 = __shfl(s, first_lane);
                    const bool mine = planes[k] != 0u && s == s0;
                    const unsigned long long mm = __ballot(mine);
                    unsigned long long sum = mine ? relax : 0ull;
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
                    if (lane == first_lane) {
                        atomicAdd(starts[s0].work, sum);
                        atomicAdd(starts[s0].work + 2, (unsigned long long)__popcll(mm));
                    }
                    rest &= ~mm;
                }
            }
            commit(first, total);
        }
        const bool whole = base >= n;
        if (whole) activity = __syncthreads_or(activity);
        if (whole && !activity && quiet && from_zero) {
            // the ring is at rest; the dead-edge cells once more if units ran since their last turn (with hand-off
            // the planner has not seen every unit: whatever was published since that turn counts for every start)
            if (handoff && t != t_special) {
                __syncthreads();
                if (tid < ns) dirty[tid] = max(dirty[tid], 1);
                __syncthreads();
            }
            const int sp = publish_specials(1);
            if (handoff) t_special = t;         // (their own entries included)
            if (sp == 0) {
                // really at rest: the deferred bits (push_improved) become pend bits; none: done
                int moved = 0;
                for (int idx = tid; idx < n; idx += NT) {
                    const int4 it = list[idx];
                    if (it.y < 0) continue;
                    unsigned *const df = reinterpret_cast<unsigned *>(flags0 + (long long)(it.x & 0xffff) * flags_stride) + it.y;
                    const unsigned d = atomicExch(df, 0u);
                    if (d) { atomicOr(df + 2 * nflag, d); moved = 1; }
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (!__syncthreads_or(moved)) {
                    if (tid == 0) atomicOr(ht, 1ull << 63);
                    return;
                }
                nodefer = 1ull << 63;
            }
        } else {
            (void)publish_specials(as.special_every);
        }
    }
}

// Worker, thread 0: the next ring entry (own ring first), or ASYNC_EXIT when every ring is done
// or the solve has failed.  *ring: the ring the entry came from.
__device__ __forceinline__ unsigned long long async_claim(const AsyncSolve &as, const int home, int *ring, const long long clock0)
{
    for (unsigned spin = 0;; spin++) {
        int ndone = 0;
        for (int probe = 0; probe < as.nrings; probe++) {
            const int q = (home + probe) % as.nrings;
            unsigned long long *const ht = as.ctl + (size_t)q * ASYNC_CTL_STRIDE;
            const unsigned long long x = ald64(ht);
            const unsigned head = (unsigned)x, tail = (unsigned)(x >> 32) & 0x7fffffffu;
            if ((int)(tail - head) > 0) {
                const unsigned j = (unsigned)atomicAdd(ht, 1ull);
                const unsigned long long *slot = as.entries + (size_t)q * (size_t)(as.cap_mask + 1) + (j & (unsigned)as.cap_mask);
                for (unsigned spin2 = 0;; spin2++) {
                    const unsigned long long e = slot_load(slot);
                    if ((e & ASYNC_ENTRY_VALID) && ((unsigned)(e >> 44) & ASYNC_TAG_MASK) == (j & ASYNC_TAG_MASK)) {
                        // read: the slot is free for the position one lap on
                        __hip_atomic_store(const_cast<unsigned long long *>(slot), 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef TTSWEEP_RING_DEBUG
                        atomicAdd(ht + 26, 1ull);
#endif
                        *ring = q;
                        return e;
                    }
                    // not published (yet): somebody else was faster and position j lies beyond the tail
                    const unsigned long long y = ald64(ht);
                    if ((y >> 63) && (int)(j - ((unsigned)(y >> 32) & 0x7fffffffu)) >= 0) break;   // ... and never will be
                    if ((spin2 & 63u) == 63u) {
                        if (ald32(as.status) != ASYNC_OK) return ASYNC_EXIT;
                        if (wall_clock64() - clock0 > as.timeout_ticks) { async_fail(as, ASYNC_ERR_TIMEOUT); return ASYNC_EXIT; }
                    }
                    __builtin_amdgcn_s_sleep(8);
                }
            } else if (x >> 63) {
                ndone++;
            }
        }
        if (ndone == as.nrings) return ASYNC_EXIT;
        if ((spin & 15u) == 15u) {
            if (ald32(as.status) != ASYNC_OK) return ASYNC_EXIT;
            if (wall_clock64() - clock0 > as.timeout_ticks) { async_fail(as, ASYNC_ERR_TIMEOUT); return ASYNC_EXIT; }
        }
        __builtin_amdgcn_s_sleep(24);
    }
}

// NS waves per workgroup split the items of a staged plane (the plan's NS shares); G staged planes per GROUP: the
// slabs of a group are asked for together, waited for together (one vmcnt(0) + barrier per group) and relaxed one
// after the other while the next group loads into the other half of the 2 G slabs.
// Throughput instances: NS = 4, two workgroups per CU, G = 1 (a plane at a time: 40 KB of slabs per workgroup).
// Latency instance (small shards, one-plane units): NS = 8, ONE workgroup per CU - a unit is relaxed by all eight
// waves of a CU at once: a wave does half the arithmetic (compute 45.5 k -> 20-23 k cycles per unit) -, G = 3: eight
// waves split a plane's ten or so items unevenly (the longest share of a plane is 1.3 times the mean, 1.1 with four
// waves) and meet at a barrier twice as often per unit of work; with three planes between two barriers - the shares
// rotated from plane to plane - a unit's waves wait 10 k instead of 17 k cycles, and a group's compute covers the load
// latency of the next (profiles/r05_lat_*.txt; two teams of four waves with half a strip each - the balance of four
// shares, three quarters of the LDS reads for half of the relaxations - came out slower: r05_lat_teams_*.txt).
template <int K, int NP, bool ASYNC, int NS, int G>
__global__ void __launch_bounds__(STRIP_TB *NS, (NS == STRIP_NS ? TTSWEEP_WGS_PER_CU : 1))
sweep_units_kernel(DevLayout L, const float *__restrict__ v, const StartDesc *__restrict__ starts,
                   const int4 *__restrict__ lists, int list_cap, int nlists, int *__restrict__ ctrl,
                   int *__restrict__ changed, const StripItem *__restrict__ items, StripPlan plan,
                   int btiles, int cstrips, UnitPassTail tail, AsyncSolve as, PlaneCounts pc,
                   int *__restrict__ flags0, long long flags_stride)
{
    constexpr int STRIP_LDS_HEAD = strip_lds_head(NS);
    constexpr int W = K + 2 * STRIP_CF;
    static_assert(K == STRIP_K && STRIP_W == 32 && W == 32, "slab rows are 8 float4 wide");
    static_assert(NP >= 1 && NP <= STRIP_PLANES && (NP * K) % NS == 0 && K % (NP * K / NS) == 0,
                  "every wave finishes a whole number of cells of one own plane");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int *head = reinterpret_cast<int *>(smem);      // [0], [1]: queue index handed to the workgroup

    const int lane = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.y);
    const int tid = wave * STRIP_TB + lane;
    const int rb = plan.rb;
    const int tb_eff = min(STRIP_TB, L.n[1]);
    const int rows = tb_eff + 2 * rb;
    const int lane_r = min(lane, tb_eff - 1);
    const int rows8 = (rows + 7) & ~7;
    constexpr int slab_floats = SLAB_BYTES / 4;     // v rows, then T rows
    const int nflag = L.n[0] * btiles * cstrips;
    float *slabs = smem + STRIP_LDS_HEAD;           // 2 G slabs: two groups of G staged planes
    float *comb = smem + STRIP_LDS_HEAD;            // [wave][plane][cell][lane], aliases the slabs
    const int nitems = plan.first[plan.nstaged];
    unsigned plane_mask = 0;                        // staged planes that have items at all
    for (int p = 0; p < plan.nstaged; p++)
        if (plan.first[p] != plan.first[p + 1]) plane_mask |= 1u << p;
    // item range [lo, hi) of wave w for staged plane p, packed lo | hi << 16, in LDS: read
    // once per plane with LDS latency instead of a chain of scalar memory loads
    int *item_range = reinterpret_cast<int *>(smem) + 16;
    if (tid < NS * 16) {
        const int w = tid >> 4, p = tid & 15;
        int packed = 0;
        // (G > 1: the shares of a plane are dealt to the waves rotated by the plane's index - the longest share of
        // every plane would otherwise go to wave 0, and a group's waits are for the SUM over its planes)
        const int sh = G > 1 ? (w + p) % NS : w;
        if (p < plan.nstaged)
            packed = (plan.first[p] + plan.wsplit[p][sh]) | ((plan.first[p] + plan.wsplit[p][sh + 1]) << 16);
        item_range[tid] = packed;
    }
    // the queue lengths are final when this kernel starts (the planner wrote them): read them
    // once - an empty queue then costs no memory round trip, and an empty pass none at all
    int *qcount = head + 2;                         // [0 .. nlists)
    if (!ASYNC && tid < UNITQ_LISTS) qcount[tid] = tid < nlists ? ctrl[tid] : 0;
#ifdef TTSWEEP_ASYNC_STATS
    if (tid == 0) head[12] = 0;
#endif
    if (tid == 0) { head[13] = 0; head[14] = 0; }
    __syncthreads();

    if (ASYNC) {
        // one launch per solve: the first workgroups plan (one ring each), the others work
        if ((int)blockIdx.x < as.nrings) {
            async_planner<NP, NS>(L, starts, as, (int)blockIdx.x, btiles, cstrips, plan.ra, pc, flags0, flags_stride,
                              reinterpret_cast<int *>(smem) + STRIP_LDS_HEAD);
            return;
        }
    } else {
        // ---- the dead-edge cells of the active starts, one wave per cell
        for (int w = blockIdx.x * NS + wave; w < tail.nactive * tail.max_box_cells; w += gridDim.x * NS) {
            const int s = tail.active[w / tail.max_box_cells];
            const StartDesc sd = starts[s];
            relax_special_cell(L, v, sd, s, w % tail.max_box_cells, changed, tail.entries, tail.nentries,
                               plan.ra, NP, lane, tail.defer_margin);
        }
    }

    PROF_T(t_k0);
    // own queue first: the one of the XCD this workgroup runs on (speed only: a start's
    // volumes then stay in one L2; any workgroup may drain any queue)
    const int home = (int)(xcc_id() % (unsigned)(ASYNC ? as.nrings : nlists));
    int probe = 0, it = 0;
    // The index of the unit after the current one is asked for while the current one is being
    // relaxed (`ahead`, held by thread 0); -1: nothing asked for yet.
    int ahead = -1;
    int flagged = -1;           // (lane 0 of a wave) the start whose "improved" bit this wave has set
    // (thread 0) relaxations of the in-unit passes, summed until the workgroup turns to another start: the planner
    // counts what it hands out, these it does not see
    unsigned long long extra_relax = 0, extra_units = 0;
    int extra_s = -1;
    const long long async_clock0 = ASYNC ? wall_clock64() : 0;
#ifdef TTSWEEP_PROFILE
    unsigned long long prof_acc[7] = {}, prof_vm = 0, prof_groups = 0;
#endif
    // (hand-off: made where it is used - at the end of a unit -, so that nothing of it lives through the relaxation)
    auto make_handoff = [&](int s, int q, float defer_margin) -> HandOff {
        HandOff ho{};
        if (ASYNC && as.handoff) {
            // a start that does not grow from one source patch is relaxed ungated (async_planner: gated[])
            const bool gated = __builtin_amdgcn_readfirstlane(flags0[(long long)s * flags_stride + 3 * nflag]) == 1;
            ho.on = as.handoff;
            ho.gate_r2 = gated ? __uint_as_float((unsigned)__builtin_amdgcn_readfirstlane(head[3])) : 3.0e38f;
            ho.ht = as.ctl + (size_t)q * ASYNC_CTL_STRIDE;
            ho.ents = as.entries + (size_t)q * (size_t)(as.cap_mask + 1);
            ho.cap_mask = (unsigned)as.cap_mask;
            // ("tell every unit at once": the entries after the ring's first flush - or no deferral at all)
            ho.tag_bits = ((unsigned long long)s << 36) | (defer_margin < -1.0e30f ? 1ull << 63 : 0ull);
            ho.status = as.status;
            ho.deadline = async_clock0 + as.timeout_ticks;
        }
        return ho;
    };
    while (ASYNC || probe < nlists) {
        // ---- take the next unit of queue q (every wave leaves through the same exit:
        // all queues exhausted)
        PROF_T(t_top);
        int s, my_unit, q;
        unsigned my_planes;
        float defer_margin = tail.defer_margin;
        if (ASYNC) {
            // a ring entry; what it stands for was released by its producers before the planner
            // could see their bits: acquire before anything of it is loaded
            if (tid == 0) {
                int ring = 0;
                const unsigned long long e = async_claim(as, home, &ring, async_clock0);
                // (hand-off: the gate of the ring's planner, as it stands now - it only ever opens)
                unsigned gate_bits = 0u;
                if (as.handoff) gate_bits = ald32(reinterpret_cast<const unsigned *>(as.ctl + (size_t)ring * ASYNC_CTL_STRIDE + 8));
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                head[0] = (int)(unsigned)e;
                head[1] = (int)(unsigned)(e >> 32);
                head[2] = ring;
                head[3] = (int)gate_bits;
                head[4] = 0; head[5] = 0;       // (hand-off: what the waves improved, per own plane)
            }
            __syncthreads();
            const unsigned e_lo = (unsigned)__builtin_amdgcn_readfirstlane(head[0]);
            const unsigned e_hi = (unsigned)__builtin_amdgcn_readfirstlane(head[1]);
            q = __builtin_amdgcn_readfirstlane(head[2]);
            if ((e_lo & e_hi) == 0xffffffffu) break;        // ASYNC_EXIT
            my_planes = e_lo & 0xffffu;
            my_unit = (int)((e_lo >> 16) | ((e_hi & 0xfu) << 16));
            s = (int)((e_hi >> 4) & 0xffu);
            defer_margin = (e_hi >> 31) ? -3.0e38f : tail.defer_margin;     // (entries after the ring's first flush)
            if ((unsigned)my_unit == ASYNC_UNIT_SPECIAL) {
                // the dead-edge cells of start s, one wave per cell
                const StartDesc sd = starts[s];
                const HandOff ho = make_handoff(s, q, defer_margin);
                for (int cell = wave; cell < tail.max_box_cells; cell += NS)
                    relax_special_cell<true>(L, v, sd, s, cell, changed, tail.entries, tail.nentries, plan.ra, NP, lane, defer_margin, ho);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (tid == 0) atomicAdd(reinterpret_cast<unsigned *>(as.ctl + (size_t)q * ASYNC_CTL_STRIDE + 16), 1u);
                continue;
            }
        } else {
        q = (home + probe) % nlists;
        const int n = __builtin_amdgcn_readfirstlane(qcount[q]);
        if (n == 0) { probe++; continue; }          // (uniform: nothing was asked of an empty queue)
        if (tid == 0) {
            int j = ahead;
            if (j < 0) j = atomicAdd(&ctrl[UNITQ_LISTS + q], 1);
            head[it & 1] = j;
        }
        __syncthreads();
        const int j = __builtin_amdgcn_readfirstlane(head[it & 1]);
        it++;
        ahead = -1;
        if (j >= n) { probe++; continue; }
        if (tid == 0) ahead = atomicAdd(&ctrl[UNITQ_LISTS + q], 1);     // (arrives during the unit)
        const const_entry_ptr entry = (const_entry_ptr)(lists + ((size_t)q * list_cap + j));
        s = entry->s;
        my_unit = entry->unit;
        my_planes = (unsigned)entry->planes;
        }
        PROF_T(t_fetch);
#ifdef TTSWEEP_PROFILE
        long long p_wait = 0, p_stage = 0, p_comp = 0, p_vm = 0, p_groups = 0;
#endif
        int u = my_unit;
        const int cs = u % cstrips;  u /= cstrips;
        const int bt = u % btiles;   u /= btiles;
        const int a0 = NP * u;                      // first own plane
        const int b0 = bt * STRIP_TB;
        const int c0 = cs * K;

        const const_start_ptr sdp = (const_start_ptr)(starts + s);
        float *const T = sdp->T;
        int *const tile_flags = sdp->tile_flags;
        const int box_lo0 = sdp->box_lo[0], box_lo1 = sdp->box_lo[1], box_lo2 = sdp->box_lo[2];
        const int box_hi0 = sdp->box_hi[0], box_hi1 = sdp->box_hi[1], box_hi2 = sdp->box_hi[2];
        // ---- staged planes to relax against, in order; the first one starts to load right
        // away.  Slab of staged plane p (plane a0 - ra + p): rows b0-rb .. b0+63+rb, columns
        // c0-CF .. c0+K+CF-1.
        unsigned todo = my_planes & plane_mask;
        int improved_all = 0, inunit_passes = 0;
        const long long src0 = (long long)(a0 - plan.ra + L.lo[0]) * L.s0
                             + (long long)(b0 - rb + L.lo[1]) * L.s1 + (c0 - STRIP_CF + L.lo[2]);
        const unsigned s1_bytes = (unsigned)(L.s1 * 4);
        // Groups of G staged planes: `unissued` are the planes whose slabs have not been asked for yet (always the
        // groups behind the one being relaxed), `half` the half of the slabs the group being relaxed lies in,
        // `in_group` the plane's position in its group.
        unsigned unissued = todo;
        int half = 0, in_group = 0;
        auto issue_group = [&](int into_half) {
#pragma unroll
            for (int i = 0; i < G; i++) {
                if (unissued) {
                    const long long src = src0 + (long long)__builtin_ctz(unissued) * L.s0;
                    stage_slab<NS>(v + src, T + src, s1_bytes, slabs + (into_half * G + i) * slab_floats, rows, rows8, wave, lane);
                    unissued &= unissued - 1;
                }
            }
        };
        issue_group(0);
        // header of the first item this wave will relax (later ones are requested one item
        // ahead, across plane boundaries)
        ItemHdr cur = load_hdr(items, min(__builtin_amdgcn_readfirstlane(
            item_range[wave * 16 + (todo ? __builtin_ctz(todo) : 0)]) & 0xffff, nitems - 1));

        // own cells: (a0 + j, b0 + lane, c0 + q)
        const long long own = (long long)(a0 + L.lo[0]) * L.s0 + (long long)(b0 + lane_r + L.lo[1]) * L.s1
                            + (c0 + L.lo[2]);
        float acc[NP][K];
        f32x2 vce[NP][K / 2];
        f32x2 vco[NP][K / 2 - 1];
#pragma unroll
        for (int jp = 0; jp < NP; jp++) {
#pragma unroll
            for (int jj = 0; jj < K / 4; jj++) {
                const float4 xx = *reinterpret_cast<const float4 *>(v + own + jp * L.s0 + 4 * jj);
                const float4 yy = *reinterpret_cast<const float4 *>(T + own + jp * L.s0 + 4 * jj);
                vce[jp][2 * jj] = f32x2{xx.x, xx.y}; vce[jp][2 * jj + 1] = f32x2{xx.z, xx.w};
                acc[jp][4 * jj + 0] = yy.x; acc[jp][4 * jj + 1] = yy.y;
                acc[jp][4 * jj + 2] = yy.z; acc[jp][4 * jj + 3] = yy.w;
            }
#pragma unroll
            for (int p = 0; p < K / 2 - 1; p++) vco[jp][p] = f32x2{vce[jp][p].y, vce[jp][p + 1].x};
        }
        // what this wave will finish and store at the end: CQ consecutive cells of one own
        // plane; their values before this unit's relaxation stay in registers
        constexpr int CQ = NP * K / NS;     // cells per wave in the epilogue
        const int fin_plane = wave * CQ / K, fin_q0 = wave * CQ % K;
        float told[CQ];
#pragma unroll
        for (int qq = 0; qq < CQ; qq++) {   // (wave-uniform selects between compile-time registers)
#define ACC_FLAT(e) acc[(e) / K][(e) % K]
            float tv = ACC_FLAT(qq);
#pragma unroll
            for (int w = 1; w < NS; w++) tv = wave == w ? ACC_FLAT(w * CQ + qq) : tv;
            told[qq] = tv;
#undef ACC_FLAT
        }

        PROF_T(t_pro);
        for (int iter = 0;; iter++) {
        while (todo) {
            const int p = __builtin_ctz(todo);
            todo &= todo - 1;
            PROF_T(t0);
            // this wave's share of the staged plane's items, and where it continues in the next
            const int range = __builtin_amdgcn_readfirstlane(item_range[wave * 16 + p]);
            const int ibeg = range & 0xffff, iend = range >> 16;
            const int next_first = todo
                ? min(__builtin_amdgcn_readfirstlane(item_range[wave * 16 + __builtin_ctz(todo)]) & 0xffff, nitems - 1)
                : 0;
            if (in_group == 0) {
                // a group begins: this wave's part of its slabs has landed ...
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef TTSWEEP_PROFILE
                p_vm += clock64() - t0; p_groups++;
#endif
                // ... and so has everybody else's; the other half of the slabs is no longer read
                __syncthreads();
            }
            PROF_T(t1);
            // the next group loads into the other half while this one is relaxed
            if (in_group == 0) issue_group(half ^ 1);
            PROF_T(t2);
#ifdef TTSWEEP_PROFILE
            p_wait += t1 - t0; p_stage += t2 - t1;
#endif

            const float *sv = slabs + (half * G + in_group) * slab_floats;
            for (int ii = ibeg; ii < iend; ii++) {
                const const_item_ptr item = (const_item_ptr)(items + ii);
                // this item's offset lengths and the next item's header travel while the
                // window is being read (ItemScalars)
                const ItemHdr nxt = load_hdr(items, ii + 1 < iend ? ii + 1 : next_first);
                typedef const __attribute__((address_space(4))) f32x16 *const_h_ptr;
                ItemScalars sc;
                sc.h0 = *reinterpret_cast<const_h_ptr>(&item->h[0][0]);
                sc.h1 = NP > 1 ? *reinterpret_cast<const_h_ptr>(&item->h[1][0]) : sc.h0;
                sc.next_rowoff = nxt.rowoff; sc.next_m0 = nxt.m0; sc.next_m1 = nxt.m1;
                float h0[16], h1[16];
#pragma unroll
                for (int t = 1; t < 16; t++) { h0[t] = sc.h0[t]; h1[t] = sc.h1[t]; }
                const int rowoff = cur.rowoff;
                const unsigned m0 = cur.m0, m1 = cur.m1;
                const int row = lane_r + rb + rowoff;
                const char *prow = reinterpret_cast<const char *>(sv) + row * (STRIP_W * 4);
                const unsigned swb = (unsigned)slab_swizzle(row) << 4;
                if (NP == 1) {
                    relax_item_single<K>(m0, h0, prow, swb, sc, vce[0], vco[0], acc[0]);
                } else {
                    f32x2 vN2[W / 2], tN2[W / 2];       // window element w is pair w/2, half w&1
                    load_window<K, 0xffu>(prow, swb, sc, vN2, tN2);
                    relax_dispatch<K>(m0, h0, vN2, tN2, vce[0], vco[0], acc[0]);
                    relax_dispatch<K>(m1, h1, vN2, tN2, vce[NP - 1], vco[NP - 1], acc[NP - 1]);
                }
                cur = nxt;
            }
            if (ibeg >= iend) cur = load_hdr(items, next_first);    // (no item of this plane was ours)
            if (++in_group == G) { in_group = 0; half ^= 1; }
#ifdef TTSWEEP_PROFILE
            p_comp += clock64() - t2;
#endif
        }
        // ---- min-combine the waves' partial results; wave w finishes cells fin_q0 .. + CQ - 1
        // of own plane fin_plane
        __syncthreads();                // the last slab is no longer read
#pragma unroll
        for (int jp = 0; jp < NP; jp++)
#pragma unroll
            for (int qq = 0; qq < K; qq++) comb[((wave * NP + jp) * K + qq) * STRIP_TB + lane] = acc[jp][qq];
        __syncthreads();
        float best[CQ];
#pragma unroll
        for (int qq = 0; qq < CQ; qq++) {
            float m = comb[((0 * NP + fin_plane) * K + fin_q0 + qq) * STRIP_TB + lane];
#pragma unroll
            for (int w = 1; w < NS; w++) m = fminf(m, comb[((w * NP + fin_plane) * K + fin_q0 + qq) * STRIP_TB + lane]);
            best[qq] = m;
        }
        // store improved cells that lie inside the grid and outside the dead-edge box
        const int a = a0 + fin_plane;
        const int b = b0 + lane;
        const bool row_ok = lane < tb_eff && b < L.n[1] && a < L.n[0];
        const bool in_box_ab = a >= box_lo0 && a <= box_hi0 && b >= box_lo1 && b <= box_hi1;
        int improved = 0;
        // zones of this lane's row along b (3 bits: ANY, LO, HI), spread to the b positions
        // of the flag word: zone set Zc of a cell becomes Zc | Zc << 3 (LO) | Zc << 6 (HI)
        const int zb_lo = lane < rb, zb_hi = lane >= STRIP_TB - rb;
        float *const Trow = T + own + fin_plane * L.s0 + fin_q0;
#pragma unroll
        for (int qq = 0; qq < CQ; qq++) {
            const int cq = fin_q0 + qq;
            const int c = c0 + cq;
            const bool special = in_box_ab && c >= box_lo2 && c <= box_hi2;
            if (row_ok && c < L.n[2] && !special && best[qq] < told[qq]) {
                // (one launch per solve: a write-through store - global_store ... sc1 -, so that no write-back of the
                // XCD's L2 is needed before the bits are pushed, only the wait for the stores themselves)
                if (ASYNC) __hip_atomic_store(&Trow[qq], best[qq], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else Trow[qq] = best[qq];
                const int zc = 1 | (cq < STRIP_CF - 1 ? 2 : 0) | (cq > K - STRIP_CF ? 4 : 0);
                improved |= zc | (zb_lo ? zc << 3 : 0) | (zb_hi ? zc << 6 : 0);
            }
        }
#pragma unroll
        for (int w = 32; w >= 1; w >>= 1) improved |= __shfl_xor(improved, w);
        improved_all |= improved;
        // ---- one launch per solve, on request (AsyncSolve::inunit): while the unit improves, relax it again against
        // its OWN planes - the values it has just stored -, up to `inunit` times: what crosses the unit's own cells
        // (64 x 16 per plane, 7 cells a time) then needs no further turn of the unit through the planner and a ring
        if (!ASYNC || iter >= as.inunit) break;
        if (improved && lane == 0) atomicOr(&head[13 + (iter & 1)], 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int again = __builtin_amdgcn_readfirstlane(head[13 + (iter & 1)]);
        if (tid == 0) head[13 + ((iter + 1) & 1)] = 0;
        if (!again) break;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // (the rows this CU staged before are stale in its L1)
#pragma unroll
        for (int qq = 0; qq < CQ; qq++) told[qq] = fminf(told[qq], best[qq]);
        todo = (((1u << NP) - 1u) << plan.ra) & plane_mask;
        inunit_passes++;
        unissued = todo;
        half = 0; in_group = 0;
        issue_group(0);
        cur = load_hdr(items, min(__builtin_amdgcn_readfirstlane(
            item_range[wave * 16 + (todo ? __builtin_ctz(todo) : 0)]) & 0xffff, nitems - 1));
        }
        PROF_T(t_loop);         // (the in-unit passes count as compute: -DTTSWEEP_PROFILE)
        const int improved = improved_all;
        const int a = a0 + fin_plane;
        if (ASYNC && tid == 0) {
            if (s != extra_s) {
                if (extra_relax) atomicAdd(starts[extra_s].work, extra_relax);
                if (extra_units) atomicAdd(starts[extra_s].work + 2, extra_units);
                extra_relax = 0;
                extra_units = 0;
                extra_s = s;
            }
            // (iter: the passes that followed the first one)
            const int wb = min(tb_eff, L.n[1] - b0), wc = max(min(K, L.n[2] - c0), 0);
            const bool two = NP > 1 && a0 + 1 < L.n[0];
            int nent = 0;
#pragma unroll
            for (int j = 0; j < NP; j++) nent += pc.n[plan.ra + j][0] + (two ? pc.n[plan.ra + j][1] : 0);
            extra_relax += (unsigned long long)inunit_passes * (unsigned long long)(wb * wc) * (unsigned long long)nent;
            if (as.handoff) {
                // (with hand-off the planner does not see every entry: the workers count the first pass too)
                int nent1 = 0;
                for (unsigned rest = my_planes & plane_mask; rest; rest &= rest - 1) {
                    const int p = __builtin_ctz(rest);
                    nent1 += pc.n[p][0] + (two ? pc.n[p][1] : 0);
                }
                extra_relax += (unsigned long long)(wb * wc) * (unsigned long long)nent1;
                extra_units += 1ull;
            }
        }
#ifdef TTSWEEP_ASYNC_STATS
        if (ASYNC && improved && lane == 0) atomicOr(&head[12], 1);      // (tuning aid: did this unit improve anything?)
#endif
        const HandOff ho = make_handoff(s, q, defer_margin);
        const bool ho_on = ASYNC && ho.on != 0;
        if (ASYNC) {
            // every wave's (write-through) stores have arrived, everybody knows it; then the bits
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            // (hand-off: the waves that finished cells of one own plane would tell the same units - the first of them
            // would take an idle one and publish it, the bits of the others would make it due once more: ONE wave per
            // own plane tells, with what all of them improved)
            if (ho_on && improved && lane == 0) atomicOr(&head[4 + fin_plane], improved);
            __syncthreads();
        }
        int told_improved = improved;
        if (ho_on) told_improved = wave == fin_plane * (NS / NP) ? __builtin_amdgcn_readfirstlane(head[4 + fin_plane]) : 0;
        if (told_improved) {    // (wave-uniform) the units that stage this plane have to look again
            const DeferRule rule{sdp->sa, sdp->sb, sdp->sc, defer_margin};
            push_improved<NP, ASYNC>(L, plan.ra, btiles, cstrips, reinterpret_cast<unsigned *>(tile_flags + 2 * nflag),
                                     a, bt, cs, told_improved, lane, rule, reinterpret_cast<unsigned *>(tile_flags), ho);
            // (the start's word: once per wave, start and pass - its few words are a hot spot)
            if (lane == 0 && s != flagged) atomicOr(&changed[s], CHANGED_IMPROVED);
            flagged = s;
        }
        if (ASYNC) {
            // the bits are out: the unit may be planned again, and only then does it count as completed
            // (a busy word counts as activity for the planner: clearing and counting need no order)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                unsigned *const myw = reinterpret_cast<unsigned *>(tile_flags + 2 * nflag) + my_unit;
                if (ho.on & ASYNC_HANDOFF_SELF) {
                    // bits that arrived while the unit was being relaxed: it goes back into the ring at once
                    const unsigned left = atomicAnd(myw, ~ASYNC_BUSY) & ~ASYNC_BUSY;
                    if (left) {
                        const unsigned planes = handoff_take(myw, left);
                        if (planes) handoff_publish(ho, (unsigned)my_unit, planes);
                    }
                } else {
                    atomicAnd(myw, ~ASYNC_BUSY);
                }
                atomicAdd(reinterpret_cast<unsigned *>(as.ctl + (size_t)q * ASYNC_CTL_STRIDE + 16), 1u);
#ifdef TTSWEEP_ASYNC_STATS
                // units / staged planes relaxed, and those of them that improved no cell (status[4 .. 7])
                const unsigned np_ = (unsigned)__builtin_popcount(my_planes);
                atomicAdd(as.status + 4, 1u);
                atomicAdd(as.status + 5, np_);
                if (!head[12]) { atomicAdd(as.status + 6, 1u); atomicAdd(as.status + 7, np_); }
                head[12] = 0;
#endif
            }
        }
#ifdef TTSWEEP_PROFILE
        {   // (summed per workgroup, added to the totals once at its end: seven atomics per unit
            // on eight words would be part of what is being measured)
            const long long t_end = clock64();
            prof_acc[0] += (unsigned long long)(t_fetch - t_top);
            prof_acc[1] += (unsigned long long)(t_pro - t_fetch);
            prof_acc[2] += (unsigned long long)p_wait;
            prof_acc[3] += (unsigned long long)p_stage;
            prof_acc[4] += (unsigned long long)p_comp;
            prof_acc[5] += (unsigned long long)(t_end - t_loop);      // (push, completion)
            prof_acc[6] += 1ull;
            prof_vm += (unsigned long long)p_vm; prof_groups += (unsigned long long)p_groups;
        }
#endif
    }
#ifdef TTSWEEP_PROFILE
    if (tid == 0) {
        for (int i = 0; i < 7; i++) if (prof_acc[6]) atomicAdd(&g_prof[i], prof_acc[i]);
        atomicAdd(&g_prof[8], prof_vm); atomicAdd(&g_prof[9], prof_groups);
        atomicAdd(&g_prof[7], (unsigned long long)(clock64() - t_k0));
    }
#endif

    if (ASYNC) {                // (the host reads the words after the one launch)
        if (tid == 0 && extra_relax) atomicAdd(starts[extra_s].work, extra_relax);
        if (tid == 0 && extra_units) atomicAdd(starts[extra_s].work + 2, extra_units);
        return;
    }
    // ---- the last workgroup to leave closes the pass: it hands the "changed" words to the
    // host (pinned memory) and clears the queue counters and the next pass's words, so a pass
    // needs no memset / copy commands around its two kernels.  All its threads take part (one
    // start each: the words are read with an atomic, a round trip apiece - done one after the
    // other by a single thread they cost 24 round trips per pass on the headline workload).
    __syncthreads();
    if (tid == 0) {
        __threadfence();
        head[0] = atomicAdd(ctrl + UNITQ_CTRL_WORDS, 1) == (int)gridDim.x - 1;
    }
    __syncthreads();
    if (head[0]) {
        for (int s = tid; s < tail.nstart; s += STRIP_TB * NS) {
            tail.changed_host[s] = atomicOr(&changed[s], 0);
            tail.changed_next[s] = 0;
        }
        if (tid < UNITQ_CTRL_WORDS) ctrl[tid] = 0;
        __threadfence_system();
        __syncthreads();
        if (tid == 0) ctrl[UNITQ_CTRL_WORDS] = 0;
    }
}

size_t units_lds_bytes(int waves)
{
    const int nslab = 2 * (waves == STRIP_NS_LAT ? STRIP_G_LAT : STRIP_G);
    size_t floats = (size_t)nslab * SLAB_BYTES / 4;                         // the slabs of v and T rows
    floats = std::max(floats, (size_t)waves * STRIP_PLANES * STRIP_K * STRIP_TB);       // combine buffer
    return (floats + strip_lds_head(waves)) * sizeof(float);
}

hipError_t launch_plan_pass(const DevLayout &L, const StartDesc *starts, const int2 *work,
                            long long nwork, int *changed, int4 *lists, int list_cap, int nlists,
                            int *ctrl, const StripPlan &plan, float gate_r2, int *flags0, long long flags_stride,
                            hipStream_t st)
{
    if (nwork <= 0) return hipSuccess;
    if (nlists < 1 || nlists > UNITQ_LISTS) return hipErrorInvalidValue;
    const int btiles = strip_btiles(L);
    // threads: waves of 64 entries, dealt over the sub-lists
    const long long per_list = (nwork + nlists - 1) / nlists;
    const long long waves = ((per_list + 63) / 64) * nlists;
    const long long nblocks = (waves + 3) / 4;
    if (nblocks > 0x7fffffffLL) return hipErrorInvalidValue;
    PlaneCounts pc;
    for (int p = 0; p < STRIP_STAGED; p++)
        for (int j = 0; j < STRIP_PLANES; j++) pc.n[p][j] = p < plan.nstaged ? plan.nent[p][j] : 0;
    hipLaunchKernelGGL(plan_pass_kernel, dim3((unsigned)nblocks), dim3(256), 0, st, L, starts, work,
                       nwork, changed, lists, list_cap, ctrl, nlists, plan.ra, plan.np, btiles, strip_cstrips(L),
                       gate_r2, pc, flags0, flags_stride);
    return hipGetLastError();
}

hipError_t launch_sweep_units(const DevLayout &L, const float *v, const StartDesc *starts,
                              const int4 *lists, int list_cap, int nlists, int *ctrl, int nblocks,
                              int *changed, const StripItem *items, const StripPlan &plan,
                              const UnitPassTail &tail, hipStream_t st)
{
    if (nblocks <= 0 || nlists < 1 || nlists > UNITQ_LISTS)
        return hipErrorInvalidValue;                    // (the last workgroup closes the pass)
    if (plan.np < 1 || plan.np > STRIP_PLANES) return hipErrorInvalidValue;
    const int btiles = strip_btiles(L);
    auto kern = plan.np == 1 ? sweep_units_kernel<STRIP_K, 1, false, STRIP_NS, STRIP_G>
                             : sweep_units_kernel<STRIP_K, 2, false, STRIP_NS, STRIP_G>;
    const size_t lds = units_lds_bytes(STRIP_NS);
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)nblocks), dim3(STRIP_TB, STRIP_NS), lds, st, L, v, starts,
                       lists, list_cap, nlists, ctrl, changed, items, plan, btiles, strip_cstrips(L), tail,
                       AsyncSolve{}, PlaneCounts{}, (int *)nullptr, 0ll);
    return hipGetLastError();
}

hipError_t launch_solve_units(const DevLayout &L, const float *v, const StartDesc *starts, int nblocks,
                              int *changed, const StripItem *items, const StripPlan &plan, int waves,
                              const UnitPassTail &tail, const AsyncSolve &as, int *flags0, long long flags_stride,
                              hipStream_t st)
{
    if (as.nrings < 1 || as.nrings > ASYNC_MAX_RINGS || nblocks <= as.nrings) return hipErrorInvalidValue;
    if (plan.np < 1 || plan.np > STRIP_PLANES) return hipErrorInvalidValue;
    const int btiles = strip_btiles(L);
    if (waves != STRIP_NS && !(waves == STRIP_NS_LAT && plan.np == 1)) return hipErrorInvalidValue;
    auto kern = waves == STRIP_NS_LAT ? sweep_units_kernel<STRIP_K, 1, true, STRIP_NS_LAT, STRIP_G_LAT>
              : plan.np == 1 ? sweep_units_kernel<STRIP_K, 1, true, STRIP_NS, STRIP_G>
                             : sweep_units_kernel<STRIP_K, 2, true, STRIP_NS, STRIP_G>;
    const size_t lds = units_lds_bytes(waves);
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    PlaneCounts pc;
    for (int p = 0; p < STRIP_STAGED; p++)
        for (int j = 0; j < STRIP_PLANES; j++) pc.n[p][j] = p < plan.nstaged ? plan.nent[p][j] : 0;
    hipLaunchKernelGGL(kern, dim3((unsigned)nblocks), dim3(STRIP_TB, waves), lds, st, L, v, starts,
                       (const int4 *)nullptr, 0, as.nrings, (int *)nullptr, changed, items, plan, btiles,
                       strip_cstrips(L), tail, as, pc, flags0, flags_stride);
    return hipGetLastError();
}

} // namespace ttsweep
