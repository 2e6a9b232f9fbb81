// ttsweep_tile.hip - sweep, variant TILE: ordered Gauss-Seidel tile sweeps for small stars.
//
// For a star of one or two cells' reach (the 6- and 26-neighbour shells) a relaxation is a
// handful of operations per 12 bytes of compulsory traffic: the
// kernel is bound by HBM, and a schedule that moves information one hop per grid pass (the
// unit-queue kernel) needs about as many passes as the grid is wide.  This variant moves
// information across the whole grid in one sweep instead, the way the reference's only
// ordered schedule does (old/wavefront-openmp/wave-multistart.c:415-530 sweeps the planes
// top-down and bottom-up; the serial loop nest serial_new/sweep-tt-multistart.c:203-205 is
// the (+,+,+) ordering): eight sweep orderings (sx, sy, sz) in {+1,-1}^3 are applied in
// turn, each as a true Gauss-Seidel pass in lexicographic order of (sx x, sy y, sz z),
// parallelised over hyperplanes at two levels:
//   * the grid is cut into tiles of 8 x 8 x 32 cells; all tiles with the same progress
//     I' + J' + K' (tile coordinates counted in sweep direction) are independent of each
//     other for a 6-neighbour star and are relaxed by ONE kernel launch, one wavefront per
//     tile; the launches of a sweep follow each other on the stream;
//   * inside a tile, lane (i', j') walks its z-column: in step d it relaxes the cell with
//     k' = d - i' - j', so every cell sees the values its three upwind neighbours got in
//     step d - 1 (a systolic hyperplane sweep, 46 steps per tile, no barrier: the tile
//     lives in the LDS of one wavefront).
// For stars with diagonal offsets some neighbours lie on the same hyperplane; they are read
// as they are (old or new).  As everywhere in this library that only affects the number of
// sweeps: every value is the length of a real path and only ever decreases, so the result
// is the reference's fixed point bit for bit (ttsweep_kernels.hip, variant CELL).
//
// Liveness (serial_new/...:160,:206 exclusive star bound; :219-221 start skip) is evaluated
// exactly, per relaxation, for the entries that are not live in both directions.
#include "ttsweep_kernels.h"

#include <algorithm>
#include <climits>

namespace ttsweep {

typedef __amdgpu_buffer_rsrc_t tile_rsrc;

__device__ __forceinline__ tile_rsrc tile_make_rsrc(const float *base)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base), 0, 0xffffffff, 0x00020000);
}

// Activity words of a tile (StartDesc::tile_flags viewed as int2): x = epoch (launch
// number) in which the tile was last relaxed, y = epoch in which it last improved.  A tile
// is due when one of its 27 neighbours (itself included) improved in or after the epoch it
// was last relaxed in; relaxing it against unchanged surroundings cannot improve anything
// (an ordering sweep relaxes every cell against its whole star).
__global__ void __launch_bounds__(256)
init_tile_state_kernel(int2 *__restrict__ state, int ntiles, int start_tile)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= ntiles) return;
    state[t] = make_int2(1, (start_tile < 0 || t == start_tile) ? 1 : 0);
}

hipError_t launch_init_tile_state(const DevLayout &L, const StartDesc &sd, bool from_box, hipStream_t st)
{
    const int NI = tile_count(L.n[0], TILE_X), NJ = tile_count(L.n[1], TILE_Y), NK = tile_count(L.n[2], TILE_Z);
    const int ntiles = NI * NJ * NK;
    const int start_tile = from_box ? -1
        : ((sd.sa / TILE_X) * NJ + sd.sb / TILE_Y) * NK + sd.sc / TILE_Z;
    hipLaunchKernelGGL(init_tile_state_kernel, dim3((ntiles + 255) / 256), dim3(256), 0, st,
                       reinterpret_cast<int2 *>(sd.tile_flags), ntiles, start_tile);
    return hipGetLastError();
}

// NE: entries relaxed (the star, padded with no-ops); EXACT: some entry is live in one
// direction only, i.e. liveness has to be evaluated.
template <int NE, bool EXACT>
__global__ void __launch_bounds__(64)
tile_sweep_kernel(TileSweep P)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x;
    const DevLayout &L = P.L;

    // ---- which tile: blockIdx -> (start slot, J', K'), I' from the hyperplane
    unsigned bid = blockIdx.x;
    const int Kp = bid % P.NK; bid /= P.NK;
    const int Jp = bid % P.NJ; bid /= P.NJ;
    const int slot = bid;
    const int Ip = P.D - Jp - Kp;
    if (Ip < 0 || Ip >= P.NI) return;
    const int I = P.sx > 0 ? Ip : P.NI - 1 - Ip;
    const int J = P.sy > 0 ? Jp : P.NJ - 1 - Jp;
    const int K = P.sz > 0 ? Kp : P.NK - 1 - Kp;
    const int s = P.active[slot];
    const StartDesc sd = P.starts[s];
    int2 *__restrict__ state = reinterpret_cast<int2 *>(sd.tile_flags);
    const int tile = (I * P.NJ + J) * P.NK + K;

    // ---- due?  (a stale read of a neighbour's word can only postpone this tile: the word
    // stays >= our last-relaxed epoch until we have been relaxed after it)
    int newest = INT_MIN;
    if (lane < 27) {
        const int ni = I + lane / 9 - 1, nj = J + (lane / 3) % 3 - 1, nk = K + lane % 3 - 1;
        if ((unsigned)ni < (unsigned)P.NI && (unsigned)nj < (unsigned)P.NJ && (unsigned)nk < (unsigned)P.NK)
            newest = state[(ni * P.NJ + nj) * P.NK + nk].y;
    }
#pragma unroll
    for (int w = 16; w >= 1; w >>= 1) newest = max(newest, __shfl_xor(newest, w));
    newest = __builtin_amdgcn_readfirstlane(newest);
    if (newest < state[tile].x) return;
    if (lane == 0) state[tile].x = P.epoch;

    // ---- stage the tile and its halo: rows (x - R .. x + 7 + R, y - R .. y + 7 + R), each
    // 10 float4 wide (z0 - 4 .. z0 + 35), by LDS-DMA: slot = row * 10 + float4, the image is
    // linear in slot order, so one wave-instruction fills 1 KiB with 64 arbitrary float4
    const int R = P.R;
    const int SY = TILE_Y + 2 * R;
    const int nslots = (TILE_X + 2 * R) * SY * TILE_QPR;
    const int niter = (nslots + 63) >> 6;
    float *vimg = lds;
    float *timg = lds + niter * 256;            // (a multiple of 1 KiB behind the v image)
    const long long g0 = (long long)(I * TILE_X + L.lo[0] - R) * L.s0
                       + (long long)(J * TILE_Y + L.lo[1] - R) * L.s1 + (K * TILE_Z + L.lo[2] - TILE_ZF);
    {
        const tile_rsrc rv = tile_make_rsrc(P.v + g0), rt = tile_make_rsrc(sd.T + g0);
        const unsigned s0b = (unsigned)(L.s0 * 4), s1b = (unsigned)(L.s1 * 4);
        for (int it = 0; it < niter; it++) {
            const int sl = min(it * 64 + lane, nslots - 1);
            const int row = sl / TILE_QPR, q = sl - row * TILE_QPR;
            const int ri = row / SY, rj = row - ri * SY;
            const unsigned off = (unsigned)ri * s0b + (unsigned)rj * s1b + (unsigned)q * 16u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, (__attribute__((address_space(3))) void *)(vimg + it * 256),
                                                     16, (int)off, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rt, (__attribute__((address_space(3))) void *)(timg + it * 256),
                                                     16, (int)off, 0, 0, 0);
        }
    }

    // ---- per-lane geometry while the loads fly
    const int ip = lane >> 3, jp = lane & 7;
    const int ci = P.sx > 0 ? ip : TILE_X - 1 - ip;
    const int cj = P.sy > 0 ? jp : TILE_Y - 1 - jp;
    const bool xy_ok = I * TILE_X + ci < L.n[0] && J * TILE_Y + cj < L.n[1];
    const int row0 = ((ci + R) * SY + (cj + R)) * TILE_PITCH + TILE_ZF;     // image index of (ci, cj, z = 0)
    const int z_cells = min(TILE_Z, L.n[2] - K * TILE_Z);                   // cells of the tile inside the grid
    // image index of the start cell, if it lies inside the image (else an index nothing has)
    int start_at = -1;
    {
        const int ra = sd.sa - I * TILE_X + R, rb = sd.sb - J * TILE_Y + R, rc = sd.sc - K * TILE_Z + TILE_ZF;
        if ((unsigned)ra < (unsigned)(TILE_X + 2 * R) && (unsigned)rb < (unsigned)SY
            && (unsigned)rc < (unsigned)TILE_PITCH)
            start_at = (ra * SY + rb) * TILE_PITCH + rc;
    }
    int del[NE];
    float hh[NE];
#pragma unroll
    for (int e = 0; e < NE; e++) {
        del[e] = (P.ent[e].da * SY + P.ent[e].db) * TILE_PITCH + P.ent[e].dc;
        hh[e] = P.ent[e].h;
    }

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the image has landed (one wave: no barrier)

    // ---- the systolic sweep
    bool improved = false;
    for (int d = 0; d < TILE_X + TILE_Y + TILE_Z - 2; d++) {
        const int kp = d - ip - jp;
        const int ck = P.sz > 0 ? kp : TILE_Z - 1 - kp;
        if (xy_ok && (unsigned)kp < (unsigned)TILE_Z && ck < z_cells) {
            const int at = row0 + ck;
            const float vc = vimg[at], tc = timg[at];
            float best = tc;
#pragma unroll
            for (int e = 0; e < NE; e++) {
                const int o = at + del[e];
                const float sum = vc + vimg[o];
                const float delay = hh[e] * sum;
                float cand = delay + timg[o];
                if (EXACT) {
                    const int fl = P.ent[e].flags;      // (uniform)
                    if (fl != (PULL_FWD | PULL_REV)) {  // an edge that exists in one direction only
                        const bool live = ((fl & PULL_FWD) && at != start_at) || ((fl & PULL_REV) && o != start_at);
                        cand = live ? cand : __builtin_inff();
                    }
                }
                best = fminf(best, cand);
            }
            if (best < tc) {
                timg[at] = best;
                improved = true;
            }
        }
        // the next step reads what this one wrote (other lanes, same wavefront: LDS
        // operations of a wavefront execute in order; keep the compiler from moving them)
        __builtin_amdgcn_wave_barrier();
    }

    // ---- write the tile back if it improved: 64 rows of 8 float4
    const bool any = __ballot(improved) != 0ull;
    if (lane == 0) {
        atomicAdd(sd.work, (unsigned long long)(min(TILE_X, L.n[0] - I * TILE_X) * min(TILE_Y, L.n[1] - J * TILE_Y)
                                                * z_cells) * (unsigned long long)P.nent);
        atomicAdd(sd.work + 2, 1ull);
    }
    if (!any) return;
    float *__restrict__ T = sd.T;
    const long long t0 = (long long)(I * TILE_X + L.lo[0]) * L.s0 + (long long)(J * TILE_Y + L.lo[1]) * L.s1
                       + (K * TILE_Z + L.lo[2]);
#pragma unroll
    for (int it = 0; it < TILE_X * TILE_Y * (TILE_Z / 4) / 64; it++) {
        const int r = it * 8 + (lane >> 3), q = lane & 7;
        const int ri = r >> 3, rj = r & 7;
        const float4 val = *reinterpret_cast<const float4 *>(
            timg + ((ri + R) * SY + (rj + R)) * TILE_PITCH + TILE_ZF + 4 * q);
        *reinterpret_cast<float4 *>(T + t0 + (long long)ri * L.s0 + (long long)rj * L.s1 + 4 * q) = val;
    }
    if (lane == 0) {
        state[tile].y = P.epoch;
        atomicOr(&P.changed[s], CHANGED_IMPROVED);
    }
}

size_t tile_lds_bytes(int R)
{
    const int nslots = (TILE_X + 2 * R) * (TILE_Y + 2 * R) * TILE_QPR;
    return (size_t)2 * ((nslots + 63) / 64) * 1024;
}

hipError_t launch_tile_sweep(const TileSweep &P, hipStream_t st)
{
    if (P.nactive <= 0) return hipSuccess;
    if (P.R < 1 || P.R > TILE_MAX_R || P.nent < 1 || P.nent > TILE_MAX_ENT) return hipErrorInvalidValue;
    const long long nblocks = (long long)P.NJ * P.NK * P.nactive;
    if (nblocks > 0x7fffffffLL) return hipErrorInvalidValue;
    const size_t lds = tile_lds_bytes(P.R);
    bool exact = false;
    for (int e = 0; e < P.nent; e++) exact |= P.ent[e].flags != (PULL_FWD | PULL_REV);
#define TILE_LAUNCH(NE, EX) \
    hipLaunchKernelGGL((tile_sweep_kernel<NE, EX>), dim3((unsigned)nblocks), dim3(64), lds, st, P)
    if (P.nent <= 6) { if (exact) TILE_LAUNCH(6, true); else TILE_LAUNCH(6, false); }
    else if (P.nent <= 18) { if (exact) TILE_LAUNCH(18, true); else TILE_LAUNCH(18, false); }
    else { if (exact) TILE_LAUNCH(TILE_MAX_ENT, true); else TILE_LAUNCH(TILE_MAX_ENT, false); }
#undef TILE_LAUNCH
    return hipGetLastError();
}

} // namespace ttsweep
