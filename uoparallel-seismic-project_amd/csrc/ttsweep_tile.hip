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
//     other for a 6-neighbour star and are relaxed by ONE launch: a grid of single-wavefront
//     workgroups, as many as the device holds at once, whose workgroups first find the tiles
//     of the hyperplane that are due among their own share of the candidates (something near
//     them changed since they were last relaxed) and then relax those (no list, no cursor,
//     no per-tile atomics: tile_candidate); the launches of a sweep follow each other on
//     the stream;
//   * inside a tile, lane (i', j') walks its z-column: in step d it relaxes the cell with
//     k' = d - i' - j' (the 6-neighbour instance: the two cells 2m, 2m + 1 with
//     m = d - i' - j', software-pipelined - sixc_sweep), so every cell sees the values its
//     three upwind neighbours got before it (a systolic hyperplane sweep, no barrier: the
//     tile lives in the LDS of one wavefront).
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
#include <cstdio>

namespace ttsweep {

typedef __amdgpu_buffer_rsrc_t tile_rsrc;

__device__ __forceinline__ tile_rsrc tile_make_rsrc(const float *base)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base), 0, 0xffffffff, 0x00020000);
}

// Activity words of a tile (StartDesc::tile_flags viewed as int2): x = epoch (launch
// number) in which the tile was last relaxed, y = the latest epoch in which one of its 27
// neighbours (itself included) improved.  A tile is due when y >= x: something in its
// surroundings improved in or after the epoch it was last relaxed in; relaxing it against
// unchanged surroundings cannot improve anything (an ordering sweep relaxes every cell against
// its whole star).  A tile that improves stamps the y words of its 27 neighbours (one store
// instruction, a lane per neighbour), so that planning reads ONE int2 per candidate - in round
// 2 the planner gathered 27 stamps per candidate in every one of the ~5700 launches of a solve,
// whether or not anything had changed.
__global__ void __launch_bounds__(256)
init_tile_state_kernel(int2 *__restrict__ state, int NJ, int NK, int ntiles, int si, int sj, int sk)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= ntiles) return;
    const int K = t % NK, J = (t / NK) % NJ, I = t / (NK * NJ);
    // (si < 0: a box that arrives with values in it - every tile is due)
    const bool near_start = si < 0 || (abs(I - si) <= 1 && abs(J - sj) <= 1 && abs(K - sk) <= 1);
    state[t] = make_int2(1, near_start ? 1 : 0);
}

// First hyperplane of the NEXT ordering sweep that can hold a due tile: a workgroup keeps the
// minimum over the tiles it improved (every neighbour of such a tile is at most three
// hyperplanes earlier) and adds it to one word at its end; the host starts the next sweep
// there (ttsweep_driver.cpp: launch_pass_tile).  Every "look again" stamp is written next to a
// call of this, so no due tile can lie in front of that hyperplane.
__device__ __forceinline__ int tile_next_plane(const TileSweep &P, int I, int J, int K)
{
    const int In = P.nsx > 0 ? I : P.NI - 1 - I, Jn = P.nsy > 0 ? J : P.NJ - 1 - J, Kn = P.nsz > 0 ? K : P.NK - 1 - K;
    return max(In + Jn + Kn - 3, 0);
}

// A tile of start-state `state` improved in this launch: its 27 neighbours have to look again.
__device__ __forceinline__ void tile_stamp_neighbours(const TileSweep &P, int2 *__restrict__ state, int I, int J, int K,
                                                      int lane)
{
    if (lane < 27) {
        const int ni = I + lane / 9 - 1, nj = J + (lane / 3) % 3 - 1, nk = K + lane % 3 - 1;
        if ((unsigned)ni < (unsigned)P.NI && (unsigned)nj < (unsigned)P.NJ && (unsigned)nk < (unsigned)P.NK)
            state[(ni * P.NJ + nj) * P.NK + nk].y = P.epoch;
    }
}

hipError_t launch_init_tile_state(const DevLayout &L, const StartDesc &sd, bool from_box, hipStream_t st)
{
    const int NI = tile_count(L.n[0], TILE_X), NJ = tile_count(L.n[1], TILE_Y), NK = tile_count(L.n[2], TILE_Z);
    const int ntiles = NI * NJ * NK;
    hipLaunchKernelGGL(init_tile_state_kernel, dim3((ntiles + 255) / 256), dim3(256), 0, st,
                       reinterpret_cast<int2 *>(sd.tile_flags), NJ, NK, ntiles,
                       from_box ? -1 : sd.sa / TILE_X, sd.sb / TILE_Y, sd.sc / TILE_Z);
    return hipGetLastError();
}

// The z faces of a padded volume (tile_face_index): one thread per face cell.
__global__ void __launch_bounds__(256)
build_tile_faces_kernel(DevLayout L, const float *__restrict__ padded, float *__restrict__ faces, int fz, long long n)
{
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    const int pb = (int)(t % L.p[1]);
    long long u = t / L.p[1];
    const int pa = (int)(u % L.p[0]);  u /= L.p[0];
    const int layer = (int)(u % fz);   u /= fz;
    const int side = (int)(u & 1), kb = (int)(u >> 1);
    const int c = TILE_Z * kb + (side ? layer : layer - fz);        // interior coordinate; the halo holds the rest
    faces[t] = padded[(long long)pa * L.s0 + (long long)pb * L.s1 + (c + L.lo[2])];
}

// The faces of a freshly initialised box (+INFINITY, 0 at the start cell) without reading it.
__global__ void __launch_bounds__(256)
init_tile_faces_kernel(DevLayout L, float *__restrict__ faces, int fz, long long n, int sa, int sb, int sc)
{
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    const int pb = (int)(t % L.p[1]);
    long long u = t / L.p[1];
    const int pa = (int)(u % L.p[0]);  u /= L.p[0];
    const int layer = (int)(u % fz);   u /= fz;
    const int side = (int)(u & 1), kb = (int)(u >> 1);
    const int c = TILE_Z * kb + (side ? layer : layer - fz);
    faces[t] = (pa == sa + L.lo[0] && pb == sb + L.lo[1] && c == sc) ? 0.0f : __builtin_inff();
}

hipError_t launch_init_tile_faces(const DevLayout &L, float *faces, int fz, int sa, int sb, int sc, hipStream_t st)
{
    if (fz < 1 || fz > TILE_ZF) return hipErrorInvalidValue;
    const long long n = tile_face_cells(L, fz);
    hipLaunchKernelGGL(init_tile_faces_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, L, faces, fz, n,
                       sa, sb, sc);
    return hipGetLastError();
}

hipError_t launch_build_tile_faces(const DevLayout &L, const float *padded, float *faces, int fz, hipStream_t st)
{
    if (fz < 1 || fz > TILE_ZF || L.lo[2] < fz) return hipErrorInvalidValue;
    const long long n = tile_face_cells(L, fz);
    hipLaunchKernelGGL(build_tile_faces_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, L, padded, faces, fz, n);
    return hipGetLastError();
}

// A value the compiler can treat as wave-uniform (it is: every lane holds the same bits).
// Without this the buffer descriptors below sit in vector registers and every LDS-DMA
// instruction is wrapped in a "waterfall" loop.
__device__ __forceinline__ int uni(int x) { return __builtin_amdgcn_readfirstlane(x); }
template <typename T>
__device__ __forceinline__ T *uni_ptr(T *p)
{
    const unsigned long long u = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    return reinterpret_cast<T *>(((unsigned long long)hi << 32) | lo);
}

// ---------------------------------------------------------------------------
// planning inside the sweep kernels: which tiles of hyperplane D are due
// ---------------------------------------------------------------------------
// A launch relaxes the due tiles of ONE hyperplane D for every active start; there is no
// separate planning kernel and no list (round 2 had both: a planner kernel and a second kernel
// boundary in front of every one of the ~5700 launches of a solve).  The candidates of a launch
// are the pairs (tile position (J', K'), active start), NJ * NK * nactive of them, and they are
// dealt so that
//   * the tiles of ALL starts at one position are relaxed at the same time on the same XCD - by
//     nactive neighbouring workgroups of the XCD that position belongs to ((J' + K') % XCDs;
//     workgroup b runs on XCD b % XCDs: observed dispatch order, a speed matter only): the
//     starts share the velocity volume, so its tile image comes from HBM once and from that
//     XCD's L2 for the other starts (dealt start by start, the same image was wanted at nactive
//     different times of a launch whose working set is ten times the L2s: measured 319 ms
//     against 247 ms per solve on 1024x1024x512 x 14 starts);
//   * every workgroup's share is the launch's average whatever start the due tiles belong to
//     and wherever they cluster: workgroup j of an XCD owns candidates j, j + W, j + 2 W, ...
//     of that XCD's (position, start) sequence, with W = the largest number of workgroups per
//     XCD that is COPRIME to the number of active starts - the tiles of one start at consecutive
//     positions, nactive apart in the sequence, then go to W different workgroups before one
//     gets a second (with W = 192 and 14 starts they went to 96 of them: the early sweeps, in
//     which every start's due tiles are one compact cluster, took 2.4 times as long).
// A workgroup evaluates up to 64 of its candidates at a time, one per lane (the tile's two
// stamps, one 8-byte load), and then relaxes the due ones in turn.  A tile is due when one of
// its 27 neighbours (itself included) improved in or after the epoch it was last relaxed in
// (they stamp its y word when they do: tile_stamp_neighbours).  Tiles of the same hyperplane
// may improve while a workgroup is still evaluating: it then either sees the new stamp (and
// relaxes a tile that would have been due in the next sweep anyway) or does not (the stamp is
// >= the tile's own, so the tile is due in a later launch): nothing is ever skipped for good.
struct TileCand {
    int a;              // index into the active list
    int tile;           // (I * NJ + J) * NK + K
    unsigned cells;     // cells of the tile inside the grid
    bool due;
};

// q: index into this XCD's (position, start) sequence; x: the XCD.  Position (J', K') belongs to
// XCD (J' + K') % nxcd - not to (J' NK + K') % nxcd = K' % nxcd: the first sweeps of a solve
// whose starts lie in the top layer only have due tiles with one K', and all of them landed on
// ONE XCD (measured: those sweeps took twice as long).  The positions of XCD x in sequence:
// J' = 0 .. NJ - 1, and for each J' the K' = (x - J') mod nxcd + nxcd m, m = 0 .. M - 1.
__device__ __forceinline__ TileCand tile_candidate(const TileSweep &P, long long q, int x, int ncand)
{
    TileCand r{0, 0, 0u, false};
    (void)ncand;
    const int M = (P.NK + P.nxcd - 1) / P.nxcd;
    const long long pos = q / P.nactive;
    const int Jp = (int)(pos / M), m = (int)(pos - (long long)Jp * M);
    if (Jp >= P.NJ) return r;
    const int Kp = ((x - Jp) % P.nxcd + P.nxcd) % P.nxcd + P.nxcd * m;
    if (Kp >= P.NK) return r;
    r.a = (int)(q - pos * P.nactive);
    int2 *__restrict__ state = reinterpret_cast<int2 *>(P.state0 + (long long)P.active[r.a] * P.state_stride);
    const int Ip = P.D - Jp - Kp;
    if (Ip < 0 || Ip >= P.NI) return r;
    const int I = P.sx > 0 ? Ip : P.NI - 1 - Ip;
    const int J = P.sy > 0 ? Jp : P.NJ - 1 - Jp;
    const int K = P.sz > 0 ? Kp : P.NK - 1 - Kp;
    r.tile = (I * P.NJ + J) * P.NK + K;
    const int2 stamps = state[r.tile];
    r.due = stamps.y >= stamps.x;
    if (r.due) state[r.tile].x = P.epoch;
    r.cells = (unsigned)(min(TILE_X, P.L.n[0] - I * TILE_X) * min(TILE_Y, P.L.n[1] - J * TILE_Y)
                         * min(TILE_Z, P.L.n[2] - K * TILE_Z));
    return r;
}

// -DTTSWEEP_TILE_PROFILE: cycle stamps per phase of a tile, summed over all tiles (tuning aid;
// the stamps go to a buffer of their own and never into a result)
#ifdef TTSWEEP_TILE_PROFILE
__device__ unsigned long long g_tile_prof[8];
__device__ __forceinline__ long long prof_now()
{
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    return (long long)t;
}
#define PROF_STAMP(x) const long long x = prof_now()
void tile_prof_dump()
{
    unsigned long long h[8] = {};
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_tile_prof), sizeof(h));
    const double n = (double)std::max<unsigned long long>(h[4], 1);
    fprintf(stderr, "tile prof (cycles per tile, %llu tiles): setup + DMA issue %.0f  DMA wait %.0f  sweep %.0f  "
            "store + tail %.0f | busy / resident %.3f\n", h[4], h[0] / n, h[1] / n, h[2] / n, h[3] / n,
            (double)(h[0] + h[1] + h[2] + h[3]) / (double)std::max<unsigned long long>(h[5], 1));
    unsigned long long z[8] = {};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_tile_prof), z, sizeof(z));
}
#else
#define PROF_STAMP(x)
#endif

// ---------------------------------------------------------------------------
// relaxing the due tiles: a persistent grid of single-wavefront workgroups
// ---------------------------------------------------------------------------
// Read-only tables written before the launch (the start descriptors) are read through the
// scalar cache: uniform addresses in the constant address space become s_load instructions, a
// few hundred cycles instead of a vector-memory round trip per tile.
typedef const __attribute__((address_space(4))) StartDesc *const_start_ptr;

// The work counters of the starts (relaxations, tiles).  10^4 workgroups adding per TILE to the
// same few words held every tile up (22 of 60 ms of a full sweep on 1024x1024x512 x 14 starts
// in round 2), and a workgroup now relaxes tiles of every start.  So lane a % 64 of a workgroup
// keeps the sums of active start a in registers, and at the workgroup's end adds them to a slot
// of its own (P.wgwork[block][a % 64]: no other writer in this launch, no atomics); the solve
// reduces the slots once, at its end (tile_reduce_work).  Active indices 64 apart share a lane:
// the lane flushes when the start it counts for changes (a workgroup's tiles come sorted by start).
struct TileWork {
    unsigned long long cells = 0;
    unsigned tiles = 0;
    int a = -1;         // active index this lane currently counts for
    bool improved = false;      // a tile of that start improved in this launch
};

__device__ __forceinline__ void tile_work_flush(const TileSweep &P, TileWork &w)
{
    if (w.a >= 0 && w.tiles) {
        const int s = P.active[w.a];
        unsigned long long *const slot = P.wgwork + ((size_t)blockIdx.x * P.nstart + s) * 2;
        slot[0] += w.cells * (unsigned long long)P.nent;
        slot[1] += w.tiles;
        // (the start's "improved" word: a plain store of the one value this kernel ever writes
        // there, once per workgroup and start, after its last tile - a store per improved tile to
        // the starts' few words sits in front of the next tile's vmcnt(0) wait and, with
        // thousands of workgroups storing, tripled the time of the sweeps in which every tile
        // improves)
        if (w.improved) P.changed[s] = CHANGED_IMPROVED;
    }
    w.cells = 0; w.tiles = 0; w.improved = false;
}

// a tile of active start `a` improved
__device__ __forceinline__ void tile_work_improved(TileWork &w, int a, int lane)
{
    if (lane == (a & 63)) w.improved = true;    // (tile_work_count has set w.a = a for this tile)
}

__device__ __forceinline__ void tile_work_count(const TileSweep &P, TileWork &w, int a, unsigned cells, int lane)
{
    if (lane == (a & 63)) {
        if (w.a != a) { tile_work_flush(P, w); w.a = a; }
        w.cells += cells;
        w.tiles++;
    }
}

// wgwork[block][start] -> the starts' work counters (one workgroup per start), slots cleared
__global__ void __launch_bounds__(256)
tile_reduce_work_kernel(unsigned long long *__restrict__ wgwork, int nblocks, int nstart, unsigned long long *__restrict__ work0)
{
    const int s = blockIdx.x;
    unsigned long long cells = 0, tiles = 0;
    for (int b = threadIdx.x; b < nblocks; b += 256) {
        unsigned long long *const slot = wgwork + ((size_t)b * nstart + s) * 2;
        cells += slot[0]; tiles += slot[1];
        slot[0] = 0; slot[1] = 0;
    }
    __shared__ unsigned long long red[2][256];
    red[0][threadIdx.x] = cells; red[1][threadIdx.x] = tiles;
    __syncthreads();
    for (int w = 128; w >= 1; w >>= 1) {
        if ((int)threadIdx.x < w) { red[0][threadIdx.x] += red[0][threadIdx.x + w]; red[1][threadIdx.x] += red[1][threadIdx.x + w]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { work0[3 * s] += red[0][0]; work0[3 * s + 2] += red[1][0]; }
}

hipError_t launch_tile_reduce_work(unsigned long long *wgwork, int nblocks, int nstart, unsigned long long *work0, hipStream_t st)
{
    if (nstart <= 0 || nblocks <= 0) return hipSuccess;
    hipLaunchKernelGGL(tile_reduce_work_kernel, dim3(nstart), dim3(256), 0, st, wgwork, nblocks, nstart, work0);
    return hipGetLastError();
}

// The 6-neighbour star with halo 1, entries in the pull star's order (sorted by offset):
// image index deltas and everything derived from them are compile-time constants.
constexpr int SIX_SY = TILE_Y + 2;

typedef float tile_f2 __attribute__((ext_vector_type(2)));

// NE: entries relaxed (the star, padded with no-ops); EXACT: some entry is live in one
// direction only, i.e. liveness has to be evaluated.  (The plain 6-neighbour star has a kernel
// of its own, tile_six_kernel.)
template <int NE, bool EXACT>
__global__ void __launch_bounds__(64)
tile_sweep_kernel(TileSweep P)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x;
    const DevLayout &L = P.L;

    const int R = P.R;
    const int SY = TILE_Y + 2 * R;
    const int nslots = (TILE_X + 2 * R) * SY * TILE_QPR;
    const int niter = (nslots + 63) >> 6;
    float *vimg = lds;
    float *timg = lds + nslots * 4;             // (directly behind the v image)

    // per-lane geometry (the same for every tile of the launch)
    const int ip = lane >> 3, jp = lane & 7;
    const int ci = P.sx > 0 ? ip : TILE_X - 1 - ip;
    const int cj = P.sy > 0 ? jp : TILE_Y - 1 - jp;
    const int row0 = ((ci + R) * SY + (cj + R)) * TILE_PITCH + TILE_ZF;     // image index of (ci, cj, z = 0)
    int del[NE];
    float hh[NE];
#pragma unroll
    for (int e = 0; e < NE; e++) {
        del[e] = (P.ent[e].da * SY + P.ent[e].db) * TILE_PITCH + P.ent[e].dc;
        hh[e] = P.ent[e].h;
    }
    const unsigned s0b = (unsigned)(L.s0 * 4), s1b = (unsigned)(L.s1 * 4);
    // staging slots of this lane: (row, float4) -> byte offset inside the tile's region
    // (the same for every tile: computed once, kept for the first iterations' worth)
    const int dat = P.sz > 0 ? 1 : -1;

#ifdef TTSWEEP_TILE_PROFILE
    unsigned long long prof_acc[5] = {};
#endif
    PROF_STAMP(t_begin);
    // (everything that selects the tile is kept in scalar registers: the start descriptor then
    // comes through the scalar cache and the buffer descriptors need no waterfall loop)
    const int ncand = P.NJ * P.NK;
    const int xcd = uni((int)blockIdx.x % P.nxcd);
    const long long W = P.wstride;              // workgroups of an XCD that take candidates (coprime to nactive)
    const long long nseq = (long long)P.NJ * ((P.NK + P.nxcd - 1) / P.nxcd) * P.nactive;   // candidates of an XCD
    TileWork work;
    int first_next = INT_MAX;       // (uniform) see tile_next_plane
    for (long long q0 = blockIdx.x / P.nxcd; q0 < nseq && (long long)(blockIdx.x / P.nxcd) < W; q0 += 64 * W) {
      const TileCand pick = tile_candidate(P, q0 + lane * W, xcd, ncand);
      unsigned long long due_lanes = __ballot(pick.due);
      while (due_lanes) {
        const int src = __builtin_ctzll(due_lanes);
        due_lanes &= due_lanes - 1;
        const int tile = __builtin_amdgcn_readlane(pick.tile, src);
        const int act = __builtin_amdgcn_readlane(pick.a, src);
        const int s = uni(P.active[act]);
        tile_work_count(P, work, act, (unsigned)__builtin_amdgcn_readlane((int)pick.cells, src), lane);
        PROF_STAMP(t_top);

        const const_start_ptr sdp = (const_start_ptr)(P.starts + s);
        float *const T = sdp->T;
        int2 *const state = reinterpret_cast<int2 *>(sdp->tile_flags);
        const int sa = sdp->sa, sb = sdp->sb, sc = sdp->sc;
        const int K = tile % P.NK, J = (tile / P.NK) % P.NJ, I = tile / (P.NK * P.NJ);

        // ---- stage the tile and its halo: rows (x - R .. x + 7 + R, y - R .. y + 7 + R), each
        // 10 float4 wide, by LDS-DMA: slot = row * 10 + float4, the image is linear in slot
        // order, so one wave-instruction fills 1 KiB with 64 arbitrary float4.  The interior
        // of a row is one whole 128-byte line of the volume; the first and the last float4 of
        // an image row (the z halo) are fetched from that same line (any values: no further
        // line is touched) and their cells next to the interior are then overwritten with the
        // z faces, which arrive in registers meanwhile.
        const long long g0 = (long long)(I * TILE_X + L.lo[0] - R) * L.s0
                           + (long long)(J * TILE_Y + L.lo[1] - R) * L.s1 + (K * TILE_Z + L.lo[2] - TILE_ZF);
        const int FZ = P.fz;
        const int nface = (TILE_X + 2 * R) * SY * 2 * FZ;           // face cells around this tile
        const float *const tface = P.tface + (long long)s * P.face_cells;
        constexpr int NFI = ((TILE_X + 2 * TILE_MAX_R) * (TILE_Y + 2 * TILE_MAX_R) * 2 * TILE_ZF + 63) / 64;
        float fv[NFI], ft[NFI];
        {
            // (the previous tile's LDS writes have retired before the image is overwritten)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const tile_rsrc rv = tile_make_rsrc(uni_ptr(P.v + g0)), rt = tile_make_rsrc(uni_ptr(T + g0));
            for (int it = 0; it < niter; it++) {
                const int sl = it * 64 + lane;
                if (sl >= nslots) break;        // (last instruction: the lanes past the image stay off)
                const int row = sl / TILE_QPR, q = min(max(sl - row * TILE_QPR, 1), TILE_QPR - 2);
                const int ri = row / SY, rj = row - ri * SY;
                const unsigned off = (unsigned)ri * s0b + (unsigned)rj * s1b + (unsigned)q * 16u;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, (__attribute__((address_space(3))) void *)(vimg + it * 256),
                                                         16, (int)off, 0, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rt, (__attribute__((address_space(3))) void *)(timg + it * 256),
                                                         16, (int)off, 0, 0, 0);
            }
#pragma unroll
            for (int it = 0; it < NFI; it++) {
                const int f = it * 64 + lane;
                fv[it] = 0.0f;
                ft[it] = 0.0f;
                if (f < nface) {
                    const int row = f / (2 * FZ), rest = f - row * (2 * FZ), side = rest / FZ, layer = rest - side * FZ;
                    const int ri = row / SY, rj = row - ri * SY;
                    // below the tile: boundary K, its lower side; above: boundary K + 1, its upper side
                    const long long fi = tile_face_index(L, FZ, K + side, side, layer, I * TILE_X + ri, J * TILE_Y + rj);
                    fv[it] = P.vface[fi];
                    ft[it] = tface[fi];
                }
            }
        }

        // steps in which this lane has a cell of the grid: k' = d - i' - j' in [klo, khi)
        const bool xy_ok = I * TILE_X + ci < L.n[0] && J * TILE_Y + cj < L.n[1];
        const int z_cells = min(TILE_Z, L.n[2] - K * TILE_Z);               // cells of the tile inside the grid
        const int klo = P.sz > 0 ? 0 : TILE_Z - z_cells;
        const unsigned span = xy_ok ? (unsigned)z_cells : 0u;
        // image index of the start cell, if it lies inside the image (else an index nothing has)
        int start_at = -1;
        if (EXACT) {
            const int ra = sa - I * TILE_X + R, rb = sb - J * TILE_Y + R, rc = sc - K * TILE_Z + TILE_ZF;
            if ((unsigned)ra < (unsigned)(TILE_X + 2 * R) && (unsigned)rb < (unsigned)SY
                && (unsigned)rc < (unsigned)TILE_PITCH)
                start_at = (ra * SY + rb) * TILE_PITCH + rc;
        }

        PROF_STAMP(t_issued);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the image has landed (one wave: no barrier)
#pragma unroll
        for (int it = 0; it < NFI; it++) {                  // the z faces into the halo cells of the rows
            const int f = it * 64 + lane;
            if (f < nface) {
                const int row = f / (2 * FZ), rest = f - row * (2 * FZ), side = rest / FZ, layer = rest - side * FZ;
                const int pos = row * TILE_PITCH + (side ? TILE_ZF + TILE_Z + layer : TILE_ZF - FZ + layer);
                vimg[pos] = fv[it];
                timg[pos] = ft[it];
            }
        }
        __builtin_amdgcn_wave_barrier();
        PROF_STAMP(t_landed);

        // ---- the systolic sweep: lane (i', j') relaxes k' = d - i' - j' in step d
        bool improved = false;
        int kp = -ip - jp;
        int at = row0 + (P.sz > 0 ? kp : TILE_Z - 1 - kp);
        for (int d = 0; d < TILE_X + TILE_Y + TILE_Z - 2; d++, kp++, at += dat) {
            if ((unsigned)(kp - klo) < span) {
                float vo[NE], to[NE];
                const float vc = vimg[at], tc = timg[at];
#pragma unroll
                for (int e = 0; e < NE; e++) {      // all neighbour reads in flight before the first use
                    vo[e] = vimg[at + del[e]];
                    to[e] = timg[at + del[e]];
                }
                float best = tc;
#pragma unroll
                for (int e = 0; e < NE; e++) {
                    const float sum = vc + vo[e];
                    const float delay = hh[e] * sum;
                    float cand = delay + to[e];
                    if (EXACT) {
                        const int fl = P.ent[e].flags;      // (uniform)
                        if (fl != (PULL_FWD | PULL_REV)) {  // an edge that exists in one direction only
                            const bool live = ((fl & PULL_FWD) && at != start_at)
                                           || ((fl & PULL_REV) && at + del[e] != start_at);
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
        PROF_STAMP(t_swept);

        // ---- write the tile back if it improved: 64 rows of 8 float4
        const bool any = __ballot(improved) != 0ull;
        if (any) {
            const long long t0 = (long long)(I * TILE_X + L.lo[0]) * L.s0 + (long long)(J * TILE_Y + L.lo[1]) * L.s1
                               + (K * TILE_Z + L.lo[2]);
#pragma unroll
            for (int it = 0; it < TILE_X * TILE_Y * (TILE_Z / 4) / 64; it++) {
                constexpr int QR = TILE_Z / 4;      // float4 per interior row
                const int r = (it * 64 + lane) / QR, q = (it * 64 + lane) % QR;
                const int ri = r >> 3, rj = r & 7;
                const float4 val = *reinterpret_cast<const float4 *>(
                    timg + ((ri + R) * SY + (rj + R)) * TILE_PITCH + TILE_ZF + 4 * q);
                *reinterpret_cast<float4 *>(T + t0 + (long long)ri * L.s0 + (long long)rj * L.s1 + 4 * q) = val;
            }
            // this column's first and last FZ cells into the faces of the tile's two boundaries
            {
                float *const tf = const_cast<float *>(tface);
                const int pa = I * TILE_X + R + ci, pb = J * TILE_Y + R + cj;
                for (int l = 0; l < FZ; l++) {
                    tf[tile_face_index(L, FZ, K, 1, l, pa, pb)] = timg[row0 + l];
                    tf[tile_face_index(L, FZ, K + 1, 0, l, pa, pb)] = timg[row0 + TILE_Z - FZ + l];
                }
            }
            tile_stamp_neighbours(P, state, I, J, K, lane);
            tile_work_improved(work, act, lane);
            first_next = min(first_next, tile_next_plane(P, I, J, K));
        }
        // (the image is overwritten by the next tile's loads: every read of it has been
        // consumed; the stores are in flight and read registers only)
#ifdef TTSWEEP_TILE_PROFILE
        {   // (summed per workgroup, added to the totals once at its end: per-tile atomics on
            // a few words would be the very cost being looked for)
            const long long t_end = prof_now();
            prof_acc[0] += (unsigned long long)(t_issued - t_top);
            prof_acc[1] += (unsigned long long)(t_landed - t_issued);
            prof_acc[2] += (unsigned long long)(t_swept - t_landed);
            prof_acc[3] += (unsigned long long)(t_end - t_swept);
            prof_acc[4] += 1ull;
        }
#endif
      }
    }
    tile_work_flush(P, work);
    if (lane == 0 && first_next != INT_MAX) atomicMin(P.dmin_next, first_next);
#ifdef TTSWEEP_TILE_PROFILE
    if (lane == 0 && prof_acc[4]) {
        for (int i = 0; i < 5; i++) atomicAdd(&g_tile_prof[i], prof_acc[i]);
        atomicAdd(&g_tile_prof[5], (unsigned long long)(prof_now() - t_begin));
    }
#endif
}

// ---------------------------------------------------------------------------
// The plain 6-neighbour star with a compact image: six tiles in flight per CU instead of five
// ---------------------------------------------------------------------------
// The 6-neighbour star reads no corner row and exactly one cell beyond either end of a row, so
// its image can be dense: 98 rows (the 10 x 10 rows but the first and the last - two corner
// rows; the other two stay, so that lateral neighbours keep constant offsets) of 32 floats,
// 8 float4 per row by LDS-DMA, and the rows' two z-halo cells in a table of their own, filled
// straight from the z faces by 4-byte LDS-DMA.  26 784 bytes with the 128 bytes in front that
// idle lanes read: 21 of the LDS's 1280-byte granules - six workgroups per CU where the
// 40-float rows of the general kernel let five fit.
constexpr int SIXC_NR = (TILE_X + 2) * SIX_SY - 2;          // rows kept
constexpr int SIXC_PAD = 32;                                // floats in front of the images
constexpr int SIXC_T = SIXC_NR * TILE_Z;                    // the T rows start here (floats behind the v rows)
constexpr int SIXC_HV = 2 * SIXC_T;                         // z-halo cells of v: [row][below, above]
constexpr int SIXC_HT = SIXC_HV + 2 * SIXC_NR;              // ... of T
constexpr int SIXC_FLOATS = SIXC_PAD + SIXC_HT + 2 * SIXC_NR;
constexpr int SIXC_NSLOTS = SIXC_NR * (TILE_Z / 4);         // float4 slots of one image
constexpr int SIXC_NITER = (SIXC_NSLOTS + 63) / 64;
constexpr int SIXC_NFACE = 2 * SIXC_NR;
constexpr int SIXC_NFIT = (SIXC_NFACE + 63) / 64;

// The systolic sweep of the plain 6-neighbour star over the compact image (`img` = the v rows,
// T rows SIXC_T behind, halo cells from the tables; `row` = the lane's image row).  Lane
// (i', j') owns the z-column (i', j') and relaxes TWO cells per step, k' = 2m and 2m + 1 with
// m = d - i' - j' (30 steps for 8 x 8 x 32): the first cell's z-upwind neighbour is the lane's
// own previous result, the second's is the first.  Software-pipelined: of a pair's inputs only
// the travel times of its four lateral neighbour pairs can have been written in the previous
// step (by other lanes; two of them really were, which two depends on the ordering); the
// z-downwind values and all velocities cannot change before this lane is past them.  So a
// step reads just those four pairs behind the previous step's writes, then - while they are on
// their way - what the NEXT pair needs that is final already; the delays of the current pair
// come from values that arrived a step ago.  The dependent chain of a step is one LDS round
// trip plus about ten vector instructions, for two cells.  A step is also short in
// instructions: one address register (every input is a compile-time offset from it: ZPOS is
// the direction along z), 64-bit LDS accesses (a pair is 8-byte aligned), and the lateral
// relaxations as packed operations over the pair.  Lanes outside their 16 steps run the same
// instructions on in-image addresses and store nothing.  Entries in the pull star's order:
// x-, y-, z-, z+, y+, x+.
template <bool ZPOS>
__device__ __forceinline__ bool sixc_sweep(float *img, int row, int ij, int klo, unsigned span, const TileSweep &P)
{
    static_assert(TILE_Z == 32, "rows are whole 128-byte lines");
    constexpr int DP = ZPOS ? 2 : -2, DX = SIX_SY * TILE_Z, DY = TILE_Z;
    constexpr int C = DX + 4;               // index of the current pair from the base below (every index >= -SIXC_PAD)
    float hxm = P.ent[0].h, hxp = P.ent[5].h, hym = P.ent[1].h, hyp = P.ent[4].h,
          hzu = P.ent[ZPOS ? 2 : 3].h, hzd = P.ent[ZPOS ? 3 : 2].h;
    asm volatile("" : "+v"(hxm), "+v"(hxp), "+v"(hym), "+v"(hyp), "+v"(hzu), "+v"(hzd));
    const tile_f2 hxm2 = {hxm, hxm}, hxp2 = {hxp, hxp}, hym2 = {hym, hym}, hyp2 = {hyp, hyp};
#define FIRST(p) (ZPOS ? (p).x : (p).y)
#define SECOND(p) (ZPOS ? (p).y : (p).x)
#define LD2(i) (*reinterpret_cast<const tile_f2 *>(vb + (i)))
    // the cells beyond the two ends of this lane's row, in sweep direction
    const tile_f2 hv = *reinterpret_cast<const tile_f2 *>(img + SIXC_HV + 2 * row);
    const tile_f2 ht = *reinterpret_cast<const tile_f2 *>(img + SIXC_HT + 2 * row);
    const float hv_up = ZPOS ? hv.x : hv.y, ht_up = ZPOS ? ht.x : ht.y;
    const float hv_dn = ZPOS ? hv.y : hv.x, ht_dn = ZPOS ? ht.y : ht.x;
    int m = -ij;                            // pair of this step: k' = 2m, 2m + 1
    const float *vb = img + (row * TILE_Z + (ZPOS ? 2 * m : TILE_Z - 2 - 2 * m) - C);
    tile_f2 vp = LD2(C), tp = LD2(SIXC_T + C);                  // own pair
    tile_f2 nv = LD2(C + DP), nt = LD2(SIXC_T + C + DP);        // own next pair
    float vzu = hv_up, tzu = ht_up;                             // own previous cell (set again when m == 0)
    tile_f2 vxm = LD2(C - DX), vxp = LD2(C + DX), vym = LD2(C - DY), vyp = LD2(C + DY);
    bool improved = false;
#pragma unroll 2
    for (int d = 0; d < TILE_X + TILE_Y + TILE_Z / 2 - 2; d++, m++, vb += DP) {
        // possibly written in the previous step: asked for first
        tile_f2 txm = LD2(SIXC_T + C - DX), txp = LD2(SIXC_T + C + DX), tym = LD2(SIXC_T + C - DY), typ = LD2(SIXC_T + C + DY);
        __builtin_amdgcn_sched_barrier(0);
        // final already, wanted by the next pair (and the pair after it along the column)
        const tile_f2 n_vxm = LD2(C + DP - DX), n_vxp = LD2(C + DP + DX), n_vym = LD2(C + DP - DY), n_vyp = LD2(C + DP + DY);
        tile_f2 n_nv = LD2(C + 2 * DP), n_nt = LD2(SIXC_T + C + 2 * DP);
        __builtin_amdgcn_sched_barrier(0);
        // the row ends: the cell before the first pair and the cell behind the last one are halo cells
        if (m == 0) { vzu = hv_up; tzu = ht_up; }
        if (m == TILE_Z / 2 - 1) {
            if (ZPOS) { nv.x = hv_dn; nt.x = ht_dn; } else { nv.y = hv_dn; nt.y = ht_dn; }
        }
        // this pair: lateral delays (both cells at once), the column's delays and candidates
        const tile_f2 lxm = hxm2 * (vp + vxm), lxp = hxp2 * (vp + vxp), lym = hym2 * (vp + vym), lyp = hyp2 * (vp + vyp);
        const float vf = FIRST(vp), vs = SECOND(vp), tf = FIRST(tp), ts = SECOND(tp);
        const float sfs = vf + vs;
        const float czu_f = hzu * (vf + vzu) + tzu;             // first cell from its z-upwind neighbour
        const float czd_f = hzd * sfs + ts;                     // first from second (old value)
        const float lzu_s = hzu * sfs;                          // second from first (this step's result)
        const float czd_s = hzd * (vs + FIRST(nv)) + FIRST(nt); // second from the next pair's first
        float pre_f = fminf(tf, fminf(czu_f, czd_f)), pre_s = fminf(ts, czd_s);
        const bool mine_f = (unsigned)(2 * m - klo) < span, mine_s = (unsigned)(2 * m + 1 - klo) < span;
        asm volatile("" : "+v"(pre_f), "+v"(pre_s), "+v"(txm), "+v"(txp), "+v"(tym), "+v"(typ));
        const tile_f2 cxm = lxm + txm, cxp = lxp + txp, cym = lym + tym, cyp = lyp + typ;
        pre_s = fminf(fminf(pre_s, fminf(SECOND(cxm), SECOND(cxp))), fminf(SECOND(cym), SECOND(cyp)));
        float best_f = fminf(fminf(pre_f, fminf(FIRST(cxm), FIRST(cxp))), fminf(FIRST(cym), FIRST(cyp)));
        best_f = mine_f ? best_f : tf;
        float best_s = fminf(pre_s, lzu_s + best_f);
        best_s = mine_s ? best_s : ts;
        asm volatile("" : "+v"(best_f), "+v"(best_s));
        if (best_f < tf || best_s < ts) {
            *reinterpret_cast<tile_f2 *>(const_cast<float *>(vb) + SIXC_T + C) = ZPOS ? tile_f2{best_f, best_s} : tile_f2{best_s, best_f};
            improved = true;
        }
        tzu = best_s; vzu = vs;
        vp = nv; tp = nt; nv = n_nv; nt = n_nt;
        vxm = n_vxm; vxp = n_vxp; vym = n_vym; vyp = n_vyp;
        __builtin_amdgcn_wave_barrier();
    }
#undef FIRST
#undef SECOND
#undef LD2
    return improved;
}

__global__ void __launch_bounds__(64)
tile_six_kernel(TileSweep P)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x;
    const DevLayout &L = P.L;
    float *const img = lds + SIXC_PAD;              // v rows; T rows SIXC_T behind; halo tables behind them

    const int ip = lane >> 3, jp = lane & 7;
    const int ci = P.sx > 0 ? ip : TILE_X - 1 - ip;
    const int cj = P.sy > 0 ? jp : TILE_Y - 1 - jp;
    const int row = (ci + 1) * SIX_SY + (cj + 1) - 1;          // this lane's image row
    const unsigned s0b = (unsigned)(L.s0 * 4), s1b = (unsigned)(L.s1 * 4);

#ifdef TTSWEEP_TILE_PROFILE
    unsigned long long prof_acc[5] = {};
#endif
    PROF_STAMP(t_begin);
    const int ncand = P.NJ * P.NK;
    const int xcd = uni((int)blockIdx.x % P.nxcd);
    const long long W = P.wstride;              // workgroups of an XCD that take candidates (coprime to nactive)
    const long long nseq = (long long)P.NJ * ((P.NK + P.nxcd - 1) / P.nxcd) * P.nactive;   // candidates of an XCD
    TileWork work;
    int first_next = INT_MAX;       // (uniform) see tile_next_plane
    for (long long q0 = blockIdx.x / P.nxcd; q0 < nseq && (long long)(blockIdx.x / P.nxcd) < W; q0 += 64 * W) {
      const TileCand pick = tile_candidate(P, q0 + lane * W, xcd, ncand);
      unsigned long long due_lanes = __ballot(pick.due);
      while (due_lanes) {
        const int src = __builtin_ctzll(due_lanes);
        due_lanes &= due_lanes - 1;
        const int tile = __builtin_amdgcn_readlane(pick.tile, src);
        const int act = __builtin_amdgcn_readlane(pick.a, src);
        const int s = uni(P.active[act]);
        tile_work_count(P, work, act, (unsigned)__builtin_amdgcn_readlane((int)pick.cells, src), lane);
        // (the start's volume and activity words from the launch arguments: no descriptor load)
        float *const T = uni_ptr(P.T0 + (long long)s * L.cells);
        int2 *const state = uni_ptr(reinterpret_cast<int2 *>(P.state0 + (long long)s * P.state_stride));
        PROF_STAMP(t_top);
        const int K = tile % P.NK, J = (tile / P.NK) % P.NJ, I = tile / (P.NK * P.NJ);

        // ---- stage: rows (x - 1 .. x + 8, y - 1 .. y + 8) but the two corner rows at the ends,
        // their 32 interior cells each (one 128-byte line), and the rows' z-halo cells from the faces
        const long long g0 = (long long)(I * TILE_X + L.lo[0] - 1) * L.s0
                           + (long long)(J * TILE_Y + L.lo[1] - 1) * L.s1 + (K * TILE_Z + L.lo[2]);
        float *const tface = P.tface + (long long)s * P.face_cells;
        {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // (the previous tile's LDS traffic has retired)
            const tile_rsrc rv = tile_make_rsrc(uni_ptr(P.v + g0)), rt = tile_make_rsrc(uni_ptr(T + g0));
#pragma unroll
            for (int it = 0; it < SIXC_NITER; it++) {
                const int sl = it * 64 + lane;
                if (sl < SIXC_NSLOTS) {
                    const int r = (sl >> 3) + 1, q = sl & 7;
                    const int ri = r / SIX_SY, rj = r - ri * SIX_SY;
                    const unsigned off = (unsigned)ri * s0b + (unsigned)rj * s1b + (unsigned)q * 16u;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, (__attribute__((address_space(3))) void *)(img + it * 256),
                                                             16, (int)off, 0, 0, 0);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rt, (__attribute__((address_space(3))) void *)(img + SIXC_T + it * 256),
                                                             16, (int)off, 0, 0, 0);
                }
            }
            const tile_rsrc fv = tile_make_rsrc(uni_ptr(P.vface)), ft = tile_make_rsrc(uni_ptr(tface));
#pragma unroll
            for (int it = 0; it < SIXC_NFIT; it++) {
                const int f = it * 64 + lane;
                if (f < SIXC_NFACE) {
                    const int r = (f >> 1) + 1, side = f & 1;
                    const int ri = r / SIX_SY, rj = r - ri * SIX_SY;
                    // below the tile: boundary K, its lower side; above: boundary K + 1, its upper side
                    const unsigned off = (unsigned)(tile_face_index(L, 1, K + side, side, 0, I * TILE_X + ri, J * TILE_Y + rj) * 4);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(fv, (__attribute__((address_space(3))) void *)(img + SIXC_HV + it * 64),
                                                             4, (int)off, 0, 0, 0);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(ft, (__attribute__((address_space(3))) void *)(img + SIXC_HT + it * 64),
                                                             4, (int)off, 0, 0, 0);
                }
            }
        }
        const bool xy_ok = I * TILE_X + ci < L.n[0] && J * TILE_Y + cj < L.n[1];
        const int z_cells = min(TILE_Z, L.n[2] - K * TILE_Z);
        const int klo = P.sz > 0 ? 0 : TILE_Z - z_cells;
        const unsigned span = xy_ok ? (unsigned)z_cells : 0u;

        PROF_STAMP(t_issued);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the image has landed (one wave: no barrier)
        PROF_STAMP(t_landed);
        const bool improved = P.sz > 0 ? sixc_sweep<true>(img, row, ip + jp, klo, span, P)
                                       : sixc_sweep<false>(img, row, ip + jp, klo, span, P);
        PROF_STAMP(t_swept);

        // ---- write the tile back if it improved: 64 rows of 8 float4
        if (__ballot(improved) != 0ull) {
            const long long t0 = (long long)(I * TILE_X + L.lo[0]) * L.s0 + (long long)(J * TILE_Y + L.lo[1]) * L.s1
                               + (K * TILE_Z + L.lo[2]);
#pragma unroll
            for (int it = 0; it < TILE_X * TILE_Y * (TILE_Z / 4) / 64; it++) {
                const int r = (it * 64 + lane) >> 3, q = lane & 7;
                const int ri = r >> 3, rj = r & 7;
                const float4 val = *reinterpret_cast<const float4 *>(
                    img + SIXC_T + ((ri + 1) * SIX_SY + (rj + 1) - 1) * TILE_Z + 4 * q);
                *reinterpret_cast<float4 *>(T + t0 + (long long)ri * L.s0 + (long long)rj * L.s1 + 4 * q) = val;
            }
            // this column's first and last cell into the faces of the tile's two boundaries
            const int pa = I * TILE_X + 1 + ci, pb = J * TILE_Y + 1 + cj;
            tface[tile_face_index(L, 1, K, 1, 0, pa, pb)] = img[SIXC_T + row * TILE_Z];
            tface[tile_face_index(L, 1, K + 1, 0, 0, pa, pb)] = img[SIXC_T + row * TILE_Z + TILE_Z - 1];
            tile_stamp_neighbours(P, state, I, J, K, lane);
            tile_work_improved(work, act, lane);
            first_next = min(first_next, tile_next_plane(P, I, J, K));
        }
#ifdef TTSWEEP_TILE_PROFILE
        {
            const long long t_end = prof_now();
            prof_acc[0] += (unsigned long long)(t_issued - t_top);
            prof_acc[1] += (unsigned long long)(t_landed - t_issued);
            prof_acc[2] += (unsigned long long)(t_swept - t_landed);
            prof_acc[3] += (unsigned long long)(t_end - t_swept);
            prof_acc[4] += 1ull;
        }
#endif
      }
    }
    tile_work_flush(P, work);
    if (lane == 0 && first_next != INT_MAX) atomicMin(P.dmin_next, first_next);
#ifdef TTSWEEP_TILE_PROFILE
    if (lane == 0 && prof_acc[4]) {
        for (int i = 0; i < 5; i++) atomicAdd(&g_tile_prof[i], prof_acc[i]);
        atomicAdd(&g_tile_prof[5], (unsigned long long)(prof_now() - t_begin));
    }
#endif
}

size_t tile_lds_bytes(int R)
{
    const int nslots = (TILE_X + 2 * R) * (TILE_Y + 2 * R) * TILE_QPR;
    return (size_t)2 * nslots * 16;
}

typedef void (*tile_sweep_fn)(TileSweep);

// the plain 6-neighbour star (entries in the pull star's sorted order), every edge live in both directions
bool tile_star_is_six(const TileEntry *ent, int nent, int R)
{
    static const int six[6][3] = {{-1, 0, 0}, {0, -1, 0}, {0, 0, -1}, {0, 0, 1}, {0, 1, 0}, {1, 0, 0}};
    bool is_six = nent == 6 && R == 1;
    for (int e = 0; e < 6 && is_six; e++)
        is_six = ent[e].flags == (PULL_FWD | PULL_REV) && ent[e].da == six[e][0] && ent[e].db == six[e][1]
              && ent[e].dc == six[e][2];
    return is_six;
}

// The instance that relaxes this star (see tile_sweep_kernel's template parameters).
static tile_sweep_fn tile_instance(const TileSweep &P)
{
    bool exact = false;
    for (int e = 0; e < P.nent; e++) exact |= P.ent[e].flags != (PULL_FWD | PULL_REV);
    // the plain 6-neighbour star has its own instance
    // (its face offsets are 32-bit byte offsets from one descriptor: a start's faces must stay
    // below 4 GiB - about a 2600^3 grid -, beyond that the general kernel's 64-bit indices serve)
    const bool is_six = tile_star_is_six(P.ent, P.nent, P.R) && tile_face_cells(P.L, 1) * 4 < 0x100000000LL;
    if (is_six) return tile_six_kernel;
    if (P.nent <= 6) return exact ? tile_sweep_kernel<6, true> : tile_sweep_kernel<6, false>;
    if (P.nent <= 18) return exact ? tile_sweep_kernel<18, true> : tile_sweep_kernel<18, false>;
    return exact ? tile_sweep_kernel<TILE_MAX_ENT, true> : tile_sweep_kernel<TILE_MAX_ENT, false>;
}

static size_t tile_instance_lds(const TileSweep &P)
{
    return tile_instance(P) == (tile_sweep_fn)tile_six_kernel ? (size_t)SIXC_FLOATS * sizeof(float) : tile_lds_bytes(P.R);
}

static bool tile_sweep_ok(const TileSweep &P)
{
    static_assert(TILE_ZF == 4, "the z halo of an image row is one float4 on either side");
    return P.R >= 1 && P.R <= TILE_MAX_R && P.nent >= 1 && P.nent <= TILE_MAX_ENT
        && P.fz >= 1 && P.fz <= TILE_ZF && P.vface && P.tface && P.state0 && P.work0 && P.T0
        && P.L.lo[0] == P.R && P.L.lo[1] == P.R && P.L.lo[2] >= TILE_ZF && P.L.lo[2] % TILE_Z == 0   // whole-line rows
        && P.L.s1 % TILE_Z == 0;
}

hipError_t tile_sweep_wgs_per_cu(const TileSweep &P, int *wgs)
{
    // What the device really holds at once (the LDS is allocated in granules: five 32 KiB
    // images do NOT fit 160 KiB).  The sweep kernel strides the list statically, so a grid
    // larger than this would run its surplus workgroups, full list shares and all, after the
    // others have finished.
    if (!tile_sweep_ok(P)) return hipErrorInvalidValue;
    int n = 0;
    const hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void *)tile_instance(P), 64,
                                                                      tile_instance_lds(P));
    if (e != hipSuccess) return e;
    *wgs = std::max(n, 1);
    return hipSuccess;
}

hipError_t launch_tile_sweep(const TileSweep &P, hipStream_t st)
{
    if (P.nactive <= 0) return hipSuccess;
    if (!tile_sweep_ok(P) || P.nxcd < 1 || P.nblocks < P.nxcd || P.nblocks % P.nxcd || !P.wgwork || P.nstart < P.nactive
        || P.wstride < 1 || P.wstride > P.nblocks / P.nxcd)
        return hipErrorInvalidValue;
    hipLaunchKernelGGL(tile_instance(P), dim3((unsigned)P.nblocks), dim3(64), tile_instance_lds(P), st, P);
    return hipGetLastError();
}

} // namespace ttsweep
