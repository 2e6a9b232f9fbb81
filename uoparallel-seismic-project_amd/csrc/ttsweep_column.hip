// ttsweep_column.hip - sweep, variant TILE, plain 6-neighbour star: ONE launch per solve ("column pipelines").
//
// Same relaxation and the same eight orderings as ttsweep_tile.hip (serial_new/sweep-tt-multistart.c:203-249 is
// the (+,+,+) ordering; old/wavefront-openmp/wave-multistart.c:415-530 the reference's plane orders), but the
// driver loop (serial_new/...:151-170) runs on the device and the hyperplane launches are gone:
//   * a COLUMN is the stack of NK tiles (8 x 8 x 32 cells each) above one tile position (I, J).  One wavefront
//     owns a column for one ordering sweep.  Lane (i', j') walks the z-row of cell column (i', j'), ONE cell per
//     step, i' + j' cells behind lane (0, 0): a systolic Gauss-Seidel sweep as in sixc_sweep, but the pipeline
//     does not drain at tile boundaries - consecutive due tiles of the column are one run of 32 steps per tile
//     (+ 14 once per run; a tile on its own is 46).  The rows live in LDS as a ring of three 16-cell chunks
//     (64 bytes of every one of the 98 image rows, v and T): while the lanes are spread over chunks j - 1 and j,
//     chunk j + 1 is in flight (LDS-DMA, 14 steps ahead) into the slot chunk j - 2 was written back from.
//     Every lane address of the 48-step ring period is a register (AX[48]); a step has no address arithmetic.
//   * Columns are claimed in sweep order - by level I' + J' of the tile position, one sequence per XCD - and
//     wait for each other through ONE 64-bit progress word per column and start (ColumnSolve, ttsweep_dev.h):
//     tile k of a column is staged when both upwind columns have finished their tile k of this sweep, a column
//     starts its sweep e when it and its four neighbours have sealed sweep e - 1.  So nobody writes what a
//     column reads while it reads it: every sweep computes exactly what the sequential ordering sweep computes.
//     Successive sweeps overlap (sweep e + 1 follows sweep e across the grid), no launch, no host round trip.
//   * A tile is relaxed in sweep e when it is DUE: a neighbour that comes later in the sweep order (or the tile
//     itself) improved in the sweep before (read from the words the columns sealed it with), or an upwind neighbour improved in THIS
//     sweep (the upwind columns' progress words carry the bits; inside the column the pipeline just goes on).
//     A sweep in which no tile of a start improved leaves no bit: the start is at rest (every column hands a flag
//     "a tile improved, here or upwind" on to its downwind columns; the sweep's last column reads it).
// Visibility between workgroups (MI355X_MICROARCH.md, "inter-workgroup visibility"): travel times are stored
// write-through (sc1), the storing wave waits for its stores (vmcnt(0)) before it publishes progress (an sc1
// store); a wave that has read progress (sc1 loads) invalidates its L1 (buffer_inv sc1) before it stages.
// Results: bit-identical to the reference's fixed point (every value is the length of a path and only ever
// decreases; rest is detected only when every due tile has been relaxed against final neighbours).
#include "ttsweep_kernels.h"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <vector>

namespace ttsweep {

namespace {

typedef __amdgpu_buffer_rsrc_t col_rsrc;
typedef float col_f2 __attribute__((ext_vector_type(2)));
typedef unsigned col_u4 __attribute__((ext_vector_type(4)));
typedef float col_f4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) char *lds_char;

constexpr int CS = 16;                          // cells of a chunk along z
constexpr int CRING = 3;                        // chunks of a row the ring holds
constexpr int CNR = (TILE_X + 2) * (TILE_Y + 2) - 2;   // image rows: 10 x 10 but the first and the last (98)
// A chunk in LDS: the float4s of an image row (cells 4 q .. 4 q + 3) come in groups of CG; group q / CG of row r
// lives in the 16-byte slots (q / CG) CKQ CG + r CG + q % CG.  CG = 4: plain rows of 64 bytes (the default).
// Neighbouring rows are a constant 16 CG (y) / 160 CG (x) bytes apart whatever the cell - one address table serves a
// lane and its four lateral neighbours.  With plain rows every LDS access of a step is a 2-way bank conflict (a
// 32-lane group falls on 16 banks: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE 0.49); CG = 2 with CKQ = 100 and CG = 1
// with CKQ = 104 (padding rows per plane) put the 32 lanes of EVERY access - own row or a neighbour's, any step of
// the ring period, either z direction - on 32 different banks (measured: the conflict counter reads 0).  But the
// LDS-DMA places lane l of an instruction at byte 16 l of its block, so consecutive lanes fetch consecutive bytes of
// the volume only within a group: an instruction touches 16 / 32 / 64 cache lines for CG = 4 / 2 / 1, and the solve
// of 1024x1024x512 x 14 takes 115.5 / 129.4 / 155.7 ms (profiles/r04_lds_layouts.txt): the staging path, not the
// LDS array, is what the conflict-free images load; the conflicts themselves cost nothing measurable (the LDS array
// is 33 % busy with them).
#ifndef TTSWEEP_COL_CG
#define TTSWEEP_COL_CG 4
#endif
constexpr int CG = TTSWEEP_COL_CG;              // float4s of a row that stay together
constexpr int CKQ = CG == 1 ? 104 : CG == 2 ? 100 : CNR;       // rows of a plane (>= CNR)
constexpr int CROWB = 16 * CG;                  // bytes from an image row to the next
constexpr int CPLANEB = CKQ * CROWB;            // bytes of a plane
constexpr int CSLOTB = (CS / 4 / CG) * CPLANEB; // bytes of a chunk (one array)
constexpr int CARRB = (CRING * CSLOTB + 255) / 256 * 256;      // the ring of one array (whole 256-byte blocks)
constexpr int CLDSB = 2 * CARRB;                // v ring, T ring
constexpr int CDX = (TILE_Y + 2) * CROWB;       // from an image row to its x + 1 neighbour
constexpr int CPER = CS * CRING;                // steps of the ring period
constexpr int CNSLOT = CSLOTB / 16;             // 16-byte slots of a chunk
constexpr int CNDMA = (CNSLOT + 63) / 64;       // LDS-DMA wave instructions per chunk and array (7)
constexpr int CTAIL = CNSLOT % 64 ? CNSLOT % 64 : 64;  // lanes of the last one
constexpr int CSIG = TILE_X + TILE_Y - 2;       // largest lane skew (14)
// wavefronts of a workgroup: each works on its own column with its own rings, they never meet (no barrier); they are
// ONE workgroup so that the CU places them on its four SIMDs, one each (single-wavefront workgroups land wherever the
// dispatcher's round robin stands: two of four on one SIMD in some launches, and the launch is then 5 % slower)
#ifndef TTSWEEP_COL_WG_WAVES
#define TTSWEEP_COL_WG_WAVES 4
#endif
constexpr int CWG = TTSWEEP_COL_WG_WAVES;
static_assert(CWG >= 1 && CWG <= COL_WAVES, "a workgroup's rings must fit the CU's LDS");
static_assert(TILE_X == 8 && TILE_Y == 8 && TILE_Z == 2 * CS, "column pipelines: 8 x 8 x 32 tiles");
static_assert((CG == 1 || CG == 2 || CG == 4) && CKQ >= CNR && COL_WAVES * CLDSB <= 160 * 1024, "ring size");
// byte offset of cell z (0 .. CS - 1) of image row r in its chunk
__host__ __device__ constexpr int col_cell_off(int r, int z) { return ((z >> 2) / CG) * CPLANEB + r * CROWB + ((z >> 2) % CG) * 16 + (z & 3) * 4; }
// 16-byte slot `slot` of a chunk: its image row (the padding rows of a plane stand for its last row) and float4
__host__ __device__ constexpr int col_slot_row(int slot) { return (slot % (CKQ * CG)) / CG < CNR ? (slot % (CKQ * CG)) / CG : CNR - 1; }
__host__ __device__ constexpr int col_slot_quad(int slot) { return (slot / (CKQ * CG) < CS / 4 / CG ? slot / (CKQ * CG) : CS / 4 / CG - 1) * CG + slot % CG; }

__device__ __forceinline__ col_rsrc col_make_rsrc(const float *base)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base), 0, 0xffffffff, 0x00020000);
}
__device__ __forceinline__ int cuni(int x) { return __builtin_amdgcn_readfirstlane(x); }
template <typename T>
__device__ __forceinline__ T *cuni_ptr(T *p)
{
    const unsigned long long u = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    return reinterpret_cast<T *>(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ unsigned long long cld64(const unsigned long long *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned cld32(const unsigned *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ long long col_clock()
{
    unsigned long long t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    return (long long)t;
}
__device__ __forceinline__ void col_fail(const ColumnSolve &P, unsigned code)
{
    atomicCAS(P.status, (unsigned)COL_RUNNING, code);
}
__device__ __forceinline__ unsigned col_flip(unsigned m, int nk) { return __brev(m) >> (32 - nk); }

// progress word: sweep << 41 | tiles finished (0xff: sealed) << 33 | "a tile improved in this sweep, in this column or in
// a column upwind of it, however far" << 32 | improved tiles of this column (sweep order).  The flag is final in the words
// of a column that has finished all its tiles (its upwind columns then have): the LAST column of a sweep - every column
// is upwind of it - reads from it whether the sweep improved anything (round 4 counted sealed and improved columns with
// an atomic per column on a word per start and sweep, and the last column waited for the count).
__device__ __forceinline__ unsigned col_key(int sweep, int cnt) { return ((unsigned)sweep << 9) | ((unsigned)cnt << 1); }

// (the kernel has no static LDS: the dynamic array starts at LDS address 0 and an image offset IS the address -
// checked on the host side of the launch; saves the addition of a base that is zero)
#define CLDS_F(off) (*reinterpret_cast<__attribute__((address_space(3))) const float *>((unsigned)(off)))
#define CLDS_W(off) (*reinterpret_cast<__attribute__((address_space(3))) float *>((unsigned)(off)))

// what a lane carries from step to step (lateral neighbours in the pairs the LDS delivers them in: ds_read2_b32)
struct ColRegs {
    float vc, tc;               // the cell of this step
    float vn, tn;               // the next cell of the row (z-downwind neighbour)
    float wz, tzu;              // the previous cell (z-upwind neighbour): the edge to it (hz (v + v')), its result
    col_f2 va, vb;              // velocities of the lateral neighbours of this step: (x-, y-), (y+, x+)
};

struct ColConst {
    col_f2 ha, hb;              // d / 2 of (x-, y-), (y+, x+)
    float hz;                   // ... of the two z entries (equal: column_solve's caller checks)
    int sigact;                 // the lane's skew i' + j', or a huge number when its cell column lies outside the grid
};

// One step: every lane relaxes one cell against its six neighbours.  Of its inputs only the travel times of the
// four lateral neighbours can have been written in the previous step (by the two upwind lanes): they are read
// first, behind that write; everything else arrived during the previous steps.
// FULL: every lane relaxes a cell of the run in this step (a block in the middle of a run): no activity test.
template <int N, bool FULL>
__device__ __forceinline__ void col_step(const lds_char lp, const int (&AX)[CPER], ColRegs &r, const ColConst &c, const int tb,
                                         const unsigned span, unsigned long long &imp)
{
    constexpr int N1 = (N + 1) % CPER, N2 = (N + 2) % CPER;
    const int a0 = AX[N], a1 = AX[N1], a2 = AX[N2];
    col_f2 ta, tb2, nva, nvb;
    ta.x = CLDS_F(a0 + CARRB);                  ta.y = CLDS_F(a0 + CARRB + CDX - CROWB);
    tb2.x = CLDS_F(a0 + CARRB + CDX + CROWB);   tb2.y = CLDS_F(a0 + CARRB + 2 * CDX);
    nva.x = CLDS_F(a1);                         nva.y = CLDS_F(a1 + CDX - CROWB);
    nvb.x = CLDS_F(a1 + CDX + CROWB);           nvb.y = CLDS_F(a1 + 2 * CDX);
    const float vnn = CLDS_F(a2 + CDX), tnn = CLDS_F(a2 + CARRB + CDX);
    // (scalar operations, not v_pk_*_f32 over the neighbour pairs: with ONE wavefront on a SIMD the packed forms
    // issue more slowly than twice the plain ones - measured 123.8 against 126.0 ms per solve on 1024x1024x512 x 14;
    // the empty asm statements keep the vectoriser from pairing them again)
    float s0 = r.vc + r.va.x, s1 = r.vc + r.va.y, s2 = r.vc + r.vb.x, s3 = r.vc + r.vb.y;
    asm volatile("" : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3));
    float l0 = c.ha.x * s0, l1 = c.ha.y * s1, l2 = c.hb.x * s2, l3 = c.hb.y * s3;
    asm volatile("" : "+v"(l0), "+v"(l1), "+v"(l2), "+v"(l3));
    // (the edge to the next cell is the next step's edge to the previous one: the same sum, the same product)
    const float wzd = c.hz * (r.vc + r.vn);
    const float czu = r.wz + r.tzu;
    const float czd = wzd + r.tn;
    const float pre = fminf(r.tc, fminf(czu, czd));
    float c0 = l0 + ta.x, c1 = l1 + ta.y, c2 = l2 + tb2.x, c3 = l3 + tb2.y;
    asm volatile("" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3));
    float best = fminf(fminf(pre, fminf(c0, c1)), fminf(c2, c3));
    if (!FULL) {
        const bool act = (unsigned)(tb + (N % CS) - c.sigact) < span;  // (tb: the block's first step less the first cell)
        best = act ? best : r.tc;
    }
    imp |= __ballot(best < r.tc);
    CLDS_W(a0 + CARRB + CDX) = best;
    r.wz = wzd; r.tzu = best;
    r.vc = r.vn; r.tc = r.tn;
    r.vn = vnn; r.tn = tnn;
    r.va = nva; r.vb = nvb;
    __builtin_amdgcn_sched_barrier(0);
}

template <int N0, int CNT, bool FULL>
struct ColSteps {
    static __device__ __forceinline__ void run(const lds_char lp, const int (&AX)[CPER], ColRegs &r, const ColConst &c,
                                               const int tb, const unsigned span, unsigned long long &imp)
    {
        col_step<N0, FULL>(lp, AX, r, c, tb, span, imp);
        ColSteps<N0 + 1, CNT - 1, FULL>::run(lp, AX, r, c, tb, span, imp);
    }
};
template <int N0, bool FULL>
struct ColSteps<N0, 0, FULL> {
    static __device__ __forceinline__ void run(const lds_char, const int (&)[CPER], ColRegs &, const ColConst &, const int,
                                               const unsigned, unsigned long long &) {}
};

// chunk (v and T) -> ring slot SLOT: 7 LDS-DMA wave instructions per array, 64 float4 each (the last one 8).
// tvalid: bit q set - this lane's row of instruction q exists in the travel-time volume (the caller's array has
// no rows around the grid: those image rows hold +INFINITY, col_fill_rows); zin: the chunk lies inside the rows
// (a chunk in front of z = 0 or behind the last cell is +INFINITY in the image instead).
// RIM: the column lies at the rim of the grid or the chunk outside the rows (the caller's array only): the general
// form; everywhere else both arrays are staged without a test.
template <int SLOT, bool RIM>
__device__ __forceinline__ void col_stage(const lds_char lp, const float *vsrc, const float *tsrc, const unsigned (&goffv)[CNDMA],
                                          const unsigned (&gofft)[CNDMA], const int lane, const unsigned tvalid, const bool zin)
{
    const col_rsrc rv = col_make_rsrc(vsrc), rt = col_make_rsrc(tsrc);
#pragma unroll
    for (int q = 0; q < CNDMA; q++) {
        if (q < CNDMA - 1 || lane < CTAIL) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, (__attribute__((address_space(3))) void *)(lp + SLOT * CSLOTB + q * 1024),
                                                     16, (int)goffv[q], 0, 0, 0);
            if (!RIM || zin) {
                if (!RIM || ((tvalid >> q) & 1u))
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rt, (__attribute__((address_space(3))) void *)(lp + CARRB + SLOT * CSLOTB + q * 1024),
                                                             16, (int)gofft[q], 0, 0, 0);
            } else {
                const float inf = __builtin_inff();
                *reinterpret_cast<__attribute__((address_space(3))) col_f4 *>(lp + CARRB + SLOT * CSLOTB + q * 1024 + lane * 16) =
                    col_f4{inf, inf, inf, inf};
            }
        }
    }
}
template <int SLOT>
__device__ __forceinline__ void col_stage_any(const lds_char lp, const float *vsrc, const float *tsrc, const unsigned (&goffv)[CNDMA],
                                              const unsigned (&gofft)[CNDMA], const int lane, const unsigned tvalid, const bool zin,
                                              const bool rim)
{
    if (rim || !zin) col_stage<SLOT, true>(lp, vsrc, tsrc, goffv, gofft, lane, tvalid, zin);
    else col_stage<SLOT, false>(lp, vsrc, tsrc, goffv, gofft, lane, tvalid, zin);
}

// the image rows that do not exist in the travel-time volume (tvalid) hold +INFINITY in every ring slot
__device__ __forceinline__ void col_fill_rows(const lds_char lp, const int lane, const unsigned tvalid)
{
    const float inf = __builtin_inff();
#pragma unroll
    for (int slot = 0; slot < CRING; slot++)
#pragma unroll
        for (int q = 0; q < CNDMA; q++)
            if ((q < CNDMA - 1 || lane < CTAIL) && !((tvalid >> q) & 1u))
                *reinterpret_cast<__attribute__((address_space(3))) col_f4 *>(lp + CARRB + slot * CSLOTB + q * 1024 + lane * 16) =
                    col_f4{inf, inf, inf, inf};
}

// the 64 interior rows of the T chunk in ring slot SLOT -> registers (as soon as the last lane has left the chunk:
// two steps before the block ends) -> the volume (at the block boundary; write-through: sc1)
template <int SLOT>
__device__ __forceinline__ void col_wb_read(const lds_char lp, const int (&wbl)[4], col_u4 (&x)[4])
{
#pragma unroll
    for (int k = 0; k < 4; k++)
        x[k] = *reinterpret_cast<__attribute__((address_space(3))) const col_u4 *>(lp + wbl[k] + SLOT * CSLOTB);
}
__device__ __forceinline__ void col_wb_store(float *tdst, const unsigned (&wbg)[4], const col_u4 (&x)[4], const unsigned wbvalid,
                                             const bool rim)
{
    const col_rsrc rt = col_make_rsrc(tdst);
    if (rim) {
#pragma unroll
        for (int k = 0; k < 4; k++)
            if ((wbvalid >> k) & 1u) __builtin_amdgcn_raw_buffer_store_b128(x[k], rt, (int)wbg[k], 0, 16 /* sc1 */);
    } else {
#pragma unroll
        for (int k = 0; k < 4; k++) __builtin_amdgcn_raw_buffer_store_b128(x[k], rt, (int)wbg[k], 0, 16 /* sc1 */);
    }
}

struct ColWork {
    unsigned long long cells = 0, tiles = 0;
    int s = -1;                 // the start this lane counts for (lane == s % 64)
};
__device__ __forceinline__ void col_work_flush(const ColumnSolve &P, ColWork &w)
{
    if (w.s >= 0 && w.tiles) {
        unsigned long long *const slot = P.wgwork + (((size_t)blockIdx.x * CWG + (threadIdx.x >> 6)) * P.nstart + w.s) * 2;
        slot[0] += w.cells * 6ull;
        slot[1] += w.tiles;
    }
    w.cells = 0; w.tiles = 0;
}
__device__ __forceinline__ void col_work_add(const ColumnSolve &P, ColWork &w, int s, unsigned long long cells, unsigned tiles, int lane)
{
    if (lane == (s & 63)) {
        if (w.s != s) { col_work_flush(P, w); w.s = s; }
        w.cells += cells;
        w.tiles += tiles;
    }
}

// Lanes 0 .. 6 poll one progress word each (pa; !valid: nothing to wait for) until its key reaches `need`.
// false: the solve is over for this column (the start is at rest, the solve has failed or ended): the launch's status
// word and the start's "at rest" word ride along in lanes 62 and 63 of EVERY round - a column claimed for a start that
// has come to rest leaves with the first one, and no round has round trips of its own for them (round 4: three
// dependent loads every fourth round).
__device__ __forceinline__ bool col_poll(const ColumnSolve &P, const unsigned long long *pa, bool valid, unsigned need,
                                         unsigned long long &pv, int s, long long deadline)
{
    const int lane = threadIdx.x & 63;
    for (unsigned spin = 0;; spin++) {
        unsigned long long x = ~0ull;
        if (valid) x = cld64(pa);
        else if (lane == 62) x = cld32(P.status);
        else if (lane == 63) x = cld32(reinterpret_cast<const unsigned *>(P.done + s));
        pv = valid ? x : ~0ull;
        if (__ballot(!valid && ((lane == 62 && (unsigned)x != (unsigned)COL_RUNNING) || (lane == 63 && (unsigned)x != 0u))) != 0ull)
            return false;
        const bool ok = !valid || (unsigned)(pv >> 32) >= need;
        if (__ballot(!ok) == 0ull) return true;
        if ((spin & 3u) == 3u && col_clock() > deadline) { col_fail(P, COL_ERR_TIMEOUT); return false; }
        __builtin_amdgcn_s_sleep(4);
    }
}

// what the progress words of the two upwind columns (lanes 1 and 2: pv) say about sweep e
// upflag: |= the upwind columns' "improved, here or upwind" flags, once they have finished all their tiles
__device__ __forceinline__ void col_upwind(unsigned long long pv, bool valid, int e, int NK, int &known, unsigned &mask, unsigned &upflag)
{
    const int sw = (int)(pv >> 41), cn = (int)((pv >> 33) & 0xffu);
    int c = NK;
    unsigned m = 0;
    int fl = 0;
    if (valid) {
        c = sw == e ? min(cn, NK) : (sw > e ? NK : 0);
        m = sw == e ? (unsigned)pv : 0u;
        fl = sw == e && cn >= NK ? (int)((pv >> 32) & 1ull) : 0;
    }
    const int c1 = __builtin_amdgcn_readlane(c, 1), c2 = __builtin_amdgcn_readlane(c, 2);
    known = min(c1, c2);
    mask = (unsigned)__builtin_amdgcn_readlane((int)m, 1) | (unsigned)__builtin_amdgcn_readlane((int)m, 2);
    upflag |= (unsigned)(__builtin_amdgcn_readlane(fl, 1) | __builtin_amdgcn_readlane(fl, 2));
}

// -DTTSWEEP_COL_PROFILE: where the wavefronts' time goes (cycles summed over all wavefronts; tuning aid, never a result)
#ifdef TTSWEEP_COL_PROFILE
__device__ unsigned long long g_col_prof[16];
__device__ unsigned long long g_col_sweep_work[64][4];  // per sweep: tiles relaxed, runs, cycles in runs, columns with a run (all starts)
__device__ long long g_col_level_time[3][512];      // the last seal of every level of sweeps 14 .. 16 of start 6 (wall clock)
__device__ long long g_col_sweep_time[64];      // (wall clock) at which the last column of sweep e of start 0 sealed
__device__ long long g_col_rest_time[64];       // (wall clock, 100 MHz) at which start s came to rest; [63]: the launch's first stamp
#define CPROF_NOW() col_cycles()
#define CPROF_ADD(i, x) prof[i] += (unsigned long long)(x)
__device__ __forceinline__ long long col_cycles()
{
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    return (long long)t;
}
#else
#define CPROF_NOW() 0ll
#define CPROF_ADD(i, x)
#endif

} // namespace

#ifdef TTSWEEP_COL_TRACE
// -DTTSWEEP_COL_TRACE: a log of the protocol's events (debugging aid): 8 words per record
__device__ unsigned g_col_trace[8u << 20];
__device__ unsigned g_col_trace_n;
#define CTRACE(type, a, b, c, d)                                                                   \
    if (lane == 0) {                                                                               \
        const unsigned at = atomicAdd(&g_col_trace_n, 1u);                                         \
        if (at < (1u << 20)) {                                                                     \
            unsigned *const w = g_col_trace + 8u * at;                                             \
            w[0] = (type); w[1] = (unsigned)s; w[2] = (unsigned)e; w[3] = (unsigned)col;           \
            w[4] = (unsigned)(a); w[5] = (unsigned)(b); w[6] = (unsigned)(c); w[7] = (unsigned)(d); \
        }                                                                                          \
    }
#else
#define CTRACE(type, a, b, c, d)
#endif

#ifdef TTSWEEP_COL_TRACE
void column_trace_dump()
{
    unsigned n = 0;
    (void)hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_col_trace_n), sizeof(n));
    n = std::min(n, 1u << 20);
    std::vector<unsigned> h((size_t)n * 8);
    if (n) (void)hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_col_trace), h.size() * 4);
    const char *path = getenv("TTSWEEP_COL_TRACE_FILE");
    if (FILE *f = fopen(path ? path : "/tmp/col_trace.bin", "wb")) { fwrite(h.data(), 4, h.size(), f); fclose(f); }
    const unsigned z = 0;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_col_trace_n), &z, sizeof(z));
}
#endif

#ifdef TTSWEEP_COL_PROFILE
void column_prof_dump()
{
    unsigned long long h[16] = {};
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_col_prof), sizeof(h));
    const double tot = (double)std::max<unsigned long long>(h[0], 1);
    fprintf(stderr, "column prof: share of resident wavefront time: claim+setup %.3f  wait(previous sweep) %.3f  wait(upwind) %.3f  "
            "run prologue %.3f  run steps %.3f  in-run waits %.3f  run end+seal %.3f | columns %llu (start at rest %llu, quiet %llu)  "
            "runs %llu  tiles %llu  blocks %llu  cycles per block %.0f  per tile (steps only) %.0f | after a wavefront's last run %.3f, before its first %.3f\n",
            h[1] / tot, h[2] / tot, h[3] / tot, h[4] / tot, h[5] / tot, h[6] / tot, h[7] / tot, h[8], h[9], h[10], h[11], h[12], h[13],
            (double)h[5] / (double)std::max<unsigned long long>(h[13], 1), (double)h[5] / (double)std::max<unsigned long long>(h[12], 1),
            h[14] / tot, h[15] / tot);
    unsigned long long z[16] = {};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_col_prof), z, sizeof(z));
    long long t[64] = {};
    (void)hipMemcpyFromSymbol(t, HIP_SYMBOL(g_col_rest_time), sizeof(t));
    fprintf(stderr, "column prof: starts at rest after (ms):");
    for (int s2 = 0; s2 < 63; s2++)
        if (t[s2]) fprintf(stderr, " %.1f", (double)(t[s2] - t[63]) / 1.0e5);
    fprintf(stderr, "\n");
    long long zz[64] = {};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_col_rest_time), zz, sizeof(zz));
    long long w[64] = {};
    (void)hipMemcpyFromSymbol(w, HIP_SYMBOL(g_col_sweep_time), sizeof(w));
    fprintf(stderr, "column prof: the last column of sweep e of start 0 sealed after (ms):");
    for (int e2 = 1; e2 < 64; e2++)
        if (w[e2]) fprintf(stderr, " %d:%.1f", e2, (double)(w[e2] - t[63]) / 1.0e5);
    fprintf(stderr, "\n");
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_col_sweep_time), zz, sizeof(zz));
    // what the sweeps were: work, and how fast the late ones cross the grid
    unsigned long long sw[64][4] = {};
    (void)hipMemcpyFromSymbol(sw, HIP_SYMBOL(g_col_sweep_work), sizeof(sw));
    fprintf(stderr, "column prof: per sweep (all starts) tiles / runs / columns with a run / microseconds per run (2.4 GHz):");
    for (int e2 = 1; e2 < 64; e2++)
        if (sw[e2][1]) fprintf(stderr, " %d: %llu/%llu/%llu/%.1f", e2, sw[e2][0], sw[e2][1], sw[e2][3], (double)sw[e2][2] / (double)sw[e2][1] / 2400.0);
    fprintf(stderr, "\n");
    static long long lt[3][512];
    (void)hipMemcpyFromSymbol(lt, HIP_SYMBOL(g_col_level_time), sizeof(lt));
    for (int q = 0; q < 3; q++) {
        fprintf(stderr, "column prof: start 6 sweep %d, every 16th level's last seal (ms):", 14 + q);
        for (int l = 0; l < 512; l += 16)
            if (lt[q][l]) fprintf(stderr, " %d:%.2f", l, (double)(lt[q][l] - t[63]) / 1.0e5);
        fprintf(stderr, "\n");
    }
    memset(lt, 0, sizeof(lt));
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_col_level_time), lt, sizeof(lt));
    unsigned long long zs[64][4] = {};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_col_sweep_work), zs, sizeof(zs));
}
#endif

__global__ void __launch_bounds__(64 * CWG)
column_solve_kernel(const ColumnSolve P)
{
    extern __shared__ __attribute__((aligned(256))) char col_lds[];
    const int lbase = cuni((int)(threadIdx.x >> 6)) * CLDSB;      // this wavefront's rings
    const lds_char lp = (lds_char)col_lds + lbase;
    const int lane = threadIdx.x & 63;
    const DevLayout &L = P.L;
    const long long clock0 = col_clock();
    const long long deadline = clock0 + P.timeout_ticks;
#ifdef TTSWEEP_COL_PROFILE
    if (blockIdx.x == 0 && threadIdx.x == 0) g_col_rest_time[63] = clock0;
#endif
    if ((unsigned)(unsigned long long)lp != (unsigned)lbase) {      // (CLDS_F: image offsets are used as LDS addresses)
        col_fail(P, COL_ERR_LDS_BASE);
        return;
    }
    const int ncol = P.NI * P.NJ;

    // per-lane constants of the staging and write-back instructions (image coordinates: the same for every ordering)
    unsigned goffv[CNDMA], gofft[CNDMA];
#pragma unroll
    for (int q = 0; q < CNDMA; q++) {
        // (the padding slots of a float4 plane fetch its last row once more: nobody reads them)
        const int sidx = 64 * q + lane, quad = col_slot_quad(sidx), r = col_slot_row(sidx);
        const int ri = (r + 1) / (TILE_Y + 2), rj = (r + 1) % (TILE_Y + 2);        // padded image coordinates 0 .. 9
        goffv[q] = (unsigned)(((long long)ri * L.s0 + (long long)rj * L.s1) * 4 + quad * 16);
        gofft[q] = (unsigned)(((long long)ri * P.ts0 + (long long)rj * P.ts1) * 4 + quad * 16);
    }
    unsigned wbg[4];
    int wbl[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int sidx = 64 * k + lane, ir = sidx >> 2, quad = sidx & 3, ci = ir >> 3, cj = ir & 7;
        wbg[k] = (unsigned)(((long long)(ci + 1) * P.ts0 + (long long)(cj + 1) * P.ts1) * 4 + quad * 16);
        wbl[k] = CARRB + col_cell_off((ci + 1) * (TILE_Y + 2) + cj, 4 * quad);
    }

    const int seq = cuni((int)blockIdx.x % P.nseq);
    const int seqlen = P.seq_len[seq];
    const int *const seqtab = P.seqtab + P.seq_off[seq];
    const long long per = (long long)seqlen * P.nstart;         // entries of one sweep in this sequence
    const bool small_claims = per > 0 && per < (1ll << 20);
    const float inv_per = 1.0f / (float)per, inv_nstart = 1.0f / (float)P.nstart;

    ColWork work;
#ifdef TTSWEEP_COL_PROFILE
    unsigned long long prof[16] = {};
    const long long prof_begin = CPROF_NOW();
#endif

    for (;;) {
        if (cld32(P.status) != COL_RUNNING) break;
        const long long pt0 = CPROF_NOW();
        (void)pt0;
        // ---- claim the next column of this sequence
        unsigned long long q = 0;
        if (lane == 0) q = atomicAdd(P.claim + seq * 16, 1ull);
        q = ((unsigned long long)(unsigned)cuni((int)(q >> 32)) << 32) | (unsigned)cuni((int)(unsigned)q);
        // q = (sweep, position, start): two divisions by numbers that are fixed for the launch - a 64-bit and a 32-bit
        // integer division are a hundred and more instructions per claim; a reciprocal estimate is off by at most one
        // (sweep < 4096, position x start < 2^20: the float's 24 bits leave 1e-3 of a unit), corrected exactly
        long long qs;
        int rem;
        if (small_claims) {
            qs = (long long)(unsigned)cuni((int)(unsigned)((float)q * inv_per));
            long long r = (long long)q - qs * per;
            if (r < 0) { qs--; r += per; } else if (r >= per) { qs++; r -= per; }
            rem = (int)r;
        } else {
            qs = (long long)(q / (unsigned long long)per);
            rem = (int)(q - (unsigned long long)qs * (unsigned long long)per);
        }
        if (qs >= COL_MAX_SWEEPS - 2) { col_fail(P, COL_ERR_CAP); break; }
        const int e = 1 + (int)qs;                              // sweep, 1-based
        int pos, s;
        if (small_claims) {
            pos = cuni((int)((float)rem * inv_nstart));
            s = rem - pos * P.nstart;
            if (s < 0) { pos--; s += P.nstart; } else if (s >= P.nstart) { pos++; s -= P.nstart; }
        } else {
            pos = rem / P.nstart;
            s = rem - pos * P.nstart;
        }
        CPROF_ADD(8, 1);
        const int packed = seqtab[pos];
        const int ip = packed & 0xffff, jp = packed >> 16;
        // the eight orderings in Gray-code order: successive sweeps differ in ONE sign, so sweep e + 1 starts at a corner
        // sweep e passed half-way through (x or y flipped) or at its own first corner (z flipped) and follows it across
        // the grid; in binary order every second change of sweep flips x and y together - the next sweep then starts
        // where the previous one ENDS and the two cannot overlap at all
        const unsigned long long oseq = P.ordseq[s];
        const int o = (int)(oseq >> (4 * ((e - 1) & 15))) & 7;
        const int sx = (o & 1) ? -1 : 1, sy = (o & 2) ? -1 : 1, sz = (o & 4) ? -1 : 1;
        const int I = sx > 0 ? ip : P.NI - 1 - ip, J = sy > 0 ? jp : P.NJ - 1 - jp;
        // the lane's image addresses of the ring period (made anew for every column, new ordering or not: a table that
        // is carried from column to column and only sometimes rebuilt lives twice in the registers)
        const int li = lane >> 3, lj = lane & 7;
        const int ci = sx > 0 ? li : TILE_X - 1 - li, cj = sy > 0 ? lj : TILE_Y - 1 - lj, sig = li + lj;
        // (... and only for a column that relaxes anything: before its first run)
        int AX[CPER];
        bool ax_ready = false;
        auto build_ax = [&]() {
            const int rxm = ci * (TILE_Y + 2) + cj;             // image row of the x - 1 neighbour (own row: + CDX)
#pragma unroll
            for (int n = 0; n < CPER; n++) {
                const int wq = n - sig + (n < sig ? CPER : 0);
                const int slot = wq >> 4, zc = wq & (CS - 1);
                const int z = sz > 0 ? zc : CS - 1 - zc;
                AX[n] = lbase + slot * CSLOTB + col_cell_off(rxm, z);
            }
            ax_ready = true;
        };
        ColConst cc;
        cc.ha = col_f2{P.h[0], P.h[1]};
        cc.hb = col_f2{P.h[4], P.h[5]};
        cc.hz = P.h[2];
        cc.sigact = (I * TILE_X + ci < L.n[0] && J * TILE_Y + cj < L.n[1]) ? sig : 0x3fffffff;

        // ---- the progress words of this column (lane 0), its upwind (1, 2) and downwind (3, 4) neighbours.  The words
        // are kept twice, by the parity of the sweep: sweep e writes buffer e & 1, and what the columns sealed sweep
        // e - 1 with - their improved tiles - stays readable in the other buffer until sweep e + 1 (which none of the
        // five can begin before this column has sealed e).  From those masks the column derives the tiles that are due
        // in sweep e from earlier sweeps - round 4 kept them in a word of bits per column, set with three atomics (and a
        // wait in front of the seal) by every column that improved, fetched with a returning atomic by every column.
        unsigned long long *const prog = P.prog + ((size_t)(e & 1) * P.nstart + s) * ncol;              // this sweep's
        const unsigned long long *const prog_prev = P.prog + ((size_t)((e - 1) & 1) * P.nstart + s) * ncol;
        const int col = I * P.NJ + J;
        int ncolumn = col;
        bool valid = lane == 0;
        if (lane == 1 || lane == 5) { valid = ip > 0; ncolumn = (I - sx) * P.NJ + J; }
        if (lane == 2 || lane == 6) { valid = jp > 0; ncolumn = I * P.NJ + (J - sy); }
        if (lane == 3) { valid = ip < P.NI - 1; ncolumn = (I + sx) * P.NJ + J; }
        if (lane == 4) { valid = jp < P.NJ - 1; ncolumn = I * P.NJ + (J + sy); }
        if (!valid) ncolumn = col;
        const unsigned long long *const pa = prog + ncolumn;      // (lanes 1, 2: the upwind columns' words of THIS sweep)
        unsigned long long pv = 0;
        // nobody is still in sweep e - 1 around this column (lanes 0 .. 4: the five words of sweep e - 1); the same round
        // brings the upwind columns' words of this sweep (lanes 5, 6: nothing to wait for) and the status words (lanes 62, 63)
        const long long pt1 = CPROF_NOW();
        (void)pt1;
        CPROF_ADD(1, pt1 - pt0);
        if (!col_poll(P, lane < 5 ? prog_prev + ncolumn : pa, valid && lane < 7, lane < 5 ? col_key(e - 1, 0xff) : 0u, pv, s, deadline)) {
            CPROF_ADD(9, 1);
            continue;
        }
        long long pt2 = CPROF_NOW();
        (void)pt2;
        CPROF_ADD(2, pt2 - pt1);
        int prof_runs = 0;
        (void)prof_runs;
        // (the cap is raised only by a column that is about to RELAX a tile in a sweep beyond it - below -: a column of
        // sweep max_sweeps + 1 can pass its poll before the last column of sweep max_sweeps has declared the start at
        // rest, and has nothing due then)
        const bool over = e > P.max_sweeps;
        // (what the polled words stand for is acquired - the L1 invalidated - where this column first stages cells: a
        // quiet column, 44 % of the claims of a solve, stages nothing, and the invalidate with its wait is 1.7 us)
        bool need_acquire = true;
        const bool upvalid = valid && (lane == 1 || lane == 2);

        // the tiles that are due from earlier sweeps: whoever came EARLIER in sweep e - 1's order than an improved tile
        // and touches it has to look again - the column itself (the improved tiles and the tile below each in that
        // order) and the two columns it was the upwind column of: their sealed words (absolute K -> this sweep's order)
        unsigned mask0 = 0;
        if (e == 1) {
            if (lane == 0) mask0 = cld32(P.due + (size_t)s * ncol + col);       // (the first sweep: what column_init marked)
            mask0 = (unsigned)cuni((int)mask0);
        } else {
            const int op = (int)(oseq >> (4 * ((e - 2) & 15))) & 7;             // the ordering of sweep e - 1
            const int sxp = (op & 1) ? -1 : 1, syp = (op & 2) ? -1 : 1, szp = (op & 4) ? -1 : 1;
            const unsigned mlo = (unsigned)pv;
            const bool lv = valid;
            const unsigned own = (unsigned)__builtin_amdgcn_readlane((int)mlo, 0);
            // (the column this one was upwind of along x in sweep e - 1 is its downwind neighbour of this sweep when x kept
            // its sign - lane 3 -, its upwind neighbour when x flipped - lane 1; y: lanes 4 / 2)
            const unsigned m1 = __builtin_amdgcn_readlane(lv ? (int)mlo : 0, 1), m2 = __builtin_amdgcn_readlane(lv ? (int)mlo : 0, 2);
            const unsigned m3 = __builtin_amdgcn_readlane(lv ? (int)mlo : 0, 3), m4 = __builtin_amdgcn_readlane(lv ? (int)mlo : 0, 4);
            unsigned m = own | (own >> 1) | (sxp == sx ? m3 : m1) | (syp == sy ? m4 : m2);
            if (szp < 0) m = col_flip(m, P.NK);
            mask0 = m & (P.NK >= 32 ? ~0u : ((1u << P.NK) - 1u));
        }
        if (sz < 0) mask0 = col_flip(mask0, P.NK);

        // what the upwind columns have finished of THIS sweep (their words came with the poll: lanes 5, 6 -> 1, 2)
        {
            const unsigned lo5 = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)pv, 5), hi5 = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(pv >> 32), 5);
            const unsigned lo6 = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)pv, 6), hi6 = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(pv >> 32), 6);
            if (lane == 1) pv = ((unsigned long long)hi5 << 32) | lo5;
            if (lane == 2) pv = ((unsigned long long)hi6 << 32) | lo6;
        }
        int known_up = 0;
        unsigned upmask = 0, upflag = 0;
        col_upwind(pv, upvalid, e, P.NK, known_up, upmask, upflag);
        CTRACE(1u, mask0, known_up, upmask, (unsigned)col_clock());

        const float *const vcol = P.v + (long long)(I * TILE_X) * L.s0 + (long long)(J * TILE_Y) * L.s1 + L.lo[2];
        // (image row (0, 0) of the column: one row and one column in front of the tile; the caller's array has none
        // there - the address is then only ever the base of offsets that lead back into the array)
        float *const tcol = cuni_ptr(P.tptr[s]) + (long long)(I * TILE_X - 1 + P.tpad) * P.ts0
                          + (long long)(J * TILE_Y - 1 + P.tpad) * P.ts1 + P.tlo;
        const int dx = min(TILE_X, L.n[0] - I * TILE_X), dy = min(TILE_Y, L.n[1] - J * TILE_Y);
        const bool whole = dx == TILE_X && dy == TILE_Y;       // every lane's cell column lies inside the grid
        // rows of the image that do not exist in the caller's array (columns at the rim of the grid)
        unsigned tvalid = (1u << CNDMA) - 1u, wbvalid = 0xfu;
        const bool rim = P.tpad == 0 && (I == 0 || J == 0 || I * TILE_X + TILE_X + 1 > L.n[0] || J * TILE_Y + TILE_Y + 1 > L.n[1]);
        if (rim) {
            tvalid = 0u;
#pragma unroll
            for (int q = 0; q < CNDMA; q++) {
                const int r = col_slot_row(64 * q + lane);
                const int x = I * TILE_X + (r + 1) / (TILE_Y + 2) - 1, y = J * TILE_Y + (r + 1) % (TILE_Y + 2) - 1;
                if ((unsigned)x < (unsigned)L.n[0] && (unsigned)y < (unsigned)L.n[1]) tvalid |= 1u << q;
            }
            wbvalid = 0u;
#pragma unroll
            for (int kk = 0; kk < 4; kk++) {
                const int ir = (64 * kk + lane) >> 2;
                if (I * TILE_X + (ir >> 3) < L.n[0] && J * TILE_Y + (ir & 7) < L.n[1]) wbvalid |= 1u << kk;
            }
            col_fill_rows(lp, lane, tvalid);
        }

        unsigned mymask = 0, pend = 0;      // improved tiles; tiles made due by a late improvement below them
        int published = 0;
        bool alive = true;
        int k = 0;
        while (k < P.NK) {
            if (known_up <= k) {
                // the upwind columns have not finished tile k: tell what is done here, then wait for them
                if (published < k) {
                    if (lane == 0) __hip_atomic_store(prog + col, ((unsigned long long)col_key(e, k) << 32) | mymask,
                                                      __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    published = k;
                }
                const long long pw0 = CPROF_NOW();
                (void)pw0;
                if (!col_poll(P, pa, upvalid, col_key(e, k + 1), pv, s, deadline)) { alive = false; break; }
                need_acquire = true;
                col_upwind(pv, upvalid, e, P.NK, known_up, upmask, upflag);
                CPROF_ADD(3, CPROF_NOW() - pw0);
            }
            const unsigned avail = known_up >= 32 ? ~0u : ((1u << known_up) - 1u);
            const unsigned dm = (mask0 | upmask | pend) & avail & ~((1u << k) - 1u);
            if (dm == 0u) { k = known_up; continue; }
            if (over) {         // a tile is due beyond the cap - unless the start has been declared at rest meanwhile
                if (cld32(reinterpret_cast<const unsigned *>(P.done + s)) == 0u) col_fail(P, COL_ERR_CAP);
                alive = false;
                break;
            }
            const int k0 = __builtin_ctz(dm);
            if (published < k0) {
                if (lane == 0) __hip_atomic_store(prog + col, ((unsigned long long)col_key(e, k0) << 32) | mymask,
                                                  __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                published = k0;
            }

            // ================= a run: tiles k0, k0 + 1, ... of the column as one pipeline =================
            // relative cell w of lane sigma at step t: w = t - sigma; chunk cr = w >> 4 lives in ring slot cr mod 3
            auto zlo = [&](int cr) { return sz > 0 ? TILE_Z * k0 + CS * cr : TILE_Z * (P.NK - k0) - CS * (cr + 1); };
            int wend = TILE_Z;                                  // cells of the run (grows while the tiles go on being due)
            // cells of the run that lie inside the grid: wlo <= w < whi
            const int wlo = sz > 0 ? 0 : max(TILE_Z * (P.NK - k0) - L.n[2], 0);
            auto whi = [&]() { return sz > 0 ? min(wend, L.n[2] - TILE_Z * k0) : wend; };
            const long long pr0 = CPROF_NOW();
            (void)pr0;
            // (the caller's rows end where the grid ends: a chunk in front of or behind them is not staged)
            auto zin = [&](int cr) { return P.tpad != 0 || (unsigned)zlo(cr) < (unsigned)L.n[2]; };
            if (need_acquire) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                need_acquire = false;
            }
            if (!ax_ready) build_ax();
            col_stage_any<2>(lp, cuni_ptr(vcol + zlo(-1)), cuni_ptr(tcol + zlo(-1)), goffv, gofft, lane, tvalid, zin(-1), rim);
            col_stage_any<0>(lp, cuni_ptr(vcol + zlo(0)), cuni_ptr(tcol + zlo(0)), goffv, gofft, lane, tvalid, zin(0), rim);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            ColRegs r;
            r.tzu = CLDS_F(AX[CPER - 1] + CARRB + CDX);
            r.vc = CLDS_F(AX[0] + CDX); r.tc = CLDS_F(AX[0] + CARRB + CDX);
            r.wz = cc.hz * (CLDS_F(AX[CPER - 1] + CDX) + r.vc);
            r.vn = CLDS_F(AX[1] + CDX); r.tn = CLDS_F(AX[1] + CARRB + CDX);
            r.va.x = CLDS_F(AX[0]); r.va.y = CLDS_F(AX[0] + CDX - CROWB);
            r.vb.x = CLDS_F(AX[0] + CDX + CROWB); r.vb.y = CLDS_F(AX[0] + 2 * CDX);

            const long long pr1 = CPROF_NOW();
            (void)pr1;
            CPROF_ADD(4, pr1 - pr0);
            unsigned long long prof_inwait = 0;
            (void)prof_inwait;
            bool imp1 = false;                  // a cell improved in block j - 1
            col_u4 wb[4];                       // chunk j - 1 on its way to the volume
            bool wb_pending = false;
            bool late = false;                  // ... after the run had been closed
            bool closed = false;
            unsigned tilebits = 0;              // tiles of the run that improved (bit: tile index in the run)
            int pub_pending = -1;               // tiles finished (absolute count) to publish behind the next vmcnt(0)
            unsigned long long pvn = pv;        // upwind progress asked for a block ahead

#define COL_BLOCK(JM)                                                                                              \
            {                                                                                                      \
                /* ---- boundary j: chunk j - 2 is behind every lane (in registers, if it improved), chunk j + 1 is  \
                   wanted in 15 steps */                                                                           \
                if (wb_pending) {                                                                                  \
                    col_wb_store(cuni_ptr(tcol + zlo(j - 2)), wbg, wb, wbvalid, rim);                              \
                    wb_pending = false;                                                                            \
                }                                                                                                  \
                if ((j & 1) && j >= 3) pub_pending = k0 + (j - 1) / 2;      /* tile (j - 3) / 2 is complete */     \
                if ((j & 1) && !closed) {                                                                          \
                    /* the first lane enters tile k0 + tr + 1 in 16 steps: is it part of the run? */              \
                    const int kt = k0 + (j + 1) / 2;                                                               \
                    if (kt >= P.NK) closed = true;                                                                 \
                    else {                                                                                         \
                        if (known_up <= kt) {                                                                      \
                            col_upwind(pvn, upvalid, e, P.NK, known_up, upmask, upflag);                                   \
                            if (known_up <= kt) {                                                                  \
                                const long long pq0 = CPROF_NOW();                                                 \
                                if (!col_poll(P, pa, upvalid, col_key(e, kt + 1), pv, s, deadline)) { alive = false; closed = true; } \
                                else col_upwind(pv, upvalid, e, P.NK, known_up, upmask, upflag);                           \
                                prof_inwait += (unsigned long long)(CPROF_NOW() - pq0);                            \
                            }                                                                                      \
                            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");                                     \
                        }                                                                                          \
                        if (alive && ((((mask0 | upmask | pend) >> kt) & 1u) || imp1)) wend += TILE_Z;             \
                        else closed = true;                                                                        \
                    }                                                                                              \
                }                                                                                                  \
                if (CS * (j + 1) <= wend)                                                                          \
                    col_stage_any<(JM + 1) % 3>(lp, cuni_ptr(vcol + zlo(j + 1)), cuni_ptr(tcol + zlo(j + 1)), goffv, gofft, lane, tvalid, zin(j + 1), rim); \
                if (upvalid) pvn = cld64(pa);                                                                      \
                const int tb = CS * j - wlo;                                                                       \
                const unsigned span = (unsigned)max(whi() - wlo, 0);                                               \
                unsigned long long imp = 0;                                                                        \
                /* (a block in the middle of a run, every cell column inside the grid: no lane is ever idle) */    \
                const bool full = whole && tb >= CSIG && tb + CS - 1 < (int)span;                                  \
                if (full) ColSteps<CS * JM, CS - 2, true>::run(lp, AX, r, cc, tb, span, imp);                      \
                else ColSteps<CS * JM, CS - 2, false>::run(lp, AX, r, cc, tb, span, imp);                          \
                /* (the block's last two steps read the next chunk: a lane's next cell but one, its neighbours' next) */ \
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                   \
                if (pub_pending >= 0) {                                                                            \
                    if (lane == 0) __hip_atomic_store(prog + col, ((unsigned long long)col_key(e, pub_pending) << 32) \
                                                      | mymask | (tilebits << k0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); \
                    published = pub_pending;                                                                       \
                    pub_pending = -1;                                                                              \
                }                                                                                                  \
                /* (the last lane has just left chunk j - 1) */                                                    \
                if (j >= 1 && CS * (j - 1) < wend && (imp1 || imp != 0ull)) {                                      \
                    col_wb_read<(JM + 2) % 3>(lp, wbl, wb);                                                        \
                    wb_pending = true;                                                                             \
                }                                                                                                  \
                if (full) ColSteps<CS * JM + CS - 2, 2, true>::run(lp, AX, r, cc, tb, span, imp);                  \
                else ColSteps<CS * JM + CS - 2, 2, false>::run(lp, AX, r, cc, tb, span, imp);                      \
                imp1 = imp != 0ull;                                                                                \
                if (imp1) {                                                                                        \
                    tilebits |= 1u << (j >> 1);                                                                    \
                    if (j >= 1) tilebits |= 1u << ((j - 1) >> 1);                                                  \
                    if (closed) late = true;                                                                       \
                }                                                                                                  \
                j++;                                                                                               \
            }

            int j = 0;
            for (;;) {
                if (CS * j >= wend + CSIG + 1) break;
                COL_BLOCK(0)
                if (CS * j >= wend + CSIG + 1) break;
                COL_BLOCK(1)
                if (CS * j >= wend + CSIG + 1) break;
                COL_BLOCK(2)
            }
#undef COL_BLOCK
            // ---- the last chunk(s) of the run: blocks 0 .. j - 1 ran, chunks up to j - 3 are written back
            const int nt = wend / TILE_Z;                       // tiles of the run
            const long long pr2 = CPROF_NOW();
            (void)pr2;
            CPROF_ADD(5, (unsigned long long)(pr2 - pr1) - prof_inwait);
#ifdef TTSWEEP_COL_PROFILE
            prof[14] = (unsigned long long)pr2;         // (the end of this wavefront's last run, so far)
            if (prof[15] == 0ull) prof[15] = (unsigned long long)pr1;      // (the beginning of its first)
#endif
            CPROF_ADD(6, prof_inwait);
            CPROF_ADD(11, 1);
            CPROF_ADD(12, nt);
#ifdef TTSWEEP_COL_PROFILE
            if (lane == 0 && e < 64) {
                atomicAdd(&g_col_sweep_work[e][0], (unsigned long long)nt);
                atomicAdd(&g_col_sweep_work[e][1], 1ull);
                atomicAdd(&g_col_sweep_work[e][2], (unsigned long long)(pr2 - pr0));
                if (prof_runs == 0) atomicAdd(&g_col_sweep_work[e][3], 1ull);
            }
#endif
            CPROF_ADD(13, j);
            prof_runs++;
            tilebits &= nt >= 32 ? ~0u : ((1u << nt) - 1u);
            // (chunk j - 2 = 2 nt - 1, the run's last one, was read in block j - 1 = 2 nt)
            if (wb_pending) col_wb_store(cuni_ptr(tcol + zlo(j - 2)), wbg, wb, wbvalid, rim);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            mymask |= tilebits << k0;
            const int kend = k0 + nt;
            if (late && kend < P.NK) pend |= 1u << kend;
            // (kend == NK: the upwind columns have finished too - the flag is final)
            if (lane == 0) __hip_atomic_store(prog + col, ((unsigned long long)(col_key(e, kend) | (kend >= P.NK && (mymask != 0u || upflag != 0u) ? 1u : 0u)) << 32) | mymask,
                                              __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            published = kend;
            CTRACE(2u, k0 | (nt << 8) | (kend << 16), tilebits, mask0 | upmask | pend, (unsigned)col_clock());
            {   // work: the cells of the run's tiles that lie inside the grid
                unsigned long long cells = 0;
                for (int t = k0; t < kend; t++) {
                    const int K = sz > 0 ? t : P.NK - 1 - t;
                    cells += (unsigned long long)(dx * dy * min(TILE_Z, L.n[2] - K * TILE_Z));
                }
                col_work_add(P, work, s, cells, (unsigned)nt, lane);
            }
            k = kend;
            CPROF_ADD(7, CPROF_NOW() - pr2);
            if (!alive) break;
        }
        if (!alive) continue;
        const long long ps0 = CPROF_NOW();
        (void)ps0;
        if (prof_runs == 0) CPROF_ADD(10, 1);

        // ---- seal: the column is done with sweep e.  (What it improved is in the word it seals with: the columns around
        // it read their due tiles from it when they begin sweep e + 1.)
        if (mymask != 0u && lane == 0) P.changed[s] = CHANGED_IMPROVED;
        CTRACE(3u, mymask, o, upmask, (unsigned)col_clock());
        const unsigned anyimp = mymask != 0u || upflag != 0u ? 1u : 0u;     // (every upwind column has finished: final)
        if (lane == 0)
            __hip_atomic_store(prog + col, ((unsigned long long)(col_key(e, 0xff) | anyimp) << 32) | mymask, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
#ifdef TTSWEEP_COL_PROFILE
        if (ip == P.NI - 1 && jp == P.NJ - 1 && s == 0 && e < 64 && lane == 0) g_col_sweep_time[e] = col_clock();
        if (s == 6 && e >= 14 && e <= 16 && ip + jp < 512 && lane == 0) atomicMax((unsigned long long *)&g_col_level_time[e - 14][ip + jp], (unsigned long long)col_clock());
#endif
        if (ip == P.NI - 1 && jp == P.NJ - 1 && anyimp == 0u && lane == 0) {
            // the sweep's last column - every column is upwind of it, and has finished its tiles -: no tile of the start
            // improved in sweep e: the start is at rest.  (Sweep e + 1 may be under way behind sweep e and come to the
            // same end before its last column hears of this one: ONE of them counts.)
            CTRACE(4u, 0, 0, 0, (unsigned)col_clock());
            if (atomicCAS(P.done + s, 0, e) == 0) {
#ifdef TTSWEEP_COL_PROFILE
                if (s < 63) g_col_rest_time[s] = col_clock();
#endif
                if (atomicSub(P.status + 1, 1u) == 1u) atomicCAS(P.status, (unsigned)COL_RUNNING, (unsigned)COL_DONE);
            }
        }
        CPROF_ADD(7, CPROF_NOW() - ps0);
    }
    col_work_flush(P, work);
#ifdef TTSWEEP_COL_PROFILE
    prof[0] = (unsigned long long)(CPROF_NOW() - prof_begin);
    // (14: from the wavefront's last run to its exit - the tail of the launch as this wavefront saw it; 15: from its
    // entry to its first run)
    prof[14] = prof[14] ? (unsigned long long)CPROF_NOW() - prof[14] : prof[0];
    prof[15] = prof[15] ? prof[15] - (unsigned long long)prof_begin : 0ull;
    if (lane == 0)
        for (int i = 0; i < 16; i++) atomicAdd(&g_col_prof[i], prof[i]);
#endif
}

// ---------------------------------------------------------------------------
// first state of a solve: every column sealed in "sweep 0", the start's tile (and its neighbours) due
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
column_init_kernel(ColumnSolve P, const StartDesc *__restrict__ starts, int from_box)
{
    const int ncol = P.NI * P.NJ;
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t < (long long)P.nstart * ncol) {
        const int s = (int)(t / ncol), col = (int)(t - (long long)s * ncol);
        const int I = col / P.NJ, J = col - I * P.NJ;
        const StartDesc &sd = starts[s];
        const int si = sd.sa / TILE_X, sj = sd.sb / TILE_Y, sk = sd.sc / TILE_Z;
        unsigned m = 0;
        if (from_box) m = P.NK >= 32 ? ~0u : ((1u << P.NK) - 1u);
        else if (abs(I - si) <= 1 && abs(J - sj) <= 1)
            for (int K = max(sk - 1, 0); K <= min(sk + 1, P.NK - 1); K++) m |= 1u << K;
        P.due[t] = m;
        P.prog[t] = (unsigned long long)col_key(0, 0xff) << 32;                             // (both buffers: "sweep 0" sealed,
        P.prog[(size_t)P.nstart * ncol + t] = (unsigned long long)col_key(0, 0xff) << 32;   //  nothing improved)
    }
    if (t < P.nstart) { P.done[t] = 0; P.changed[t] = 0; }
    if (t < COL_SEQS * 16) P.claim[t] = 0ull;
    if (t == 0) { P.status[0] = COL_RUNNING; P.status[1] = (unsigned)P.nstart; }
}

// ---------------------------------------------------------------------------
// what the default choice of a start's sequence of orderings looks at: the velocity values along the vertical line
// through the start - the value at the start, the least and the largest of the line (out[3 s ..])
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(64)
column_line_kernel(const float *__restrict__ v, DevLayout L, const StartDesc *__restrict__ starts, float *__restrict__ out)
{
    const StartDesc &sd = starts[blockIdx.x];
    const float *const row = v + dev_index(L, sd.sa, sd.sb, 0);
    float lo = INFINITY, hi = -INFINITY;
    for (int c = threadIdx.x; c < L.n[2]; c += 64) {
        const float x = row[c];
        lo = fminf(lo, x);
        hi = fmaxf(hi, x);
    }
    for (int off = 32; off; off >>= 1) {
        lo = fminf(lo, __shfl_xor(lo, off));
        hi = fmaxf(hi, __shfl_xor(hi, off));
    }
    if (threadIdx.x == 0) {
        out[3 * blockIdx.x] = row[sd.sc];
        out[3 * blockIdx.x + 1] = lo;
        out[3 * blockIdx.x + 2] = hi;
    }
}

hipError_t launch_column_line(const float *v, const DevLayout &L, const StartDesc *starts, int nstart, float *out, hipStream_t st)
{
    hipLaunchKernelGGL(column_line_kernel, dim3((unsigned)nstart), dim3(64), 0, st, v, L, starts, out);
    return hipGetLastError();
}

// The default (TTSWEEP_OPT_TILE_ORDER = -1): by where the start lies in the velocity profile of its vertical line.
// The volume holds delay per distance (the reference adds h (v + v') along an edge): a start where that is LARGE - a
// source at the surface of a medium that gets faster with depth, the reference's own data - sends rays that dive and
// come back, and each lateral quadrant wants its downward and its upward ordering next to each other (table 5: z flips
// with every sweep); a start at the FAST end of its line is served better by the x-fastest cyclic code (table 1), and so
// is a start at the slow end of a grid that is at least as deep as it is wide.  Measured on nine geometries and four
// depths of the starts (profiles/r05_col_sweep_order.txt, second half): 1024 x 1024 x 512 x 14 from the surface 87 -> 78
// ms, 1024 x 1024 x 256: 87 -> 58, 241 x 241 x 51 x 24: 5.8 -> 5.0; the cubes keep table 1.
int column_order_default(const int (&n)[3], float at_start, float least, float largest)
{
    const float range = largest - least;
    const float f = range > 0.f ? (at_start - least) / range : 0.5f;       // 0: the fast end of the line, 1: the slow end
    if (f < 0.15f) return 111;
    if (f > 0.85f && n[2] >= std::max(n[0], n[1])) return 111;
    return 115;
}

int column_solve_wg_waves() { return CWG; }

// The opt-in to more than 64 KB of dynamic LDS belongs to the CURRENT device: every context (one per device in
// ttsweep_solve_multi / _multi_device, each on a thread of its own) asks for it with its device bound, and
// launch_column_solve asks again on every launch, as launch_solve_units does - no process-wide flag.
static hipError_t column_raise_lds()
{
    return hipFuncSetAttribute((const void *)column_solve_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, CWG * CLDSB);
}

hipError_t column_solve_wgs_per_cu(int *wgs)
{
    hipError_t e = column_raise_lds();
    if (e != hipSuccess) return e;
    int n = 0;
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void *)column_solve_kernel, 64 * CWG, CWG * CLDSB);
    if (e != hipSuccess) return e;
    *wgs = std::min(std::max(n, 1), (int)COL_WAVES / CWG);
    return hipSuccess;
}

hipError_t launch_column_init(const ColumnSolve &P, const StartDesc *starts, bool from_box, hipStream_t st)
{
    const long long n = std::max<long long>(std::max<long long>((long long)P.nstart * P.NI * P.NJ, (long long)P.nstart * COL_MAX_SWEEPS),
                                            COL_SEQS * 16);
    hipLaunchKernelGGL(column_init_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, P, starts, from_box ? 1 : 0);
    return hipGetLastError();
}

// The orderings of a start's successive sweeps (bit 0: x backwards, bit 1: y, bit 2: z), sixteen nibbles, repeated.
// Every sequence visits all eight; what differs is how much work the solve is and what the step from one sweep to the
// next costs.  A sweep follows the one before across the grid as closely as its columns' neighbours have sealed that
// one: when only z flips (or nothing) it starts at the same corner, two levels behind; when x or y flips it starts at a
// corner the sweep before passes HALF-WAY through - it lags (NI + NJ) / 2 levels, and in the sparse sweeps of a solve,
// which are nothing but latency from level to level, that lag is the sweep's whole cost; x and y together: a whole
// traversal.  The order along z is the column's own business.  What matters most, though, is the work: a start's
// first sweeps should run into its largest octants (profiles/r05_col_sweep_order.txt: six-FS 1024x1024x512 x 14, from
// 4.37 to 2.86 grid sweeps of due tiles per start and from 116 to 88 ms with the x-fastest cyclic code begun at the
// corner nearest to each start; the z-fastest codes from the same corner: 2.95 - 3.15, 86 - 92 ms).
//   `which` is read as three decimal digits.  The last one: the table - the sequence as seen from the corner (0, 0, 0);
//   the tens: the corner each start's first sweep begins at - 0: (0, 0, 0) for every start, 1: the corner NEAREST
//     to the start (its first sweeps run into the largest octants), 2: the farthest;
//   the hundreds: 0 - x, y, z as the table says; 1 - per start, the lateral axis along which the start lies nearer to
//     the middle of the grid plays the table's x, 2 - the other one; 3 - the three axes by how near to the middle the
//     start lies along them play x, y, z (the nearest: x), 4 - the other way round.
bool column_order_valid(int which) { return which >= 0 && which % 10 < COL_ORDER_SEQUENCES && which / 10 % 10 < 3 && which / 100 < 5; }

void column_order_sequence(int which, const int (&n)[3], const int (&at)[3], unsigned long long *seq)
{
    static const unsigned char table[COL_ORDER_SEQUENCES][16] = {
        // 0: the reflected Gray code of rounds 3 - 5, x fastest (x flips 8 times in 16 sweeps, y 4, z 2, 2 repeats)
        {0, 1, 3, 2, 6, 7, 5, 4, 4, 5, 7, 6, 2, 3, 1, 0},
        // 1: the cyclic Gray code, x fastest (x 4 of 8, y 2, z 2)
        {0, 1, 3, 2, 6, 7, 5, 4, 0, 1, 3, 2, 6, 7, 5, 4},
        // 2: the cyclic Gray code, z fastest, then x
        {0, 4, 5, 1, 3, 7, 6, 2, 0, 4, 5, 1, 3, 7, 6, 2},
        // 3: reflected, z fastest
        {0, 4, 5, 1, 3, 7, 6, 2, 2, 6, 7, 3, 1, 5, 4, 0},
        // 4: cyclic, z fastest, then y
        {0, 4, 6, 2, 3, 7, 5, 1, 0, 4, 6, 2, 3, 7, 5, 1},
        // 5: z flips with EVERY sweep, x or y with every second one (that costs what the x or y flip costs alone)
        {0, 4, 1, 5, 3, 7, 2, 6, 0, 4, 1, 5, 3, 7, 2, 6},
        // 6: as 5, pairs in the other order
        {0, 4, 5, 1, 7, 3, 2, 6, 0, 4, 5, 1, 7, 3, 2, 6},
        // 7: as 5, the quadrants of the second eight sweeps in the opposite order
        {0, 4, 1, 5, 3, 7, 2, 6, 2, 6, 3, 7, 1, 5, 0, 4},
        // 8: the first eight sweeps as 1, the next eight with z fastest (the late sweeps of a solve are chains of
        // single tiles across the grid: a sweep that flips only z rides on the one before)
        {0, 1, 3, 2, 6, 7, 5, 4, 0, 4, 6, 2, 3, 7, 5, 1},
        // 9: two quadrants down, the same two up
        {0, 1, 5, 4, 2, 3, 7, 6, 0, 1, 5, 4, 2, 3, 7, 6},
    };
    if (!column_order_valid(which)) which = 0;
    const int base = which % 10, first = which / 10 % 10, axes = which / 100;
    unsigned x = 0;
    for (int d = 0; d < 3 && first; d++)
        if ((2 * at[d] >= n[d]) == (first == 1)) x |= 1u << d;
    // how far from the nearer face, as a share of the axis: the larger, the more evenly the start splits the axis
    const double cx = (double)std::min(at[0], n[0] - 1 - at[0]) / n[0], cy = (double)std::min(at[1], n[1] - 1 - at[1]) / n[1];
    const double cz = (double)std::min(at[2], n[2] - 1 - at[2]) / n[2];
    int role[3] = {0, 1, 2};        // the axis that plays the table's x, y, z
    if (axes == 1 ? cy > cx : axes == 2 ? cx > cy : false) std::swap(role[0], role[1]);
    if (axes == 3 || axes == 4) {
        const double c[3] = {cx, cy, cz};
        std::stable_sort(role, role + 3, [&](int a, int b) { return axes == 3 ? c[a] > c[b] : c[a] < c[b]; });
    }
    unsigned long long q = 0;
    for (int e = 0; e < 16; e++) {
        unsigned o = 0;
        for (int d = 0; d < 3; d++)
            if (table[base][e] >> d & 1) o |= 1u << role[d];
        q |= (unsigned long long)(o ^ x) << (4 * e);
    }
    *seq = q;
}

hipError_t launch_column_solve(const ColumnSolve &P, int nblocks, hipStream_t st)
{
    if (P.nstart < 1 || P.NK < 1 || P.NK > COL_MAX_NK || P.nseq < 1 || P.nseq > COL_SEQS || nblocks < P.nseq
        || P.L.lo[0] != 1 || P.L.lo[1] != 1 || P.L.lo[2] < CS || P.L.lo[2] % CS || P.L.s1 % CS
        || P.L.p[2] - P.L.lo[2] - P.NK * TILE_Z < CS || !P.tptr || P.ts1 % CS || P.ts0 % CS || P.tlo % CS
        || (P.tpad != 0 && P.tpad != 1) || (P.tpad == 0 && P.L.n[2] % TILE_Z))
        return hipErrorInvalidValue;
    const hipError_t e = column_raise_lds();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(column_solve_kernel, dim3((unsigned)nblocks), dim3(64 * CWG), CWG * CLDSB, st, P);
    return hipGetLastError();
}

} // namespace ttsweep
