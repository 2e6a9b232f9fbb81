// ttsweep_dev.h - structures shared by the host API and the HIP kernels.
//
// Device data layout (DESIGN.md "Data layout in HBM").  The library never
// computes on the caller's FLOATBOX arrays directly: it keeps padded copies in
// which every cell has a full halo of R = max|offset| cells on every side, so
// the relaxation loops need no bounds tests:
//   * travel-time halo cells hold +INFINITY  -> a candidate through them is
//     +INFINITY and never wins the min;
//   * velocity halo cells hold 0             -> delay stays finite, no NaN.
// "Device axes" (a, b, c) are a permutation of the user's (x, y, z); c is the
// stride-1 axis of the padded arrays.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ttsweep {

struct DevLayout {
    int n[3];           // extents along device axes a, b, c (interior)
    int lo[3];          // halo cells in front of the interior, per axis
    int p[3];           // padded extents
    long long s0, s1;   // strides (floats) of axes a and b; axis c has stride 1
    long long cells;    // padded volume = p0*p1*p2
    int perm[3];        // device axis d is user axis perm[d] (0=x, 1=y, 2=z)
    int un[3];          // user extents nx, ny, nz
};

__host__ __device__ inline long long dev_index(const DevLayout &L, int a, int b, int c)
{
    return (long long)(a + L.lo[0]) * L.s0 + (long long)(b + L.lo[1]) * L.s1 + (c + L.lo[2]);
}

// flags of a pull entry (same meaning as ttsweep_pull_entry.flags)
enum : int {
    PULL_FWD = 1,   // edge exists as (centre = this cell, offset +e): dead if this cell is the start
    PULL_REV = 2,   // edge exists as (centre = neighbour, offset -e): dead if the neighbour is the start
};

// One pull entry for the per-cell kernel: neighbour = cell + delta (floats).
struct CellEntry {
    int delta;
    float h;
    int flags;
    int pad_;
};

// One forward star entry in device axes, for the validator kernel.
struct FwdEntry {
    int da, db, dc;
    float h;
};

// Per-start device record.
struct StartDesc {
    float *T;               // padded travel-time volume of this start
    long long sidx;         // padded linear index of the start cell
    int sa, sb, sc;         // start cell, device-axis interior coordinates
    int pad_;
    // Bounding box (inclusive, device axes, clipped to the grid) of the cells
    // that have at least one dead edge: the STRIP kernel does not store into
    // them, the exact wave-per-cell kernel owns them.  Empty if lo > hi.
    int box_lo[3], box_hi[3];
    // STRIP activity tracking (see plan_pass_kernel, push_improved): two blocks of patch flags
    // (plane, lane tile, strip) the initialisation uses - the second one says which patches
    // are sources -, then one word per unit of staged-plane bits it has to relax (pend[]: pushed
    // by whoever improves a plane, kept while the gate holds the unit back), then the number of
    // source patches.
    int *tile_flags;
    unsigned long long *work;   // cells actually relaxed for this start (sum over passes)
};

// ---------------------------------------------------------------------------
// STRIP kernel geometry (compile-time)
// ---------------------------------------------------------------------------
constexpr int STRIP_K = 16;                         // cells per thread along c
constexpr int STRIP_NS = 4;                         // waves per workgroup (throughput instances: two workgroups per CU)
constexpr int STRIP_NS_LAT = 8;                     // ... of the latency instance (one workgroup per CU relaxes one unit with all
                                                    // of the CU's eight waves: small shards, ttsweep_driver.cpp)
constexpr int STRIP_NS_MAX = 8;
constexpr int STRIP_G = 1;                          // staged planes per barrier group of the throughput instances (2 G slabs in LDS)
#ifndef TTSWEEP_G_LAT
#define TTSWEEP_G_LAT 3
#endif
constexpr int STRIP_G_LAT = TTSWEEP_G_LAT;          // ... of the latency instance
constexpr int STRIP_TB = 64;                        // lanes along b
constexpr int STRIP_CF = 8;                         // halo in front of a strip window (>= max|dc|, multiple of 4)
constexpr int STRIP_W = STRIP_K + 2 * STRIP_CF;     // neighbour window per (cell strip, column)
constexpr int STRIP_MAX_RA = 7;                     // plane offsets handled: |da| <= 7

// A unit of the STRIP kernel owns np = 1 or 2 neighbouring planes (np A .. np A + np - 1) of one
// lane tile and strip.  With two, every staged neighbour plane q is used twice, with plane
// offset q - 2A for the first own plane and q - 2A - 1 for the second: slab traffic, barriers,
// window loads and per-unit overheads are shared by two planes of output (the throughput
// mode, many starts).  With one, units are half as long and twice as many (the latency mode:
// few starts cannot fill the machine with long units, and a pass is never shorter than its
// longest unit).  An item is one slab row offset db of a staged plane together with the
// offsets (differing only in dc) that either own plane relaxes against it: one register
// window is loaded per item and serves both planes.
constexpr int STRIP_PLANES = 2;                         // own planes per unit, at most
constexpr int STRIP_STAGED = 2 * STRIP_MAX_RA + STRIP_PLANES;   // staged planes per unit, at most

struct StripItem {
    int rowoff;             // db: row offset inside the staged slab
    unsigned mask[STRIP_PLANES];    // bit t (1..15) set: own plane j relaxes offset dc = t - 8
    int pad_;
    float h[STRIP_PLANES][16];      // h[j][t] = d/2 of offset (q - 2A - j, db, t - 8)
};

struct StripPlan {
    int ra, rb;                         // max |da|, max |db| over the star
    int np;                             // own planes per unit (1 or 2)
    int nstaged;                        // 2 ra + np: staged plane p is plane np A - ra + p
    int first[STRIP_STAGED + 1];        // items of staged plane p are [first[p], first[p+1])
    int nent[STRIP_STAGED][STRIP_PLANES];   // offsets own plane j relaxes against staged plane p
    // the items of a staged plane are laid out so that wave w of the workgroup relaxes
    // [first[p] + wsplit[p][w], first[p] + wsplit[p][w+1]) - ns shares of nearly equal cost
    unsigned char wsplit[STRIP_STAGED][STRIP_NS_MAX + 1];
    int ns;                             // waves per workgroup the shares are made for
};

// Unit grid of one start: plane groups (np planes each) along a x lane tiles along b x strips
// along c.
__host__ __device__ inline int strip_agroups(const DevLayout &L, int np) { return (L.n[0] + np - 1) / np; }
__host__ __device__ inline int strip_btiles(const DevLayout &L) { return (L.n[1] + STRIP_TB - 1) / STRIP_TB; }
__host__ __device__ inline int strip_cstrips(const DevLayout &L) { return (L.n[2] + STRIP_K - 1) / STRIP_K; }
__host__ __device__ inline int strip_units(const DevLayout &L, int np) { return strip_agroups(L, np) * strip_btiles(L) * strip_cstrips(L); }

// "changed" word of a start and pass: bit 0 = a travel time improved, bit 1 = units are held
// back by the distance gate (the start is not converged, but nothing has improved for it).
enum : int { CHANGED_IMPROVED = 1, CHANGED_PENDING = 2 };

// Unit queues of one pass (sweep_units_kernel): one list per XCD of the device (at most
// UNITQ_LISTS; the count is measured at run time), filled by plan_pass_kernel.
// ctrl[0..7] = entries in list x, ctrl[8..15] = next entry to hand out.
constexpr int UNITQ_LISTS = 8;
constexpr int UNITQ_CTRL_WORDS = 2 * UNITQ_LISTS;       // followed by one "workgroups done" word

// What sweep_units_kernel does besides draining the queues: the dead-edge cells of the
// active starts at its beginning, the hand-over of the pass at its end.
struct UnitPassTail {
    const int *active;          // indices of the active starts
    int nactive;
    const CellEntry *entries;   // whole pull star, for the dead-edge cells
    int nentries;
    int max_box_cells;          // cells of the largest dead-edge box
    int nstart;                 // "changed" words per pass
    int *changed_host;          // pinned host copy of this pass's words (written at the end)
    int *changed_next;          // the next pass's device words (cleared at the end)
    float defer_margin;         // DeferRule::margin of the solve (ttsweep_kernels.hip, push_improved)
};

// ---------------------------------------------------------------------------
// STRIP, one launch per solve (sweep_units_kernel<.., ASYNC = true>, DESIGN.md 4.1 "No passes")
// ---------------------------------------------------------------------------
// The pass structure is gone: the grid is launched ONCE per solve.  The starts are dealt over
// `nrings` rings (at most one per XCD); the first nrings workgroups are PLANNERS, one per ring:
// a planner scans its ring's static unit list (every unit of its starts, nearest to the start
// first) for units whose pend word holds staged-plane bits, marks them busy and publishes them
// into the ring, nearest first, but only while the ring holds fewer than `high` entries - so
// far units are handed out only when near work has run dry (a priority queue by distance, made
// of a scan order and a bounded FIFO).  All other workgroups are WORKERS: they claim ring
// entries (own XCD's ring first), relax the unit exactly as a pass does, RELEASE their stores
// (agent scope), push the plane bits to the neighbours' pend words, clear the unit's busy bit
// and count the unit as completed.  A ring is finished when a whole scan found no bit, nothing
// was queued or running when the scan began, and the dead-edge cells of its starts have been
// relaxed since the last unit: its planner sets the done bit; a worker leaves when every ring
// is done.  Every word that workgroups exchange is accessed with agent-scope atomics only; the
// travel times themselves cross workgroups behind a release (storing side) / acquire (after a
// ring entry has been read) pair of agent-scope fences.
constexpr int ASYNC_MAX_RINGS = 8;
constexpr int ASYNC_CTL_STRIDE = 32;                    // 64-bit words per ring: [0] head | tail << 32 | done << 63,
                                                        // [8] (low half) squared gate radius of the ring's planner (float
                                                        // bits; read by workers that hand units on), [16] (low half) units
                                                        // completed
constexpr int ASYNC_RING_STARTS = 32;                   // starts per ring, at most
constexpr unsigned ASYNC_BUSY = 0x80000000u;            // pend word: the unit is queued or being relaxed
constexpr unsigned ASYNC_UNIT_SPECIAL = 0xfffffu;       // ring entry: relax the start's dead-edge cells
constexpr unsigned long long ASYNC_EXIT = ~0ull;
constexpr int ASYNC_MAX_STARTS = 255;                   // (8 bits of a ring entry; 255 keeps ASYNC_EXIT apart)
// ring entry: planes (16) | unit (20) << 16 | start (8) << 36 | position in the ring mod 2^18 << 44 | valid << 62 |
// "tell every unit at once" << 63 (set from the ring's first flush of deferred bits on).  A slot is EMPTY (0) before
// its first entry and again as soon as the worker that claimed its position has READ the entry (the worker stores 0
// back); a producer writes a slot only when it is empty.  So a slot is never reused before its entry has been read,
// however long a worker is held up between its claim and its look at the slot (round 4 bounded the reuse by a count
// of completed entries, which a single pre-empted worker could slip through: the solve then ended at its wall-clock
// limit).
constexpr unsigned long long ASYNC_ENTRY_VALID = 1ull << 62;
constexpr unsigned ASYNC_TAG_MASK = 0x3ffffu;
enum : int { ASYNC_OK = 0, ASYNC_ERR_TIMEOUT = 1, ASYNC_ERR_CAP = 2 };
// AsyncSolve::handoff bits: a worker that has improved a plane does not only set the plane bit in the pend words of
// the units that stage it - it also takes the ones that are idle and inside the gate (compare-and-swap to BUSY) and
// publishes them into the ring itself, instead of leaving them to the planner's next scan (1); the same for its own
// unit when bits have arrived while it was relaxed (2).  The planner stays for everything else: the first units,
// units behind the gate, rest detection.
enum : int { ASYNC_HANDOFF_NEIGHBOURS = 1, ASYNC_HANDOFF_SELF = 2 };

struct AsyncSolve {
    int nrings;
    int cap_mask;                   // ring capacity - 1 (a power of two)
    int low, high;                  // the planner refills a ring that holds <= low entries up to high
    int special_every;              // units of a start between two relaxations of its dead-edge cells
    const int4 *list;               // the rings' unit lists, one after the other: (start | index in ring << 16, unit,
                                    // squared distance from the start to the unit in cells (float bits), 0)
    int policy;                     // 0: every refill scans from the nearest unit on (strict priority by distance);
                                    // 1: the scan goes round and round the list (a unit is handed out at most once
                                    //    per round) behind a distance gate that opens by gate_speed cells per round
                                    // 2: as 1, but the gate of a start lies `window` cells beyond the nearest of its
                                    //    units that the previous round found with anything to do
    float gate_r0, gate_speed;      // policy 1; speed <= 0: no gate
    float gate_fast;                // policy 1: cells per round while the ring is empty at the beginning of a round
    float window;                   // policy 2; <= 0: no gate
    int scan_slack;                 // list entries in front of the first unit with anything to do that a round still scans
    int inunit;                     // times a unit that improved is relaxed again against its own planes before it is handed back
    int handoff;                    // ASYNC_HANDOFF_* (0: only the planners publish)
    int ring_off[ASYNC_MAX_RINGS], ring_len[ASYNC_MAX_RINGS];
    int ring_start_off[ASYNC_MAX_RINGS + 1];    // ring r serves starts ring_starts[ring_start_off[r] .. [r + 1])
    const int *ring_starts;
    unsigned long long *entries;    // nrings x (cap_mask + 1), zero (every slot empty) before the launch
    unsigned long long *ctl;        // nrings x ASYNC_CTL_STRIDE, zero before the launch
    unsigned *status;               // [0]: ASYNC_OK or the first error
    long long timeout_ticks;        // wall-clock ticks (100 MHz) after which every wait gives up
    long long max_entries;          // entries a ring may publish before the solve counts as not converging
};

// ---------------------------------------------------------------------------
// TILE kernel (ordered tile sweeps for small stars, ttsweep_tile.hip)
// ---------------------------------------------------------------------------
#ifndef TTSWEEP_TILE_Z
#define TTSWEEP_TILE_Z 32
#endif
constexpr int TILE_X = 8, TILE_Y = 8, TILE_Z = TTSWEEP_TILE_Z;      // cells of a tile along the device axes a, b, c
constexpr int TILE_ZF = 4;                              // cells staged in front of / behind a tile row (>= max |dc|)
constexpr int TILE_PITCH = TILE_Z + 2 * TILE_ZF;        // floats per staged row
constexpr int TILE_QPR = TILE_PITCH / 4;                // float4 per staged row
constexpr int TILE_MAX_R = 2;                           // max |da|, |db| of the star
constexpr int TILE_MAX_ENT = 26;                        // pull entries (the 26-neighbour shell)

__host__ __device__ inline int tile_count(int n, int t) { return (n + t - 1) / t; }

// z faces (TILE layout): tile rows are whole 128-byte lines of the volumes (the z halo in front
// of the interior is a whole tile wide), so the FZ = max |dc| cells a tile needs beyond either
// end of its rows would cost two more lines per row.  They come from compact copies instead:
// for every tile boundary kb = 0 .. NK along c, side 0 holds the FZ cells below it
// (c = TILE_Z kb - FZ + layer), side 1 the FZ cells above it (c = TILE_Z kb + layer), for every
// padded (a, b) - a few lines per tile instead of 2 x 100.  The velocity faces are written
// once, the travel-time faces of a start when its box is initialised and by every tile that
// improves (its own first and last FZ layers).
__host__ __device__ inline long long tile_face_index(const DevLayout &L, int fz, int kb, int side, int layer,
                                                     int pa, int pb)
{
    return ((((long long)kb * 2 + side) * fz + layer) * L.p[0] + pa) * L.p[1] + pb;
}
__host__ __device__ inline long long tile_face_cells(const DevLayout &L, int fz)
{
    return (long long)(tile_count(L.n[2], TILE_Z) + 1) * 2 * fz * L.p[0] * L.p[1];
}

struct TileEntry {
    int da, db, dc;     // offset in device axes
    float h;            // d / 2
    int flags;          // PULL_FWD | PULL_REV
};

// Arguments of one launch of tile_sweep_kernel: the tiles of hyperplane D of a sweep with
// ordering (sx, sy, sz), for every active start.  The grid is `nblocks` single-wavefront
// workgroups (a multiple of nxcd); the candidates ((J', K'), active start) are dealt over them
// position by position (ttsweep_tile.hip: tile_candidate).
struct TileSweep {
    DevLayout L;
    const float *v;
    const StartDesc *starts;
    const int *active;      // indices of the active starts
    int *changed;           // "changed" words of this sweep, per start
    int nactive;
    int nblocks;            // workgroups of the launch
    int nxcd;               // XCDs of the device (workgroup b is taken to run on XCD b % nxcd)
    int wstride;            // workgroups per XCD that take candidates: <= nblocks / nxcd, coprime to nactive
    int nstart;             // starts of the solve (slots per workgroup in wgwork)
    unsigned long long *wgwork;     // [nblocks][nstart][2]: private work sums of the workgroups (relaxations, tiles)
    int NI, NJ, NK;         // tiles along a, b, c
    int R;                  // max |da|, |db| (halo of the staged image along a and b)
    int sx, sy, sz;         // sweep ordering, +1 / -1 per axis
    int nsx, nsy, nsz;      // ordering of the NEXT sweep
    int *dmin_next;         // first hyperplane of the next sweep that can hold a due tile (atomicMin; the host
                            // sets it to a large value before a sweep and starts the next sweep there)
    int D;                  // hyperplane: tiles with I' + J' + K' == D (coordinates in sweep direction)
    int epoch;              // launch number within the solve (>= 2)
    int nent;               // pull entries in use; ent[nent..] are no-ops (h = 0 onto the cell itself)
    float *T0;              // padded travel-time volume of start 0; start s: + s * L.cells
    int *state0;            // activity words of start 0 (StartDesc::tile_flags); start s: + s * state_stride ints
    long long state_stride; //   (the planner computes a start's addresses instead of loading its descriptor)
    unsigned long long *work0;  // work counters of start 0; start s: + 3 s
    int fz;                 // max |dc|: layers per z face
    const float *vface;     // velocity faces
    float *tface;           // travel-time faces of start 0; start s: + s * face_cells
    long long face_cells;
    TileEntry ent[TILE_MAX_ENT];
};

// ---------------------------------------------------------------------------
// TILE, plain 6-neighbour star, one launch per solve: column pipelines (ttsweep_column.hip)
// ---------------------------------------------------------------------------
// A COLUMN is the stack of NK tiles above one tile position (I, J).  One wavefront owns a column for one
// ordering sweep and relaxes its due tiles as ONE systolic pipeline along z (lane (i', j') is i' + j' cells
// behind the first lane; the pipeline does not drain between consecutive due tiles), through a ring of
// three 16-cell chunks per image row in LDS.  Columns are claimed in the order of the sweep (by tile level
// I' + J', per XCD sequence) and wait for each other through one 64-bit progress word per column and
// start: sweep (23 bits) | tiles finished, 0xff = sealed (8) | a tile improved in this sweep here or upwind (1) |
// tiles improved in this sweep (32, in sweep order).  Which tiles are due in the NEXT sweep the columns read from
// the words their neighbours sealed this one with.
constexpr int COL_MAX_NK = 32;              // tiles of a column (bits of a mask word)
constexpr int COL_MAX_SWEEPS = 4096;        // per-start, per-sweep seal counters
constexpr int COL_SEQS = 8;                 // claim sequences (one per XCD)
constexpr int COL_WAVES = 4;                // wavefronts (columns in flight) per workgroup = per CU
enum : unsigned { COL_RUNNING = 0, COL_DONE = 1, COL_ERR_TIMEOUT = 2, COL_ERR_CAP = 3, COL_ERR_LDS_BASE = 4 };

struct ColumnSolve {
    DevLayout L;
    const float *v;                 // padded velocity volume (layout L)
    // The travel times are relaxed where they lie: in the padded volumes of the library (layout L: tpad = 1,
    // tlo = L.lo[2], strides L.s0 / L.s1) or - when the caller's rows are whole tiles long (nz % 32 == 0) - in the
    // caller's own FLOATBOX arrays (tpad = 0, tlo = 0, strides ny nz / nz): no copy in, no copy out; rows and
    // chunks that lie outside the grid are then not staged but set to +INFINITY in the image.
    float *const *tptr;             // [nstart] travel-time volume of every start
    long long ts0, ts1;             // its strides (floats) along x and y
    int tpad;                       // rows / columns in front of the grid
    int tlo;                        // floats in front of z = 0 in a row
    int nstart;
    int NI, NJ, NK;
    int nseq;                       // claim sequences in use (<= COL_SEQS)
    int seq_off[COL_SEQS], seq_len[COL_SEQS];
    const int *seqtab;              // sequence x: positions I' | J' << 16 in sweep order, from seq_off[x] on
    unsigned long long *prog;       // [2][nstart][NI * NJ] progress words, a buffer per sweep parity
    unsigned *due;                  // [nstart][NI * NJ] the tiles due in the FIRST sweep (column_init); later sweeps read
                                    // them from the sealed progress words of the sweep before
    int *done;                      // [nstart]: the sweep after which the start was at rest (0: running)
    unsigned long long *claim;      // [COL_SEQS][16]: next entry of each sequence (128 bytes apart)
    unsigned *status;               // [0]: COL_RUNNING / COL_DONE / error; [1]: starts still running
    unsigned long long *wgwork;     // [waves][nstart][2]: private work sums (relaxations, tiles)
    int *changed;                   // per start: CHANGED_IMPROVED when anything improved
    float h[6];                     // d / 2 of the entries x-, y-, z-, z+, y+, x+
    int max_sweeps;
    const unsigned long long *ordseq;   // [nstart] the orderings of the start's sweeps 1, 2, ...: a nibble each (bit 0: x
                                    // backwards, 1: y, 2: z), repeated with period 16 - column_order_sequence()
    long long timeout_ticks;        // wall-clock ticks (100 MHz) after which every wait gives up
};

} // namespace ttsweep
