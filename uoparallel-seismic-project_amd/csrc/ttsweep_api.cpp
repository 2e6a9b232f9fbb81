// ttsweep_api.cpp - host side of libttsweep.so: the C ABI of include/ttsweep.h.
//
// Replaces the driver loop of serial_new/sweep-tt-multistart.c:151-170 and its
// callee sweepXYZ (:198-256) by device-resident relaxation to convergence.
// There is no CPU fallback in this file: every solve runs HIP kernels or fails.
#include "../../include/ttsweep.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "pullstar.h"
#include "ttsweep_dev.h"
#include "ttsweep_kernels.h"

using namespace ttsweep;

// ---------------------------------------------------------------------------
// error reporting
// ---------------------------------------------------------------------------
static thread_local std::string g_last_error;

static int set_error(const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return -1;
}

#define HIPCHK(expr)                                                                   \
    do {                                                                               \
        hipError_t e_ = (expr);                                                        \
        if (e_ != hipSuccess)                                                          \
            return set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),    \
                             __FILE__, __LINE__);                                      \
    } while (0)

// ---------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------
// Passes are enqueued one ahead of the convergence test (see ttsweep_solve_device), so the
// per-start "changed" words exist once per pass in flight.
constexpr int PASS_SLOTS = 3;

struct ttsweep_ctx {
    int device = 0;
    int nx = 0, ny = 0, nz = 0;
    hipStream_t stream = nullptr;

    std::vector<ttsweep_pull_entry> pull;   // user-axis pull star
    int radius = 0;
    long long relax_per_sweep = 0;

    DevLayout L{};
    int kernel = TTSWEEP_KERNEL_CELL;

    float *d_v = nullptr;                   // padded velocity
    bool have_v = false;
    CellEntry *d_cell_entries = nullptr;
    int n_cell_entries = 0;
    FwdEntry *d_fwd_entries = nullptr;      // forward star entries (validator)
    int n_fwd_entries = 0;

    // STRIP kernel: (da, db) columns of the star, dead-edge boxes
    // STRIP kernel: the star's items for units of one plane (latency mode, few starts) and of
    // two planes (throughput mode); `np` is the mode of the solve in progress
    StripItem *d_strip_items[STRIP_PLANES] = {nullptr, nullptr};
    StripPlan plans[STRIP_PLANES]{};
    int np = STRIP_PLANES;
    int pair_min_starts = -1;               // two-plane units from this many starts on; -1: by the supply of units
    long long pair_min_units = 80000;       //   (starts x one-plane units of a start; measured crossover)
    std::vector<std::array<int, 3>> special_offsets;   // device-axis offsets e: cell start - e owns a dead edge
    bool start_is_special = false;
    int max_box_cells = 0;                  // of the current solve
    // STRIP: static work list, entry -> (start, unit), XCD-interleaved (build_worklist)
    int2 *d_worklist = nullptr;
    size_t worklist_cap = 0;
    long long worklist_len = 0;
    // unit queues of a sparse pass (plan_pass_kernel -> sweep_units_kernel)
    int4 *d_unitq = nullptr;                // UNITQ_LISTS lists of unitq_cap entries
    size_t unitq_cap = 0;
    int *d_unitq_ctrl = nullptr;            // UNITQ_CTRL_WORDS (counts, cursors)
    int nlists = UNITQ_LISTS;               // unit queues = XCDs of the device (census at create)
    int unitq_blocks = 0;                   // persistent grid: workgroups the device holds at once
    std::vector<std::vector<int>> unit_order;       // per start: unit ids, nearest to the start first
    std::vector<long long> unit_order_key;          // start cell the cached order belongs to
    // Distance gate (see plan_pass_kernel): radius of the first pass and cells it opens per
    // pass.  Defaults follow the star's reach: final values spread at about half the reach
    // per pass (measured, 818-offset star: 3.5 cells/pass gives the shortest solve).
    double gate_speed = 0.0;                // 0: no gate
    double gate_r0 = 0.0;
    // TILE kernel: the star in device axes, halo of the staged image, launch counter
    TileEntry tile_ent[TILE_MAX_ENT];
    int tile_nent = 0, tile_R = 1, tile_fz = 1;     // entries, max |da|,|db|, max |dc| of the star
    float *d_vface = nullptr, *d_tface = nullptr;   // z faces of v and of every start's T (TILE layout)
    int tile_epoch = 1;
    int2 *d_tile_list = nullptr;            // due tiles of the launch in flight
    size_t tile_list_cap = 0;
    int *d_tile_ctrl = nullptr;             // number of due tiles, one word per launch of a sweep
    size_t tile_ctrl_cap = 0;
    int tile_blocks = 0;                    // persistent grid of the sweep kernel
    int *d_tile_flags = nullptr;            // capacity_starts x activity words (flag_words)
    unsigned long long *d_work = nullptr;   // capacity_starts
    unsigned long long *h_work = nullptr;   // pinned
    int pass_index = 0;

    // per-solve pools (grown on demand, reused between solves)
    float *d_T = nullptr;                   // capacity_starts padded volumes
    int capacity_starts = 0;
    StartDesc *d_starts = nullptr;
    int *d_active = nullptr;
    int *d_changed = nullptr;
    StartDesc *h_starts = nullptr;          // pinned
    int *h_active = nullptr;                // pinned
    int *h_changed = nullptr;               // pinned

    // options
    bool timing = false;
    long long max_sweeps = 100000;
    int max_batch = 0;                      // cap on starts per ttsweep_solve batch (0: by memory)

    hipEvent_t ev_solve0 = nullptr, ev_solve1 = nullptr;
    hipEvent_t ev_flags[PASS_SLOTS] = {nullptr, nullptr, nullptr};     // "changed" words of a pass are on the host
    std::vector<hipEvent_t> ev_pool;        // pairs around sweep launches
    size_t ev_used = 0;

    ttsweep_stats stats{};
};

static int ctx_bind(const ttsweep_ctx *ctx)
{
    HIPCHK(hipSetDevice(ctx->device));
    return 0;
}

// Counts the XCDs of the device (each has its own L2) by asking many workgroups where they
// run: the STRIP kernel keeps one unit queue per XCD.  Falls back to 1 queue on any doubt
// (queues are a locality device, never a correctness one).
static int count_xcds(ttsweep_ctx *ctx)
{
    unsigned *d_seen = nullptr, h_seen = 0;
    HIPCHK(hipMalloc((void **)&d_seen, sizeof(unsigned)));
    hipError_t e = hipMemsetAsync(d_seen, 0, sizeof(unsigned), ctx->stream);
    if (e == hipSuccess) e = launch_xcc_census(d_seen, 4096, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(&h_seen, d_seen, sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_seen);
    if (e != hipSuccess) return set_error("XCD census failed: %s", hipGetErrorString(e));
    const int n = __builtin_popcount(h_seen);
    ctx->nlists = std::min(std::max(n, 1), (int)UNITQ_LISTS);
    return 0;
}

// Padded layout for the CELL kernel: identity axis order, halo R on all sides.
static void make_layout_cell(ttsweep_ctx *ctx)
{
    DevLayout &L = ctx->L;
    const int R = std::max(ctx->radius, 1);
    const int n[3] = {ctx->nx, ctx->ny, ctx->nz};
    for (int d = 0; d < 3; d++) {
        L.perm[d] = d;
        L.n[d] = n[d];
        L.un[d] = n[d];
        L.lo[d] = R;
        L.p[d] = n[d] + 2 * R;
    }
    L.s1 = L.p[2];
    L.s0 = (long long)L.p[1] * L.p[2];
    L.cells = L.s0 * L.p[0];
}

// Padded layout for the STRIP kernel: picks which user axis becomes the plane axis
// a, the lane axis b and the strip (stride-1) axis c.
static void make_layout_strip(ttsweep_ctx *ctx)
{
    DevLayout &L = ctx->L;
    const int n[3] = {ctx->nx, ctx->ny, ctx->nz};
    // Lane axis b: an axis that fits into one wave (extent <= 64, the largest such)
    // makes the activity units thin in that direction; otherwise the axis that
    // fills 64-lane tiles best.  Strip axis c: of the two remaining axes the one
    // that fills K-cell strips best.  The last axis is the plane axis a (untiled).
    auto util = [](int m, int q) { return (double)m / (double)(((m + q - 1) / q) * q); };
    int bax = -1;
    for (int d = 0; d < 3; d++)
        if (n[d] <= STRIP_TB && (bax < 0 || n[d] > n[bax])) bax = d;
    if (bax < 0) {
        bax = 0;
        for (int d = 1; d < 3; d++)
            if (util(n[d], STRIP_TB) > util(n[bax], STRIP_TB) + 1e-12) bax = d;
    }
    int rest[2], k = 0;
    for (int d = 0; d < 3; d++)
        if (d != bax) rest[k++] = d;
    int cax = rest[1], aax = rest[0];       // ties: keep the user's fastest axis as c
    if (util(n[rest[0]], STRIP_K) > util(n[rest[1]], STRIP_K) + 1e-12) { cax = rest[0]; aax = rest[1]; }
    L.perm[0] = aax;
    L.perm[1] = bax;
    L.perm[2] = cax;
    int r[3] = {0, 0, 0};
    for (const auto &e : ctx->pull) {
        const int u[3] = {e.di, e.dj, e.dk};
        for (int d = 0; d < 3; d++) r[d] = std::max(r[d], std::abs(u[L.perm[d]]));
    }
    for (int d = 0; d < 3; d++) {
        L.n[d] = n[L.perm[d]];
        L.un[d] = n[d];
    }
    L.lo[0] = std::max(r[0], 1);
    L.p[0] = L.n[0] + 2 * L.lo[0];
    L.lo[1] = std::max(r[1], 1);
    L.p[1] = ((L.n[1] + STRIP_TB - 1) / STRIP_TB) * STRIP_TB + 2 * L.lo[1];
    L.lo[2] = STRIP_CF;
    L.p[2] = ((L.n[2] + STRIP_K - 1) / STRIP_K) * STRIP_K + 2 * STRIP_CF;
    L.s1 = L.p[2];
    L.s0 = (long long)L.p[1] * L.p[2];
    L.cells = L.s0 * L.p[0];
    for (StripPlan &plan : ctx->plans) {
        plan.ra = r[0];
        plan.rb = L.lo[1];
    }
}

// Padded layout for the TILE kernel: identity axis order (z stays the stride-1 axis),
// whole tiles, halo R along x and y, one tile of halo in front of and behind every row so
// that a tile's rows are whole 128-byte lines (the allocation is at least that aligned).
static void make_layout_tile(ttsweep_ctx *ctx)
{
    DevLayout &L = ctx->L;
    const int n[3] = {ctx->nx, ctx->ny, ctx->nz};
    const int t[3] = {TILE_X, TILE_Y, TILE_Z};
    int r[2] = {1, 1};
    for (const auto &e : ctx->pull) {
        r[0] = std::max(r[0], std::abs(e.di));
        r[1] = std::max(r[1], std::abs(e.dj));
    }
    ctx->tile_R = std::max(r[0], r[1]);
    ctx->tile_fz = 1;
    for (const auto &e : ctx->pull) ctx->tile_fz = std::max(ctx->tile_fz, std::abs(e.dk));
    for (int d = 0; d < 3; d++) {
        L.perm[d] = d;
        L.n[d] = n[d];
        L.un[d] = n[d];
        L.lo[d] = d < 2 ? ctx->tile_R : TILE_Z;         // (a whole tile in front: tile rows are whole lines)
        L.p[d] = tile_count(n[d], t[d]) * t[d] + 2 * L.lo[d];
    }
    L.s1 = L.p[2];
    L.s0 = (long long)L.p[1] * L.p[2];
    L.cells = L.s0 * L.p[0];
    ctx->tile_nent = (int)ctx->pull.size();
    for (int e = 0; e < TILE_MAX_ENT; e++) {
        TileEntry &te = ctx->tile_ent[e];
        if (e < ctx->tile_nent) {
            const ttsweep_pull_entry &p = ctx->pull[e];
            te = TileEntry{p.di, p.dj, p.dk, p.h, p.flags};
        } else {
            te = TileEntry{0, 0, 0, 0.0f, PULL_FWD | PULL_REV};     // no-op: candidate = the cell's own value
        }
    }
}

// Can the TILE kernel handle this star?  (the small stars of the HBM-bound regime: reach of
// at most 2 cells along x and y, 4 along z, at most 26 pull entries)
static bool tile_supported(const ttsweep_ctx *ctx)
{
    if (ctx->pull.empty() || (int)ctx->pull.size() > TILE_MAX_ENT) return false;
    for (const auto &e : ctx->pull)
        if (std::abs(e.di) > TILE_MAX_R || std::abs(e.dj) > TILE_MAX_R || std::abs(e.dk) > TILE_ZF) return false;
    return true;
}

static bool kernel_available(const ttsweep_ctx *ctx, int k);
static void make_layout(ttsweep_ctx *ctx);
static int auto_kernel(const ttsweep_ctx *ctx);

// Can the STRIP kernel handle this star?  (plane and strip offsets within +-7)
static bool strip_supported(const ttsweep_ctx *ctx)
{
    return !ctx->pull.empty() && ctx->radius <= STRIP_MAX_RA && ctx->radius < STRIP_CF;
}

static bool kernel_available(const ttsweep_ctx *ctx, int k)
{
    return k == TTSWEEP_KERNEL_CELL || (k == TTSWEEP_KERNEL_STRIP && strip_supported(ctx))
        || (k == TTSWEEP_KERNEL_TILE && tile_supported(ctx));
}

// Small stars: ordered tile sweeps; everything within +-7: LDS-staged unit relaxation;
// otherwise the per-cell kernel.
static int auto_kernel(const ttsweep_ctx *ctx)
{
    return tile_supported(ctx) ? TTSWEEP_KERNEL_TILE
         : strip_supported(ctx) ? TTSWEEP_KERNEL_STRIP : TTSWEEP_KERNEL_CELL;
}

static void make_layout(ttsweep_ctx *ctx)
{
    if (ctx->kernel == TTSWEEP_KERNEL_STRIP) make_layout_strip(ctx);
    else if (ctx->kernel == TTSWEEP_KERNEL_TILE) make_layout_tile(ctx);
    else make_layout_cell(ctx);
}

static int upload_strip_plan_np(ttsweep_ctx *ctx, int np);

static int upload_strip_plan(ttsweep_ctx *ctx)
{
    for (int np = 1; np <= STRIP_PLANES; np++)
        if (upload_strip_plan_np(ctx, np)) return -1;
    return 0;
}

static int upload_strip_plan_np(ttsweep_ctx *ctx, int np)
{
    const DevLayout &L = ctx->L;
    StripPlan &plan = ctx->plans[np - 1];
    plan.np = np;
    // (da, db) columns of the pull star: all offsets that differ only in dc
    struct Col { int db; unsigned mask; float h[16]; };
    std::vector<std::vector<Col>> per_da(2 * plan.ra + 1);
    ctx->special_offsets.clear();
    ctx->start_is_special = false;
    for (const auto &e : ctx->pull) {
        const int u[3] = {e.di, e.dj, e.dk};
        const int da = u[L.perm[0]], db = u[L.perm[1]], dc = u[L.perm[2]];
        if (e.flags == PULL_FWD) ctx->start_is_special = true;              // dead when the centre is the start
        if (e.flags == PULL_REV) ctx->special_offsets.push_back({da, db, dc});   // dead when the neighbour is the start
        std::vector<Col> &cols = per_da[da + plan.ra];
        const int t = dc + STRIP_CF;
        Col *col = nullptr;
        for (auto &c : cols)
            if (c.db == db && !(c.mask & (1u << t))) { col = &c; break; }
        if (!col) {     // (a second column for the same (da,db) only if an offset repeats with another length)
            cols.push_back(Col{});
            col = &cols.back();
            col->db = db;
        }
        col->mask |= 1u << t;
        col->h[t] = e.h;
    }
    // Items of staged plane p (plane np A - ra + p of a unit that owns planes np A ..): own
    // plane j relaxes it with plane offset da = p - ra - j.  Columns of two own planes with
    // the same row offset share an item (one window load serves both).
    plan.nstaged = 2 * plan.ra + np;
    std::vector<StripItem> flat;
    for (int p = 0; p < plan.nstaged; p++) {
        plan.first[p] = (int)flat.size();
        std::vector<StripItem> its;
        for (int j = 0; j < STRIP_PLANES; j++) {
            plan.nent[p][j] = 0;
            const int da = p - plan.ra - j;
            if (j >= np || da < -plan.ra || da > plan.ra) continue;
            for (const Col &c : per_da[da + plan.ra]) {
                plan.nent[p][j] += __builtin_popcount(c.mask);
                StripItem *it = nullptr;
                for (auto &x : its)
                    if (x.rowoff == c.db && x.mask[j] == 0) { it = &x; break; }
                if (!it) {
                    its.push_back(StripItem{});
                    it = &its.back();
                    it->rowoff = c.db;
                }
                it->mask[j] = c.mask;
                for (int t = 0; t < 16; t++) it->h[j][t] = c.h[t];
            }
        }
        // Four shares of nearly equal cost for the unit kernel's waves (longest processing
        // time first; an item costs its offsets plus a fixed part for the window load), each
        // share contiguous in the flat list.
        auto cost = [](const StripItem &x) { return __builtin_popcount(x.mask[0]) + __builtin_popcount(x.mask[1]) + 3; };
        std::stable_sort(its.begin(), its.end(), [&](const StripItem &x, const StripItem &y) { return cost(x) > cost(y); });
        std::vector<StripItem> share[STRIP_NS];
        int load[STRIP_NS] = {};
        for (const auto &x : its) {
            int w = 0;
            for (int k = 1; k < STRIP_NS; k++)
                if (load[k] < load[w]) w = k;
            share[w].push_back(x);
            load[w] += cost(x);
        }
        if (its.size() > 255) return set_error("star has too many columns per plane offset");
        plan.wsplit[p][0] = 0;
        for (int w = 0; w < STRIP_NS; w++) {
            flat.insert(flat.end(), share[w].begin(), share[w].end());
            plan.wsplit[p][w + 1] = (unsigned char)(plan.wsplit[p][w] + share[w].size());
        }
    }
    plan.first[plan.nstaged] = (int)flat.size();
    if (flat.size() > 0xffff) return set_error("star has too many columns");
    StripItem *&d_items = ctx->d_strip_items[np - 1];
    if (d_items) HIPCHK(hipFree(d_items));
    d_items = nullptr;
    if (!flat.empty()) {
        HIPCHK(hipMalloc((void **)&d_items, flat.size() * sizeof(StripItem)));
        HIPCHK(hipMemcpy(d_items, flat.data(), flat.size() * sizeof(StripItem), hipMemcpyHostToDevice));
    }
    return 0;
}

// Dead-edge box of one start (device axes, clipped, inclusive).
static void fill_special_box(const ttsweep_ctx *ctx, StartDesc &sd)
{
    const DevLayout &L = ctx->L;
    const int st[3] = {sd.sa, sd.sb, sd.sc};
    int lo[3] = {1, 1, 1}, hi[3] = {0, 0, 0};
    bool any = false;
    auto add = [&](const int p[3]) {
        for (int d = 0; d < 3; d++)
            if (p[d] < 0 || p[d] >= L.n[d]) return;
        for (int d = 0; d < 3; d++) {
            lo[d] = any ? std::min(lo[d], p[d]) : p[d];
            hi[d] = any ? std::max(hi[d], p[d]) : p[d];
        }
        any = true;
    };
    if (ctx->start_is_special) add(st);
    for (const auto &e : ctx->special_offsets) {
        const int p[3] = {st[0] - e[0], st[1] - e[1], st[2] - e[2]};
        add(p);
    }
    for (int d = 0; d < 3; d++) {
        sd.box_lo[d] = lo[d];
        sd.box_hi[d] = hi[d];
    }
}

static int upload_star(ttsweep_ctx *ctx)
{
    const DevLayout &L = ctx->L;
    std::vector<CellEntry> ce(ctx->pull.size());
    for (size_t e = 0; e < ctx->pull.size(); e++) {
        const ttsweep_pull_entry &p = ctx->pull[e];
        const int u[3] = {p.di, p.dj, p.dk};
        const long long delta = (long long)u[L.perm[0]] * L.s0 + (long long)u[L.perm[1]] * L.s1
                              + u[L.perm[2]];
        if (delta > 0x7fffffffLL || delta < -0x7fffffffLL)
            return set_error("grid too large for 32-bit neighbour offsets");
        ce[e].delta = (int)delta;
        ce[e].h = p.h;
        ce[e].flags = p.flags;
        ce[e].pad_ = 0;
    }
    // order by address so consecutive entries touch neighbouring cache lines
    std::sort(ce.begin(), ce.end(),
              [](const CellEntry &x, const CellEntry &y) { return x.delta < y.delta; });
    {   // forward entries in device axes for the validator: exactly the entries whose
        // edge is centred on the cell (PULL_FWD), i.e. the reference's (cell, l) pairs
        std::vector<FwdEntry> fe;
        for (const auto &q : ctx->pull) {
            if (!(q.flags & PULL_FWD)) continue;
            const int u[3] = {q.di, q.dj, q.dk};
            fe.push_back(FwdEntry{u[L.perm[0]], u[L.perm[1]], u[L.perm[2]], q.h});
        }
        if (ctx->d_fwd_entries) HIPCHK(hipFree(ctx->d_fwd_entries));
        ctx->d_fwd_entries = nullptr;
        ctx->n_fwd_entries = (int)fe.size();
        if (!fe.empty()) {
            HIPCHK(hipMalloc((void **)&ctx->d_fwd_entries, fe.size() * sizeof(FwdEntry)));
            HIPCHK(hipMemcpy(ctx->d_fwd_entries, fe.data(), fe.size() * sizeof(FwdEntry),
                             hipMemcpyHostToDevice));
        }
    }
    if (ctx->d_cell_entries) HIPCHK(hipFree(ctx->d_cell_entries));
    ctx->d_cell_entries = nullptr;
    ctx->n_cell_entries = (int)ce.size();
    if (!ce.empty()) {
        HIPCHK(hipMalloc((void **)&ctx->d_cell_entries, ce.size() * sizeof(CellEntry)));
        HIPCHK(hipMemcpy(ctx->d_cell_entries, ce.data(), ce.size() * sizeof(CellEntry),
                         hipMemcpyHostToDevice));
    }
    return 0;
}

// Activity words of one start: two parities of unit flags, the held-back plane bits and the
// number of source units (see plan_pass_kernel).
static size_t flag_words(const DevLayout &L)
{
    const size_t strip = 3 * (size_t)std::max(strip_units(L, 1), 1) + 4;      // (one-plane units: the larger grid)
    const size_t tile = 2 * (size_t)tile_count(L.n[0], TILE_X) * tile_count(L.n[1], TILE_Y) * tile_count(L.n[2], TILE_Z);
    return (std::max(strip, tile) + 1) & ~(size_t)1;       // (even: the TILE kernel views them as int2)
}

static int ensure_capacity(ttsweep_ctx *ctx, int nstart)
{
    if (nstart <= ctx->capacity_starts && ctx->d_T) return 0;
    nstart = std::max(nstart, ctx->capacity_starts);
    if (ctx->d_T) HIPCHK(hipFree(ctx->d_T));
    if (ctx->d_starts) HIPCHK(hipFree(ctx->d_starts));
    if (ctx->d_active) HIPCHK(hipFree(ctx->d_active));
    if (ctx->d_changed) HIPCHK(hipFree(ctx->d_changed));
    if (ctx->h_starts) HIPCHK(hipHostFree(ctx->h_starts));
    if (ctx->h_active) HIPCHK(hipHostFree(ctx->h_active));
    if (ctx->h_changed) HIPCHK(hipHostFree(ctx->h_changed));
    if (ctx->d_tile_flags) HIPCHK(hipFree(ctx->d_tile_flags));
    if (ctx->d_work) HIPCHK(hipFree(ctx->d_work));
    if (ctx->h_work) HIPCHK(hipHostFree(ctx->h_work));
    if (ctx->d_tface) HIPCHK(hipFree(ctx->d_tface));
    ctx->d_tface = nullptr;
    ctx->d_tile_flags = nullptr; ctx->d_work = nullptr; ctx->h_work = nullptr;
    ctx->d_T = nullptr; ctx->d_starts = nullptr; ctx->d_active = nullptr; ctx->d_changed = nullptr;
    ctx->h_starts = nullptr; ctx->h_active = nullptr; ctx->h_changed = nullptr;
    ctx->capacity_starts = 0;
    HIPCHK(hipMalloc((void **)&ctx->d_T, (size_t)nstart * ctx->L.cells * sizeof(float)));
    HIPCHK(hipMalloc((void **)&ctx->d_starts, nstart * sizeof(StartDesc)));
    HIPCHK(hipMalloc((void **)&ctx->d_active, nstart * sizeof(int)));
    HIPCHK(hipMalloc((void **)&ctx->d_changed, PASS_SLOTS * nstart * sizeof(int)));
    HIPCHK(hipHostMalloc((void **)&ctx->h_starts, nstart * sizeof(StartDesc)));
    HIPCHK(hipHostMalloc((void **)&ctx->h_active, nstart * sizeof(int)));
    HIPCHK(hipHostMalloc((void **)&ctx->h_changed, PASS_SLOTS * nstart * sizeof(int)));
    HIPCHK(hipMalloc((void **)&ctx->d_tile_flags,
                     (size_t)nstart * flag_words(ctx->L) * sizeof(int)));
    HIPCHK(hipMalloc((void **)&ctx->d_work, 3 * nstart * sizeof(unsigned long long)));
    HIPCHK(hipHostMalloc((void **)&ctx->h_work, 3 * nstart * sizeof(unsigned long long)));
    if (ctx->kernel == TTSWEEP_KERNEL_TILE)
        HIPCHK(hipMalloc((void **)&ctx->d_tface,
                         (size_t)nstart * tile_face_cells(ctx->L, ctx->tile_fz) * sizeof(float)));
    ctx->capacity_starts = nstart;
    return 0;
}

static int timed_event(ttsweep_ctx *ctx, hipEvent_t *out)
{
    if (ctx->ev_used == ctx->ev_pool.size()) {
        hipEvent_t e;
        HIPCHK(hipEventCreate(&e));
        ctx->ev_pool.push_back(e);
    }
    *out = ctx->ev_pool[ctx->ev_used++];
    HIPCHK(hipEventRecord(*out, ctx->stream));
    return 0;
}

// STRIP work list (static order of the units, input of plan_pass_kernel).  There is one
// queue per XCD of the device (ctx->nlists, counted at create; 8 on a whole MI355X), a
// workgroup drains the queue of the XCD it runs on first, and entry i of the list belongs
// to queue i % nlists.  Every active start is given a set of XCDs (one XCD when there are
// at least as many starts as XCDs, several when there are fewer)
// and its units are listed for those XCDs nearest to the start point first.  The units of
// one start therefore mostly share one L2, and a unit usually runs after the units between
// it and the start have finished their update of this pass: fresh travel times then cross
// several units in ONE pass instead of one unit per pass.  Correctness never depends on
// this order.
static int build_worklist(ttsweep_ctx *ctx, int nactive)
{
    const auto t_begin = std::chrono::steady_clock::now();
    const int nunits = strip_units(ctx->L, ctx->np);
    const int NX = ctx->nlists;
    std::vector<std::vector<int2>> per_xcd(NX);
    if (nactive >= NX) {
        // XCD x serves starts x, x+8, ...; interleave them rank by rank
        for (int x = 0; x < NX; x++)
            for (int k = 0; k < nunits; k++)
                for (int a = x; a < nactive; a += NX) {
                    const int s = ctx->h_active[a];
                    per_xcd[x].push_back(make_int2(s, ctx->unit_order[s][k]));
                }
    } else {
        // start a owns XCDs a, a+nactive, ...; deal its units over them (dealing whole
        // sectors around the start to one XCD each was measured: no less work, worse balance)
        for (int a = 0; a < nactive; a++) {
            const int s = ctx->h_active[a];
            std::vector<int> mine;
            for (int x = a; x < NX; x += nactive) mine.push_back(x);
            for (int k = 0; k < nunits; k++)
                per_xcd[mine[k % mine.size()]].push_back(make_int2(s, ctx->unit_order[s][k]));
        }
    }
    size_t longest = 0;
    for (const auto &q : per_xcd) longest = std::max(longest, q.size());
    std::vector<int2> flat(longest * NX, make_int2(0, -1));
    for (int x = 0; x < NX; x++)
        for (size_t j = 0; j < per_xcd[x].size(); j++) flat[j * NX + x] = per_xcd[x][j];
    if (flat.size() > ctx->worklist_cap) {
        if (ctx->d_worklist) HIPCHK(hipFree(ctx->d_worklist));
        ctx->d_worklist = nullptr;
        HIPCHK(hipMalloc((void **)&ctx->d_worklist, flat.size() * sizeof(int2)));
        ctx->worklist_cap = flat.size();
    }
    HIPCHK(hipMemcpyAsync(ctx->d_worklist, flat.data(), flat.size() * sizeof(int2),
                          hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));      // `flat` is a stack-lifetime buffer
    ctx->worklist_len = (long long)flat.size();
    // queues that can hold every unit of the list
    if (longest > ctx->unitq_cap) {
        if (ctx->d_unitq) HIPCHK(hipFree(ctx->d_unitq));
        ctx->d_unitq = nullptr;
        HIPCHK(hipMalloc((void **)&ctx->d_unitq, longest * UNITQ_LISTS * sizeof(int4)));   // (room for 8 lists)
        ctx->unitq_cap = longest;
    }
    if (!ctx->d_unitq_ctrl) HIPCHK(hipMalloc((void **)&ctx->d_unitq_ctrl, (UNITQ_CTRL_WORDS + 1) * sizeof(int)));
    if (ctx->unitq_blocks == 0) {
        hipDeviceProp_t prop;
        HIPCHK(hipGetDeviceProperties(&prop, ctx->device));
        const int wgs = units_wgs_per_cu() * std::max(prop.multiProcessorCount, 1);     // 2 workgroups (8 waves) per CU:
            // measured optimum - a third one adds no throughput, lengthens every unit and lets
            // fewer units see their neighbours' updates of the same pass
        ctx->unitq_blocks = ((wgs + ctx->nlists - 1) / ctx->nlists) * ctx->nlists;
    }
#ifdef TTSWEEP_DEBUG_ENV
    if (getenv("TTSWEEP_TRACE"))
        fprintf(stderr, "ttsweep work list for %d starts: %.0f us\n", nactive,
                std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_begin).count());
#else
    (void)t_begin;
#endif
    return 0;
}

// Units of one start ordered by distance (unit centre to start point).  The queues hand
// the units out in this order, so a unit usually starts after the units between it and the
// start have finished their update of this pass and sees their fresh values (measured:
// 8 % less work and 6 % less time than an order that keeps runs of neighbouring planes
// together for cache locality).
static void order_units(const ttsweep_ctx *ctx, const StartDesc &sd, std::vector<int> &order)
{
    const DevLayout &L = ctx->L;
    const int btiles = strip_btiles(L), cstrips = strip_cstrips(L);
    const int np = ctx->np;
    const int nunits = strip_units(L, np);
    std::vector<std::pair<long long, int>> key(nunits);
    for (int t = 0; t < nunits; t++) {
        const int cs = t % cstrips, bt = (t / cstrips) % btiles, A = t / (cstrips * btiles);
        const long long cb = std::min(bt * STRIP_TB + STRIP_TB / 2, L.n[1] - 1);
        const long long cc = std::min(cs * STRIP_K + STRIP_K / 2, L.n[2] - 1);
        // (distances in half cells: a unit of two planes is centred between them)
        const long long da = 2 * (np * A - sd.sa) + (np - 1), db = 2 * (cb - sd.sb), dc = 2 * (cc - sd.sc);
        key[t] = {da * da + db * db + dc * dc, t};
    }
    std::sort(key.begin(), key.end());
    order.resize(nunits);
    for (int t = 0; t < nunits; t++) order[t] = key[t].second;
}

// Squared radius (cells) of the distance gate for the pass about to be launched.
static float gate_r2(const ttsweep_ctx *ctx)
{
    if (ctx->gate_speed <= 0) return 3.0e38f;       // gate disabled
    const double r = ctx->gate_r0 + ctx->gate_speed * (double)ctx->pass_index;
    return (float)(r * r);
}

// One full-grid pass for the active starts.
// STRIP: the pass's "changed" words arrive in h_changed_slot without a copy command, and
// d_changed_next is cleared for the pass after this one (UnitPassTail).
static int launch_pass(ttsweep_ctx *ctx, int nactive, int nstart, int *d_changed, int *h_changed_slot,
                       int *d_changed_next)
{
    hipEvent_t e0, e1;
    if (ctx->timing && timed_event(ctx, &e0)) return -1;
    if (ctx->kernel == TTSWEEP_KERNEL_STRIP) {
        HIPCHK(launch_plan_pass(ctx->L, ctx->d_starts, ctx->d_worklist, ctx->worklist_len, d_changed,
                                ctx->d_unitq, (int)ctx->unitq_cap, ctx->nlists, ctx->d_unitq_ctrl,
                                ctx->plans[ctx->np - 1], ctx->pass_index & 1, gate_r2(ctx), ctx->stream));
        UnitPassTail tail;
        tail.active = ctx->d_active;
        tail.nactive = nactive;
        tail.entries = ctx->d_cell_entries;
        tail.nentries = ctx->n_cell_entries;
        tail.max_box_cells = (int)ctx->max_box_cells;
        tail.nstart = nstart;
        tail.changed_host = h_changed_slot;
        tail.changed_next = d_changed_next;
        HIPCHK(launch_sweep_units(ctx->L, ctx->d_v, ctx->d_starts, ctx->d_unitq, (int)ctx->unitq_cap,
                                  ctx->nlists, ctx->d_unitq_ctrl, ctx->unitq_blocks, d_changed,
                                  ctx->d_strip_items[ctx->np - 1], ctx->plans[ctx->np - 1], ctx->pass_index & 1,
                                  tail, ctx->stream));
    } else if (ctx->kernel == TTSWEEP_KERNEL_TILE) {
        // one ordering sweep: the tile hyperplanes in stream order, one launch each
        TileSweep P;
        P.L = ctx->L;
        P.v = ctx->d_v;
        P.starts = ctx->d_starts;
        P.active = ctx->d_active;
        P.changed = d_changed;
        P.nactive = nactive;
        P.NI = tile_count(ctx->L.n[0], TILE_X);
        P.NJ = tile_count(ctx->L.n[1], TILE_Y);
        P.NK = tile_count(ctx->L.n[2], TILE_Z);
        P.R = ctx->tile_R;
        const int o = ctx->pass_index & 7;          // the eight orderings in turn
        P.sx = (o & 1) ? -1 : 1;
        P.sy = (o & 2) ? -1 : 1;
        P.sz = (o & 4) ? -1 : 1;
        P.nent = ctx->tile_nent;
        P.T0 = ctx->d_T;
        P.state0 = ctx->d_tile_flags;
        P.state_stride = (long long)flag_words(ctx->L);
        P.work0 = ctx->d_work;
        P.fz = ctx->tile_fz;
        P.vface = ctx->d_vface;
        P.tface = ctx->d_tface;
        P.face_cells = tile_face_cells(ctx->L, ctx->tile_fz);
        for (int e = 0; e < TILE_MAX_ENT; e++) P.ent[e] = ctx->tile_ent[e];
        const int nsteps = P.NI + P.NJ + P.NK - 2;
        const size_t need_list = (size_t)P.NJ * P.NK * nactive;
        if (need_list > ctx->tile_list_cap) {
            HIPCHK(hipStreamSynchronize(ctx->stream));
            if (ctx->d_tile_list) HIPCHK(hipFree(ctx->d_tile_list));
            ctx->d_tile_list = nullptr;
            HIPCHK(hipMalloc((void **)&ctx->d_tile_list, need_list * sizeof(int2)));
            ctx->tile_list_cap = need_list;
        }
        if ((size_t)nsteps > ctx->tile_ctrl_cap) {
            HIPCHK(hipStreamSynchronize(ctx->stream));
            if (ctx->d_tile_ctrl) HIPCHK(hipFree(ctx->d_tile_ctrl));
            ctx->d_tile_ctrl = nullptr;
            HIPCHK(hipMalloc((void **)&ctx->d_tile_ctrl, (size_t)nsteps * sizeof(int)));
            ctx->tile_ctrl_cap = nsteps;
        }
        if (ctx->tile_blocks == 0) {
            hipDeviceProp_t prop;
            HIPCHK(hipGetDeviceProperties(&prop, ctx->device));
            // as many single-wavefront workgroups as the device holds at once
#ifdef TTSWEEP_TILE_WGS_PER_CU
            int per_cu = TTSWEEP_TILE_WGS_PER_CU;
#else
            int per_cu = 1;
            HIPCHK(tile_sweep_wgs_per_cu(P, &per_cu));
#endif
            ctx->tile_blocks = per_cu * std::max(prop.multiProcessorCount, 1);
        }
        HIPCHK(hipMemsetAsync(ctx->d_tile_ctrl, 0, (size_t)nsteps * sizeof(int), ctx->stream));
        for (int D = 0; D < nsteps; D++) {
            P.D = D;
            P.epoch = ++ctx->tile_epoch;
            HIPCHK(launch_tile_sweep(P, ctx->d_tile_list, ctx->d_tile_ctrl + D, ctx->tile_blocks, ctx->stream));
        }
        ctx->stats.launches += nsteps - 1;
    } else {
        HIPCHK(launch_sweep_cell(ctx->L, ctx->d_v, ctx->d_starts, ctx->d_active, nactive,
                                 d_changed, ctx->d_cell_entries, ctx->n_cell_entries,
                                 ctx->stream));
    }
    if (ctx->timing && timed_event(ctx, &e1)) return -1;
    ctx->stats.launches++;
    ctx->pass_index++;
    return 0;
}

static int solve_device_body(ttsweep_ctx *ctx, int nstart, const ttsweep_start *starts,
                             float *const *tt_dev, int init);

// ---------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------
extern "C" {

int ttsweep_abi_version(void) { return TTSWEEP_ABI_VERSION; }

const char *ttsweep_last_error(void) { return g_last_error.c_str(); }

int ttsweep_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return set_error("hipGetDeviceCount: %s", hipGetErrorString(e));
    return n;
}

int ttsweep_build_pull_star(const ttsweep_fs *fs, int starstart, int starstop,
                            ttsweep_pull_entry *out, int cap)
{
    if (!fs || starstart < 0 || starstop < starstart) return set_error("bad star range");
    std::vector<ttsweep_pull_entry> p = build_pull_star(fs, starstart, starstop);
    for (int e = 0; e < (int)p.size() && e < cap; e++) out[e] = p[e];
    return (int)p.size();
}

long long ttsweep_relaxations_per_sweep(int nx, int ny, int nz, const ttsweep_fs *fs,
                                        int starstart, int starstop)
{
    long long total = 0;
    for (int l = starstart; l < starstop; l++) {
        long long a = std::max(nx - std::abs(fs[l].i), 0);
        long long b = std::max(ny - std::abs(fs[l].j), 0);
        long long c = std::max(nz - std::abs(fs[l].k), 0);
        total += a * b * c;
    }
    return total;
}

ttsweep_ctx *ttsweep_create(int device, int nx, int ny, int nz, const ttsweep_fs *fs,
                            int starstart, int starstop)
{
    if (nx <= 0 || ny <= 0 || nz <= 0 || !fs || starstart < 0 || starstop < starstart) {
        set_error("ttsweep_create: bad arguments");
        return nullptr;
    }
    int ndev = ttsweep_device_count();
    if (ndev <= 0) {
        if (ndev == 0) set_error("ttsweep_create: no HIP device (there is no CPU fallback)");
        return nullptr;
    }
    if (device < 0 || device >= ndev) {
        set_error("ttsweep_create: device %d out of range (%d devices)", device, ndev);
        return nullptr;
    }
    ttsweep_ctx *ctx = new ttsweep_ctx();
    ctx->device = device;
    ctx->nx = nx; ctx->ny = ny; ctx->nz = nz;
    ctx->pull = build_pull_star(fs, starstart, starstop);
    ctx->radius = pull_star_radius(ctx->pull);
    ctx->gate_speed = std::max(1.0, 0.5 * ctx->radius);
    ctx->gate_r0 = ctx->radius + 1.0;
    ctx->relax_per_sweep = ttsweep_relaxations_per_sweep(nx, ny, nz, fs, starstart, starstop);
    ctx->kernel = auto_kernel(ctx);
    make_layout(ctx);

    bool ok = hipSetDevice(device) == hipSuccess
           && hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) == hipSuccess
           && hipEventCreate(&ctx->ev_solve0) == hipSuccess
           && hipEventCreate(&ctx->ev_solve1) == hipSuccess
           && hipEventCreateWithFlags(&ctx->ev_flags[0], hipEventDisableTiming) == hipSuccess
           && hipEventCreateWithFlags(&ctx->ev_flags[1], hipEventDisableTiming) == hipSuccess
           && hipEventCreateWithFlags(&ctx->ev_flags[2], hipEventDisableTiming) == hipSuccess
           && hipMalloc((void **)&ctx->d_v, (size_t)ctx->L.cells * sizeof(float)) == hipSuccess;
    if (!ok) {
        set_error("ttsweep_create: HIP setup failed: %s", hipGetErrorString(hipGetLastError()));
        ttsweep_destroy(ctx);
        return nullptr;
    }
    if (count_xcds(ctx) || upload_star(ctx)
        || (ctx->kernel == TTSWEEP_KERNEL_STRIP && upload_strip_plan(ctx))) {
        ttsweep_destroy(ctx);
        return nullptr;
    }
    return ctx;
}

void ttsweep_destroy(ttsweep_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(ctx->d_v);
    (void)hipFree(ctx->d_cell_entries);
    (void)hipFree(ctx->d_fwd_entries);
    for (StripItem *d : ctx->d_strip_items) (void)hipFree(d);
    (void)hipFree(ctx->d_T);
    (void)hipFree(ctx->d_starts);
    (void)hipFree(ctx->d_active);
    (void)hipFree(ctx->d_changed);
    (void)hipFree(ctx->d_tile_flags);
    (void)hipFree(ctx->d_worklist);
    (void)hipFree(ctx->d_unitq);
    (void)hipFree(ctx->d_unitq_ctrl);
    (void)hipFree(ctx->d_work);
    (void)hipFree(ctx->d_tile_list);
    (void)hipFree(ctx->d_vface);
    (void)hipFree(ctx->d_tface);
    (void)hipFree(ctx->d_tile_ctrl);
    if (ctx->h_work) (void)hipHostFree(ctx->h_work);
    if (ctx->h_starts) (void)hipHostFree(ctx->h_starts);
    if (ctx->h_active) (void)hipHostFree(ctx->h_active);
    if (ctx->h_changed) (void)hipHostFree(ctx->h_changed);
    for (hipEvent_t e : ctx->ev_pool) (void)hipEventDestroy(e);
    if (ctx->ev_solve0) (void)hipEventDestroy(ctx->ev_solve0);
    if (ctx->ev_solve1) (void)hipEventDestroy(ctx->ev_solve1);
    for (hipEvent_t e : ctx->ev_flags)
        if (e) (void)hipEventDestroy(e);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int ttsweep_set_option(ttsweep_ctx *ctx, int key, long long value)
{
    if (!ctx) return set_error("null context");
    switch (key) {
    case TTSWEEP_OPT_TIMING: ctx->timing = value != 0; return 0;
    case TTSWEEP_OPT_KERNEL: {
        int k = (int)value;
        if (k == TTSWEEP_KERNEL_AUTO) k = auto_kernel(ctx);
        if (!kernel_available(ctx, k))
            return set_error("kernel variant %lld not available for this star", value);
        if (k == ctx->kernel) return 0;
        // the padded layout depends on the kernel: rebuild it and drop device copies
        if (ctx_bind(ctx)) return -1;
        HIPCHK(hipStreamSynchronize(ctx->stream));
        ctx->kernel = k;
        make_layout(ctx);
        HIPCHK(hipFree(ctx->d_v));
        ctx->d_v = nullptr;
        if (ctx->d_vface) HIPCHK(hipFree(ctx->d_vface));
        ctx->d_vface = nullptr;
        ctx->have_v = false;
        if (ctx->d_T) HIPCHK(hipFree(ctx->d_T));
        ctx->d_T = nullptr;
        ctx->capacity_starts = 0;
        ctx->unit_order_key.assign(ctx->unit_order_key.size(), -1);     // orders belong to the old layout
        HIPCHK(hipMalloc((void **)&ctx->d_v, (size_t)ctx->L.cells * sizeof(float)));
        if (upload_star(ctx)) return -1;
        if (k == TTSWEEP_KERNEL_STRIP && upload_strip_plan(ctx)) return -1;
        return 0;
    }
    case TTSWEEP_OPT_MAX_SWEEPS:
        if (value <= 0) return set_error("max sweeps must be positive");
        ctx->max_sweeps = value;
        return 0;
    case TTSWEEP_OPT_MAX_BATCH:
        if (value < 0) return set_error("max batch must be >= 0");
        ctx->max_batch = (int)value;
        return 0;
    case TTSWEEP_OPT_GATE_SPEED_MILLI:
        if (value < 0) return set_error("gate speed must be >= 0");
        ctx->gate_speed = (double)value / 1000.0;
        return 0;
    case TTSWEEP_OPT_PAIR_MIN_STARTS:
        if (value < 0) return set_error("start count must be >= 0");
        ctx->pair_min_starts = (int)std::min<long long>(value, 1 << 30);      // (>= 0: the explicit rule from now on)
        return 0;
    case TTSWEEP_OPT_GATE_R0_MILLI:
        if (value < 0) return set_error("gate start radius must be >= 0");
        ctx->gate_r0 = (double)value / 1000.0;
        return 0;
    default: return set_error("unknown option %d", key);
    }
}

int ttsweep_set_velocity_device(ttsweep_ctx *ctx, const float *v_dev)
{
    if (!ctx || !v_dev) return set_error("null argument");
    if (ctx_bind(ctx)) return -1;
    // every cell must be a positive finite number (positive delays: SURVEY.md section 8-a)
    if (ensure_capacity(ctx, 1)) return -1;
    unsigned long long *d_bad = ctx->d_work, h_bad = 0;
    const long long n = (long long)ctx->nx * ctx->ny * ctx->nz;
    HIPCHK(hipMemsetAsync(d_bad, 0, sizeof(unsigned long long), ctx->stream));
    HIPCHK(launch_count_bad_velocity(v_dev, n, d_bad, ctx->stream));
    HIPCHK(hipMemcpyAsync(&h_bad, d_bad, sizeof h_bad, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(launch_pack(ctx->L, v_dev, ctx->d_v, 0.0f, ctx->stream));
    if (ctx->kernel == TTSWEEP_KERNEL_TILE) {
        if (!ctx->d_vface)
            HIPCHK(hipMalloc((void **)&ctx->d_vface, (size_t)tile_face_cells(ctx->L, ctx->tile_fz) * sizeof(float)));
        HIPCHK(launch_build_tile_faces(ctx->L, ctx->d_v, ctx->d_vface, ctx->tile_fz, ctx->stream));
    }
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (h_bad) {
        ctx->have_v = false;
        return set_error("velocity volume holds %llu cells that are not positive finite numbers", h_bad);
    }
    ctx->have_v = true;
    return 0;
}

int ttsweep_set_velocity(ttsweep_ctx *ctx, const float *v_host)
{
    if (!ctx || !v_host) return set_error("null argument");
    if (ctx_bind(ctx)) return -1;
    const size_t bytes = (size_t)ctx->nx * ctx->ny * ctx->nz * sizeof(float);
    float *tmp = nullptr;
    HIPCHK(hipMalloc((void **)&tmp, bytes));
    // copy and consumer kernels are ordered by the same stream
    hipError_t e = hipMemcpyAsync(tmp, v_host, bytes, hipMemcpyHostToDevice, ctx->stream);
    int rc = 0;
    if (e != hipSuccess) rc = set_error("velocity upload failed: %s", hipGetErrorString(e));
    else rc = ttsweep_set_velocity_device(ctx, tmp);     // (synchronises the stream)
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(tmp);
    return rc;
}

int ttsweep_solve_device(ttsweep_ctx *ctx, int nstart, const ttsweep_start *starts,
                         float *const *tt_dev, int init)
{
    if (!ctx || !starts || !tt_dev || nstart < 0) return set_error("bad arguments");
    if (!ctx->have_v) return set_error("velocity not set");
    if (ctx_bind(ctx)) return -1;
    ctx->stats = ttsweep_stats{};
    ctx->stats.nstart = nstart;
    ctx->stats.cells = (long long)ctx->nx * ctx->ny * ctx->nz;
    ctx->stats.relaxations_per_sweep = ctx->relax_per_sweep;
    ctx->stats.kernel_variant = ctx->kernel;
    ctx->ev_used = 0;
    if (nstart == 0) return 0;
    for (int s = 0; s < nstart; s++)        // before anything is queued on the stream
        if (starts[s].i < 0 || starts[s].i >= ctx->nx || starts[s].j < 0 || starts[s].j >= ctx->ny
            || starts[s].k < 0 || starts[s].k >= ctx->nz)
            return set_error("start %d (%d,%d,%d) outside the grid", s, starts[s].i, starts[s].j, starts[s].k);
    if (ensure_capacity(ctx, nstart)) return -1;
    const int rc = solve_device_body(ctx, nstart, starts, tt_dev, init);
    if (rc < 0) {
        // A failed launch, copy or convergence cap leaves passes queued and the pass state
        // half-updated: drain the stream (keeping the first error's text) and put the
        // context back into its between-solves state so that it can be used again.
        const std::string first = g_last_error;
        (void)hipStreamSynchronize(ctx->stream);
        if (ctx->d_unitq_ctrl)
            (void)hipMemset(ctx->d_unitq_ctrl, 0, (UNITQ_CTRL_WORDS + 1) * sizeof(int));
        ctx->pass_index = 0;
        g_last_error = first;
    }
    return rc;
}

} // extern "C"

static int solve_device_body(ttsweep_ctx *ctx, int nstart, const ttsweep_start *starts,
                             float *const *tt_dev, int init)
{
    const DevLayout &L = ctx->L;
    HIPCHK(hipEventRecord(ctx->ev_solve0, ctx->stream));
    // units of two planes when there are starts enough to fill the machine with them
    const bool pairs = ctx->pair_min_starts >= 0
        ? nstart >= ctx->pair_min_starts
        : (long long)nstart * strip_units(L, 1) >= ctx->pair_min_units;
    const int np = pairs ? STRIP_PLANES : 1;
    if (np != ctx->np) ctx->unit_order_key.assign(ctx->unit_order_key.size(), -1);     // orders belong to the other unit grid
    ctx->np = np;

    for (int s = 0; s < nstart; s++) {
        const int u[3] = {starts[s].i, starts[s].j, starts[s].k};
        StartDesc &sd = ctx->h_starts[s];
        sd.T = ctx->d_T + (size_t)s * L.cells;
        sd.sa = u[L.perm[0]];
        sd.sb = u[L.perm[1]];
        sd.sc = u[L.perm[2]];
        sd.sidx = dev_index(L, sd.sa, sd.sb, sd.sc);
        sd.pad_ = 0;
        fill_special_box(ctx, sd);
        {
            long long vol = 1;
            for (int d = 0; d < 3; d++) vol *= std::max(sd.box_hi[d] - sd.box_lo[d] + 1, 0);
            ctx->max_box_cells = std::max<long long>(s == 0 ? 0 : ctx->max_box_cells, vol);
        }
        sd.tile_flags = ctx->d_tile_flags + (size_t)s * flag_words(L);
        sd.work = ctx->d_work + 3 * s;
        if (init) HIPCHK(launch_init_tt(L, sd.T, sd.sidx, ctx->stream));
        else HIPCHK(launch_pack(L, tt_dev[s], sd.T, INFINITY, ctx->stream));
        if (ctx->kernel == TTSWEEP_KERNEL_STRIP)
            HIPCHK(launch_init_tile_flags(L, sd, /*from_box=*/!init, ctx->stream));
        if (ctx->kernel == TTSWEEP_KERNEL_TILE) {
            HIPCHK(launch_init_tile_state(L, sd, /*from_box=*/!init, ctx->stream));
            float *const faces = ctx->d_tface + (size_t)s * tile_face_cells(L, ctx->tile_fz);
            if (init) HIPCHK(launch_init_tile_faces(L, faces, ctx->tile_fz, sd.sa, sd.sb, sd.sc, ctx->stream));
            else HIPCHK(launch_build_tile_faces(L, sd.T, faces, ctx->tile_fz, ctx->stream));
        }
        ctx->h_active[s] = s;
    }
    if (ctx->kernel == TTSWEEP_KERNEL_STRIP) {
        // host work while the device initialises the boxes: every start's units, nearest
        // first (kept from the previous solve when the start point is the same)
        if ((int)ctx->unit_order.size() < nstart) {
            ctx->unit_order.resize(nstart);
            ctx->unit_order_key.resize(nstart, -1);
        }
        for (int s = 0; s < nstart; s++) {
            const StartDesc &sd = ctx->h_starts[s];
            if (ctx->unit_order_key[s] == sd.sidx && !ctx->unit_order[s].empty()) continue;
            order_units(ctx, sd, ctx->unit_order[s]);
            ctx->unit_order_key[s] = sd.sidx;
        }
    }
    HIPCHK(hipMemcpyAsync(ctx->d_starts, ctx->h_starts, nstart * sizeof(StartDesc),
                          hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(ctx->d_active, ctx->h_active, nstart * sizeof(int),
                          hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemsetAsync(ctx->d_work, 0, 3 * nstart * sizeof(unsigned long long), ctx->stream));
    ctx->pass_index = 0;
    ctx->tile_epoch = 1;
    if (ctx->kernel == TTSWEEP_KERNEL_STRIP) {
        if (build_worklist(ctx, nstart)) return -1;
        // the passes keep these cleared themselves from here on
        HIPCHK(hipMemsetAsync(ctx->d_changed, 0, (size_t)PASS_SLOTS * nstart * sizeof(int), ctx->stream));
        HIPCHK(hipMemsetAsync(ctx->d_unitq_ctrl, 0, (UNITQ_CTRL_WORDS + 1) * sizeof(int), ctx->stream));
    }

    // driver loop: serial_new/...:151-170 without the break (:168-169).  Passes are
    // enqueued ONE AHEAD of the convergence test: pass k+1 is already running while the
    // host waits for the "changed" words of pass k, so the GPU never idles between
    // passes.  A start whose pass-k words show no change is converged; the pass k+1
    // that was launched speculatively for it finds all its units inactive.
    std::vector<int> sweeps(nstart, 0);
    std::vector<char> done(nstart, 0);          // converged: the pass launched one ahead for it is not counted
#ifdef TTSWEEP_DEBUG_ENV
    const bool trace = getenv("TTSWEEP_TRACE") != nullptr;
#else
    const bool trace = false;
#endif
    unsigned long long trace_prev = 0, trace_prev_un = 0;
    std::vector<int> snapshot[PASS_SLOTS];      // active starts of each pass in flight
    int nactive = nstart, launched = 0, processed = 0;
    bool anychange_ever = false;
    auto t_pass = std::chrono::steady_clock::now();
    while (processed < launched || nactive > 0) {
        if (nactive > 0 && launched - processed < 2) {          // enqueue the next pass
            const int slot = launched % PASS_SLOTS;
            int *dch = ctx->d_changed + (size_t)slot * nstart;
            int *hch_slot = ctx->h_changed + (size_t)slot * nstart;
            const bool strip = ctx->kernel == TTSWEEP_KERNEL_STRIP;
            if (!strip) HIPCHK(hipMemsetAsync(dch, 0, nstart * sizeof(int), ctx->stream));
            const auto t_enq = std::chrono::steady_clock::now();
            if (launch_pass(ctx, nactive, nstart, dch, hch_slot,
                            ctx->d_changed + (size_t)((launched + 1) % PASS_SLOTS) * nstart))
                return -1;
            if (trace)
                fprintf(stderr, "   (host: %.0f us to enqueue pass %d)\n",
                        std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_enq).count(),
                        launched + 1);
            if (!strip)
                HIPCHK(hipMemcpyAsync(hch_slot, dch, nstart * sizeof(int), hipMemcpyDeviceToHost,
                                      ctx->stream));
            HIPCHK(hipEventRecord(ctx->ev_flags[slot], ctx->stream));
            snapshot[slot].assign(ctx->h_active, ctx->h_active + nactive);
            launched++;
            if (launched - processed < 2 && nactive > 0 && launched == 1) continue;   // prime the pipeline
        }
        // examine the oldest pass in flight
        const int slot = processed % PASS_SLOTS;
        HIPCHK(hipEventSynchronize(ctx->ev_flags[slot]));
        const int *hch = ctx->h_changed + (size_t)slot * nstart;
        if (trace) {    // TTSWEEP_TRACE=1: per-pass activity on stderr (serialises the passes)
            HIPCHK(hipStreamSynchronize(ctx->stream));
            HIPCHK(hipMemcpy(ctx->h_work, ctx->d_work, 3 * nstart * sizeof(unsigned long long),
                             hipMemcpyDeviceToHost));
            unsigned long long tot = 0, un = 0;
            for (int s = 0; s < nstart; s++) { tot += ctx->h_work[3 * s]; un += ctx->h_work[3 * s + 2]; }
            const double us = std::chrono::duration<double, std::micro>(
                                  std::chrono::steady_clock::now() - t_pass).count();
            t_pass = std::chrono::steady_clock::now();
            fprintf(stderr, "ttsweep pass %d: %d active starts, %.3f full-sweep equivalents relaxed, "
                    "%llu units, %.0f us\n", processed + 1, (int)snapshot[slot].size(),
                    (double)(tot - trace_prev) / (double)ctx->stats.cells
                        / (double)std::max<size_t>(ctx->pull.size(), 1),
                    un - trace_prev_un, us);
            trace_prev = tot;
            trace_prev_un = un;
            if (ctx->kernel == TTSWEEP_KERNEL_TILE && ctx->d_tile_ctrl) {      // due tiles per launch of the sweep
                std::vector<int> cnt(ctx->tile_ctrl_cap);
                HIPCHK(hipMemcpy(cnt.data(), ctx->d_tile_ctrl, cnt.size() * sizeof(int), hipMemcpyDeviceToHost));
                long long empty = 0, small = 0, big = 0, tiles_small = 0, tiles_big = 0;
                for (int c : cnt) {
                    if (c == 0) empty++;
                    else if (c < ctx->tile_blocks) { small++; tiles_small += c; }
                    else { big++; tiles_big += c; }
                }
                fprintf(stderr, "   launches: %lld empty, %lld below one round (%lld tiles), %lld larger (%lld tiles)\n",
                        empty, small, tiles_small, big, tiles_big);
            }
        }
        bool dropped = false;
        for (int s : snapshot[slot]) {
            if (done[s]) continue;
            sweeps[s]++;
            if (hch[s]) {           // improved, or units still held back by the gate
                if (hch[s] & CHANGED_IMPROVED) anychange_ever = true;
                if (sweeps[s] >= ctx->max_sweeps)
                    return set_error("start %d did not converge in %lld sweeps", s, ctx->max_sweeps);
            } else {
                // converged: remove it from the active list
                done[s] = 1;
                int *end = std::remove(ctx->h_active, ctx->h_active + nactive, s);
                if (end != ctx->h_active + nactive) dropped = true;
                nactive = (int)(end - ctx->h_active);
            }
        }
        processed++;
        if (dropped && nactive > 0) {
            // (the uploads are stream-ordered behind the pass in flight; the stream is
            // synchronised before h_active is touched again)
            HIPCHK(hipMemcpyAsync(ctx->d_active, ctx->h_active, nactive * sizeof(int),
                                  hipMemcpyHostToDevice, ctx->stream));
            if (ctx->kernel == TTSWEEP_KERNEL_STRIP) {
                if (build_worklist(ctx, nactive)) return -1;
            } else {
                HIPCHK(hipStreamSynchronize(ctx->stream));
            }
        }
    }

    for (int s = 0; s < nstart; s++)
        HIPCHK(launch_unpack(L, ctx->h_starts[s].T, tt_dev[s], ctx->stream));
    HIPCHK(hipMemcpyAsync(ctx->h_work, ctx->d_work, 3 * nstart * sizeof(unsigned long long),
                          hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipEventRecord(ctx->ev_solve1, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));

#ifdef TTSWEEP_PROFILE
    prof_dump();
#endif
#ifdef TTSWEEP_TILE_PROFILE
    if (ctx->kernel == TTSWEEP_KERNEL_TILE) tile_prof_dump();
#endif
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, ctx->ev_solve0, ctx->ev_solve1));
    ctx->stats.solve_ms = ms;
    for (size_t e = 0; e + 1 < ctx->ev_used; e += 2) {
        HIPCHK(hipEventElapsedTime(&ms, ctx->ev_pool[e], ctx->ev_pool[e + 1]));
        ctx->stats.sweep_kernel_ms += ms;
    }
    for (int s = 0; s < nstart; s++) {
        ctx->stats.sweeps_total += sweeps[s];
        ctx->stats.sweeps_max = std::max(ctx->stats.sweeps_max, sweeps[s]);
        // CELL kernel relaxes every cell in every pass; STRIP counts its active tiles
        // (STRIP counts cells x offsets actually relaxed; convert to whole-star cell relaxations)
        ctx->stats.cells_relaxed += ctx->kernel != TTSWEEP_KERNEL_CELL
            ? (long long)(ctx->h_work[3 * s] / std::max<size_t>(ctx->pull.size(), 1))
            : (long long)sweeps[s] * ctx->stats.cells;
    }
    return anychange_ever ? 1 : 0;
}

extern "C" {

int ttsweep_solve(ttsweep_ctx *ctx, int nstart, const ttsweep_start *starts,
                  float *const *tt_host)
{
    if (!ctx || !starts || !tt_host || nstart < 0) return set_error("bad arguments");
    if (ctx_bind(ctx)) return -1;
    if (nstart == 0) return 0;
    const size_t cells = (size_t)ctx->nx * ctx->ny * ctx->nz;

    // Starts are independent: solve them in batches that fit the device memory
    // (per start: one staging box in the caller's layout + one padded volume).
    size_t free_b = 0, total_b = 0;
    HIPCHK(hipMemGetInfo(&free_b, &total_b));
    if (ctx->d_T) free_b += (size_t)ctx->capacity_starts * ctx->L.cells * sizeof(float);   // reusable pool
    const size_t per_start = (cells + (size_t)ctx->L.cells) * sizeof(float);
    int batch = (int)std::min<size_t>((size_t)nstart, (size_t)(0.85 * (double)free_b) / per_start);
    if (batch < 1) return set_error("not enough device memory for one travel-time volume");
    if (ctx->max_batch > 0) batch = std::min(batch, ctx->max_batch);

    ttsweep_stats total{};
    int any = 0;
    for (int first = 0; first < nstart; first += batch) {
        const int n = std::min(batch, nstart - first);
        float *stage = nullptr;
        HIPCHK(hipMalloc((void **)&stage, (size_t)n * cells * sizeof(float)));
        // The caller's boxes are pinned for the duration of the batch (the reference's CUDA
        // host does the same, cuda/cudasweep-tt-multistart.cu:273-277): the copies then run at
        // the full PCIe rate and are ordered on the library's stream with the kernels that
        // consume / produce the staged boxes.  Where pinning is refused the copies still
        // work (the runtime stages them).
        std::vector<float *> ptrs(n);
        std::vector<char> pinned(n, 0);
        int rc = 0;
#ifdef TTSWEEP_DEBUG_ENV
        const bool trace = getenv("TTSWEEP_TRACE") != nullptr;
#else
        const bool trace = false;
#endif
        auto t_phase = std::chrono::steady_clock::now();
        auto lap = [&](const char *what) {
            if (!trace) return;
            (void)hipStreamSynchronize(ctx->stream);
            const auto now = std::chrono::steady_clock::now();
            fprintf(stderr, "ttsweep_solve: %s %.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t_phase).count());
            t_phase = now;
        };
        for (int s = 0; s < n; s++) {
            ptrs[s] = stage + (size_t)s * cells;
            if (hipHostRegister(tt_host[first + s], cells * sizeof(float), hipHostRegisterDefault) == hipSuccess)
                pinned[s] = 1;
            else
                (void)hipGetLastError();
        }
        lap("pin the caller's boxes");
        for (int s = 0; s < n && rc == 0; s++) {
            hipError_t e = hipMemcpyAsync(ptrs[s], tt_host[first + s], cells * sizeof(float),
                                          hipMemcpyHostToDevice, ctx->stream);
            if (e != hipSuccess) rc = set_error("travel-time upload failed: %s", hipGetErrorString(e));
        }
        lap("upload");
        if (rc == 0) rc = ttsweep_solve_device(ctx, n, starts + first, ptrs.data(), 0);
        lap("solve");
        if (rc > 0) {       // (rc == 0: nothing was stored, the caller's boxes are the result already)
            for (int s = 0; s < n; s++) {
                hipError_t e = hipMemcpyAsync(tt_host[first + s], ptrs[s], cells * sizeof(float),
                                              hipMemcpyDeviceToHost, ctx->stream);
                if (e != hipSuccess) {
                    rc = set_error("travel-time download failed: %s", hipGetErrorString(e));
                    break;
                }
            }
        }
        {
            hipError_t e = hipStreamSynchronize(ctx->stream);
            if (e != hipSuccess && rc >= 0) rc = set_error("travel-time transfer failed: %s", hipGetErrorString(e));
        }
        lap("download");
        for (int s = 0; s < n; s++)
            if (pinned[s]) (void)hipHostUnregister(tt_host[first + s]);
        (void)hipFree(stage);
        lap("unpin, free");
        if (rc < 0) return rc;
        any |= rc;
        // accumulate the per-batch counters into one report
        const ttsweep_stats &b = ctx->stats;
        total.nstart += b.nstart;
        total.sweeps_max = std::max(total.sweeps_max, b.sweeps_max);
        total.sweeps_total += b.sweeps_total;
        total.cells_relaxed += b.cells_relaxed;
        total.cells = b.cells;
        total.relaxations_per_sweep = b.relaxations_per_sweep;
        total.launches += b.launches;
        total.sweep_kernel_ms += b.sweep_kernel_ms;
        total.solve_ms += b.solve_ms;
        total.kernel_variant = b.kernel_variant;
    }
    ctx->stats = total;
    return any;
}

int ttsweep_validate_device(ttsweep_ctx *ctx, const ttsweep_start *start, const float *tt_dev,
                            long long *open_edges, long long *cells_infinite,
                            long long *cells_unsupported)
{
    if (!ctx || !start || !tt_dev) return set_error("null argument");
    if (!ctx->have_v) return set_error("velocity not set");
    if (start->i < 0 || start->i >= ctx->nx || start->j < 0 || start->j >= ctx->ny || start->k < 0
        || start->k >= ctx->nz)
        return set_error("start outside the grid");
    if (ctx_bind(ctx)) return -1;
    if (ensure_capacity(ctx, 1)) return -1;
    const DevLayout &L = ctx->L;
    const int u[3] = {start->i, start->j, start->k};
    const long long sidx = dev_index(L, u[L.perm[0]], u[L.perm[1]], u[L.perm[2]]);
    unsigned long long *d_counts = ctx->d_work;     // first words of the per-solve counters
    HIPCHK(hipMemsetAsync(d_counts, 0, 3 * sizeof(unsigned long long), ctx->stream));
    HIPCHK(launch_pack(L, tt_dev, ctx->d_T, INFINITY, ctx->stream));
    HIPCHK(launch_validate(L, ctx->d_v, ctx->d_T, sidx, ctx->d_fwd_entries, ctx->n_fwd_entries,
                           ctx->d_cell_entries, ctx->n_cell_entries, d_counts, ctx->stream));
    unsigned long long h[3] = {0, 0, 0};
    HIPCHK(hipMemcpyAsync(h, d_counts, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (open_edges) *open_edges = (long long)h[0];
    if (cells_infinite) *cells_infinite = (long long)h[1];
    if (cells_unsupported) *cells_unsupported = (long long)h[2];
    return 0;
}

int ttsweep_get_stats(const ttsweep_ctx *ctx, ttsweep_stats *out)
{
    if (!ctx || !out) return set_error("null argument");
    *out = ctx->stats;
    return 0;
}

int ttsweep_solve_multi(int ndev, const int *devices, int nx, int ny, int nz,
                        const ttsweep_fs *fs, int starstart, int starstop, const float *v_host,
                        int nstart, const ttsweep_start *starts, float *const *tt_host)
{
    if (ndev <= 0 || !devices || !starts || !tt_host || nstart < 0) return set_error("bad arguments");
    std::vector<int> rc(ndev, 0);
    std::vector<std::string> err(ndev);
    std::vector<std::thread> workers;
    // Shards balanced by estimated cost (distance from the start to the farthest corner of
    // the grid: the number of passes grows with it), longest first onto the least loaded
    // device, at most ceil(nstart / ndev) starts per device (multistart.all_shards).
    std::vector<std::vector<int>> shard(ndev);
    {
        std::vector<double> cost(nstart), load(ndev, 0.0);
        std::vector<int> order(nstart);
        const int n[3] = {nx, ny, nz};
        for (int s = 0; s < nstart; s++) {
            const int c[3] = {starts[s].i, starts[s].j, starts[s].k};
            double d2 = 0;
            for (int a = 0; a < 3; a++) {
                const double far = std::max(c[a], n[a] - 1 - c[a]);
                d2 += far * far;
            }
            cost[s] = std::sqrt(d2);
            order[s] = s;
        }
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return cost[a] > cost[b]; });
        const size_t cap = (size_t)(nstart + ndev - 1) / ndev;
        for (int s : order) {
            int best = -1;
            for (int d = 0; d < ndev; d++)
                if (shard[d].size() < cap && (best < 0 || load[d] < load[best])) best = d;
            shard[best].push_back(s);
            load[best] += cost[s];
        }
    }
    for (int d = 0; d < ndev; d++) {
        workers.emplace_back([&, d]() {
            std::vector<ttsweep_start> my_starts;
            std::vector<float *> my_boxes;
            for (int s : shard[d]) {
                my_starts.push_back(starts[s]);
                my_boxes.push_back(tt_host[s]);
            }
            if (my_starts.empty()) return;
            ttsweep_ctx *ctx = ttsweep_create(devices[d], nx, ny, nz, fs, starstart, starstop);
            int r = ctx ? ttsweep_set_velocity(ctx, v_host) : -1;
            if (r == 0)
                r = ttsweep_solve(ctx, (int)my_starts.size(), my_starts.data(), my_boxes.data());
            if (r < 0) err[d] = ttsweep_last_error();       // thread-local text
            ttsweep_destroy(ctx);
            rc[d] = r;
        });
    }
    for (auto &w : workers) w.join();
    int any = 0;
    for (int d = 0; d < ndev; d++) {
        if (rc[d] < 0) return set_error("device %d: %s", devices[d], err[d].c_str());
        any |= rc[d];
    }
    return any;
}

int ttsweep_sweepXYZ(const float *v, float *tt, int nx, int ny, int nz, const ttsweep_fs *fs,
                     int starstart, int starstop, int si, int sj, int sk)
{
    ttsweep_ctx *ctx = ttsweep_create(0, nx, ny, nz, fs, starstart, starstop);
    if (!ctx) return -1;
    int rc = ttsweep_set_velocity(ctx, v);
    if (rc == 0) {
        ttsweep_start st = {si, sj, sk};
        float *boxes[1] = {tt};
        rc = ttsweep_solve(ctx, 1, &st, boxes);
    }
    ttsweep_destroy(ctx);
    return rc;
}

} // extern "C"
