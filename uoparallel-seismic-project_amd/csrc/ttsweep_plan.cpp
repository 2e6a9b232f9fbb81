// ttsweep_plan.cpp - what the host decides once per context or per solve (see ttsweep_ctx.h):
// padded layouts, the kernel that relaxes a star, the STRIP kernel's items and unit order,
// device pools.
#include "ttsweep_ctx.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <mutex>

namespace ttsweep {

// Counts the XCDs of the device (each has its own L2) by asking many workgroups where they
// run: the STRIP kernel keeps one unit queue per XCD.  Falls back to 1 queue on any doubt
// (queues are a locality device, never a correctness one).  Measured once per device and
// process (several contexts share the answer: multi-device solves, the pre-pass context, the
// warm-up thread of ttsweep_warmup).
static std::mutex g_xcd_mutex;
static int g_xcd_count[64];         // 0: not measured yet

static int measure_xcds(hipStream_t stream, int *out)
{
    unsigned *d_seen = nullptr, h_seen = 0;
    HIPCHK(hipMalloc((void **)&d_seen, sizeof(unsigned)));
    hipError_t e = hipMemsetAsync(d_seen, 0, sizeof(unsigned), stream);
    if (e == hipSuccess) e = launch_xcc_census(d_seen, 4096, stream);
    if (e == hipSuccess) e = hipMemcpyAsync(&h_seen, d_seen, sizeof(unsigned), hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    (void)hipFree(d_seen);
    if (e != hipSuccess) return set_error("XCD census failed: %s", hipGetErrorString(e));
    *out = std::min(std::max(__builtin_popcount(h_seen), 1), (int)UNITQ_LISTS);
    return 0;
}

int device_xcds(int device, hipStream_t stream, int *out)
{
    std::lock_guard<std::mutex> lock(g_xcd_mutex);
    if (device >= 0 && device < 64 && g_xcd_count[device] > 0) {
        *out = g_xcd_count[device];
        return 0;
    }
    if (measure_xcds(stream, out)) return -1;
    if (device >= 0 && device < 64) g_xcd_count[device] = *out;
    return 0;
}

int count_xcds(ttsweep_ctx *ctx)
{
    return device_xcds(ctx->device, ctx->stream, &ctx->nlists);
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
    ctx->plan_lat.ra = r[0];
    ctx->plan_lat.rb = L.lo[1];
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


// Can the STRIP kernel handle this star?  (plane and strip offsets within +-7)
static bool strip_supported(const ttsweep_ctx *ctx)
{
    return !ctx->pull.empty() && ctx->radius <= STRIP_MAX_RA && ctx->radius < STRIP_CF;
}

bool kernel_available(const ttsweep_ctx *ctx, int k)
{
    return k == TTSWEEP_KERNEL_CELL || (k == TTSWEEP_KERNEL_STRIP && strip_supported(ctx))
        || (k == TTSWEEP_KERNEL_TILE && tile_supported(ctx));
}

// Small stars: ordered tile sweeps; everything within +-7: LDS-staged unit relaxation;
// otherwise the per-cell kernel.
int auto_kernel(const ttsweep_ctx *ctx)
{
    return tile_supported(ctx) ? TTSWEEP_KERNEL_TILE
         : strip_supported(ctx) ? TTSWEEP_KERNEL_STRIP : TTSWEEP_KERNEL_CELL;
}

void make_layout(ttsweep_ctx *ctx)
{
    if (ctx->kernel == TTSWEEP_KERNEL_STRIP) make_layout_strip(ctx);
    else if (ctx->kernel == TTSWEEP_KERNEL_TILE) make_layout_tile(ctx);
    else make_layout_cell(ctx);
}

static int upload_strip_plan_np(ttsweep_ctx *ctx, int np, int ns, StripPlan &plan, StripItem *&d_items);

int upload_strip_plan(ttsweep_ctx *ctx)
{
    for (int np = 1; np <= STRIP_PLANES; np++)
        if (upload_strip_plan_np(ctx, np, STRIP_NS, ctx->plans[np - 1], ctx->d_strip_items[np - 1])) return -1;
    return upload_strip_plan_np(ctx, 1, STRIP_NS_LAT, ctx->plan_lat, ctx->d_strip_items_lat);
}

static int upload_strip_plan_np(ttsweep_ctx *ctx, int np, int ns, StripPlan &plan, StripItem *&d_items)
{
    const DevLayout &L = ctx->L;
    plan.np = np;
    plan.ns = ns;
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
        std::vector<StripItem> share[STRIP_NS_MAX];
        int load[STRIP_NS_MAX] = {};
        for (const auto &x : its) {
            int w = 0;
            for (int k = 1; k < ns; k++)
                if (load[k] < load[w]) w = k;
            share[w].push_back(x);
            load[w] += cost(x);
        }
        if (its.size() > 255) return set_error("star has too many columns per plane offset");
        plan.wsplit[p][0] = 0;
        for (int w = 0; w < STRIP_NS_MAX; w++) {
            if (w < ns) flat.insert(flat.end(), share[w].begin(), share[w].end());
            plan.wsplit[p][w + 1] = (unsigned char)(plan.wsplit[p][w] + (w < ns ? share[w].size() : 0));
        }
    }
    plan.first[plan.nstaged] = (int)flat.size();
    if (flat.size() > 0xffff) return set_error("star has too many columns");
    if (d_items) HIPCHK(hipFree(d_items));
    d_items = nullptr;
    if (!flat.empty()) {
        HIPCHK(hipMalloc((void **)&d_items, flat.size() * sizeof(StripItem)));
        HIPCHK(hipMemcpy(d_items, flat.data(), flat.size() * sizeof(StripItem), hipMemcpyHostToDevice));
    }
    return 0;
}

// Dead-edge box of one start (device axes, clipped, inclusive).
void fill_special_box(const ttsweep_ctx *ctx, StartDesc &sd)
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

int upload_star(ttsweep_ctx *ctx)
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
size_t flag_words(const DevLayout &L, int kernel)
{
    const size_t strip = kernel == TTSWEEP_KERNEL_STRIP
        ? 3 * (size_t)std::max(strip_units(L, 1), 1) + 4 : 0;      // (one-plane units: the larger grid)
    const size_t tile = 2 * (size_t)tile_count(L.n[0], TILE_X) * tile_count(L.n[1], TILE_Y) * tile_count(L.n[2], TILE_Z);
    return (std::max(strip, tile) + 31) & ~(size_t)31;     // (even: the TILE kernel views them as int2; whole lines)
}

// Device bytes ensure_capacity allocates per start (ttsweep_solve sizes its batches with it).
size_t per_start_device_bytes(const ttsweep_ctx *ctx)
{
    size_t b = (size_t)ctx->L.cells * sizeof(float)                 // padded travel-time volume
             + flag_words(ctx->L, ctx->kernel) * sizeof(int)                     // activity words
             + sizeof(StartDesc) + (2 + PASS_SLOTS) * sizeof(int) + 3 * sizeof(unsigned long long);
    if (ctx->kernel == TTSWEEP_KERNEL_TILE)
        b += (size_t)tile_face_cells(ctx->L, ctx->tile_fz) * sizeof(float);     // z faces
    if (ctx->kernel == TTSWEEP_KERNEL_STRIP)                        // static work list + unit queues
        b += (size_t)strip_units(ctx->L, 1) * (sizeof(int2) + sizeof(int4));
    return b;
}

int ensure_capacity(ttsweep_ctx *ctx, int nstart)
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
    HIPCHK(hipMalloc((void **)&ctx->d_changed, (PASS_SLOTS + 1) * nstart * sizeof(int)));     // (+ 1: the flush of deferred bits)
    HIPCHK(hipHostMalloc((void **)&ctx->h_starts, nstart * sizeof(StartDesc)));
    HIPCHK(hipHostMalloc((void **)&ctx->h_active, nstart * sizeof(int)));
    HIPCHK(hipHostMalloc((void **)&ctx->h_changed, (PASS_SLOTS + 1) * nstart * sizeof(int)));
    HIPCHK(hipMalloc((void **)&ctx->d_tile_flags,
                     (size_t)nstart * flag_words(ctx->L, ctx->kernel) * sizeof(int)));
    HIPCHK(hipMalloc((void **)&ctx->d_work, 3 * nstart * sizeof(unsigned long long)));
    HIPCHK(hipHostMalloc((void **)&ctx->h_work, 3 * nstart * sizeof(unsigned long long)));
    if (ctx->kernel == TTSWEEP_KERNEL_TILE)
        HIPCHK(hipMalloc((void **)&ctx->d_tface,
                         (size_t)nstart * tile_face_cells(ctx->L, ctx->tile_fz) * sizeof(float)));
    ctx->capacity_starts = nstart;
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
int ensure_unit_grid(ttsweep_ctx *ctx)
{
    if (ctx->unitq_blocks == 0) {
        hipDeviceProp_t prop;
        HIPCHK(hipGetDeviceProperties(&prop, ctx->device));
        const int wgs = units_wgs_per_cu() * std::max(prop.multiProcessorCount, 1);     // 2 workgroups (8 waves) per CU:
            // measured optimum - a third one adds no throughput, lengthens every unit and lets
            // fewer units see their neighbours' updates of the same pass
        ctx->unitq_blocks = ((wgs + ctx->nlists - 1) / ctx->nlists) * ctx->nlists;
        // (the latency instance: one eight-wave workgroup per CU)
        ctx->unitq_blocks_lat = ((std::max(prop.multiProcessorCount, 1) + ctx->nlists - 1) / ctx->nlists) * ctx->nlists;
    }
    return 0;
}

int build_worklist(ttsweep_ctx *ctx, int nactive)
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
    if (ensure_unit_grid(ctx)) return -1;
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
void order_units(const ttsweep_ctx *ctx, const StartDesc &sd, std::vector<int> &order)
{
    const DevLayout &L = ctx->L;
    const int btiles = strip_btiles(L), cstrips = strip_cstrips(L);
    const int np = ctx->np;
    const int nunits = strip_units(L, np);
    // (round 5, for the volumes that leave the caches: within shells of 4 ... 32 cells around the start the units
    // ordered by position - plane group, lane tile, strip - so that units handed out one after the other stage
    // overlapping windows: 512x512x256 x 8 291 -> 339 ... 362 ms, 6.8 -> 8.0 ... 8.6 sweep equivalents of work at the
    // same speed per relaxation - the order by distance is what keeps the work low, and the staging traffic is not what
    // limits that grid: profiles/r05_unit_order_shells.txt; removed)
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

} // namespace ttsweep
