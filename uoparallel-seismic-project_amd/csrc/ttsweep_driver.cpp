// ttsweep_driver.cpp - the driver loop of serial_new/sweep-tt-multistart.c:151-170 without the
// break (:168-169), i.e. old/sweep-serial/sweep-tt-multistart.c:189-211, on device-resident
// boxes: passes of the kernel variant in use, enqueued one ahead of the convergence test.
#include "ttsweep_ctx.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace ttsweep {

#ifdef TTSWEEP_DEBUG_ENV
static int strip_flag_words_host(const DevLayout &L) { return L.n[0] * strip_btiles(L) * strip_cstrips(L); }
#endif

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

// Squared radius (cells) of the distance gate for the pass about to be launched.
static float gate_r2(const ttsweep_ctx *ctx)
{
    if (ctx->gate_speed <= 0) return 3.0e38f;       // gate disabled
    const double r = ctx->gate_r0 + ctx->gate_speed * (double)ctx->pass_index;
    return (float)(r * r);
}

// ---- one pass of each kernel variant -----------------------------------------------------

// STRIP: plan_pass_kernel + sweep_units_kernel.  The pass's "changed" words arrive in
// h_changed_slot without a copy command, and d_changed_next is cleared for the pass after this
// one (UnitPassTail).
static int launch_pass_strip(ttsweep_ctx *ctx, int nactive, int nstart, int *d_changed, int *h_changed_slot,
                             int *d_changed_next)
{
    const StripPlan &plan = ctx->plans[ctx->np - 1];
    HIPCHK(launch_plan_pass(ctx->L, ctx->d_starts, ctx->d_worklist, ctx->worklist_len, d_changed,
                            ctx->d_unitq, (int)ctx->unitq_cap, ctx->nlists, ctx->d_unitq_ctrl,
                            plan, gate_r2(ctx), ctx->d_tile_flags, (long long)flag_words(ctx->L, ctx->kernel), ctx->stream));
    UnitPassTail tail;
    tail.active = ctx->d_active;
    tail.nactive = nactive;
    tail.entries = ctx->d_cell_entries;
    tail.nentries = ctx->n_cell_entries;
    tail.max_box_cells = (int)ctx->max_box_cells;
    tail.nstart = nstart;
    tail.changed_host = h_changed_slot;
    tail.changed_next = d_changed_next;
    tail.defer_margin = ctx->defer_suspended ? -3.0e38f : ctx->defer_margin;
    HIPCHK(launch_sweep_units(ctx->L, ctx->d_v, ctx->d_starts, ctx->d_unitq, (int)ctx->unitq_cap,
                              ctx->nlists, ctx->d_unitq_ctrl, ctx->unitq_blocks, d_changed,
                              ctx->d_strip_items[ctx->np - 1], plan, tail, ctx->stream));
    return 0;
}

// TILE: what stays the same for every launch of a solve (filled once per solve).
static int prepare_tile_sweep(ttsweep_ctx *ctx)
{
    TileSweep &P = ctx->tile_sweep;
    P = TileSweep{};
    P.L = ctx->L;
    P.v = ctx->d_v;
    P.starts = ctx->d_starts;
    P.active = ctx->d_active;
    P.NI = tile_count(ctx->L.n[0], TILE_X);
    P.NJ = tile_count(ctx->L.n[1], TILE_Y);
    P.NK = tile_count(ctx->L.n[2], TILE_Z);
    P.R = ctx->tile_R;
    P.nent = ctx->tile_nent;
    P.T0 = ctx->d_T;
    P.state0 = ctx->d_tile_flags;
    P.state_stride = (long long)flag_words(ctx->L, ctx->kernel);
    P.work0 = ctx->d_work;
    P.fz = ctx->tile_fz;
    P.vface = ctx->d_vface;
    P.tface = ctx->d_tface;
    P.face_cells = tile_face_cells(ctx->L, ctx->tile_fz);
    P.sx = P.sy = P.sz = 1;
    P.nblocks = 1;
    for (int e = 0; e < TILE_MAX_ENT; e++) P.ent[e] = ctx->tile_ent[e];
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
    // the workgroups' private work sums: [tile_blocks][nstart][2], zero between solves
    P.nblocks = ctx->tile_blocks;
    P.nxcd = std::max(ctx->nlists, 1);
    if (ctx->tile_blocks % P.nxcd) P.nxcd = 1;
    P.nstart = ctx->stats.nstart;
    const size_t need = (size_t)ctx->tile_blocks * (size_t)P.nstart * 2;
    if (need > ctx->tile_wgwork_cap) {
        if (ctx->d_tile_wgwork) HIPCHK(hipFree(ctx->d_tile_wgwork));
        ctx->d_tile_wgwork = nullptr;
        ctx->tile_wgwork_cap = 0;
        HIPCHK(hipMalloc((void **)&ctx->d_tile_wgwork, need * sizeof(unsigned long long)));
        ctx->tile_wgwork_cap = need;
    }
    HIPCHK(hipMemsetAsync(ctx->d_tile_wgwork, 0, need * sizeof(unsigned long long), ctx->stream));
    P.wgwork = ctx->d_tile_wgwork;
    if (!ctx->d_tile_dmin) HIPCHK(hipMalloc((void **)&ctx->d_tile_dmin, 2 * sizeof(int)));
    if (!ctx->h_tile_dmin) HIPCHK(hipHostMalloc((void **)&ctx->h_tile_dmin, 2 * sizeof(int)));
    ctx->h_tile_dmin[0] = ctx->h_tile_dmin[1] = 0;
    return 0;
}

// TILE: one ordering sweep = the tile hyperplanes in stream order, one launch each (the
// launch finds its due tiles itself: ttsweep_tile.hip, tile_candidate).
static int launch_pass_tile(ttsweep_ctx *ctx, int nactive, int *d_changed)
{
    TileSweep &P = ctx->tile_sweep;
    P.changed = d_changed;
    P.nactive = nactive;
    // the resident grid (no more workgroups than candidates; whole XCD rounds)
    P.nblocks = (int)std::min<long long>(ctx->tile_blocks, (long long)P.NJ * P.NK * nactive);
    P.nblocks = std::max(P.nblocks / P.nxcd, 1) * P.nxcd;
    // workgroups per XCD that take candidates: coprime to the number of active starts, so that
    // one start's tiles at consecutive positions go to different workgroups (tile_candidate)
    auto gcd = [](int a, int b) { while (b) { const int t = a % b; a = b; b = t; } return a; };
    P.wstride = P.nblocks / P.nxcd;
    while (P.wstride > 1 && gcd(P.wstride, nactive) != 1) P.wstride--;
    const int o = ctx->pass_index & 7, on = (ctx->pass_index + 1) & 7;     // the eight orderings in turn
    P.sx = (o & 1) ? -1 : 1;
    P.sy = (o & 2) ? -1 : 1;
    P.sz = (o & 4) ? -1 : 1;
    P.nsx = (on & 1) ? -1 : 1;
    P.nsy = (on & 2) ? -1 : 1;
    P.nsz = (on & 4) ? -1 : 1;
    // Hyperplanes in front of the first one that can hold a due tile are not launched: the
    // previous sweep recorded, for THIS ordering, the earliest hyperplane next to a tile it
    // improved (tile_next_plane; its word is on the host: TILE sweeps are enqueued one at a
    // time, each after the previous one's words have arrived).  The first sweep of a solve
    // starts at 0; a sweep that finds the word untouched has nothing to do at all.
    const int nsteps = P.NI + P.NJ + P.NK - 2;
    const int slot = ctx->pass_index & 1;
    const int first = ctx->pass_index == 0 ? 0 : std::min(ctx->h_tile_dmin[slot ^ 1], nsteps);
    P.dmin_next = ctx->d_tile_dmin + slot;
    HIPCHK(hipMemsetAsync(P.dmin_next, 0x7f, sizeof(int), ctx->stream));
    for (int D = first; D < nsteps; D++) {
        P.D = D;
        P.epoch = ++ctx->tile_epoch;
        HIPCHK(launch_tile_sweep(P, ctx->stream));
    }
    HIPCHK(hipMemcpyAsync(ctx->h_tile_dmin + slot, P.dmin_next, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    ctx->stats.launches += nsteps - first - 1;
    return 0;
}

// One full-grid pass for the active starts.
static int launch_pass(ttsweep_ctx *ctx, int nactive, int nstart, int *d_changed, int *h_changed_slot,
                       int *d_changed_next)
{
    hipEvent_t e0, e1;
    if (ctx->timing && timed_event(ctx, &e0)) return -1;
    if (ctx->kernel == TTSWEEP_KERNEL_STRIP) {
        if (launch_pass_strip(ctx, nactive, nstart, d_changed, h_changed_slot, d_changed_next)) return -1;
    } else if (ctx->kernel == TTSWEEP_KERNEL_TILE) {
        if (launch_pass_tile(ctx, nactive, d_changed)) return -1;
    } else {
        HIPCHK(launch_sweep_cell(ctx->L, ctx->d_v, ctx->d_starts, ctx->d_active, nactive,
                                 d_changed, ctx->d_cell_entries, ctx->n_cell_entries, ctx->exact_half,
                                 ctx->stream));
    }
    if (ctx->timing && timed_event(ctx, &e1)) return -1;
    ctx->stats.launches++;
    ctx->pass_index++;
    return 0;
}

// STRIP, one launch per solve (AsyncSolve, ttsweep_dev.h): can this solve run that way, and should it?
static bool use_async(ttsweep_ctx *ctx, int nstart)
{
    if (ctx->kernel != TTSWEEP_KERNEL_STRIP || ctx->async_mode == 0) return false;
    if (nstart > ASYNC_MAX_STARTS || strip_units(ctx->L, ctx->np) >= (int)ASYNC_UNIT_SPECIAL) return false;
    // (planners take up to ASYNC_MAX_RINGS workgroups of the resident grid: a device that holds only a handful
    // keeps the pass driver)
    if (ensure_unit_grid(ctx) || ctx->unitq_blocks < 4 * ASYNC_MAX_RINGS) return false;
    // (a ring serves at most ASYNC_RING_STARTS starts, and there are as many rings as XCDs: a device or a partition
    // with fewer XCDs holds fewer starts per launch - the pass driver takes what does not fit)
    const int nrings = std::max(std::min(std::min(nstart, ctx->nlists), (int)ASYNC_MAX_RINGS), 1);
    if ((nstart + nrings - 1) / nrings > ASYNC_RING_STARTS) return false;
    return true;        // (-1: wherever it can run; 1: the same)
}

// The whole driver loop as one launch: the rings' lists (every unit of a ring's starts, nearest
// first, the starts interleaved rank by rank), the ring memory, the launch, its verdict.
// sweeps[s]: whole-grid equivalents of units relaxed for start s (what a pass count is for the
// pass driver).  Returns 1 / 0 (something improved / nothing did) or < 0.
static int solve_async_strip(ttsweep_ctx *ctx, int nstart, std::vector<int> &sweeps)
{
    const int nunits = strip_units(ctx->L, ctx->np);
    if (ensure_unit_grid(ctx)) return -1;
    // The latency instance (one-plane units relaxed by all eight waves of a CU, ttsweep_kernels.hip): for solves that
    // offer too few units to fill the machine anyway - what bounds them is how long a hop of the front takes.
    const bool lat = ctx->np == 1 && ctx->unitq_blocks_lat >= 4 * ASYNC_MAX_RINGS
        && (ctx->async_waves >= 0 ? ctx->async_waves == STRIP_NS_LAT
                                  : (long long)nstart * strip_units(ctx->L, 1) < ctx->lat_max_units);
    const StripPlan &plan = lat ? ctx->plan_lat : ctx->plans[ctx->np - 1];
    const StripItem *const d_items = lat ? ctx->d_strip_items_lat : ctx->d_strip_items[ctx->np - 1];
    const int nblocks = lat ? ctx->unitq_blocks_lat : ctx->unitq_blocks;
    AsyncSolve as{};
    as.nrings = std::min(std::min(nstart, ctx->nlists), (int)ASYNC_MAX_RINGS);
    const int cap = 1 << 14;
    as.cap_mask = cap - 1;
    const int workers = std::max((nblocks - as.nrings) / as.nrings, 1);
    // fill marks of a ring: half / twice its workers where the units of a ring are a few thousand (241 x 241 x 51 x 24:
    // 5.8 k per ring - fuller rings relax MORE there: 27.9 / 28.3 / 29.3 / 31.1 ms with 126 / 256 / 512 / 1024); on the
    // grids that leave the caches the front is thousands of units wide and a fuller ring is less work and less time
    // (512 x 512 x 256 x 8, 33 k units per ring: 288 / 276 / 273 / 269 / 264 / 258 ms with 126 / 256 / 512 / 1024 / 2048 /
    // 4096; 1024 x 1024 x 512 x 14: 3.95 -> 3.83 s with 1024) - profiles/r05_big_grid_ring_marks.txt
    const long long units_per_ring = (long long)nstart * nunits / std::max(as.nrings, 1);
    const int high0 = units_per_ring >= 16384 ? std::min(4096, cap / 4) : 2 * workers;
    as.low = ctx->async_low > 0 ? ctx->async_low : std::max(units_per_ring >= 16384 ? high0 / 4 : workers / 2, 8);
    as.high = ctx->async_high > 0 ? ctx->async_high : high0;
    as.high = std::max(as.high, as.low + 1);
    as.special_every = ctx->async_special_every;
    // lists
    as.policy = ctx->async_policy;
    as.gate_r0 = (float)ctx->gate_r0;
    as.gate_speed = ctx->gate_speed > 0 ? ctx->async_gate_speed : 0.f;     // (per start: only for boxes that grow from one unit)
    as.gate_fast = std::max(ctx->async_gate_fast, as.gate_speed);
    as.window = ctx->async_window;
    const int btiles = strip_btiles(ctx->L), cstrips = strip_cstrips(ctx->L);
    auto gate_d2 = [&](int s, int unit) {       // squared distance from the start to the unit's cells (plan_pass_kernel)
        const StartDesc &sd = ctx->h_starts[s];
        const int cs = unit % cstrips, bt = (unit / cstrips) % btiles, a0 = ctx->np * (unit / (cstrips * btiles));
        const int b0 = bt * STRIP_TB, cb0 = cs * STRIP_K, tb_eff = std::min((int)STRIP_TB, ctx->L.n[1]);
        const float da = (float)std::max(std::max(a0 - sd.sa, sd.sa - (a0 + ctx->np - 1)), 0);
        const float db = (float)std::max(std::max(b0 - sd.sb, sd.sb - (b0 + tb_eff - 1)), 0);
        const float dc = (float)std::max(std::max(cb0 - sd.sc, sd.sc - (cb0 + STRIP_K - 1)), 0);
        const float d2 = da * da + db * db + dc * dc;
        int bits;
        memcpy(&bits, &d2, sizeof bits);
        return bits;
    };
    // (the lists depend on the start cells and the unit grid only: a solve of the same starts keeps them)
    std::vector<long long> key;
    key.reserve(nstart + 2);
    key.push_back(ctx->np);
    key.push_back(as.nrings);
    for (int s = 0; s < nstart; s++) key.push_back(ctx->h_starts[s].sidx);
    const bool cached = key == ctx->async_list_key && ctx->d_async_list;
    std::vector<int4> flat;
    if (!cached) flat.reserve((size_t)nstart * nunits);
    std::vector<int> ring_starts;
    if (cached) {
        for (int r = 0; r < ASYNC_MAX_RINGS; r++) { as.ring_off[r] = ctx->async_rings.ring_off[r]; as.ring_len[r] = ctx->async_rings.ring_len[r]; }
        for (int r = 0; r <= ASYNC_MAX_RINGS; r++) as.ring_start_off[r] = ctx->async_rings.ring_start_off[r];
    }
    for (int r = 0; r < as.nrings && !cached; r++) {
        as.ring_off[r] = (int)flat.size();
        as.ring_start_off[r] = (int)ring_starts.size();
        std::vector<int> mine;
        for (int a = r; a < nstart; a += as.nrings) mine.push_back(a);
        if ((int)mine.size() > ASYNC_RING_STARTS) return set_error("too many starts for a one-launch solve");
        for (int a : mine) ring_starts.push_back(a);
        for (int k = 0; k < nunits; k++)
            for (size_t i = 0; i < mine.size(); i++)
                flat.push_back(make_int4(mine[i] | ((int)i << 16), ctx->unit_order[mine[i]][k],
                                         gate_d2(mine[i], ctx->unit_order[mine[i]][k]), 0));
        as.ring_len[r] = (int)flat.size() - as.ring_off[r];
    }
    if (!cached) as.ring_start_off[as.nrings] = (int)ring_starts.size();
    as.scan_slack = 16384 * ((nstart + as.nrings - 1) / as.nrings);
    // a unit that improved is relaxed again against its own planes, at once (measured: 24 starts 34.5 -> 29.4 ms with
    // two such passes - the planner hands out a third less -, 3 starts 7.85 -> 7.77, one start 6.38 -> 6.52:
    // profiles/r04_inunit.txt)
    // (the latency instance: none - its units come round again faster than a pass over their own planes pays:
    // 3 starts 7.24 -> 7.13 ms, 1 start 5.16 -> 5.02, profiles/r05_lat_8shares_g3.txt)
    as.inunit = ctx->async_inunit >= 0 ? ctx->async_inunit : (lat ? 0 : (nstart >= 2 ? 2 : 0));
    // Direct hand-off (ttsweep_dev.h, ASYNC_HANDOFF_*): for solves that cannot fill the machine - what bounds them is
    // how fast good values travel from unit to unit, and every hop through the planner's scan costs a round of it.
    // Every unit of a ring can sit in it at once (taken by a worker, not yet claimed): the rings must hold that.
    {
        long long longest_ring = 0;
        for (int r = 0; r < as.nrings; r++) longest_ring = std::max<long long>(longest_ring, as.ring_len[r]);
        const bool can = (as.policy == 1 || as.policy == 0) && longest_ring + 2 * ASYNC_RING_STARTS <= cap / 2;
        const int want = ctx->async_handoff >= 0 ? ctx->async_handoff
                       : ((long long)nstart * strip_units(ctx->L, 1) < ctx->handoff_max_units ? ctx->handoff_default : 0);
        as.handoff = can ? want : 0;
    }
    if (!cached && flat.size() > ctx->async_list_cap) {
        if (ctx->d_async_list) HIPCHK(hipFree(ctx->d_async_list));
        ctx->d_async_list = nullptr;
        ctx->async_list_cap = 0;
        HIPCHK(hipMalloc((void **)&ctx->d_async_list, flat.size() * sizeof(int4)));
        ctx->async_list_cap = flat.size();
    }
    if (!ctx->d_async_ring_starts) HIPCHK(hipMalloc((void **)&ctx->d_async_ring_starts, ASYNC_MAX_STARTS * sizeof(int)));
    if (!ctx->d_async_entries) HIPCHK(hipMalloc((void **)&ctx->d_async_entries, (size_t)ASYNC_MAX_RINGS * cap * sizeof(unsigned long long)));
    if (!ctx->d_async_ctl) HIPCHK(hipMalloc((void **)&ctx->d_async_ctl, (size_t)ASYNC_MAX_RINGS * ASYNC_CTL_STRIDE * sizeof(unsigned long long)));
    if (!ctx->d_async_status) HIPCHK(hipMalloc((void **)&ctx->d_async_status, 8 * sizeof(unsigned)));
    if (!ctx->h_async_status) HIPCHK(hipHostMalloc((void **)&ctx->h_async_status, 8 * sizeof(unsigned)));
    if (!cached) {
        ctx->async_list_key.clear();        // (until the upload below has completed)
        HIPCHK(hipMemcpyAsync(ctx->d_async_list, flat.data(), flat.size() * sizeof(int4), hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(hipMemcpyAsync(ctx->d_async_ring_starts, ring_starts.data(), ring_starts.size() * sizeof(int),
                              hipMemcpyHostToDevice, ctx->stream));
    }
    HIPCHK(hipMemsetAsync(ctx->d_async_entries, 0, (size_t)ASYNC_MAX_RINGS * cap * sizeof(unsigned long long), ctx->stream));
    HIPCHK(hipMemsetAsync(ctx->d_async_ctl, 0, (size_t)ASYNC_MAX_RINGS * ASYNC_CTL_STRIDE * sizeof(unsigned long long), ctx->stream));
    HIPCHK(hipMemsetAsync(ctx->d_async_status, 0, 8 * sizeof(unsigned), ctx->stream));
    HIPCHK(hipMemsetAsync(ctx->d_changed, 0, (size_t)nstart * sizeof(int), ctx->stream));
    if (!cached) {
        HIPCHK(hipStreamSynchronize(ctx->stream));  // (`flat` is a stack-lifetime buffer)
        ctx->async_list_key = key;
        ctx->async_rings = as;
    }
    as.list = ctx->d_async_list;
    as.ring_starts = ctx->d_async_ring_starts;
    as.entries = ctx->d_async_entries;
    as.ctl = ctx->d_async_ctl;
    as.status = ctx->d_async_status;
    // every wait inside the launch gives up after this much wall clock (100 MHz ticks): ten seconds plus
    // twenty times what the solve should take at a tenth of the machine's rate (1024x1024x512 x 14 with
    // the 818-offset star: 5.6 s measured, limit about 130 s) - a protocol error must not hang the device
    const double expect_s = (double)ctx->relax_per_sweep * (double)nstart * 8.0 / 1.0e12;
    as.timeout_ticks = (long long)((10.0 + 20.0 * expect_s) * 1.0e8);
    if (ctx->async_timeout_ms > 0) as.timeout_ticks = (long long)ctx->async_timeout_ms * 100000ll;
    long long longest = 0;
    for (int r = 0; r < as.nrings; r++) longest = std::max<long long>(longest, as.ring_len[r]);
    as.max_entries = (long long)std::min<double>((double)ctx->max_sweeps * (double)longest, 2.0e9);

    UnitPassTail tail{};
    tail.entries = ctx->d_cell_entries;
    tail.nentries = ctx->n_cell_entries;
    tail.max_box_cells = (int)ctx->max_box_cells;
    tail.nstart = nstart;
    tail.defer_margin = ctx->defer_margin;
    hipEvent_t e0, e1;
    if (ctx->timing && timed_event(ctx, &e0)) return -1;
    HIPCHK(launch_solve_units(ctx->L, ctx->d_v, ctx->d_starts, nblocks, ctx->d_changed,
                              d_items, plan, lat ? STRIP_NS_LAT : STRIP_NS, tail, as, ctx->d_tile_flags,
                              (long long)flag_words(ctx->L, ctx->kernel), ctx->stream));
    if (ctx->timing && timed_event(ctx, &e1)) return -1;
    ctx->stats.launches++;
    HIPCHK(hipMemcpyAsync(ctx->h_async_status, ctx->d_async_status, 8 * sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(ctx->h_changed, ctx->d_changed, (size_t)nstart * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(ctx->h_work, ctx->d_work, 3 * nstart * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
#ifdef TTSWEEP_ASYNC_STATS
    fprintf(stderr, "one-launch solve: %u units (%u staged planes) relaxed, %u units (%u planes) improved nothing\n",
            ctx->h_async_status[4], ctx->h_async_status[5], ctx->h_async_status[6], ctx->h_async_status[7]);
#endif
    if (ctx->h_async_status[0] == ASYNC_ERR_CAP)
        return set_error("a start did not converge in %lld sweeps", ctx->max_sweeps);
    if (ctx->h_async_status[0] != ASYNC_OK) {
#ifdef TTSWEEP_DEBUG_ENV
        {   // what the rings and the activity words looked like when the launch gave up
            std::vector<unsigned long long> ctl((size_t)ASYNC_MAX_RINGS * ASYNC_CTL_STRIDE);
            (void)hipMemcpy(ctl.data(), ctx->d_async_ctl, ctl.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
            for (int r = 0; r < as.nrings; r++) {
                const unsigned long long x = ctl[(size_t)r * ASYNC_CTL_STRIDE];
                float g;
                const unsigned gb = (unsigned)ctl[(size_t)r * ASYNC_CTL_STRIDE + 8];
                memcpy(&g, &gb, sizeof g);
                fprintf(stderr, "  ring %d: head %u tail %u done %d completed %u gate_r2 %g | stored by the planner %llu, handed off %llu (reserved %llu), consumed %llu, reserved by the planner %llu (negative %llu)\n", r, (unsigned)x,
                        (unsigned)(x >> 32) & 0x7fffffffu, (int)(x >> 63), (unsigned)ctl[(size_t)r * ASYNC_CTL_STRIDE + 16], g,
                        ctl[(size_t)r * ASYNC_CTL_STRIDE + 24], ctl[(size_t)r * ASYNC_CTL_STRIDE + 25], ctl[(size_t)r * ASYNC_CTL_STRIDE + 27],
                        ctl[(size_t)r * ASYNC_CTL_STRIDE + 26], ctl[(size_t)r * ASYNC_CTL_STRIDE + 28], ctl[(size_t)r * ASYNC_CTL_STRIDE + 29]);
            }
            const size_t fw = flag_words(ctx->L, ctx->kernel);
            const int nflag = strip_flag_words_host(ctx->L);
            std::vector<int> fl(fw);
            for (int s = 0; s < nstart; s++) {
                (void)hipMemcpy(fl.data(), ctx->d_tile_flags + (size_t)s * fw, fw * sizeof(int), hipMemcpyDeviceToHost);
                int busy = 0, bits = 0, defer = 0, first_busy = -1, first_bits = -1;
                for (int u = 0; u < nunits; u++) {
                    const unsigned w = (unsigned)fl[2 * nflag + u];
                    if (w & ASYNC_BUSY) { busy++; if (first_busy < 0) first_busy = u; }
                    if (w & ~ASYNC_BUSY) { bits++; if (first_bits < 0) first_bits = u; }
                    if (fl[u]) defer++;
                }
                fprintf(stderr, "  start %d: %d units busy (first %d), %d with plane bits (first %d, word %08x), %d with deferred bits\n",
                        s, busy, first_busy, bits, first_bits, first_bits >= 0 ? (unsigned)fl[2 * nflag + first_bits] : 0u, defer);
            }
            std::vector<unsigned long long> ent((size_t)(as.cap_mask + 1));
            for (int r = 0; r < as.nrings; r++) {
                (void)hipMemcpy(ent.data(), ctx->d_async_entries + (size_t)r * (as.cap_mask + 1), ent.size() * 8, hipMemcpyDeviceToHost);
                int full = 0;
                for (size_t i = 0; i < ent.size(); i++) {
                    const unsigned long long e = ent[i];
                    if (e == 0ull) continue;
                    full++;
                    fprintf(stderr, "  ring %d slot %zu: planes %04x unit %u start %u tag %u valid %d nodefer %d\n", r, i,
                            (unsigned)(e & 0xffffu), (unsigned)((e >> 16) & 0xfffffu), (unsigned)((e >> 36) & 0xffu),
                            (unsigned)((e >> 44) & ASYNC_TAG_MASK), (int)((e >> 62) & 1), (int)(e >> 63));
                }
                fprintf(stderr, "  ring %d: %d slots hold an entry\n", r, full);
            }
        }
#endif
        set_error("the one-launch solve gave up (code %u)", ctx->h_async_status[0]);
        return -2;      // (the boxes hold valid upper bounds: the caller goes on with the pass driver)
    }
    bool any = false;
    for (int s = 0; s < nstart; s++) {
        any |= (ctx->h_changed[s] & CHANGED_IMPROVED) != 0;
        sweeps[s] = (int)((ctx->h_work[3 * s + 2] + (unsigned long long)nunits - 1) / (unsigned long long)nunits);
    }
    return any ? 1 : 0;
}


// the caller's FLOATBOX array as a layout without halo (include/floatbox.h:127-129: x * ny * nz + y * nz + z)
static DevLayout user_layout(const DevLayout &L)
{
    DevLayout U = L;
    for (int d = 0; d < 3; d++) { U.lo[d] = 0; U.n[d] = U.p[d] = L.un[d]; U.perm[d] = d; }
    U.s1 = U.p[2];
    U.s0 = (long long)U.p[1] * U.p[2];
    U.cells = U.s0 * U.p[0];
    return U;
}

// TILE, plain 6-neighbour star: can the solve run as ONE launch of column pipelines (ColumnSolve, ttsweep_dev.h)?
static bool use_column(ttsweep_ctx *ctx, int nstart)
{
    if (ctx->kernel != TTSWEEP_KERNEL_TILE || ctx->async_mode == 0) return false;
    if (!tile_star_is_six(ctx->tile_ent, ctx->tile_nent, ctx->tile_R)) return false;
    // (the two z entries have one length - both stand for the same two star entries -: the column kernel carries the
    // z edge of a cell to the next step, where it is the edge back)
    if (ctx->tile_ent[2].h != ctx->tile_ent[3].h) return false;
    const DevLayout &L = ctx->L;
    const int NI = tile_count(L.n[0], TILE_X), NJ = tile_count(L.n[1], TILE_Y), NK = tile_count(L.n[2], TILE_Z);
    if (NK > COL_MAX_NK || NI > 32767 || NJ > 32767 || nstart > 32767) return false;
    // (the staging instructions address a column's rows with 32-bit byte offsets)
    if ((9 * L.s0 + 9 * L.s1) * 4 + 64 >= 0x7fffffffLL) return false;
    if (ctx->col_blocks == 0) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, ctx->device) != hipSuccess) return false;
        int per_cu = 0;
        if (column_solve_wgs_per_cu(&per_cu) != hipSuccess) return false;
        ctx->col_blocks = per_cu * std::max(prop.multiProcessorCount, 1);
    }
    return ctx->col_blocks >= COL_SEQS;
}

// The whole driver loop as one launch: claim sequences, first state, the launch, its verdict.  sweeps[s]: ordering
// sweeps start s took part in.  Returns 1 / 0 (something improved / nothing did), -2 (a wait inside the launch ran
// into its wall-clock limit: the boxes hold valid upper bounds, the caller goes on with the hyperplane launches) or < 0.
// Can the column driver relax the travel times in the caller's own arrays (no padded copy, no copy back)?  Their rows
// have to be whole tiles long and 64-byte aligned.
static bool column_in_place(const ttsweep_ctx *ctx, int nstart, float *const *tt_dev)
{
    const DevLayout &L = ctx->L;
    if (ctx->col_in_place_off || L.n[2] % TILE_Z) return false;
    if ((9ll * L.n[1] * L.n[2] + 9ll * L.n[2]) * 4 + 64 >= 0x7fffffffLL) return false;
    for (int s = 0; s < nstart; s++)
        if (!tt_dev[s] || (reinterpret_cast<uintptr_t>(tt_dev[s]) & 63u)) return false;
    return true;
}

static int solve_column(ttsweep_ctx *ctx, int nstart, bool from_box, float *const *tt_dev, bool in_place, std::vector<int> &sweeps)
{
    const DevLayout &L = ctx->L;
    ColumnSolve &C = ctx->col;
    C = ColumnSolve{};
    C.L = L;
    C.v = ctx->d_v;
    C.nstart = nstart;
    C.NI = tile_count(L.n[0], TILE_X);
    C.NJ = tile_count(L.n[1], TILE_Y);
    C.NK = tile_count(L.n[2], TILE_Z);
    const int ncol = C.NI * C.NJ;
    C.nseq = std::min((int)COL_SEQS, C.NI);
    if (ctx->nlists >= 1 && ctx->nlists < C.nseq) C.nseq = ctx->nlists;
    // sequence x: the positions (I', J') with I' % nseq == x in the order of the sweep: by level I' + J', then by I'
    std::vector<int> tab;
    tab.reserve(ncol);
    for (int x = 0; x < C.nseq; x++) {
        C.seq_off[x] = (int)tab.size();
        for (int lev = 0; lev <= C.NI + C.NJ - 2; lev++)
            for (int ip = x; ip < C.NI; ip += C.nseq) {
                const int jp = lev - ip;
                if (jp >= 0 && jp < C.NJ) tab.push_back(ip | (jp << 16));
            }
        C.seq_len[x] = (int)tab.size() - C.seq_off[x];
    }
    if (ctx->col_seq_key[0] != C.NI || ctx->col_seq_key[1] != C.NJ || ctx->col_seq_key[2] != C.nseq || !ctx->d_col_seqtab) {
        if (ctx->d_col_seqtab) HIPCHK(hipFree(ctx->d_col_seqtab));
        ctx->d_col_seqtab = nullptr;
        HIPCHK(hipMalloc((void **)&ctx->d_col_seqtab, tab.size() * sizeof(int)));
        HIPCHK(hipMemcpy(ctx->d_col_seqtab, tab.data(), tab.size() * sizeof(int), hipMemcpyHostToDevice));
        ctx->col_seq_key[0] = C.NI; ctx->col_seq_key[1] = C.NJ; ctx->col_seq_key[2] = C.nseq;
    }
    if (nstart > ctx->col_cap_starts) {
        if (ctx->d_col_prog) HIPCHK(hipFree(ctx->d_col_prog));
        if (ctx->d_col_due) HIPCHK(hipFree(ctx->d_col_due));
        if (ctx->d_col_done) HIPCHK(hipFree(ctx->d_col_done));
        if (ctx->h_col_done) HIPCHK(hipHostFree(ctx->h_col_done));
        ctx->d_col_prog = nullptr; ctx->d_col_due = nullptr; ctx->d_col_done = nullptr;
        ctx->h_col_done = nullptr;
        ctx->col_cap_starts = 0;
        HIPCHK(hipMalloc((void **)&ctx->d_col_prog, (size_t)2 * nstart * ncol * sizeof(unsigned long long)));    // (two buffers, by sweep parity)
        HIPCHK(hipMalloc((void **)&ctx->d_col_due, (size_t)nstart * ncol * sizeof(unsigned)));
        HIPCHK(hipMalloc((void **)&ctx->d_col_done, (size_t)nstart * sizeof(int)));
        HIPCHK(hipHostMalloc((void **)&ctx->h_col_done, (size_t)nstart * sizeof(int)));
        ctx->col_cap_starts = nstart;
    }
    if (nstart > ctx->col_cap_tptr) {
        if (ctx->d_col_tptr) HIPCHK(hipFree(ctx->d_col_tptr));
        if (ctx->h_col_tptr) HIPCHK(hipHostFree(ctx->h_col_tptr));
        if (ctx->d_col_ordseq) HIPCHK(hipFree(ctx->d_col_ordseq));
        if (ctx->h_col_ordseq) HIPCHK(hipHostFree(ctx->h_col_ordseq));
        if (ctx->d_col_line) HIPCHK(hipFree(ctx->d_col_line));
        if (ctx->h_col_line) HIPCHK(hipHostFree(ctx->h_col_line));
        ctx->d_col_line = nullptr; ctx->h_col_line = nullptr;
        ctx->d_col_tptr = nullptr; ctx->h_col_tptr = nullptr;
        ctx->d_col_ordseq = nullptr; ctx->h_col_ordseq = nullptr;
        ctx->col_cap_tptr = 0;
        HIPCHK(hipMalloc((void **)&ctx->d_col_tptr, (size_t)nstart * sizeof(float *)));
        HIPCHK(hipHostMalloc((void **)&ctx->h_col_tptr, (size_t)nstart * sizeof(float *)));
        HIPCHK(hipMalloc((void **)&ctx->d_col_ordseq, (size_t)nstart * sizeof(unsigned long long)));
        HIPCHK(hipHostMalloc((void **)&ctx->h_col_ordseq, (size_t)nstart * sizeof(unsigned long long)));
        HIPCHK(hipMalloc((void **)&ctx->d_col_line, (size_t)nstart * 3 * sizeof(float)));
        HIPCHK(hipHostMalloc((void **)&ctx->h_col_line, (size_t)nstart * 3 * sizeof(float)));
        ctx->col_cap_tptr = nstart;
    }
    for (int s = 0; s < nstart; s++) ctx->h_col_tptr[s] = in_place ? tt_dev[s] : ctx->d_T + (size_t)s * L.cells;
    HIPCHK(hipMemcpyAsync(ctx->d_col_tptr, ctx->h_col_tptr, (size_t)nstart * sizeof(float *), hipMemcpyHostToDevice, ctx->stream));
    C.tptr = ctx->d_col_tptr;
    C.ts0 = in_place ? (long long)L.n[1] * L.n[2] : L.s0;
    C.ts1 = in_place ? (long long)L.n[2] : L.s1;
    C.tpad = in_place ? 0 : 1;
    C.tlo = in_place ? 0 : L.lo[2];
    if (!ctx->d_col_claim) HIPCHK(hipMalloc((void **)&ctx->d_col_claim, COL_SEQS * 16 * sizeof(unsigned long long)));
    if (!ctx->d_col_status) HIPCHK(hipMalloc((void **)&ctx->d_col_status, 8 * sizeof(unsigned)));
    if (!ctx->h_col_status) HIPCHK(hipHostMalloc((void **)&ctx->h_col_status, 8 * sizeof(unsigned)));
    // the resident grid: no more workgroups than a sweep has columns (a workgroup that finds nothing to do waits
    // inside a later sweep), whole rounds of the sequences
    const int wgw = column_solve_wg_waves();
    long long blocks = std::min<long long>(ctx->col_blocks, ((long long)ncol * nstart + wgw - 1) / wgw);
    blocks = std::max<long long>(blocks / C.nseq, 1) * C.nseq;
    const size_t need = (size_t)blocks * wgw * (size_t)nstart * 2;
    if (need > ctx->tile_wgwork_cap) {
        if (ctx->d_tile_wgwork) HIPCHK(hipFree(ctx->d_tile_wgwork));
        ctx->d_tile_wgwork = nullptr;
        ctx->tile_wgwork_cap = 0;
        HIPCHK(hipMalloc((void **)&ctx->d_tile_wgwork, need * sizeof(unsigned long long)));
        ctx->tile_wgwork_cap = need;
    }
    HIPCHK(hipMemsetAsync(ctx->d_tile_wgwork, 0, need * sizeof(unsigned long long), ctx->stream));
    C.seqtab = ctx->d_col_seqtab;
    C.prog = ctx->d_col_prog;
    C.due = ctx->d_col_due;
    C.done = ctx->d_col_done;
    C.claim = ctx->d_col_claim;
    C.status = ctx->d_col_status;
    C.wgwork = ctx->d_tile_wgwork;
    C.changed = ctx->d_changed;
    static const int order[6] = {0, 1, 2, 3, 4, 5};     // (tile_star_is_six: x-, y-, z-, z+, y+, x+)
    for (int e = 0; e < 6; e++) C.h[e] = ctx->tile_ent[order[e]].h;
    C.max_sweeps = (int)std::min<long long>(ctx->max_sweeps, COL_MAX_SWEEPS - 2);
    if (ctx->col_order < 0) {       // the default: chosen per start from the velocity profile of its vertical line
        HIPCHK(launch_column_line(ctx->d_v, L, ctx->d_starts, nstart, ctx->d_col_line, ctx->stream));
        HIPCHK(hipMemcpyAsync(ctx->h_col_line, ctx->d_col_line, (size_t)nstart * 3 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
    }
    for (int s = 0; s < nstart; s++) {
        const StartDesc &sd = ctx->h_starts[s];
        const int at[3] = {sd.sa, sd.sb, sd.sc}, nn[3] = {L.n[0], L.n[1], L.n[2]};
        const int which = ctx->col_order >= 0 ? ctx->col_order
                        : column_order_default(nn, ctx->h_col_line[3 * s], ctx->h_col_line[3 * s + 1], ctx->h_col_line[3 * s + 2]);
        column_order_sequence(which, nn, at, &ctx->h_col_ordseq[s]);
#ifdef TTSWEEP_DEBUG_ENV
        if (const char *q = getenv("TTSWEEP_COL_ORDSEQ")) ctx->h_col_ordseq[s] = strtoull(q, nullptr, 16);   // (sweep 1: the lowest nibble)
#endif
    }
    HIPCHK(hipMemcpyAsync(ctx->d_col_ordseq, ctx->h_col_ordseq, (size_t)nstart * sizeof(unsigned long long), hipMemcpyHostToDevice, ctx->stream));
    C.ordseq = ctx->d_col_ordseq;
    // every wait inside the launch gives up after this much wall clock (100 MHz ticks): ten seconds plus twenty
    // times what forty sweep equivalents should take at a quarter of the memory rate
    const double expect_s = (double)L.n[0] * L.n[1] * L.n[2] * (double)nstart * 12.0 * 40.0 / 2.0e12;
    C.timeout_ticks = (long long)((10.0 + 20.0 * expect_s) * 1.0e8);
    if (ctx->async_timeout_ms > 0) C.timeout_ticks = (long long)ctx->async_timeout_ms * 100000ll;

    HIPCHK(launch_column_init(C, ctx->d_starts, from_box, ctx->stream));
    hipEvent_t e0, e1;
    if (ctx->timing && timed_event(ctx, &e0)) return -1;
    {   // (a launch that the device refuses - e.g. the LDS opt-in - leaves the boxes untouched: the hyperplane driver runs)
        const hipError_t le = launch_column_solve(C, (int)blocks, ctx->stream);
        if (le != hipSuccess) {
            (void)hipGetLastError();
            set_error("the column launch failed (%s)", hipGetErrorString(le));
            return -2;
        }
    }
    if (ctx->timing && timed_event(ctx, &e1)) return -1;
    ctx->stats.launches++;
    HIPCHK(hipMemcpyAsync(ctx->h_col_status, ctx->d_col_status, 8 * sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(ctx->h_col_done, ctx->d_col_done, (size_t)nstart * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(ctx->h_changed, ctx->d_changed, (size_t)nstart * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (ctx->h_col_status[0] == COL_ERR_CAP) {
        // (the launch counts sweeps in COL_MAX_SWEEPS slots: a caller's cap beyond that is the hyperplane driver's to
        // honour - it takes over from the boxes as they are)
        if (ctx->max_sweeps > (long long)C.max_sweeps) {
            set_error("the one-launch solve ran out of sweep slots (%d)", C.max_sweeps);
            return -2;
        }
        return set_error("a start did not converge in %lld sweeps", ctx->max_sweeps);
    }
    if (ctx->h_col_status[0] != COL_DONE) {
        set_error("the one-launch solve gave up (code %u)", ctx->h_col_status[0]);
        return -2;
    }
    bool any = false;
    for (int s = 0; s < nstart; s++) {
        any |= (ctx->h_changed[s] & CHANGED_IMPROVED) != 0;
        sweeps[s] = ctx->h_col_done[s];
    }
    ctx->tile_blocks_used = (int)blocks * wgw;
    return any ? 1 : 0;
}

int solve_device_body(ttsweep_ctx *ctx, int nstart, const ttsweep_start *starts,
                             float *const *tt_dev, int init)
{
    const DevLayout &L = ctx->L;
    HIPCHK(hipEventRecord(ctx->ev_solve0, ctx->stream));
    // units of two planes when there are starts enough to fill the machine with them
    const bool pairs = ctx->pair_min_starts >= 0
        ? nstart >= ctx->pair_min_starts
        : (long long)nstart * strip_units(L, 1) >= (ctx->async_mode != 0 ? ctx->pair_min_units / 2 : ctx->pair_min_units);
        // (measured crossovers on 241x241x51: the pass driver from 21 starts on, the one-launch driver - whose
        // shorter units pay the same claim and completion - from about 10: 8 starts 14.7 (one plane) / 15.7 ms (two),
        // 12 starts 24.1 / 22.8, 16 starts 28.5 / 26.3, 24 starts 42.1 / 36.9)
    const int np = pairs ? STRIP_PLANES : 1;
    if (np != ctx->np) ctx->unit_order_key.assign(ctx->unit_order_key.size(), -1);     // orders belong to the other unit grid
    ctx->np = np;
    const bool column = use_column(ctx, nstart);
    bool in_place = column && column_in_place(ctx, nstart, tt_dev);    // TILE, column driver: the caller's arrays ARE the volumes

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
        sd.tile_flags = ctx->d_tile_flags + (size_t)s * flag_words(L, ctx->kernel);
        sd.work = ctx->d_work + 3 * s;
        // (STRIP, fresh boxes: one launch for all starts further down)
        const bool batched = init && ctx->kernel == TTSWEEP_KERNEL_STRIP && L.cells % 4 == 0 && nstart <= 65535;
        if (batched) { ctx->h_active[s] = s; continue; }
        if (in_place) {
            // (a fresh box is initialised where it lies: the reference's state, serial_new/...:139-144)
            if (init) HIPCHK(launch_init_tt(user_layout(L), tt_dev[s], ((long long)starts[s].i * L.un[1] + starts[s].j) * L.un[2] + starts[s].k,
                                            ctx->stream));
        } else if (init) HIPCHK(launch_init_tt(L, sd.T, sd.sidx, ctx->stream));
        else HIPCHK(launch_pack(L, tt_dev[s], sd.T, INFINITY, ctx->stream));
        if (ctx->kernel == TTSWEEP_KERNEL_STRIP)
            HIPCHK(launch_init_tile_flags(L, sd, /*from_box=*/!init, ctx->plans[np - 1].ra, np, ctx->stream));
        if (ctx->kernel == TTSWEEP_KERNEL_TILE && !column) {
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
    if (init && ctx->kernel == TTSWEEP_KERNEL_STRIP && L.cells % 4 == 0 && nstart <= 65535) {
        HIPCHK(launch_init_tt_batch(L, ctx->d_T, ctx->d_starts, nstart, ctx->stream));
        HIPCHK(launch_init_tile_flags_batch(L, ctx->d_tile_flags, (long long)flag_words(L, ctx->kernel), ctx->d_starts, nstart,
                                            ctx->plans[np - 1].ra, np, ctx->stream));
    }
    HIPCHK(hipMemcpyAsync(ctx->d_active, ctx->h_active, nstart * sizeof(int),
                          hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemsetAsync(ctx->d_work, 0, 3 * nstart * sizeof(unsigned long long), ctx->stream));
    ctx->pass_index = 0;
    ctx->defer_suspended = false;
    ctx->tile_epoch = 1;
    if (ctx->kernel == TTSWEEP_KERNEL_TILE && !column && prepare_tile_sweep(ctx)) return -1;
    bool async = use_async(ctx, nstart);
    if (ctx->kernel == TTSWEEP_KERNEL_STRIP && !async) {
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
    ctx->batch_changed.assign(nstart, 0);
#ifdef TTSWEEP_DEBUG_ENV
    const bool trace = getenv("TTSWEEP_TRACE") != nullptr;
#else
    const bool trace = false;
#endif
    unsigned long long trace_prev = 0, trace_prev_un = 0;
    std::vector<int> snapshot[PASS_SLOTS];      // active starts of each pass in flight
    int nactive = nstart, launched = 0, processed = 0;
    // (TILE: a sweep is milliseconds long and the host decides where the next one starts from
    // what this one recorded, so sweeps are enqueued one at a time; STRIP / CELL: one ahead)
    const int depth = ctx->kernel == TTSWEEP_KERNEL_TILE ? 1 : 2;
    bool anychange_ever = false;
    bool fell_back = false;     // the one-launch form gave up: the other driver finishes the solve
    auto t_pass = std::chrono::steady_clock::now();
    ctx->tile_blocks_used = ctx->tile_blocks;
    if (column) {
        HIPCHK(hipMemsetAsync(ctx->d_changed, 0, (size_t)nstart * sizeof(int), ctx->stream));
        const int rc = solve_column(ctx, nstart, /*from_box=*/!init, tt_dev, in_place, sweeps);
        if (rc == -2) {
            // A wait inside the launch ran into its wall-clock limit.  Every travel time in the boxes is the length
            // of a real path: the hyperplane launches take over from there, with every tile due.
            for (int s = 0; s < nstart; s++) {
                if (in_place) HIPCHK(launch_pack(L, tt_dev[s], ctx->h_starts[s].T, INFINITY, ctx->stream));
                HIPCHK(launch_init_tile_state(L, ctx->h_starts[s], /*from_box=*/true, ctx->stream));
                HIPCHK(launch_build_tile_faces(L, ctx->h_starts[s].T, ctx->d_tface + (size_t)s * tile_face_cells(L, ctx->tile_fz),
                                               ctx->tile_fz, ctx->stream));
                sweeps[s] = 0;
            }
            if (prepare_tile_sweep(ctx)) return -1;
            ctx->tile_blocks_used = ctx->tile_blocks;
            anychange_ever = true;
            fell_back = true;
            in_place = false;       // (the boxes go back to the caller's arrays at the end)
        } else {
            if (rc < 0) return rc;
            anychange_ever = rc > 0;
            nactive = 0;
        }
        for (int s = 0; s < nstart; s++) ctx->batch_changed[s] |= (ctx->h_changed[s] & CHANGED_IMPROVED) != 0;
    }
    if (async) {
        const int rc = solve_async_strip(ctx, nstart, sweeps);
        if (rc == -2) {
            // A wait inside the launch ran into its wall-clock limit.  Every travel time in the boxes is
            // the length of a real path, so the pass driver takes over from there: activity words as for
            // boxes that arrive with values in them, then passes until nothing changes.
            for (int s = 0; s < nstart; s++)
                HIPCHK(launch_init_tile_flags(L, ctx->h_starts[s], /*from_box=*/true, ctx->plans[np - 1].ra, np, ctx->stream));
            if (build_worklist(ctx, nstart)) return -1;
            HIPCHK(hipMemsetAsync(ctx->d_changed, 0, (size_t)PASS_SLOTS * nstart * sizeof(int), ctx->stream));
            HIPCHK(hipMemsetAsync(ctx->d_unitq_ctrl, 0, (UNITQ_CTRL_WORDS + 1) * sizeof(int), ctx->stream));
            for (int s = 0; s < nstart; s++) sweeps[s] = 0;
            anychange_ever = true;
            async = false;
            fell_back = true;
        } else {
            if (rc < 0) return rc;
            anychange_ever = rc > 0;
            nactive = 0;
        }
        // (also when the launch gave up: what it had improved by then stays improved)
        for (int s = 0; s < nstart; s++) ctx->batch_changed[s] |= (ctx->h_changed[s] & CHANGED_IMPROVED) != 0;
    }
    for (;;) {
    while (processed < launched || nactive > 0) {
        if (nactive > 0 && launched - processed < depth) {      // enqueue the next pass
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
            if (launched - processed < depth && nactive > 0 && launched == 1) continue;   // prime the pipeline
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
        }
        bool dropped = false;
        for (int s : snapshot[slot]) {
            if (done[s]) continue;
            sweeps[s]++;
            if (hch[s]) {           // improved, or units still held back by the gate
                if (hch[s] & CHANGED_IMPROVED) { anychange_ever = true; ctx->batch_changed[s] = 1; }
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

        // STRIP: every start is at rest; the bits that were deferred (push_improved: units behind the
        // front) become due now, and the starts that had any go on
        if (ctx->kernel != TTSWEEP_KERNEL_STRIP || async || ctx->defer_margin < -1.0e30f) break;
        for (int s = 0; s < nstart; s++) ctx->h_active[s] = s;
        int *const dfl = ctx->d_changed + (size_t)PASS_SLOTS * nstart, *const hfl = ctx->h_changed + (size_t)PASS_SLOTS * nstart;
        HIPCHK(hipMemcpyAsync(ctx->d_active, ctx->h_active, nstart * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(hipMemsetAsync(dfl, 0, nstart * sizeof(int), ctx->stream));
        HIPCHK(launch_flush_deferred(L, ctx->np, ctx->d_tile_flags, (long long)flag_words(L, ctx->kernel), ctx->d_active,
                                     nstart, dfl, ctx->stream));
        HIPCHK(hipMemcpyAsync(hfl, dfl, nstart * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        nactive = 0;
        for (int s = 0; s < nstart; s++)
            if (hfl[s]) { ctx->h_active[nactive++] = s; done[s] = 0; }
        if (nactive == 0) break;
        ctx->defer_suspended = true;        // (one flush per solve: from here on every unit is told at once)
        HIPCHK(hipMemcpyAsync(ctx->d_active, ctx->h_active, nactive * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
        if (build_worklist(ctx, nactive)) return -1;
    }

    if (in_place) {
        // (the travel times were relaxed in the caller's arrays)
    } else if (ctx->kernel == TTSWEEP_KERNEL_STRIP && nstart <= 65535 && nstart > 1) {
        // (all boxes in one launch; the boxes' addresses go through the pinned copy of the "changed" words,
        // which the driver loop is done with: PASS_SLOTS + 1 ints per start hold a pointer per start)
        float **const hp = reinterpret_cast<float **>(ctx->h_changed);
        float **const dp = reinterpret_cast<float **>(ctx->d_changed);
        for (int s = 0; s < nstart; s++) hp[s] = tt_dev[s];
        HIPCHK(hipMemcpyAsync(dp, hp, nstart * sizeof(float *), hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(launch_unpack_batch(L, ctx->d_T, dp, nstart, ctx->stream));
    } else {
        for (int s = 0; s < nstart; s++)
            HIPCHK(launch_unpack(L, ctx->h_starts[s].T, tt_dev[s], ctx->stream));
    }
    if (ctx->kernel == TTSWEEP_KERNEL_TILE)
        HIPCHK(launch_tile_reduce_work(ctx->d_tile_wgwork, ctx->tile_blocks_used, nstart, ctx->d_work, ctx->stream));
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
#ifdef TTSWEEP_COL_TRACE
    if (ctx->kernel == TTSWEEP_KERNEL_TILE) column_trace_dump();
#endif
#ifdef TTSWEEP_COL_PROFILE
    if (ctx->kernel == TTSWEEP_KERNEL_TILE) column_prof_dump();
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
    if (fell_back) {
        ctx->stats.fallbacks++;
        set_error("%s", "");    // (the solve succeeded: the one-launch form's complaint is not an error of this call)
    }
    return anychange_ever ? 1 : 0;
}

} // namespace ttsweep
