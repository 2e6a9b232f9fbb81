// ttsweep_api.cpp - the C ABI of include/ttsweep.h (host side of libttsweep.so).
//
// Replaces the driver loop of serial_new/sweep-tt-multistart.c:151-170 and its
// callee sweepXYZ (:198-256) by device-resident relaxation to convergence.
// There is no CPU fallback in this library: every solve runs HIP kernels or fails.
#include "ttsweep_ctx.h"

#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <thread>

using namespace ttsweep;

// ---------------------------------------------------------------------------
// error reporting
// ---------------------------------------------------------------------------
static thread_local std::string g_last_error;

int ttsweep::set_error(const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return -1;
}

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

// ---- warm-up ----------------------------------------------------------------------------
// The first HIP call of a process pays for the runtime (driver, device, code objects): a few
// hundred milliseconds that a host program otherwise spends inside its first sweepXYZ call.
// ttsweep_warmup starts that work on a thread of its own and returns at once, so that it runs
// beside the host's file reading and box initialisation; ttsweep_create waits for it.
static std::mutex g_warm_mutex;
static std::thread g_warm_thread;
static struct WarmupAtExit {        // (a process that ends early - a usage error - waits for the thread)
    ~WarmupAtExit()
    {
        if (g_warm_thread.joinable()) g_warm_thread.join();
    }
} g_warm_at_exit;

static void join_warmup()
{
    std::lock_guard<std::mutex> lock(g_warm_mutex);
    if (g_warm_thread.joinable()) g_warm_thread.join();
}

int ttsweep_warmup(int device)
{
    std::lock_guard<std::mutex> lock(g_warm_mutex);
    if (g_warm_thread.joinable()) return 0;         // already under way
    g_warm_thread = std::thread([device]() {
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return;
        if (hipSetDevice(device) != hipSuccess) return;
        hipStream_t st = nullptr;
        if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) return;
        int n = 0;
        (void)device_xcds(device, st, &n);          // (first kernel launch: loads the code object)
        (void)hipStreamDestroy(st);
    });
    return 0;
}

ttsweep_ctx *ttsweep_create(int device, int nx, int ny, int nz, const ttsweep_fs *fs,
                            int starstart, int starstop)
{
    join_warmup();
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
    ctx->fs_copy.assign(fs, fs + starstop);
    ctx->starstart = starstart;
    ctx->starstop = starstop;
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
           && hipMalloc((void **)&ctx->d_scratch, 4 * sizeof(unsigned long long)) == hipSuccess
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
    if (ctx->pre) ttsweep_destroy(ctx->pre);
    ctx->pre = nullptr;
    (void)hipFree(ctx->d_v);
    (void)hipFree(ctx->d_scratch);
    (void)hipFree(ctx->d_stage);
    (void)hipFree(ctx->d_cell_entries);
    (void)hipFree(ctx->d_fwd_entries);
    for (StripItem *d : ctx->d_strip_items) (void)hipFree(d);
    (void)hipFree(ctx->d_strip_items_lat);
    (void)hipFree(ctx->d_T);
    (void)hipFree(ctx->d_starts);
    (void)hipFree(ctx->d_active);
    (void)hipFree(ctx->d_changed);
    (void)hipFree(ctx->d_tile_flags);
    (void)hipFree(ctx->d_worklist);
    (void)hipFree(ctx->d_unitq);
    (void)hipFree(ctx->d_unitq_ctrl);
    (void)hipFree(ctx->d_async_list);
    (void)hipFree(ctx->d_async_ring_starts);
    (void)hipFree(ctx->d_async_entries);
    (void)hipFree(ctx->d_async_ctl);
    (void)hipFree(ctx->d_async_status);
    if (ctx->h_async_status) (void)hipHostFree(ctx->h_async_status);
    (void)hipFree(ctx->d_work);
    (void)hipFree(ctx->d_tile_wgwork);
    (void)hipFree(ctx->d_col_prog);
    (void)hipFree(ctx->d_col_claim);
    (void)hipFree(ctx->d_col_due);
    (void)hipFree(ctx->d_col_status);
    (void)hipFree(ctx->d_col_done);
    (void)hipFree(ctx->d_col_ordseq);
    (void)hipFree(ctx->d_col_line);
    if (ctx->h_col_line) (void)hipHostFree(ctx->h_col_line);
    if (ctx->h_col_ordseq) (void)hipHostFree(ctx->h_col_ordseq);
    (void)hipFree(ctx->d_col_seqtab);
    (void)hipFree(ctx->d_col_tptr);
    if (ctx->h_col_tptr) (void)hipHostFree(ctx->h_col_tptr);
    if (ctx->h_col_status) (void)hipHostFree(ctx->h_col_status);
    if (ctx->h_col_done) (void)hipHostFree(ctx->h_col_done);
    (void)hipFree(ctx->d_tile_dmin);
    if (ctx->h_tile_dmin) (void)hipHostFree(ctx->h_tile_dmin);
    (void)hipFree(ctx->d_vface);
    (void)hipFree(ctx->d_tface);
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

// the padded layout depends on the kernel: rebuild it and drop the device copies
static int switch_kernel(ttsweep_ctx *ctx, int k)
{
    if (k == ctx->kernel) return 0;
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
    ctx->async_list_key.clear();
    HIPCHK(hipMalloc((void **)&ctx->d_v, (size_t)ctx->L.cells * sizeof(float)));
    if (upload_star(ctx)) return -1;
    if (k == TTSWEEP_KERNEL_STRIP && upload_strip_plan(ctx)) return -1;
    return 0;
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
        if (ctx->exact_half) {          // (a velocity volume with sub-limit values: the CELL kernel's exact instance stays)
            ctx->kernel_wanted = k;
            return 0;
        }
        return switch_kernel(ctx, k);
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
    case TTSWEEP_OPT_ASYNC:
        if (value < -1 || value > 1) return set_error("async mode must be -1, 0 or 1");
        ctx->async_mode = (int)value;
        return 0;
    case TTSWEEP_OPT_ASYNC_LOW:
    case TTSWEEP_OPT_ASYNC_HIGH:
        if (value < 0 || value > 8192) return set_error("ring fill marks must lie in [0, 8192]");
        (key == TTSWEEP_OPT_ASYNC_LOW ? ctx->async_low : ctx->async_high) = (int)value;
        return 0;
    case TTSWEEP_OPT_DEFER_MARGIN_MILLI:
        ctx->defer_margin = value <= -1000000000ll ? -3.0e38f : (float)((double)value / 1000.0);
        return 0;
    case TTSWEEP_OPT_ASYNC_POLICY:
        if (value < 0 || value > 2) return set_error("ring policy must be 0, 1 or 2");
        ctx->async_policy = (int)value;
        return 0;
    case TTSWEEP_OPT_ASYNC_GATE_MILLI:
        if (value < 0) return set_error("gate speed must be >= 0");
        ctx->async_gate_speed = (float)((double)value / 1000.0);
        return 0;
    case TTSWEEP_OPT_ASYNC_GATE_FAST_MILLI:
        if (value < 0) return set_error("gate speed must be >= 0");
        ctx->async_gate_fast = (float)((double)value / 1000.0);
        return 0;
    case TTSWEEP_OPT_ASYNC_TIMEOUT_MILLI:
        if (value < 0 || value > 3600000) return set_error("time limit must lie in [0, 3600000] ms");
        ctx->async_timeout_ms = (int)value;
        return 0;
    case TTSWEEP_OPT_ASYNC_WINDOW_MILLI:
        if (value < 0) return set_error("window must be >= 0");
        ctx->async_window = (float)((double)value / 1000.0);
        return 0;
    case TTSWEEP_OPT_TILE_IN_PLACE: ctx->col_in_place_off = value == 0; return 0;
    case TTSWEEP_OPT_TILE_ORDER:
        if (value != -1 && (value < 0 || value > 999 || !ttsweep::column_order_valid((int)value)))
            return set_error("sweep order: -1 (by the model) or table 0 .. %d + 10 x first corner 0 .. 2 + 100 x axis roles 0 .. 4", (int)ttsweep::COL_ORDER_SEQUENCES - 1);
        ctx->col_order = (int)value;
        return 0;
    case TTSWEEP_OPT_ASYNC_INUNIT:
        if (value < -1 || value > 8) return set_error("in-unit passes must be -1 (default rule) or 0 .. 8");
        ctx->async_inunit = (int)value;
        return 0;
    case TTSWEEP_OPT_ASYNC_WAVES:
        if (value != -1 && value != ttsweep::STRIP_NS && value != ttsweep::STRIP_NS_LAT)
            return set_error("waves per unit must be -1 (default rule), %d or %d", (int)ttsweep::STRIP_NS, (int)ttsweep::STRIP_NS_LAT);
        ctx->async_waves = (int)value;
        return 0;
    case TTSWEEP_OPT_ASYNC_HANDOFF:
        if (value < -1 || value > 3) return set_error("hand-off must be -1 (default rule) or a sum of 1 (neighbours) and 2 (own unit)");
        ctx->async_handoff = (int)value;
        return 0;
    case TTSWEEP_OPT_QUEUES:
        if (value < 1 || value > ttsweep::UNITQ_LISTS) return set_error("queues must be 1 .. %d", (int)ttsweep::UNITQ_LISTS);
        ctx->nlists = (int)value;
        ctx->unitq_blocks = 0;          // (the grid is whole rounds of the queues: sized again at the next solve)
        return 0;
    case TTSWEEP_OPT_ASYNC_SPECIAL:
        if (value < 1) return set_error("dead-edge interval must be positive");
        ctx->async_special_every = (int)std::min<long long>(value, 1 << 30);
        return 0;
    case TTSWEEP_OPT_GATE_R0_MILLI:
        if (value < 0) return set_error("gate start radius must be >= 0");
        ctx->gate_r0 = (double)value / 1000.0;
        return 0;
    case TTSWEEP_OPT_PREPASS_ENTRIES: {
        if (value < 0) return set_error("pre-pass entry count must be >= 0");
        const int m = (int)std::min<long long>(value, ctx->starstop - ctx->starstart);
        if (ctx_bind(ctx)) return -1;
        if (ctx->pre) ttsweep_destroy(ctx->pre);
        ctx->pre = nullptr;
        ctx->prepass_entries = 0;
        if (m == 0 || m == ctx->starstop - ctx->starstart) return 0;     // off (or the whole star: nothing to gain)
        ctx->pre = ttsweep_create(ctx->device, ctx->nx, ctx->ny, ctx->nz, ctx->fs_copy.data(), ctx->starstart,
                                  ctx->starstart + m);
        if (!ctx->pre) return -1;
        ctx->prepass_entries = m;
        if (ctx->have_v) {      // hand the velocity volume on (the context only keeps its padded copy)
            float *tmp = nullptr;
            const size_t bytes = (size_t)ctx->nx * ctx->ny * ctx->nz * sizeof(float);
            HIPCHK(hipMalloc((void **)&tmp, bytes));
            hipError_t e = launch_unpack(ctx->L, ctx->d_v, tmp, ctx->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
            const int rc = e == hipSuccess ? ttsweep_set_velocity_device(ctx->pre, tmp)
                                           : set_error("pre-pass velocity copy failed: %s", hipGetErrorString(e));
            (void)hipFree(tmp);
            if (rc < 0) return -1;
        }
        return 0;
    }
    default: return set_error("unknown option %d", key);
    }
}

int ttsweep_set_velocity_device(ttsweep_ctx *ctx, const float *v_dev)
{
    if (!ctx || !v_dev) return set_error("null argument");
    if (ctx_bind(ctx)) return -1;
    ctx->solved.clear();
    // every cell must be a finite number >= 0: delays are then >= 0, the relaxation has a least
    // fixed point and no NaN can arise.  Zero is accepted, as the
    // reference accepts it (zero delays: serial_new/sweep-tt-multistart.c:216 has no test);
    // a negative velocity would make the reference loop forever and is refused here.
    // Positive velocities so small that a delay could be a denormal number: the reference
    // halves the ROUNDED product d * (v[c] + v[o]) (:216), the fast kernels multiply by d / 2, and
    // the two differ in the last bit once the product is denormal (found by
    // tests/test_gpu_parity.py::test_tiny_velocities).  With the smallest offset length d_min every
    // product is >= 2^-124 when v >= 2^-124 / d_min (a factor two above the smallest product that
    // halves exactly).  A volume with a positive value below that is accepted all the same and solved
    // by the CELL kernel's EXACT instance, which rounds as the reference does (product, then half):
    // slow, but what the reference accepts the boundary accepts, with the same bits.
    unsigned long long *d_bad = ctx->d_scratch, h_bad[2] = {0, 0};
    const long long n = (long long)ctx->nx * ctx->ny * ctx->nz;
    HIPCHK(hipMemsetAsync(d_bad, 0, 2 * sizeof(unsigned long long), ctx->stream));
    float tiny = 0.0f;
    {
        float dmin = 0.0f;
        for (int l = ctx->starstart; l < ctx->starstop; l++) {
            const float d = ctx->fs_copy[l].d;
            if (d > 0.0f && (dmin == 0.0f || d < dmin)) dmin = d;
        }
        if (dmin > 0.0f) tiny = (float)std::min(std::ldexp(1.0, -124) / (double)dmin, 1.0e30);
    }
    HIPCHK(launch_count_bad_velocity(v_dev, n, tiny, d_bad, ctx->stream));
    HIPCHK(hipMemcpyAsync(h_bad, d_bad, sizeof h_bad, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (h_bad[0]) {
        ctx->have_v = false;
        return set_error("velocity volume holds %llu cells that are negative or not finite", h_bad[0]);
    }
    const bool exact = h_bad[1] != 0;
    if (exact != ctx->exact_half) {
        if (exact) {
            ctx->kernel_wanted = ctx->kernel;
            if (switch_kernel(ctx, TTSWEEP_KERNEL_CELL)) return -1;
            ctx->exact_half = true;
        } else {
            ctx->exact_half = false;
            if (switch_kernel(ctx, ctx->kernel_wanted)) return -1;
        }
    }
    HIPCHK(launch_pack(ctx->L, v_dev, ctx->d_v, 0.0f, ctx->stream));
    if (ctx->kernel == TTSWEEP_KERNEL_TILE) {
        if (!ctx->d_vface)
            HIPCHK(hipMalloc((void **)&ctx->d_vface, (size_t)tile_face_cells(ctx->L, ctx->tile_fz) * sizeof(float)));
        HIPCHK(launch_build_tile_faces(ctx->L, ctx->d_v, ctx->d_vface, ctx->tile_fz, ctx->stream));
    }
    HIPCHK(hipStreamSynchronize(ctx->stream));
    ctx->have_v = true;
    if (ctx->pre && ttsweep_set_velocity_device(ctx->pre, v_dev) < 0) return -1;
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
    ctx->changed_last.assign(nstart, 0);
    if (nstart == 0) return 0;
    for (int s = 0; s < nstart; s++)        // before anything is queued on the stream
        if (starts[s].i < 0 || starts[s].i >= ctx->nx || starts[s].j < 0 || starts[s].j >= ctx->ny
            || starts[s].k < 0 || starts[s].k >= ctx->nz)
            return set_error("start %d (%d,%d,%d) outside the grid", s, starts[s].i, starts[s].j, starts[s].k);
    // Pre-pass (TTSWEEP_OPT_PREPASS_ENTRIES; old/wavefront-openmp/wave-multistart.c:210-215 sweeps
    // a short sub-range of the star before the whole one): the first entries of the star are
    // relaxed to THEIR fixed point first, then the whole star from that state.  Every
    // travel time the pre-pass leaves is the length of a path of the whole graph (its edges
    // are a subset), so the fixed point of the second phase is the same, bit for bit.
    int pre_rc = 0;
    ttsweep_stats pre_stats{};
    if (ctx->pre) {
        ctx->pre->timing = ctx->timing;
        ctx->pre->max_sweeps = ctx->max_sweeps;
        pre_rc = ttsweep_solve_device(ctx->pre, nstart, starts, tt_dev, init);
        if (pre_rc < 0) return pre_rc;
        pre_stats = ctx->pre->stats;
        init = 0;
        if (ctx_bind(ctx)) return -1;
    }
    if (ensure_capacity(ctx, nstart)) return -1;
    int rc = solve_device_body(ctx, nstart, starts, tt_dev, init);
    if (rc >= 0) {
        for (int s = 0; s < nstart && s < (int)ctx->batch_changed.size(); s++) ctx->changed_last[s] = ctx->batch_changed[s];
        if (ctx->pre)
            for (int s = 0; s < nstart && s < (int)ctx->pre->changed_last.size(); s++) ctx->changed_last[s] |= ctx->pre->changed_last[s];
    }
    if (rc >= 0 && ctx->pre) {
        rc |= pre_rc;
        // (work of the pre-pass in units of the whole star, so that cells_relaxed keeps its meaning)
        const double share = (double)ctx->pre->pull.size() / (double)std::max<size_t>(ctx->pull.size(), 1);
        ctx->stats.cells_relaxed += (long long)((double)pre_stats.cells_relaxed * share);
        ctx->stats.sweeps_total += pre_stats.sweeps_total;
        ctx->stats.launches += pre_stats.launches;
        ctx->stats.sweep_kernel_ms += pre_stats.sweep_kernel_ms;
        ctx->stats.solve_ms += pre_stats.solve_ms;
    }
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

extern "C" {

// 128-bit digest of a box (the bit patterns of its floats), for the "nothing has changed since
// this context returned these boxes" test of ttsweep_solve: four independent multiply-xorshift
// lanes over 64-bit words (a lane's step is a bijection of its state for every word, and of the
// word for every state), memory-bound; the 256 bits of lane state are folded into two 64-bit
// words in two different ways.  (Round 4 kept 64 bits: the one place where the boundary could,
// with probability 2^-64 per edited box, answer "0" for a box the caller has changed.)
typedef ttsweep_ctx::Digest Digest;
static Digest box_digest(const float *box, size_t cells)
{
    const uint64_t K = 0x9E3779B97F4A7C15ull, K2 = 0xC2B2AE3D27D4EB4Full;
    uint64_t h[4] = {K, K ^ 0x1111111111111111ull, K ^ 0x2222222222222222ull, K ^ 0x3333333333333333ull};
    const size_t words = cells / 2;
    size_t i = 0;
    for (; i + 4 <= words; i += 4) {
        uint64_t w[4];
        std::memcpy(w, box + 2 * i, sizeof w);
        for (int k = 0; k < 4; k++) {
            h[k] = (h[k] ^ w[k]) * K;
            h[k] ^= h[k] >> 29;
        }
    }
    for (; 2 * i < cells; i++) {       // the last words (a half word for an odd cell count)
        uint64_t w = 0;
        std::memcpy(&w, box + 2 * i, std::min<size_t>(8, (cells - 2 * i) * sizeof(float)));
        h[i & 3] = (h[i & 3] ^ w) * K;
        h[i & 3] ^= h[i & 3] >> 29;
    }
    Digest out{cells, ~(uint64_t)cells};
    for (int k = 0; k < 4; k++) out.a = (out.a ^ h[k]) * K + (out.a >> 31);
    for (int k = 3; k >= 0; k--) {
        const uint64_t r = (h[k] << 23) | (h[k] >> 41);
        out.b = (out.b ^ r) * K2 + (out.b >> 29);
    }
    return out;
}

// Digests of n boxes, a few host threads side by side.
static std::vector<Digest> box_digests(float *const *boxes, int n, size_t cells)
{
    std::vector<Digest> out(n);
    const int nthreads = std::max(1, std::min({n, 4, (int)std::thread::hardware_concurrency()}));
    std::vector<std::thread> pool;
    for (int t = 0; t < nthreads; t++)
        pool.emplace_back([&, t]() {
            for (int s = t; s < n; s += nthreads) out[s] = box_digest(boxes[s], cells);
        });
    for (auto &th : pool) th.join();
    return out;
}

int ttsweep_solve(ttsweep_ctx *ctx, int nstart, const ttsweep_start *starts,
                  float *const *tt_host)
{
    if (!ctx || !starts || !tt_host || nstart < 0) return set_error("bad arguments");
    if (ctx_bind(ctx)) return -1;
    if (nstart == 0) return 0;
    const size_t cells = (size_t)ctx->nx * ctx->ny * ctx->nz;

    // The reference driver calls its sweep until nothing changes (serial_new/...:151-170), so
    // a drop-in host calls this function once more with the boxes it was just given back, to
    // hear "0".  When every box of the call is bit for bit what this context last wrote into
    // that very array for that very start (velocity unchanged since; the fixed point depends
    // on nothing else a context can change), it is the fixed point already and a solve would
    // store nothing: answer without touching the device.  (The comparison is a 128-bit digest
    // of every box, not a promise by the caller.)
    {
        bool known = !ctx->solved.empty();
        for (int s = 0; s < nstart && known; s++) {
            const auto it = ctx->solved.find(tt_host[s]);
            known = it != ctx->solved.end() && it->second.start.i == starts[s].i
                 && it->second.start.j == starts[s].j && it->second.start.k == starts[s].k;
        }
        if (known) {
            const std::vector<Digest> now = box_digests(tt_host, nstart, cells);
            for (int s = 0; s < nstart && known; s++) known = now[s] == ctx->solved[tt_host[s]].digest;
        }
        if (known) {
            ctx->changed_last.assign(nstart, 0);
            ctx->stats = ttsweep_stats{};
            ctx->stats.nstart = nstart;
            ctx->stats.cells = (long long)cells;
            ctx->stats.relaxations_per_sweep = ctx->relax_per_sweep;
            ctx->stats.kernel_variant = ctx->kernel;
            return 0;
        }
        for (int s = 0; s < nstart; s++) ctx->solved.erase(tt_host[s]);     // (until this call has succeeded)
    }

    // Starts are independent: solve them in batches that fit the device memory (per start:
    // one staging box in the caller's layout + what ensure_capacity allocates: the padded
    // volume, activity words, z faces).  The pools are kept by the context between calls.
    size_t free_b = 0, total_b = 0;
    HIPCHK(hipMemGetInfo(&free_b, &total_b));
    const size_t per_start = cells * sizeof(float) + per_start_device_bytes(ctx)
                           + (ctx->pre ? per_start_device_bytes(ctx->pre) : 0);
    free_b += (size_t)ctx->capacity_starts * per_start_device_bytes(ctx) + ctx->stage_cap * cells * sizeof(float);
    int batch = (int)std::min<size_t>((size_t)nstart, (size_t)(0.85 * (double)free_b) / per_start);
    if (batch < 1) return set_error("not enough device memory for one travel-time volume");
    if (ctx->max_batch > 0) batch = std::min(batch, ctx->max_batch);

    ttsweep_stats total{};
    int any = 0;
    std::vector<int> changed_all(nstart, 0);
    for (int first = 0; first < nstart; first += batch) {
        const int n = std::min(batch, nstart - first);
        if ((size_t)n > ctx->stage_cap) {
            HIPCHK(hipStreamSynchronize(ctx->stream));
            if (ctx->d_stage) HIPCHK(hipFree(ctx->d_stage));
            ctx->d_stage = nullptr;
            ctx->stage_cap = 0;
            HIPCHK(hipMalloc((void **)&ctx->d_stage, (size_t)n * cells * sizeof(float)));
            ctx->stage_cap = (size_t)n;
        }
        float *const stage = ctx->d_stage;
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
        for (int s = 0; s < n; s++) ptrs[s] = stage + (size_t)s * cells;
        {   // (page-locking is host work per page: a few threads side by side)
            const int nthreads = std::max(1, std::min({n, 4, (int)std::thread::hardware_concurrency()}));
            std::vector<std::thread> pool;
            for (int t = 0; t < nthreads; t++)
                pool.emplace_back([&, t]() {
                    if (hipSetDevice(ctx->device) != hipSuccess) return;
                    for (int s = t; s < n; s += nthreads) {
                        if (hipHostRegister(tt_host[first + s], cells * sizeof(float), hipHostRegisterDefault) == hipSuccess)
                            pinned[s] = 1;
                        else
                            (void)hipGetLastError();
                    }
                });
            for (auto &th : pool) th.join();
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
        // what the caller will hold afterwards: the fixed point of every start - digested box by
        // box as the downloads complete (an event per box, a few host threads)
        std::vector<Digest> dig(n);
        std::vector<hipEvent_t> landed;
        if (rc > 0) {       // (rc == 0: nothing was stored, the caller's boxes are the result already)
            for (int s = 0; s < n; s++) {
                hipError_t e = hipMemcpyAsync(tt_host[first + s], ptrs[s], cells * sizeof(float),
                                              hipMemcpyDeviceToHost, ctx->stream);
                hipEvent_t ev = nullptr;
                if (e == hipSuccess) e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
                if (e == hipSuccess) e = hipEventRecord(ev, ctx->stream);
                if (ev) landed.push_back(ev);
                if (e != hipSuccess) {
                    rc = set_error("travel-time download failed: %s", hipGetErrorString(e));
                    break;
                }
            }
        }
        if (rc >= 0) {
            const bool by_event = (int)landed.size() == n;
            if (!by_event) (void)hipStreamSynchronize(ctx->stream);
            const int nthreads = std::max(1, std::min({n, 4, (int)std::thread::hardware_concurrency()}));
            std::vector<std::thread> pool;
            for (int t = 0; t < nthreads; t++)
                pool.emplace_back([&, t]() {
                    if (hipSetDevice(ctx->device) != hipSuccess) return;
                    for (int s = t; s < n; s += nthreads) {
                        if (by_event && hipEventSynchronize(landed[s]) != hipSuccess) continue;
                        dig[s] = box_digest(tt_host[first + s], cells);
                    }
                });
            for (auto &th : pool) th.join();
        }
        {
            hipError_t e = hipStreamSynchronize(ctx->stream);
            if (e != hipSuccess && rc >= 0) rc = set_error("travel-time transfer failed: %s", hipGetErrorString(e));
        }
        for (hipEvent_t ev : landed) (void)hipEventDestroy(ev);
        lap("download + digests");
        if (rc >= 0) {
            if (ctx->solved.size() + (size_t)n > 65536) ctx->solved.clear();       // (bounded memory)
            for (int s = 0; s < n; s++) ctx->solved[tt_host[first + s]] = ttsweep_ctx::SolvedBox{starts[first + s], dig[s]};
        }
        for (int s = 0; s < n; s++)
            if (pinned[s]) (void)hipHostUnregister(tt_host[first + s]);
        lap("unpin");
        if (rc < 0) return rc;
        any |= rc;
        for (int s = 0; s < n && s < (int)ctx->changed_last.size(); s++) changed_all[first + s] = ctx->changed_last[s];
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
        total.fallbacks += b.fallbacks;
    }
    ctx->stats = total;
    ctx->changed_last = changed_all;
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
    unsigned long long *d_counts = ctx->d_scratch;
    HIPCHK(hipMemsetAsync(d_counts, 0, 3 * sizeof(unsigned long long), ctx->stream));
    HIPCHK(launch_pack(L, tt_dev, ctx->d_T, INFINITY, ctx->stream));
    HIPCHK(launch_validate(L, ctx->d_v, ctx->d_T, sidx, ctx->d_fwd_entries, ctx->n_fwd_entries,
                           ctx->d_cell_entries, ctx->n_cell_entries, d_counts, ctx->exact_half, ctx->stream));
    unsigned long long h[3] = {0, 0, 0};
    HIPCHK(hipMemcpyAsync(h, d_counts, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (open_edges) *open_edges = (long long)h[0];
    if (cells_infinite) *cells_infinite = (long long)h[1];
    if (cells_unsupported) *cells_unsupported = (long long)h[2];
    return 0;
}

int ttsweep_get_changed(const ttsweep_ctx *ctx, int *out, int n)
{
    if (!ctx || !out || n < 0) return set_error("bad arguments");
    const int m = std::min(n, (int)ctx->changed_last.size());
    for (int s = 0; s < m; s++) out[s] = ctx->changed_last[s];
    return m;
}

int ttsweep_get_stats(const ttsweep_ctx *ctx, ttsweep_stats *out)
{
    if (!ctx || !out) return set_error("null argument");
    *out = ctx->stats;
    return 0;
}

// Shards balanced by estimated cost (distance from the start to the farthest corner of the grid: the number of
// passes grows with it), longest first onto the least loaded device, at most ceil(nstart / ndev) starts per device
// (multistart.all_shards).
static std::vector<std::vector<int>> cost_balanced_shards(int ndev, int nx, int ny, int nz, int nstart, const ttsweep_start *starts)
{
    std::vector<std::vector<int>> shard(ndev);
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
    return shard;
}

// ---- RCCL, loaded at run time (the library does not link it: a host without librccl still has the peer copies) ----
extern "C++" {
namespace {
typedef void *rccl_comm;
struct Rccl {
    void *lib = nullptr;
    int (*CommInitAll)(rccl_comm *, int, const int *) = nullptr;
    int (*CommDestroy)(rccl_comm) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void *, size_t, int, int, rccl_comm, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, rccl_comm, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    bool ok() const { return CommInitAll && CommDestroy && GroupStart && GroupEnd && Send && Recv; }
};
constexpr int RCCL_FLOAT = 7;       // ncclFloat32 (rccl.h: ncclDataType_t)

const Rccl &rccl()
{
    static Rccl r = [] {
        Rccl x;
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            x.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (x.lib) break;
        }
        if (x.lib) {
            x.CommInitAll = (int (*)(rccl_comm *, int, const int *))dlsym(x.lib, "ncclCommInitAll");
            x.CommDestroy = (int (*)(rccl_comm))dlsym(x.lib, "ncclCommDestroy");
            x.GroupStart = (int (*)())dlsym(x.lib, "ncclGroupStart");
            x.GroupEnd = (int (*)())dlsym(x.lib, "ncclGroupEnd");
            x.Send = (int (*)(const void *, size_t, int, int, rccl_comm, hipStream_t))dlsym(x.lib, "ncclSend");
            x.Recv = (int (*)(void *, size_t, int, int, rccl_comm, hipStream_t))dlsym(x.lib, "ncclRecv");
            x.GetErrorString = (const char *(*)(int))dlsym(x.lib, "ncclGetErrorString");
        }
        return x;
    }();
    return r;
}
} // namespace
} // extern "C++"

int ttsweep_solve_multi_device(int ndev, const int *devices, int nx, int ny, int nz, const ttsweep_fs *fs, int starstart,
                               int starstop, const float *v_host, int nstart, const ttsweep_start *starts,
                               float *const *tt_root, int flags, int *changed, int *gather_path)
{
    if (ndev <= 0 || !devices || !starts || !tt_root || !v_host || nstart < 0) return set_error("bad arguments");
    if (gather_path) *gather_path = TTSWEEP_GATHER_NONE;
    if (nstart == 0) return 0;
    const size_t cells = (size_t)nx * ny * nz;
    const bool loopback = (flags & TTSWEEP_MULTI_LOOPBACK) != 0;
    const std::vector<std::vector<int>> shard = cost_balanced_shards(ndev, nx, ny, nz, nstart, starts);
    std::vector<int> rc(ndev, 0);
    std::vector<std::string> err(ndev);
    std::vector<float *> local(ndev, nullptr);          // per device: its boxes, one after the other (not the root's,
                                                        // which are solved where they belong - unless they loop back)
    std::vector<int> changed_all(nstart, 0);
    std::vector<std::thread> workers;
    for (int d = 0; d < ndev; d++) {
        workers.emplace_back([&, d]() {
            if (shard[d].empty()) return;
            const int n = (int)shard[d].size();
            ttsweep_ctx *ctx = ttsweep_create(devices[d], nx, ny, nz, fs, starstart, starstop);
            int r = ctx ? ttsweep_set_velocity(ctx, v_host) : -1;
            std::vector<ttsweep_start> my_starts;
            std::vector<float *> my_boxes;
            if (r == 0 && (d != 0 || loopback)) {
                if (hipSetDevice(devices[d]) != hipSuccess || hipMalloc((void **)&local[d], (size_t)n * cells * sizeof(float)) != hipSuccess) {
                    set_error("no device memory for %d boxes on device %d", n, devices[d]);
                    r = -1;
                }
            }
            for (int k = 0; k < n && r == 0; k++) {
                my_starts.push_back(starts[shard[d][k]]);
                my_boxes.push_back(local[d] ? local[d] + (size_t)k * cells : tt_root[shard[d][k]]);
            }
            if (r == 0) r = ttsweep_solve_device(ctx, n, my_starts.data(), my_boxes.data(), /*init=*/1);
            if (r >= 0 && ctx) {
                std::vector<int> ch(n, 0);
                (void)ttsweep_get_changed(ctx, ch.data(), n);
                for (int k = 0; k < n; k++) changed_all[shard[d][k]] = ch[k];
            }
            if (r < 0) err[d] = ttsweep_last_error();       // thread-local text
            if (ctx) ttsweep_destroy(ctx);
            rc[d] = r;
        });
    }
    for (auto &w : workers) w.join();
    auto release = [&]() {
        for (int d = 0; d < ndev; d++)
            if (local[d]) { (void)hipSetDevice(devices[d]); (void)hipFree(local[d]); }
    };
    int any = 0;
    for (int d = 0; d < ndev; d++) {
        if (rc[d] < 0) { release(); return set_error("device %d: %s", devices[d], err[d].c_str()); }
        any |= rc[d];
    }
    if (changed) for (int s = 0; s < nstart; s++) changed[s] = changed_all[s];

    // ---- the gather: every box that was solved elsewhere goes into its slot on devices[0]
    bool anything = false;
    for (int d = 0; d < ndev; d++) anything |= local[d] != nullptr;
    if (!anything) return any;
    std::vector<hipStream_t> streams(ndev, nullptr);
    auto fail = [&](const char *what, const char *why) {
        for (int d = 0; d < ndev; d++)
            if (streams[d]) { (void)hipSetDevice(devices[d]); (void)hipStreamDestroy(streams[d]); }
        release();
        return set_error("gather: %s: %s", what, why);
    };
    for (int d = 0; d < ndev; d++) {
        if (hipSetDevice(devices[d]) != hipSuccess || hipStreamCreate(&streams[d]) != hipSuccess)
            return fail("hipStreamCreate", hipGetErrorString(hipGetLastError()));
    }
    bool distinct = true;
    for (int a = 0; a < ndev; a++)
        for (int b = a + 1; b < ndev; b++) distinct &= devices[a] != devices[b];
    bool done = false;
    if (!(flags & TTSWEEP_MULTI_NO_RCCL) && distinct && rccl().ok()) {
        // one communicator per listed device, ONE group of send / receive pairs: all inbound xGMI links of the root at once
        const Rccl &R = rccl();
        std::vector<rccl_comm> comm(ndev, nullptr);
        int e = R.CommInitAll(comm.data(), ndev, devices);
        if (e == 0) {
            e = R.GroupStart();
            for (int d = 0; d < ndev && e == 0; d++) {
                if (!local[d]) continue;
                for (size_t k = 0; k < shard[d].size() && e == 0; k++) {
                    e = R.Send(local[d] + k * cells, cells, RCCL_FLOAT, 0, comm[d], streams[d]);
                    if (e == 0) e = R.Recv(tt_root[shard[d][k]], cells, RCCL_FLOAT, d, comm[0], streams[0]);
                }
            }
            const int e2 = R.GroupEnd();
            if (e == 0) e = e2;
            for (int d = 0; d < ndev && e == 0; d++) {
                (void)hipSetDevice(devices[d]);
                if (hipStreamSynchronize(streams[d]) != hipSuccess) e = -1;
            }
            for (int d = 0; d < ndev; d++)
                if (comm[d]) (void)R.CommDestroy(comm[d]);
            done = e == 0;
        }
        if (done && gather_path) *gather_path = TTSWEEP_GATHER_RCCL;
        // (a refused communicator or a failed group: the peer copies below move every box again)
    }
    if (!done) {
        // (a group that failed part-way may have receives into tt_root in flight on any stream: every device is
        // drained before the peer copies write the same slots)
        for (int d = 0; d < ndev; d++) {
            (void)hipSetDevice(devices[d]);
            (void)hipStreamSynchronize(streams[d]);
            (void)hipDeviceSynchronize();
            (void)hipGetLastError();
        }
        for (int d = 0; d < ndev; d++) {
            if (!local[d]) continue;
            (void)hipSetDevice(devices[0]);
            for (size_t k = 0; k < shard[d].size(); k++)
                if (hipMemcpyPeerAsync(tt_root[shard[d][k]], devices[0], local[d] + k * cells, devices[d], cells * sizeof(float),
                                       streams[0]) != hipSuccess)
                    return fail("hipMemcpyPeerAsync", hipGetErrorString(hipGetLastError()));
        }
        (void)hipSetDevice(devices[0]);
        if (hipStreamSynchronize(streams[0]) != hipSuccess) return fail("hipStreamSynchronize", hipGetErrorString(hipGetLastError()));
        if (gather_path) *gather_path = TTSWEEP_GATHER_PEER;
    }
    for (int d = 0; d < ndev; d++) { (void)hipSetDevice(devices[d]); (void)hipStreamDestroy(streams[d]); }
    release();
    return any;
}

int ttsweep_solve_multi(int ndev, const int *devices, int nx, int ny, int nz,
                        const ttsweep_fs *fs, int starstart, int starstop, const float *v_host,
                        int nstart, const ttsweep_start *starts, float *const *tt_host)
{
    return ttsweep_solve_multi_changed(ndev, devices, nx, ny, nz, fs, starstart, starstop, v_host, nstart, starts, tt_host,
                                       nullptr);
}

int ttsweep_solve_multi_changed(int ndev, const int *devices, int nx, int ny, int nz,
                                const ttsweep_fs *fs, int starstart, int starstop, const float *v_host,
                                int nstart, const ttsweep_start *starts, float *const *tt_host, int *changed)
{
    if (ndev <= 0 || !devices || !starts || !tt_host || nstart < 0) return set_error("bad arguments");
    std::vector<int> changed_all(std::max(nstart, 0), 0);
    std::vector<int> rc(ndev, 0);
    std::vector<std::string> err(ndev);
    std::vector<std::thread> workers;
    const std::vector<std::vector<int>> shard = cost_balanced_shards(ndev, nx, ny, nz, nstart, starts);
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
            if (r >= 0 && ctx) {        // (the shard's boxes are distinct starts: no two threads write one slot)
                std::vector<int> ch(my_starts.size(), r);
                (void)ttsweep_get_changed(ctx, ch.data(), (int)ch.size());
                for (size_t k = 0; k < ch.size(); k++) changed_all[shard[d][k]] = ch[k];
            }
            if (r < 0) err[d] = ttsweep_last_error();       // thread-local text
            if (ctx) ttsweep_destroy(ctx);
            rc[d] = r;
        });
    }
    for (auto &w : workers) w.join();
    int any = 0;
    for (int d = 0; d < ndev; d++) {
        if (rc[d] < 0) return set_error("device %d: %s", devices[d], err[d].c_str());
        any |= rc[d];
    }
    if (changed) for (int s = 0; s < nstart; s++) changed[s] = changed_all[s];
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
