// ttsweep_api.cpp - host side of libttsweep.so: the C ABI of include/ttsweep.h.
//
// Replaces the driver loop of serial_new/sweep-tt-multistart.c:151-170 and its
// callee sweepXYZ (:198-256) by device-resident relaxation to convergence.
// There is no CPU fallback in this file: every solve runs HIP kernels or fails.
#include "../../include/ttsweep.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
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
struct ttsweep_ctx {
    int device = 0;
    int nx = 0, ny = 0, nz = 0;
    hipStream_t stream = nullptr;

    std::vector<ttsweep_pull_entry> pull;   // user-axis pull star
    int radius = 0;
    long long relax_per_sweep = 0;

    DevLayout L{};
    int kernel = TTSWEEP_KERNEL_CELL;
    int forced_kernel = TTSWEEP_KERNEL_AUTO;

    float *d_v = nullptr;                   // padded velocity
    bool have_v = false;
    CellEntry *d_cell_entries = nullptr;
    int n_cell_entries = 0;

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
    int batch_sweeps = 1;

    hipEvent_t ev_solve0 = nullptr, ev_solve1 = nullptr;
    std::vector<hipEvent_t> ev_pool;        // pairs around sweep launches
    size_t ev_used = 0;

    ttsweep_stats stats{};
};

static int ctx_bind(const ttsweep_ctx *ctx)
{
    HIPCHK(hipSetDevice(ctx->device));
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

static int ensure_capacity(ttsweep_ctx *ctx, int nstart)
{
    if (nstart <= ctx->capacity_starts) return 0;
    if (ctx->d_T) HIPCHK(hipFree(ctx->d_T));
    if (ctx->d_starts) HIPCHK(hipFree(ctx->d_starts));
    if (ctx->d_active) HIPCHK(hipFree(ctx->d_active));
    if (ctx->d_changed) HIPCHK(hipFree(ctx->d_changed));
    if (ctx->h_starts) HIPCHK(hipHostFree(ctx->h_starts));
    if (ctx->h_active) HIPCHK(hipHostFree(ctx->h_active));
    if (ctx->h_changed) HIPCHK(hipHostFree(ctx->h_changed));
    ctx->d_T = nullptr; ctx->d_starts = nullptr; ctx->d_active = nullptr; ctx->d_changed = nullptr;
    ctx->h_starts = nullptr; ctx->h_active = nullptr; ctx->h_changed = nullptr;
    ctx->capacity_starts = 0;
    HIPCHK(hipMalloc((void **)&ctx->d_T, (size_t)nstart * ctx->L.cells * sizeof(float)));
    HIPCHK(hipMalloc((void **)&ctx->d_starts, nstart * sizeof(StartDesc)));
    HIPCHK(hipMalloc((void **)&ctx->d_active, nstart * sizeof(int)));
    HIPCHK(hipMalloc((void **)&ctx->d_changed, nstart * sizeof(int)));
    HIPCHK(hipHostMalloc((void **)&ctx->h_starts, nstart * sizeof(StartDesc)));
    HIPCHK(hipHostMalloc((void **)&ctx->h_active, nstart * sizeof(int)));
    HIPCHK(hipHostMalloc((void **)&ctx->h_changed, nstart * sizeof(int)));
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

// One full-grid pass for the active starts.
static int launch_pass(ttsweep_ctx *ctx, int nactive)
{
    hipEvent_t e0, e1;
    if (ctx->timing && timed_event(ctx, &e0)) return -1;
    HIPCHK(launch_sweep_cell(ctx->L, ctx->d_v, ctx->d_starts, ctx->d_active, nactive,
                             ctx->d_changed, ctx->d_cell_entries, ctx->n_cell_entries,
                             ctx->stream));
    if (ctx->timing && timed_event(ctx, &e1)) return -1;
    ctx->stats.launches++;
    return 0;
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
    ctx->relax_per_sweep = ttsweep_relaxations_per_sweep(nx, ny, nz, fs, starstart, starstop);
    make_layout_cell(ctx);

    bool ok = hipSetDevice(device) == hipSuccess
           && hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) == hipSuccess
           && hipEventCreate(&ctx->ev_solve0) == hipSuccess
           && hipEventCreate(&ctx->ev_solve1) == hipSuccess
           && hipMalloc((void **)&ctx->d_v, (size_t)ctx->L.cells * sizeof(float)) == hipSuccess;
    if (!ok) {
        set_error("ttsweep_create: HIP setup failed: %s", hipGetErrorString(hipGetLastError()));
        ttsweep_destroy(ctx);
        return nullptr;
    }
    if (upload_star(ctx)) {
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
    (void)hipFree(ctx->d_T);
    (void)hipFree(ctx->d_starts);
    (void)hipFree(ctx->d_active);
    (void)hipFree(ctx->d_changed);
    if (ctx->h_starts) (void)hipHostFree(ctx->h_starts);
    if (ctx->h_active) (void)hipHostFree(ctx->h_active);
    if (ctx->h_changed) (void)hipHostFree(ctx->h_changed);
    for (hipEvent_t e : ctx->ev_pool) (void)hipEventDestroy(e);
    if (ctx->ev_solve0) (void)hipEventDestroy(ctx->ev_solve0);
    if (ctx->ev_solve1) (void)hipEventDestroy(ctx->ev_solve1);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int ttsweep_set_option(ttsweep_ctx *ctx, int key, long long value)
{
    if (!ctx) return set_error("null context");
    switch (key) {
    case TTSWEEP_OPT_TIMING: ctx->timing = value != 0; return 0;
    case TTSWEEP_OPT_KERNEL:
        if (value != TTSWEEP_KERNEL_AUTO && value != TTSWEEP_KERNEL_CELL)
            return set_error("kernel variant %lld not available", value);
        ctx->forced_kernel = (int)value;
        return 0;
    case TTSWEEP_OPT_MAX_SWEEPS:
        if (value <= 0) return set_error("max sweeps must be positive");
        ctx->max_sweeps = value;
        return 0;
    case TTSWEEP_OPT_BATCH_SWEEPS:
        if (value <= 0 || value > 1024) return set_error("batch sweeps out of range");
        ctx->batch_sweeps = (int)value;
        return 0;
    default: return set_error("unknown option %d", key);
    }
}

int ttsweep_set_velocity_device(ttsweep_ctx *ctx, const float *v_dev)
{
    if (!ctx || !v_dev) return set_error("null argument");
    if (ctx_bind(ctx)) return -1;
    HIPCHK(launch_pack(ctx->L, v_dev, ctx->d_v, 0.0f, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
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
    hipError_t e = hipMemcpy(tmp, v_host, bytes, hipMemcpyHostToDevice);
    int rc = 0;
    if (e != hipSuccess) rc = set_error("velocity upload failed: %s", hipGetErrorString(e));
    else rc = ttsweep_set_velocity_device(ctx, tmp);
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
    if (ensure_capacity(ctx, nstart)) return -1;

    const DevLayout &L = ctx->L;
    HIPCHK(hipEventRecord(ctx->ev_solve0, ctx->stream));

    for (int s = 0; s < nstart; s++) {
        const int u[3] = {starts[s].i, starts[s].j, starts[s].k};
        if (u[0] < 0 || u[0] >= ctx->nx || u[1] < 0 || u[1] >= ctx->ny || u[2] < 0
            || u[2] >= ctx->nz)
            return set_error("start %d (%d,%d,%d) outside the grid", s, u[0], u[1], u[2]);
        StartDesc &sd = ctx->h_starts[s];
        sd.T = ctx->d_T + (size_t)s * L.cells;
        sd.sa = u[L.perm[0]];
        sd.sb = u[L.perm[1]];
        sd.sc = u[L.perm[2]];
        sd.sidx = dev_index(L, sd.sa, sd.sb, sd.sc);
        sd.pad_ = 0;
        if (init) HIPCHK(launch_init_tt(L, sd.T, sd.sidx, ctx->stream));
        else HIPCHK(launch_pack(L, tt_dev[s], sd.T, INFINITY, ctx->stream));
        ctx->h_active[s] = s;
    }
    HIPCHK(hipMemcpyAsync(ctx->d_starts, ctx->h_starts, nstart * sizeof(StartDesc),
                          hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(ctx->d_active, ctx->h_active, nstart * sizeof(int),
                          hipMemcpyHostToDevice, ctx->stream));

    // driver loop: serial_new/...:151-170 without the break (:168-169)
    std::vector<int> sweeps(nstart, 0);
    int nactive = nstart;
    bool anychange_ever = false;
    while (nactive > 0) {
        HIPCHK(hipMemsetAsync(ctx->d_changed, 0, nstart * sizeof(int), ctx->stream));
        if (launch_pass(ctx, nactive)) return -1;
        HIPCHK(hipMemcpyAsync(ctx->h_changed, ctx->d_changed, nstart * sizeof(int),
                              hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        int keep = 0;
        for (int a = 0; a < nactive; a++) {
            const int s = ctx->h_active[a];
            sweeps[s]++;
            if (ctx->h_changed[s]) {
                anychange_ever = true;
                if (sweeps[s] >= ctx->max_sweeps)
                    return set_error("start %d did not converge in %lld sweeps", s, ctx->max_sweeps);
                ctx->h_active[keep++] = s;
            }
        }
        if (keep != nactive && keep > 0)
            HIPCHK(hipMemcpyAsync(ctx->d_active, ctx->h_active, keep * sizeof(int),
                                  hipMemcpyHostToDevice, ctx->stream));
        nactive = keep;
    }

    for (int s = 0; s < nstart; s++)
        HIPCHK(launch_unpack(L, ctx->h_starts[s].T, tt_dev[s], ctx->stream));
    HIPCHK(hipEventRecord(ctx->ev_solve1, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));

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
    }
    return anychange_ever ? 1 : 0;
}

int ttsweep_solve(ttsweep_ctx *ctx, int nstart, const ttsweep_start *starts,
                  float *const *tt_host)
{
    if (!ctx || !starts || !tt_host || nstart < 0) return set_error("bad arguments");
    if (ctx_bind(ctx)) return -1;
    if (nstart == 0) return 0;
    const size_t cells = (size_t)ctx->nx * ctx->ny * ctx->nz;
    float *stage = nullptr;
    HIPCHK(hipMalloc((void **)&stage, (size_t)nstart * cells * sizeof(float)));
    std::vector<float *> ptrs(nstart);
    int rc = 0;
    for (int s = 0; s < nstart && rc == 0; s++) {
        ptrs[s] = stage + (size_t)s * cells;
        hipError_t e = hipMemcpy(ptrs[s], tt_host[s], cells * sizeof(float), hipMemcpyHostToDevice);
        if (e != hipSuccess) rc = set_error("travel-time upload failed: %s", hipGetErrorString(e));
    }
    if (rc == 0) rc = ttsweep_solve_device(ctx, nstart, starts, ptrs.data(), 0);
    if (rc >= 0) {
        for (int s = 0; s < nstart; s++) {
            hipError_t e = hipMemcpy(tt_host[s], ptrs[s], cells * sizeof(float), hipMemcpyDeviceToHost);
            if (e != hipSuccess) {
                rc = set_error("travel-time download failed: %s", hipGetErrorString(e));
                break;
            }
        }
    }
    (void)hipFree(stage);
    return rc;
}

int ttsweep_get_stats(const ttsweep_ctx *ctx, ttsweep_stats *out)
{
    if (!ctx || !out) return set_error("null argument");
    *out = ctx->stats;
    return 0;
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
