// ttsweep_ctx.h - the solver context behind the opaque `ttsweep_ctx` of include/ttsweep.h and the
// host-side pieces that work on it.  The host side of libttsweep.so is split by concern:
//   ttsweep_api.cpp     the C ABI (argument checks, host <-> device staging, multi-device threads)
//   ttsweep_plan.cpp    what is decided once per context or solve: padded layouts, which kernel
//                       relaxes a star, the STRIP kernel's items and unit order, capacity
//   ttsweep_driver.cpp  the driver loop of serial_new/sweep-tt-multistart.c:151-170 (without the
//                       break, :168-169): passes enqueued one ahead of the convergence test
// There is no CPU relaxation in any of them: every solve runs HIP kernels or fails.
#pragma once

#include "../../include/ttsweep.h"

#include <hip/hip_runtime.h>

#include <array>
#include <string>
#include <unordered_map>
#include <vector>

#include "pullstar.h"
#include "ttsweep_dev.h"
#include "ttsweep_kernels.h"

namespace ttsweep {

// error text of the calling thread (ttsweep_last_error); returns -1
int set_error(const char *fmt, ...) __attribute__((format(printf, 1, 2)));

} // namespace ttsweep

#define HIPCHK(expr)                                                                   \
    do {                                                                               \
        hipError_t e_ = (expr);                                                        \
        if (e_ != hipSuccess)                                                          \
            return ttsweep::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),    \
                                      __FILE__, __LINE__);                             \
    } while (0)

// ---------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------
// Passes are enqueued one ahead of the convergence test (see ttsweep_solve_device), so the
// per-start "changed" words exist once per pass in flight.
constexpr int PASS_SLOTS = 3;

struct ttsweep_ctx {
    // (types of namespace ttsweep: ttsweep_dev.h)
    using DevLayout = ttsweep::DevLayout; using CellEntry = ttsweep::CellEntry; using FwdEntry = ttsweep::FwdEntry;
    using StripItem = ttsweep::StripItem; using StripPlan = ttsweep::StripPlan; using TileEntry = ttsweep::TileEntry;
    using StartDesc = ttsweep::StartDesc;
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
    long long handoff_max_units = 0;            // (default rule) starts x one-plane units of a start below which workers hand on
    int handoff_default = 3;
    int async_handoff = -1;                     // workers publish successor units themselves (ASYNC_HANDOFF_*; -1: by the size of the solve)
    int async_inunit = -1;                      // in-unit passes of a one-launch STRIP solve (-1: by the number of starts)
    bool exact_half = false;                    // the velocity volume holds sub-limit values: CELL kernel, reference rounding
    int kernel_wanted = 0;                      // ... and the kernel to go back to when a volume without them arrives
    unsigned long long *d_scratch = nullptr;    // four counters for one-off kernels (velocity check, validator)
    CellEntry *d_cell_entries = nullptr;
    int n_cell_entries = 0;
    FwdEntry *d_fwd_entries = nullptr;      // forward star entries (validator)
    int n_fwd_entries = 0;

    // STRIP kernel: (da, db) columns of the star, dead-edge boxes
    // STRIP kernel: the star's items for units of one plane (latency mode, few starts) and of
    // two planes (throughput mode); `np` is the mode of the solve in progress
    StripItem *d_strip_items[ttsweep::STRIP_PLANES] = {nullptr, nullptr};
    StripPlan plans[ttsweep::STRIP_PLANES]{};
    // the latency instance (one-plane units, STRIP_NS_LAT waves): the same items in eight shares
    StripItem *d_strip_items_lat = nullptr;
    StripPlan plan_lat{};
    int unitq_blocks_lat = 0;               // its resident grid (one workgroup per CU)
    int async_waves = -1;                   // TTSWEEP_OPT_ASYNC_WAVES: 4 / 8 waves per unit; -1: by the size of the solve
    long long lat_max_units = 12000;        // (default rule) starts x one-plane units of a start below which the latency instance runs
                                            // (241x241x51: up to 3 starts; measured 1 / 2 / 3 starts 6.0 -> 5.0, 6.8 -> 6.4, 7.6 -> 7.2 ms; 4 starts 8.4 -> 8.7,
                                            // 6 starts 10.3 -> 12.7: profiles/r05_small_shards_and_special.txt)
    int np = ttsweep::STRIP_PLANES;
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
    int4 *d_unitq = nullptr;                // ttsweep::UNITQ_LISTS lists of unitq_cap entries
    size_t unitq_cap = 0;
    int *d_unitq_ctrl = nullptr;            // UNITQ_CTRL_WORDS (counts, cursors)
    int nlists = ttsweep::UNITQ_LISTS;               // unit queues = XCDs of the device (census at create)
    int unitq_blocks = 0;                   // persistent grid: workgroups the device holds at once
    std::vector<std::vector<int>> unit_order;       // per start: unit ids, nearest to the start first
    std::vector<long long> unit_order_key;          // start cell the cached order belongs to
    // Distance gate (see plan_pass_kernel): radius of the first pass and cells it opens per
    // pass.  Defaults follow the star's reach: final values spread at about half the reach
    // per pass (measured, 818-offset star: 3.5 cells/pass gives the shortest solve).
    double gate_speed = 0.0;                // 0: no gate
    double gate_r0 = 0.0;
    // STRIP: bits for units nearer to the start than the improved cells by more than this many cells are
    // deferred until the start is otherwise at rest (push_improved); < -1e30: off
    float defer_margin = 0.375f;                // (cells; measured with the in-unit passes, profiles/r04_schedule_knobs.txt)
    bool defer_suspended = false;           // (pass driver) the solve in progress has flushed once: no more deferral
    // STRIP, one launch per solve (AsyncSolve, ttsweep_dev.h)
    int async_mode = -1;                    // TTSWEEP_OPT_ASYNC
    int async_low = 0, async_high = 0;      // 0: defaults (solve_async_strip)
    int async_special_every = 32;           // (128 until round 5: 24 starts 28.5 -> 28.1 ms with 32, 1 and 8 the same, 512 29.6: profiles/r05_small_shards_and_special.txt)
    int async_timeout_ms = 0;               // TTSWEEP_OPT_ASYNC_TIMEOUT_MILLI (0: from the size of the solve)
    int async_policy = 1;                   // TTSWEEP_OPT_ASYNC_POLICY
    float async_gate_speed = 0.5f;          // cells per round (policy 1; TTSWEEP_OPT_ASYNC_GATE_MILLI)
    float async_gate_fast = 2.0f;           // ... while the workers are running dry (TTSWEEP_OPT_ASYNC_GATE_FAST_MILLI)
    float async_window = 16.f;              // TTSWEEP_OPT_ASYNC_WINDOW_MILLI
    int4 *d_async_list = nullptr;           // the rings' unit lists
    size_t async_list_cap = 0;
    std::vector<long long> async_list_key;  // unit grid, rings and start cells the lists on the device were made for
    ttsweep::AsyncSolve async_rings{};      // ... and their ring offsets
    int *d_async_ring_starts = nullptr;     // ASYNC_MAX_STARTS
    unsigned long long *d_async_entries = nullptr, *d_async_ctl = nullptr;
    unsigned *d_async_status = nullptr, *h_async_status = nullptr;     // (pinned)
    // TILE kernel: the star in device axes, halo of the staged image, launch counter
    TileEntry tile_ent[ttsweep::TILE_MAX_ENT];
    int tile_nent = 0, tile_R = 1, tile_fz = 1;     // entries, max |da|,|db|, max |dc| of the star
    float *d_vface = nullptr, *d_tface = nullptr;   // z faces of v and of every start's T (TILE layout)
    int tile_epoch = 1;
    int tile_blocks = 0;                    // workgroups of the sweep kernel the device holds at once
    ttsweep::TileSweep tile_sweep{};        // launch arguments of the solve in progress
    unsigned long long *d_tile_wgwork = nullptr;    // private work sums of the sweep kernel's workgroups
    size_t tile_wgwork_cap = 0;
    int *d_tile_dmin = nullptr, *h_tile_dmin = nullptr;    // first hyperplane with a due tile, per sweep parity (device / pinned)
    // TILE, plain 6-neighbour star, one launch per solve (ColumnSolve, ttsweep_dev.h)
    ttsweep::ColumnSolve col{};
    unsigned long long *d_col_prog = nullptr, *d_col_claim = nullptr;
    unsigned *d_col_due = nullptr, *d_col_status = nullptr, *h_col_status = nullptr;   // (h_: pinned)
    int *d_col_done = nullptr, *h_col_done = nullptr, *d_col_seqtab = nullptr;
    float **d_col_tptr = nullptr, **h_col_tptr = nullptr;  // the starts' travel-time volumes (h_: pinned)
    float *d_col_line = nullptr, *h_col_line = nullptr;     // per start: velocity at the start, least / largest of its vertical line (sized with tptr)
    unsigned long long *d_col_ordseq = nullptr, *h_col_ordseq = nullptr;   // the starts' sequences of orderings (sized with tptr)
    int col_cap_tptr = 0;
    int col_order = -1;                     // TTSWEEP_OPT_TILE_ORDER: which sequence of the eight orderings each start's sweeps follow (column_order_sequence)
    bool col_in_place_off = false;          // TTSWEEP_OPT_TILE_IN_PLACE = 0: always relax in the library's padded volumes
    int col_cap_starts = 0;                 // starts the buffers above were sized for
    int col_seq_key[3] = {0, 0, 0};         // NI, NJ, sequences the table on the device was made for
    int col_blocks = 0;                     // single-wavefront workgroups the device holds at once
    int tile_blocks_used = 0;               // workgroups whose private work sums the solve in progress has to add up
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

    // ttsweep_solve (host boxes): device staging stack in the caller's layout, kept between calls,
    // and what successful calls returned since the velocity was set: host array -> (start, digest
    // of the converged box)
    float *d_stage = nullptr;
    size_t stage_cap = 0;                   // boxes
    struct Digest {
        unsigned long long a = 0, b = 0;
        bool operator==(const Digest &o) const { return a == o.a && b == o.b; }
    };
    struct SolvedBox { ttsweep_start start; Digest digest; };
    std::unordered_map<const float *, SolvedBox> solved;

    // TTSWEEP_OPT_PREPASS: a context of its own for the sub-star that is relaxed first
    ttsweep_ctx *pre = nullptr;
    int prepass_entries = 0;
    std::vector<ttsweep_fs> fs_copy;        // the caller's star entries [0, starstop)
    int starstart = 0, starstop = 0;

    // options
    bool timing = false;
    long long max_sweeps = 100000;
    int max_batch = 0;                      // cap on starts per ttsweep_solve batch (0: by memory)

    hipEvent_t ev_solve0 = nullptr, ev_solve1 = nullptr;
    hipEvent_t ev_flags[PASS_SLOTS] = {nullptr, nullptr, nullptr};     // "changed" words of a pass are on the host
    std::vector<hipEvent_t> ev_pool;        // pairs around sweep launches
    size_t ev_used = 0;

    ttsweep_stats stats{};
    std::vector<int> batch_changed;         // per start of the device solve in progress: a travel time improved
    std::vector<int> changed_last;          // ... of the last ttsweep_solve / ttsweep_solve_device call
};

namespace ttsweep {

inline int ctx_bind(const ttsweep_ctx *ctx)
{
    HIPCHK(hipSetDevice(ctx->device));
    return 0;
}

// ---- ttsweep_plan.cpp ------------------------------------------------------
int count_xcds(ttsweep_ctx *ctx);                       // ctx->nlists = XCDs of the device
int device_xcds(int device, hipStream_t stream, int *out);   // the same, measured once per device and process
bool kernel_available(const ttsweep_ctx *ctx, int k);   // can kernel variant k relax this star?
int auto_kernel(const ttsweep_ctx *ctx);                // the variant the library picks for it
void make_layout(ttsweep_ctx *ctx);                     // padded layout of ctx->kernel
int upload_star(ttsweep_ctx *ctx);                      // pull star (CELL kernel, dead-edge cells, validator)
int upload_strip_plan(ttsweep_ctx *ctx);                // the STRIP kernel's items, one- and two-plane units
void fill_special_box(const ttsweep_ctx *ctx, StartDesc &sd);   // dead-edge box of one start
size_t flag_words(const DevLayout &L, int kernel);      // activity words per start (of kernel variant `kernel`)
int ensure_capacity(ttsweep_ctx *ctx, int nstart);      // per-solve pools for nstart starts
size_t per_start_device_bytes(const ttsweep_ctx *ctx);  // what ensure_capacity allocates per start
int build_worklist(ttsweep_ctx *ctx, int nactive);      // STRIP: static unit list of the active starts
int ensure_unit_grid(ttsweep_ctx *ctx);                 // STRIP: ctx->unitq_blocks = workgroups the device holds at once
void order_units(const ttsweep_ctx *ctx, const StartDesc &sd, std::vector<int> &order);

// ---- ttsweep_driver.cpp ----------------------------------------------------
// The driver loop on device-resident boxes; returns 1 / 0 / < 0 like ttsweep_solve_device.
int solve_device_body(ttsweep_ctx *ctx, int nstart, const ttsweep_start *starts,
                      float *const *tt_dev, int init);

} // namespace ttsweep
