/* ttsweep.h - C ABI of the MI355X travel-time sweep library (libttsweep.so).
 *
 * This is the drop-in boundary for the hot path of the reference program
 * serial_new/sweep-tt-multistart.c: the file-local call
 *     changed[s] += sweepXYZ(nx, ny, nz, s, 0, starsize-1);        (:160)
 * inside the `while (anychange)` driver loop (:151-170), which reads the
 * file-scope globals fs[], start[], vbox and ttboxes[] (:62-66).  Nothing but
 * plain pointers, ints and the two small POD structs below crosses this
 * boundary; no FLOATBOX/VELOCITYBOX struct and no torch type does.
 *
 * Data layout at the boundary is the reference's FLOATBOX layout
 * (include/floatbox.h:127-129,160): a contiguous float32 array indexed
 * x*ny*nz + y*nz + z.  The library keeps its own padded/permuted device
 * copies; caller buffers are never re-allocated or freed by the library.
 *
 * Semantics.  One library "solve" is the reference's driver loop run to its
 * end: it relaxes every live edge of the forward star until no travel time
 * can improve (the `while (anychange)` loop without the temporary `break` of
 * :168-169; see old/sweep-serial/sweep-tt-multistart.c:189-211).  The
 * converged box does not depend on the order of relaxation, so the GPU is free
 * to use its own schedule and reproduces the serial fixed point bit for bit,
 * including the reference's two quirks (exclusive upper star bound at :160/:206
 * and the skipped edges centred on the start point, :219-221).
 *
 * Error convention: functions returning int return a negative value on a
 * HIP/argument error (text via ttsweep_last_error()); the reference's own
 * "0 = failure" convention stays on the header surface (floatbox.h etc.).
 * There is NO CPU fallback: without a usable HIP device every solve fails.
 */
#ifndef TTSWEEP_H
#define TTSWEEP_H

#ifdef __cplusplus
extern "C" {
#endif

#define TTSWEEP_ABI_VERSION 6      /* 6 (round 5): + ttsweep_solve_multi_changed, options 22-23; 5 (round 4): + ttsweep_get_changed,
                                      ttsweep_solve_multi_device, stats.fallbacks (in the struct's former padding),
                                      options 19-21; every version-4 caller runs unchanged */

/* Forward-star entry: same layout as `struct FS`
 * (serial_new/sweep-tt-multistart.c:46-49).  d must already hold
 * delta * |offset| exactly as the reference main() prepares it (:122,:127). */
typedef struct ttsweep_fs {
    int i, j, k;
    float d;
} ttsweep_fs;

/* Start point: same layout as `struct START` (serial_new/...:56-58). */
typedef struct ttsweep_start {
    int i, j, k;
} ttsweep_start;

/* Counters of the most recent solve on a context. */
typedef struct ttsweep_stats {
    int nstart;                 /* starts in the solve */
    int sweeps_max;             /* passes executed for the slowest start */
    long long sweeps_total;     /* sum over starts of passes launched (a pass relaxes only
                                   the units that are due: whose inputs changed and that the
                                   distance gate has reached; TILE kernel: a pass is one
                                   ordering sweep over the tiles that are due) */
    long long cells_relaxed;    /* cells actually relaxed against the whole star, summed
                                   over passes and starts (= sweeps_total * cells when
                                   nothing is skipped) */
    long long cells;            /* nx*ny*nz */
    long long relaxations_per_sweep; /* in-bounds (cell, offset) pairs one pass relaxes */
    long long launches;         /* sweep-kernel launches (TILE: one per tile hyperplane of a sweep) */
    double sweep_kernel_ms;     /* sum of sweep-kernel durations (HIP events on the
                                   library's stream; 0 unless timing is enabled) */
    double solve_ms;            /* device time of the whole solve (events) */
    int kernel_variant;         /* which sweep kernel ran (TTSWEEP_KERNEL_*) */
    int fallbacks;              /* solves whose one-launch form gave up (a wait inside it ran into its
                                   wall-clock limit) and that the pass / hyperplane driver then finished:
                                   the result is the same, the time is not */
} ttsweep_stats;

typedef struct ttsweep_ctx ttsweep_ctx;

/* option keys for ttsweep_set_option */
#define TTSWEEP_OPT_TIMING        1   /* 1: time every sweep launch with HIP events */
#define TTSWEEP_OPT_KERNEL        2   /* force a kernel variant (TTSWEEP_KERNEL_*) */
#define TTSWEEP_OPT_MAX_SWEEPS    3   /* safety cap on passes per solve (default 100000) */
#define TTSWEEP_OPT_MAX_BATCH     4   /* ttsweep_solve: at most this many starts per device
                                         batch (0 = as many as device memory holds) */
#define TTSWEEP_OPT_GATE_SPEED_MILLI 5 /* schedule only, never the result: cells (x 1/1000) by
                                          which the distance gate of the STRIP kernel opens per
                                          pass; 0 switches the gate off (default: half the
                                          star's reach) */
#define TTSWEEP_OPT_PAIR_MIN_STARTS 7 /* schedule only: the STRIP kernel relaxes units of two planes
                                         from this many starts per solve on, units of one plane
                                         below (0: always two, a huge value: never).  Default
                                         without this option: two planes when the solve offers
                                         enough units to keep the device busy with them (starts x
                                         one-plane units of a start >= 80 000), else one */
#define TTSWEEP_OPT_GATE_R0_MILLI 6   /* schedule only: gate radius of the first pass, cells x 1/1000
                                         (default: the star's reach + 1) */

#define TTSWEEP_OPT_PREPASS_ENTRIES 8  /* schedule only, never the result: relax the first N entries of
                                         the star (fs[starstart .. starstart+N-1]) to their own fixed
                                         point first, then the whole star from that state - the
                                         pre-processing of old/wavefront-openmp/wave-multistart.c:210-215
                                         (there: N = fsindex[3], the entries no longer than 4 cells,
                                         146 of the 818).  0 (default) switches it off. */

#define TTSWEEP_OPT_ASYNC          9   /* schedule only, never the result: 1 = the STRIP kernel - and the TILE kernel for
                                          the plain 6-neighbour star (column pipelines) - runs a solve as ONE
                                         launch (no passes: planner workgroups hand the due units, nearest to
                                         their start first, to the working workgroups through rings in device
                                         memory, and detect convergence on the device); 0 = a launch pair per
                                         pass; -1 (default) = the library chooses */
#define TTSWEEP_OPT_ASYNC_LOW     10  /* schedule only: a ring is refilled when it holds at most this many ... */
#define TTSWEEP_OPT_ASYNC_HIGH    11  /* ... up to this many units (0: defaults from the grid of workgroups) */
#define TTSWEEP_OPT_ASYNC_SPECIAL 12  /* schedule only: units of a start between two relaxations of its
                                         dead-edge cells in a one-launch solve (default 32) */
#define TTSWEEP_OPT_ASYNC_POLICY  13  /* schedule only: how a planner hands units out in a one-launch solve.
                                         0 = every refill of its ring starts at the unit nearest to the start
                                         (strict priority by distance); 1 (default) = the scan goes round and
                                         round the list, a unit at most once per round, behind the distance
                                         gate (TTSWEEP_OPT_GATE_*: per round instead of per pass) */
#define TTSWEEP_OPT_DEFER_MARGIN_MILLI 14 /* schedule only, never the result (STRIP kernel): an improvement is
                                         reported at once only to the units that are not nearer to the start
                                         than the improved cells by more than this many cells (x 1/1000, may be
                                         negative); the units behind the front hear of it when the start is
                                         otherwise at rest, once, instead of in every pass.  Default 375 (3/8 of a cell);
                                         <= -1000000000 switches the deferral off */
#define TTSWEEP_OPT_ASYNC_WINDOW_MILLI 15 /* schedule only: ring policy 2 - cells (x 1/1000) beyond the nearest unit
                                         with anything to do up to which a start's units are handed out; 0 = no
                                         gate */
#define TTSWEEP_OPT_ASYNC_GATE_MILLI 16 /* schedule only: ring policy 1 - cells (x 1/1000) by which the distance gate
                                         opens per round of a one-launch solve (default 500; 0 = no gate;
                                         TTSWEEP_OPT_GATE_SPEED_MILLI = 0 switches this gate off as well) */
#define TTSWEEP_OPT_ASYNC_GATE_FAST_MILLI 17 /* schedule only: ring policy 1 - cells (x 1/1000) by which the gate opens
                                         in a round that begins with an empty ring (the workers are running dry);
                                         never less than TTSWEEP_OPT_ASYNC_GATE_MILLI */
#define TTSWEEP_OPT_ASYNC_TIMEOUT_MILLI 18 /* wall-clock limit (ms) of every wait inside a one-launch solve; when one
                                         runs into it the launch drains and the pass driver finishes the solve from
                                         the boxes as they stand (same result).  0 (default): ten seconds plus
                                         twenty times the solve's expected duration */

#define TTSWEEP_OPT_TILE_IN_PLACE 19 /* schedule only, never the result (TILE kernel, one launch per solve): 1 (default) =
                                         relax the travel times in the caller's own device arrays when their rows are
                                         whole tiles long (nz % 32 == 0) and 64-byte aligned - no padded copy, no copy
                                         back; 0 = always in the library's padded volumes */

#define TTSWEEP_OPT_QUEUES 20        /* schedule only, never the result: unit queues / planner rings / claim sequences of
                                         a solve, 1 .. 8 (default: the XCDs the device shows, counted at create - 8 on a
                                         whole MI355X, fewer on a partition).  A one-launch solve serves at most 32 starts
                                         per ring: with more starts per queue than that the launch-per-pass driver runs */

#define TTSWEEP_OPT_ASYNC_INUNIT 21   /* schedule only, never the result (STRIP kernel, one launch per solve): how often
                                         a unit that improved is relaxed again, at once, against its OWN planes - the
                                         values it has just stored - before it is handed back (-1, the default: 2 for
                                         solves of 2 and more starts, 0 for a single start and for the eight-wave instance of small shards; 0 .. 8) */

#define TTSWEEP_OPT_ASYNC_HANDOFF 22  /* schedule only, never the result (STRIP kernel, one launch per solve): direct
                                         hand-off - a worker that has improved a plane not only tells the units that
                                         stage it, it also puts the idle ones among them (inside the distance gate) into
                                         the ring itself instead of leaving them to the planner's next scan (1), and
                                         likewise its own unit when bits arrived while it was being relaxed (2; 3 = both).
                                         -1 (default): the library chooses by the size of the solve (small shards: 3);
                                         0 = only the planners publish */

#define TTSWEEP_OPT_ASYNC_WAVES 23    /* schedule only, never the result (STRIP kernel, one launch per solve, one-plane
                                         units): wavefronts that relax a unit - 4 (two workgroups per CU, two units per CU
                                         at a time: throughput) or 8 (one workgroup per CU, all of a CU's wavefronts on
                                         one unit, four planes staged ahead: a hop of the front takes about half as
                                         long - for shards too small to fill the machine).  -1 (default): by the size
                                         of the solve */

#define TTSWEEP_OPT_TILE_ORDER 24     /* schedule only, never the result (TILE kernel, one launch per solve): which
                                         sequence of the eight orderings (+-x, +-y, +-z) the sweeps of each start follow:
                                         table (0 .. 9) + 10 x the corner the first sweep begins at (0: the grid's origin,
                                         1: the corner nearest to the start, 2: the farthest) + 100 x which axis plays
                                         which role of the table (0 .. 4) - column_order_sequence() in
                                         csrc/ttsweep_column.hip (0: the sequence of rounds 3 - 4).  -1 (default): per start, by where
                                         the start lies in the velocity profile of its vertical line (115: z flips with
                                         every sweep - sources at the slow end of a medium that gets faster with depth;
                                         111 for a start at the fast end) - column_order_default() */

#define TTSWEEP_KERNEL_AUTO       0
#define TTSWEEP_KERNEL_CELL       1   /* one thread per cell, star from global memory */
#define TTSWEEP_KERNEL_STRIP      2   /* LDS-staged plane slabs, register strips */
#define TTSWEEP_KERNEL_TILE       3   /* ordered (8-ordering Gauss-Seidel) tile sweeps for small
                                         stars: the HBM-bound regime */

/* ---- information ------------------------------------------------------- */
int ttsweep_abi_version(void);
/* number of HIP devices, or a negative value when HIP cannot be initialised */
int ttsweep_device_count(void);
/* text of the most recent error on this thread ("" if none) */
const char *ttsweep_last_error(void);

/* Optional: start initialising the HIP runtime for `device` on a background thread and
 * return at once (ttsweep_create waits for it).  A host program calls it first thing, so
 * that the few hundred milliseconds a process pays at its first HIP call run beside its
 * own file reading (serial_new/...:77-147) instead of inside its first sweepXYZ call.
 * Never needed for correctness.  Returns 0. */
int ttsweep_warmup(int device);

/* ---- context ----------------------------------------------------------- */
/* Create a solver for an nx*ny*nz grid and the star entries
 * fs[starstart .. starstop-1] (EXCLUSIVE upper bound, as the reference call
 * site passes starsize-1, :160).  Uploads the star to `device`.
 * Replaces: the globals fs[] / nx,ny,nz and the (starstart, starstop)
 * arguments of sweepXYZ (:198).  Returns NULL on failure. */
ttsweep_ctx *ttsweep_create(int device, int nx, int ny, int nz,
                            const ttsweep_fs *fs, int starstart, int starstop);
void ttsweep_destroy(ttsweep_ctx *ctx);

int ttsweep_set_option(ttsweep_ctx *ctx, int key, long long value);

/* Velocity volume (the global `vbox.box.flat`, :65), host or device memory,
 * FLOATBOX layout.  The library keeps its own device copy.  Every value must be finite
 * and >= 0 (zero is accepted as the reference accepts it; a negative velocity, for which
 * the reference's loop :151-170 need not terminate, Inf and NaN are refused: < 0).
 * A volume with a positive value below 2^-124 / (smallest fs[].d of the star) - about 4.7e-39
 * for the reference's delta of 10 - is accepted and solved bit for bit like every other, but
 * slowly: there a delay d * (v[c] + v[o]) can be a denormal number, where the reference's
 * "/ 2.0" of the rounded product (:216) and the fast kernels' multiplication by d / 2 no
 * longer agree in the last bit, so such a volume goes to the per-cell kernel's instance that
 * rounds as the reference does (ttsweep_stats.kernel_variant reads TTSWEEP_KERNEL_CELL, whatever
 * TTSWEEP_OPT_KERNEL asked for; the next volume without such values gets the chosen kernel back). */
int ttsweep_set_velocity(ttsweep_ctx *ctx, const float *v_host);
int ttsweep_set_velocity_device(ttsweep_ctx *ctx, const float *v_dev);

/* ---- the hot path ------------------------------------------------------ */
/* Relax nstart travel-time boxes to convergence.
 *   starts[s]  : the start point of box s (global start[], :63)
 *   tt[s]      : box s (global ttboxes[s].flat, :66), FLOATBOX layout; read as
 *                the initial state and overwritten with the converged state.
 * Returns 1 if any travel time improved, 0 if every box was already converged
 * (the two outcomes `anychange != 0` / `== 0` of :163-166), < 0 on error.
 * Replaces: the whole `while (anychange)` loop of :151-170 over all starts.
 * A call with exactly the arrays and starts of the previous successful call on this context,
 * their contents bit for bit as that call left them (checked with a 128-bit digest of every
 * box; velocity, star and kernel unchanged), is answered with 0 without any device work: it
 * is the confirming pass of a reference-style driver loop. */
int ttsweep_solve(ttsweep_ctx *ctx, int nstart, const ttsweep_start *starts,
                  float *const *tt_host);

/* Same, with the boxes in device memory (pointers are device addresses held
 * in a host array).  If init != 0 the incoming contents are ignored and every
 * box starts from the reference initial state (all +INFINITY, start = 0;
 * serial_new/...:139-144), which is then done on the device. */
int ttsweep_solve_device(ttsweep_ctx *ctx, int nstart, const ttsweep_start *starts,
                         float *const *tt_dev, int init);

int ttsweep_get_stats(const ttsweep_ctx *ctx, ttsweep_stats *out);

/* Per-start outcome of the last ttsweep_solve / ttsweep_solve_device call of this context: out[s] = 1 when a
 * travel time of start s improved, 0 when its box was at its fixed point already - what the reference's driver
 * prints and sums as changed[s] (serial_new/sweep-tt-multistart.c:158-164), where the call's return value is the
 * OR over the starts.  Writes min(n, starts of that call) entries and returns their number (< 0 on bad
 * arguments). */
int ttsweep_get_changed(const ttsweep_ctx *ctx, int *out, int n);

/* On-device fixed-point check in the spirit of testconvergence
 * (old/wavefront-openmp/wave-multistart.c:300-347) on serial_new's edge set:
 *   open_edges         (cell, offset) pairs through which one more reference sweep would
 *                      still store (serial_new/...:219-249, evaluated without modifying
 *                      the box): 0 iff nothing can improve any more;
 *   cells_infinite     cells still at INFINITY;
 *   cells_unsupported  cells (other than the start) whose travel time is smaller than
 *                      every candidate their live edges offer, i.e. that no store of
 *                      :222-223 / :246-247 can have produced: 0 iff nothing is too small.
 * open_edges == 0 and cells_unsupported == 0 together pin the box to the one fixed point
 * of the relaxation (all delays positive), which is what the reference's loop :151-170
 * converges to.  tt_dev: device memory, FLOATBOX layout.  For grids where no CPU oracle
 * run is feasible (one reference sweep of 1024x1024x512 takes ~40 min).  Any of the
 * three result pointers may be NULL.  Returns 0 on success, < 0 on error. */
int ttsweep_validate_device(ttsweep_ctx *ctx, const ttsweep_start *start, const float *tt_dev,
                            long long *open_edges, long long *cells_infinite,
                            long long *cells_unsupported);

/* Multi-GPU form of ttsweep_solve for a host program: the start points are
 * independent (serial_new/...:158-162; mpi/backup.c:351-363 runs one start per
 * rank), so the starts are dealt over the devices, longest first by estimated cost
 * (distance to the farthest grid corner), at most ceil(nstart / ndev) per device; every device gets its own
 * context and copy of the velocity volume, there is no communication while
 * sweeping, and each device writes its converged boxes straight into the caller's
 * host arrays.  devices may name the same GPU more than once.  Returns 1 / 0 / < 0
 * like ttsweep_solve.  (With the boxes resident in HBM the gather is a collective
 * instead: see bench.py / multistart.py, RCCL over xGMI.) */
int ttsweep_solve_multi(int ndev, const int *devices, int nx, int ny, int nz,
                        const ttsweep_fs *fs, int starstart, int starstop, const float *v_host,
                        int nstart, const ttsweep_start *starts, float *const *tt_host);
/* ... with the per-start outcome (ABI 6): changed[s] = 1 when a travel time of start s improved, 0 when its box was
 * at its fixed point already - what the reference's driver prints and sums as changed[s]
 * (serial_new/sweep-tt-multistart.c:158-164); changed may be NULL (then this is ttsweep_solve_multi). */
int ttsweep_solve_multi_changed(int ndev, const int *devices, int nx, int ny, int nz,
                                const ttsweep_fs *fs, int starstart, int starstop, const float *v_host,
                                int nstart, const ttsweep_start *starts, float *const *tt_host, int *changed);

/* The same with the result set RESIDENT ON A DEVICE - the step the reference's MPI version left as a TODO
 * (mpi/backup.c:381-386: "gather the ttboxes"; its CUDA version moves boxes between devices with peer copies,
 * cuda/cudasweep-tt-multistart.cu:359-369).  The starts are sharded over `devices` as above; every device
 * initialises and solves its shard in its own memory (fresh boxes: +INFINITY, 0 at the start), and the converged
 * boxes are gathered on devices[0], the root: tt_root[s] = device address ON THE ROOT of box s (nx*ny*nz floats,
 * allocated by the caller: the root holds the whole set - for result sets beyond its memory use ttsweep_solve_multi,
 * whose boxes live in host memory).  The root's own starts are solved in their slots.  The gather is ONE group of
 * ncclSend / ncclRecv pairs (RCCL over xGMI: one communicator per listed device, ncclCommInitAll; librccl is
 * loaded at run time) or - where RCCL is missing, refuses the list (a device listed twice) or fails - peer copies
 * (hipMemcpyPeerAsync).  flags: TTSWEEP_MULTI_*.  changed (may be NULL): per start, as ttsweep_get_changed.
 * gather_path (may be NULL): TTSWEEP_GATHER_*.  Returns 1 / 0 / < 0 like ttsweep_solve. */
#define TTSWEEP_MULTI_LOOPBACK 1    /* testing aid: the root's own boxes travel too (a send to itself) - the collective
                                       path then runs on a single device */
#define TTSWEEP_MULTI_NO_RCCL  2    /* peer copies even where RCCL is available */
#define TTSWEEP_GATHER_NONE 0       /* every box was solved on the root */
#define TTSWEEP_GATHER_RCCL 1
#define TTSWEEP_GATHER_PEER 2
int ttsweep_solve_multi_device(int ndev, const int *devices, int nx, int ny, int nz, const ttsweep_fs *fs, int starstart,
                               int starstop, const float *v_host, int nstart, const ttsweep_start *starts,
                               float *const *tt_root, int flags, int *changed, int *gather_path);

/* One-call drop-in for the reference's
 *   int sweepXYZ(int nx,int ny,int nz,int s,int starstart,int starstop)  (:198)
 * with the globals it reads passed explicitly: v = vbox.box.flat,
 * tt = ttboxes[s].flat, fs = fs, (si,sj,sk) = start[s].  Runs to convergence
 * on device 0 and returns > 0 if anything improved, 0 if not (so the
 * reference driver loop terminates after the next call), < 0 on error.
 * Creates and destroys a context per call; use the context API for batches. */
int ttsweep_sweepXYZ(const float *v, float *tt, int nx, int ny, int nz,
                     const ttsweep_fs *fs, int starstart, int starstop,
                     int si, int sj, int sk);

/* ---- host-only helpers (no device needed; used by the CPU test tier) ---- */
/* Build the pull form of the star: fills up to cap entries of
 * (di,dj,dk,flags,h) and returns the entry count (also when cap is too
 * small), < 0 on error.  flags bit0: live unless the centre cell is the
 * start; bit1: live unless the neighbour is the start. */
typedef struct ttsweep_pull_entry {
    int di, dj, dk;
    int flags;
    float h;        /* d/2, so delay = h * (v[c] + v[o]) */
} ttsweep_pull_entry;
int ttsweep_build_pull_star(const ttsweep_fs *fs, int starstart, int starstop,
                            ttsweep_pull_entry *out, int cap);

/* In-bounds (cell, offset) pairs of one reference pass: the closed form
 * sum over l in [starstart,starstop) of prod_axis max(n_axis - |off|, 0). */
long long ttsweep_relaxations_per_sweep(int nx, int ny, int nz, const ttsweep_fs *fs,
                                        int starstart, int starstop);

#ifdef __cplusplus
}
#endif

#endif /* TTSWEEP_H */
