// ttsweep_kernels.h - launchers of the HIP kernels (ttsweep_kernels.hip).
#pragma once

#include "ttsweep_dev.h"

namespace ttsweep {

// ---- layout conversion / initialisation -----------------------------------
// user FLOATBOX array -> padded device volume; halo cells get `halo_value`.
hipError_t launch_pack(const DevLayout &L, const float *user, float *padded,
                       float halo_value, hipStream_t st);
// padded device volume -> user FLOATBOX array (interior only).
hipError_t launch_unpack(const DevLayout &L, const float *padded, float *user,
                         hipStream_t st);
// reference initial state: everything +INFINITY, the start cell 0
// (serial_new/sweep-tt-multistart.c:139-144).
hipError_t launch_init_tt(const DevLayout &L, float *padded, long long sidx,
                          hipStream_t st);

// the same for all starts of a solve in one launch each: volume s at T0 + s * L.cells (L.cells % 4 == 0), start
// cell starts[s].sidx; users[s] = device address of box s (a device array)
hipError_t launch_init_tt_batch(const DevLayout &L, float *T0, const StartDesc *starts, int nstart, hipStream_t st);
hipError_t launch_unpack_batch(const DevLayout &L, const float *padded0, float *const *users, int nstart, hipStream_t st);

// ---- device census / input check -------------------------------------------
// *seen |= 1 << (XCD id) for every workgroup of an `nblocks`-workgroup launch
hipError_t launch_xcc_census(unsigned *seen, int nblocks, hipStream_t st);
// *bad += cells of the caller's n-cell velocity volume that are negative, not finite, or
// positive but below `tiny` (bad[1]; bad[0]: negative, infinite, NaN)
hipError_t launch_count_bad_velocity(const float *v, long long n, float tiny, unsigned long long *bad, hipStream_t st);

// ---- sweep, variant CELL ---------------------------------------------------
// One chaotic in-place pull pass over the whole grid for the `nactive` starts
// listed in `active`; changed[s] is OR-ed with 1 when any cell of start s
// improved.  exact: delays rounded as the reference rounds them - product first, then halved - for velocity
// volumes whose products can be denormal numbers (ttsweep_set_velocity decides).
hipError_t launch_sweep_cell(const DevLayout &L, const float *v, const StartDesc *starts,
                             const int *active, int nactive, int *changed,
                             const CellEntry *entries, int nentries, bool exact, hipStream_t st);

// ---- validator: counts[0] += (cell, forward entry) pairs a reference sweep would still
// store through, counts[1] += cells still at +INFINITY, counts[2] += cells whose travel
// time no live edge can have produced (T is one padded volume)
hipError_t launch_validate(const DevLayout &L, const float *v, const float *T, long long sidx,
                           const FwdEntry *entries, int nentries, const CellEntry *cell_entries,
                           int ncell_entries, unsigned long long *counts, bool exact, hipStream_t st);

// ---- sweep, variant STRIP --------------------------------------------------
// Same contract as launch_sweep_cell, but cells inside a start's dead-edge box
// (StartDesc::box_*) are left untouched by the unit relaxation; the same kernel relaxes
// exactly those cells with the full liveness rule (one wave per cell).
//
// A pass = launch_plan_pass + launch_sweep_units, nothing else: the last workgroup of
// sweep_units hands the "changed" words to the host and clears the counters (ctrl: the
// UNITQ_CTRL_WORDS queue words + one "workgroups done" word, all zero before the first
// pass) and the next pass's "changed" words.
// plan_pass: one thread per entry of the static work list `work`: work[i] = (start index,
// unit id = (A*btiles + bt)*cstrips + cs) or unit id < 0 for padding; entry i belongs to
// XCD i % nlists.  A unit is due when its pend word holds staged-plane bits (pushed by the
// units that improved those planes: push_improved) and the distance gate has reached it; due
// units are appended as (start, unit, planes) to the queue of their XCD, their word cleared.
// sweep_units: a persistent grid (`nblocks` workgroups) drains the queues, own XCD first.
size_t units_lds_bytes(int waves);
int units_wgs_per_cu();     // persistent workgroups per CU the unit kernel is built for
hipError_t launch_plan_pass(const DevLayout &L, const StartDesc *starts, const int2 *work,
                            long long nwork, int *changed, int4 *lists, int list_cap, int nlists,
                            int *ctrl, const StripPlan &plan, float gate_r2, int *flags0, long long flags_stride,
                            hipStream_t st);      // (flags0 + s * flags_stride = starts[s].tile_flags)
hipError_t launch_sweep_units(const DevLayout &L, const float *v, const StartDesc *starts,
                              const int4 *lists, int list_cap, int nlists, int *ctrl, int nblocks,
                              int *changed, const StripItem *items, const StripPlan &plan,
                              const UnitPassTail &tail, hipStream_t st);
// One launch per solve (AsyncSolve, ttsweep_dev.h): the first as.nrings workgroups plan, the others
// relax; returns when every ring is at rest.  tail: only entries / nentries / max_box_cells are used.
// waves: STRIP_NS (two workgroups per CU) or - units of one plane - STRIP_NS_LAT (the latency instance)
hipError_t launch_solve_units(const DevLayout &L, const float *v, const StartDesc *starts, int nblocks,
                              int *changed, const StripItem *items, const StripPlan &plan, int waves,
                              const UnitPassTail &tail, const AsyncSolve &as, int *flags0, long long flags_stride,
                              hipStream_t st);
// pend |= defer, defer = 0 for the units of the listed starts (np planes per unit); changed[s] |=
// CHANGED_PENDING where a bit moved (push_improved: bits for units nearer to the start are deferred)
hipError_t launch_flush_deferred(const DevLayout &L, int np, int *flags0, long long flags_stride,
                                 const int *active, int nactive, int *changed, hipStream_t st);
// First activity words of a start: from_box = false: only the start's patch is a source;
// from_box = true: every patch that holds a finite travel time is one.  (ra, np: the reach of
// the star along the plane axis and the planes per unit of the solve.)
hipError_t launch_init_tile_flags(const DevLayout &L, const StartDesc &sd, bool from_box, int ra, int np,
                                  hipStream_t st);

// ... for all starts of a solve at once (fresh boxes: the start's patch is the only source); start s's words at
// flags0 + s * stride
hipError_t launch_init_tile_flags_batch(const DevLayout &L, int *flags0, long long stride, const StartDesc *starts,
                                        int nstart, int ra, int np, hipStream_t st);

// ---- sweep, variant TILE ---------------------------------------------------
// One call = the tiles of one hyperplane of an ordering sweep (TileSweep), ONE kernel: a grid of
// P.nblocks single-wavefront workgroups (what tile_sweep_wgs_per_cu says the device holds at
// once); workgroup b evaluates the candidates (active start, J', K') number b, b + nblocks, ...
// - a tile is relaxed only if one of its 27 neighbours improved since it was last relaxed
// (StartDesc::tile_flags holds two words per tile) - and relaxes the due ones; its work sums go
// to private slots (P.wgwork) that launch_tile_reduce_work adds to the starts' counters.  A sweep = the calls D = 0 .. NI + NJ + NK - 3
// in stream order.  changed[s] |= 1 when a tile of start s improved: a whole sweep without a
// change proves convergence.
size_t tile_lds_bytes(int R);
// faces[...] = the z faces (ttsweep_dev.h: tile_face_index) of a padded volume
hipError_t launch_build_tile_faces(const DevLayout &L, const float *padded, float *faces, int fz, hipStream_t st);
// the faces of an initialised box (+INFINITY everywhere, 0 at the start cell (sa, sb, sc))
hipError_t launch_init_tile_faces(const DevLayout &L, float *faces, int fz, int sa, int sb, int sc, hipStream_t st);
hipError_t tile_sweep_wgs_per_cu(const TileSweep &P, int *wgs);
hipError_t launch_tile_sweep(const TileSweep &P, hipStream_t st);
// work0[3 s], work0[3 s + 2] += the workgroups' private sums wgwork[block][s][0 / 1]; slots cleared
hipError_t launch_tile_reduce_work(unsigned long long *wgwork, int nblocks, int nstart, unsigned long long *work0, hipStream_t st);
// from_box = false: only the start's tile counts as changed; true: every tile does.
hipError_t launch_init_tile_state(const DevLayout &L, const StartDesc &sd, bool from_box, hipStream_t st);

// ---- sweep, variant TILE, plain 6-neighbour star: one launch per solve (ttsweep_column.hip) ----
// is this the plain 6-neighbour star (pull entries x-, y-, z-, z+, y+, x+, every edge live in both directions)?
bool tile_star_is_six(const TileEntry *ent, int nent, int R);
// first state: every column sealed in "sweep 0", the tiles around a start due (from_box: every tile)
hipError_t launch_column_init(const ColumnSolve &P, const StartDesc *starts, bool from_box, hipStream_t st);
hipError_t column_solve_wgs_per_cu(int *wgs);   // workgroups a CU holds (and the LDS opt-in)
int column_solve_wg_waves();                    // wavefronts (columns in flight) of one workgroup
// the sequence of orderings the sweeps of a start follow (TTSWEEP_OPT_TILE_ORDER; n: the grid, at: the start, device axes)
constexpr int COL_ORDER_SEQUENCES = 10;
hipError_t launch_column_line(const float *v, const DevLayout &L, const StartDesc *starts, int nstart, float *out, hipStream_t st);
int column_order_default(const int (&n)[3], float at_start, float least, float largest);   // the choice for one start (-1: by the model)
bool column_order_valid(int which);        // table + 10 x first corner + 100 x axis roles
void column_order_sequence(int which, const int (&n)[3], const int (&at)[3], unsigned long long *seq);
// the whole solve: the wavefronts of `nblocks` resident workgroups claim columns until every start is at rest
hipError_t launch_column_solve(const ColumnSolve &P, int nblocks, hipStream_t st);

#ifdef TTSWEEP_TILE_PROFILE
void tile_prof_dump();   // prints and clears the phase counters of tile_sweep_kernel
#endif
#ifdef TTSWEEP_COL_TRACE
void column_trace_dump(); // writes the event log of column_solve_kernel to $TTSWEEP_COL_TRACE_FILE and clears it
#endif
#ifdef TTSWEEP_COL_PROFILE
void column_prof_dump(); // prints and clears the phase counters of column_solve_kernel
#endif
#ifdef TTSWEEP_PROFILE
void prof_dump();        // prints and clears the phase counters of sweep_units_kernel
#endif

} // namespace ttsweep
