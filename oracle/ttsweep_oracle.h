/* ttsweep_oracle.h - CPU restatement of the reference's travel-time sweep.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * build, load or call it, and only as the checker / CPU baseline.  The product
 * path (uoparallel-seismic-project_amd/, include/ttsweep.h) never links it.
 *
 * Parity status: PINNED.  The reference ships no golden vectors (SURVEY.md
 * section 4), so this restatement is pinned against outputs of the reference
 * itself: oracle/ref_wrapper.c compiles the unmodified reference translation
 * unit (serial_new/sweep-tt-multistart.c) into oracle/_ref/libttref.so and
 * tests/golden/make_golden.py recorded its converged boxes as fixtures;
 * tests/test_oracle.py checks this file against them bit for bit (and against
 * the live reference build whenever /root/reference is present).
 *
 * Every function cites the reference lines it restates (paths relative to
 * the reference checkout).
 */
#ifndef TTSWEEP_ORACLE_H
#define TTSWEEP_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* serial_new/sweep-tt-multistart.c:46-49 (struct FS) */
struct oracle_fs {
    int i, j, k;    /* offset */
    float d;        /* delta * |offset|, filled by oracle_star_prepare */
};

/* Fill fs[l].d for l in [0,starsize) from the integer offsets:
 *   d = (float)sqrt((double)(i*i+j*j+k*k));  d = delta * d   (float*float)
 * serial_new/sweep-tt-multistart.c:122,127 (delta = 10.0 at :108). */
void oracle_star_prepare(struct oracle_fs *fs, int starsize, float delta);

/* Travel-time initialisation for one start: every cell +INFINITY, start 0.
 * serial_new/sweep-tt-multistart.c:139-144. */
void oracle_tt_init(float *tt, int nx, int ny, int nz, int si, int sj, int sk);

/* One ascending i,j,k Gauss-Seidel pass for one start, offsets
 * l in [starstart, starstop) (exclusive upper bound), edges centred on the
 * start skipped, INFINITY cases as in the reference.  Returns the number of
 * stores (the reference's `change`).
 * serial_new/sweep-tt-multistart.c:198-256; indexing include/floatbox.h:127-129,160
 * (flat index x*ny*nz + y*nz + z). */
long oracle_sweepXYZ(const float *v, float *tt, int nx, int ny, int nz,
                     const struct oracle_fs *fs, int starstart, int starstop,
                     int si, int sj, int sk);

/* Same relaxation body, but the three loops run ascending (dir=+1) or
 * descending (dir=-1) per axis.  dir=(+1,+1,+1) is oracle_sweepXYZ.  Not in the
 * reference: used by tests to reach the (order-independent) fixed point in
 * fewer passes on larger grids; tests/test_oracle.py checks it converges to the
 * same bits as the reference order. */
long oracle_sweep_dir(const float *v, float *tt, int nx, int ny, int nz,
                      const struct oracle_fs *fs, int starstart, int starstop,
                      int si, int sj, int sk, int dirx, int diry, int dirz);

/* Driver loop: sweep until a pass makes no store (the `while (anychange)` loop
 * of serial_new/sweep-tt-multistart.c:151-170 without the temporary `break` at
 * :168-169, i.e. old/sweep-serial/sweep-tt-multistart.c:189-211), for ONE start.
 * order = 0: reference order every pass; order = 1: cycle through the 8
 * (+-x,+-y,+-z) orderings.  Stops after max_sweeps passes if > 0.
 * Returns the number of passes executed (the last one being the all-quiet
 * pass), or -1 if max_sweeps was hit first.  *stores_out (may be NULL)
 * receives the total number of stores. */
int oracle_converge(const float *v, float *tt, int nx, int ny, int nz,
                    const struct oracle_fs *fs, int starstart, int starstop,
                    int si, int sj, int sk, int order, int max_sweeps,
                    long *stores_out);

/* Fixed-point check in the spirit of testconvergence
 * (old/wavefront-openmp/wave-multistart.c:300-347) but on serial_new's edge
 * set: counts the (cell, offset) pairs a further reference sweep would still
 * store through, without modifying tt.  *ninf_out (may be NULL) receives the
 * number of cells still at INFINITY.  0 means converged. */
long oracle_validate(const float *v, const float *tt, int nx, int ny, int nz,
                     const struct oracle_fs *fs, int starstart, int starstop,
                     int si, int sj, int sk, long *ninf_out);

/* VBOX checksum over `count` little-endian words, signed-byte rule of
 * include/velocityboxfiler.h:240-252 (union member c4 is int8_t, :79). */
unsigned int oracle_vbox_checksum(unsigned int seed, const unsigned int *words, long count);

#ifdef __cplusplus
}
#endif

#endif /* TTSWEEP_ORACLE_H */
