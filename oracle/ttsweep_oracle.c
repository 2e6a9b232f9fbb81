/* ttsweep_oracle.c - CPU restatement of the reference's travel-time sweep.
 *
 * TEST INFRASTRUCTURE ONLY (see ttsweep_oracle.h).  Plain C, single thread,
 * built with `gcc -O3 -ffp-contract=off` and no -march/-ffast-math, i.e. the
 * reference's own flags (serial_new/Makefile:1-3) plus an explicit ban on
 * contraction so that every float operation rounds exactly once as it does in
 * the reference binary on x86-64.
 */
#include "ttsweep_oracle.h"

#include <math.h>
#include <stddef.h>
#include <stdint.h>

/* serial_new/sweep-tt-multistart.c:122,127 */
void oracle_star_prepare(struct oracle_fs *fs, int starsize, float delta)
{
    int l;
    for (l = 0; l < starsize; l++) {
        /* int sum -> double sqrt -> stored to float (:122) */
        fs[l].d = sqrt(fs[l].i * fs[l].i + fs[l].j * fs[l].j + fs[l].k * fs[l].k);
        /* float * float (:127) */
        fs[l].d = delta * fs[l].d;
    }
}

/* serial_new/sweep-tt-multistart.c:139-144 */
void oracle_tt_init(float *tt, int nx, int ny, int nz, int si, int sj, int sk)
{
    size_t n = (size_t)nx * ny * nz, c;
    for (c = 0; c < n; c++) tt[c] = INFINITY;
    tt[((size_t)si * ny + sj) * nz + sk] = 0;
}

/* The relaxation of one (cell, offset) pair: serial_new/...:208-249.
 * Returns the number of stores (0 or 1). */
static inline int relax_pair(const float *v, float *tt, int nx, int ny, int nz,
                             const struct oracle_fs *f, int i, int j, int k,
                             int si, int sj, int sk)
{
    int oi = i + f->i, oj = j + f->j, ok = k + f->k;
    size_t c, o;
    float sum, prod, delay, t, to;

    /* :210-214 neighbour outside the box */
    if (oi < 0 || oi > nx - 1 || oj < 0 || oj > ny - 1 || ok < 0 || ok > nz - 1)
        return 0;

    c = ((size_t)i * ny + j) * nz + k;      /* floatbox.h:127-129,160 */
    o = ((size_t)oi * ny + oj) * nz + ok;

    /* :216  delay = fs[l].d * (v[c] + v[o]) / 2.0
     * float add, float multiply, divide by 2.0 in double, round to float */
    sum = v[c] + v[o];
    prod = f->d * sum;
    delay = (float)((double)prod / 2.0);

    /* :219-221 edges centred on the start point are skipped */
    if (i == si && j == sj && k == sk) return 0;

    t = tt[c];
    to = tt[o];
    if (t == INFINITY && to == INFINITY) return 0;              /* :225-227 */
    if (t != INFINITY && to == INFINITY) {                      /* :228-232 */
        tt[o] = delay + t;
        return 1;
    }
    if (t == INFINITY && to != INFINITY) {                      /* :233-237 */
        tt[c] = delay + to;
        return 1;
    }
    /* :238-249 both finite */
    if ((delay + to) < t) {
        tt[c] = delay + to;
        return 1;
    } else if ((delay + t) < to) {
        tt[o] = delay + t;
        return 1;
    }
    return 0;
}

/* serial_new/sweep-tt-multistart.c:198-256 */
long oracle_sweepXYZ(const float *v, float *tt, int nx, int ny, int nz,
                     const struct oracle_fs *fs, int starstart, int starstop,
                     int si, int sj, int sk)
{
    long change = 0;
    int i, j, k, l;
    for (i = 0; i < nx; i++)
        for (j = 0; j < ny; j++)
            for (k = 0; k < nz; k++)
                for (l = starstart; l < starstop; l++)
                    change += relax_pair(v, tt, nx, ny, nz, &fs[l], i, j, k, si, sj, sk);
    return change;
}

long oracle_sweep_dir(const float *v, float *tt, int nx, int ny, int nz,
                      const struct oracle_fs *fs, int starstart, int starstop,
                      int si, int sj, int sk, int dirx, int diry, int dirz)
{
    long change = 0;
    int a, b, c, l;
    for (a = 0; a < nx; a++) {
        int i = dirx >= 0 ? a : nx - 1 - a;
        for (b = 0; b < ny; b++) {
            int j = diry >= 0 ? b : ny - 1 - b;
            for (c = 0; c < nz; c++) {
                int k = dirz >= 0 ? c : nz - 1 - c;
                for (l = starstart; l < starstop; l++)
                    change += relax_pair(v, tt, nx, ny, nz, &fs[l], i, j, k, si, sj, sk);
            }
        }
    }
    return change;
}

/* serial_new/...:151-170 minus the break at :168-169
 * (= old/sweep-serial/sweep-tt-multistart.c:189-211), one start */
int oracle_converge(const float *v, float *tt, int nx, int ny, int nz,
                    const struct oracle_fs *fs, int starstart, int starstop,
                    int si, int sj, int sk, int order, int max_sweeps,
                    long *stores_out)
{
    int sweeps = 0;
    long total = 0, changed;
    do {
        if (max_sweeps > 0 && sweeps >= max_sweeps) {
            if (stores_out) *stores_out = total;
            return -1;
        }
        if (order == 0) {
            changed = oracle_sweepXYZ(v, tt, nx, ny, nz, fs, starstart, starstop, si, sj, sk);
        } else {
            int m = sweeps & 7;
            changed = oracle_sweep_dir(v, tt, nx, ny, nz, fs, starstart, starstop, si, sj, sk,
                                       (m & 1) ? -1 : 1, (m & 2) ? -1 : 1, (m & 4) ? -1 : 1);
        }
        sweeps++;
        total += changed;
    } while (changed);
    if (stores_out) *stores_out = total;
    return sweeps;
}

long oracle_validate(const float *v, const float *tt, int nx, int ny, int nz,
                     const struct oracle_fs *fs, int starstart, int starstop,
                     int si, int sj, int sk, long *ninf_out)
{
    long open = 0, ninf = 0;
    int i, j, k, l;
    for (i = 0; i < nx; i++)
        for (j = 0; j < ny; j++)
            for (k = 0; k < nz; k++) {
                size_t c = ((size_t)i * ny + j) * nz + k;
                if (tt[c] == INFINITY) ninf++;
                if (i == si && j == sj && k == sk) continue;    /* :219-221 */
                for (l = starstart; l < starstop; l++) {
                    int oi = i + fs[l].i, oj = j + fs[l].j, ok = k + fs[l].k;
                    size_t o;
                    float sum, prod, delay, t, to;
                    if (oi < 0 || oi > nx - 1 || oj < 0 || oj > ny - 1 || ok < 0 || ok > nz - 1)
                        continue;
                    o = ((size_t)oi * ny + oj) * nz + ok;
                    sum = v[c] + v[o];
                    prod = fs[l].d * sum;
                    delay = (float)((double)prod / 2.0);
                    t = tt[c];
                    to = tt[o];
                    if (t == INFINITY && to == INFINITY) continue;
                    if (t == INFINITY || to == INFINITY) { open++; continue; }
                    if ((delay + to) < t || (delay + t) < to) open++;
                }
            }
    if (ninf_out) *ninf_out = ninf;
    return open;
}

/* include/velocityboxfiler.h:240-252 with union VBOX4BYTES::c4 being int8_t (:79) */
unsigned int oracle_vbox_checksum(unsigned int seed, const unsigned int *words, long count)
{
    uint32_t sum = seed;
    long n;
    for (n = 0; n < count; n++) {
        union { int8_t c4[4]; uint32_t u32; } fb;
        fb.u32 = words[n];
        sum += (uint32_t)fb.c4[0]
             + ((uint32_t)fb.c4[1] << 8)
             + ((uint32_t)fb.c4[2] << 16)
             + ((uint32_t)fb.c4[3] << 24);
    }
    return sum;
}
