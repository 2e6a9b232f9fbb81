/* ref_wrapper.c - drives the UNMODIFIED reference translation unit.
 *
 * TEST INFRASTRUCTURE ONLY.  Compiled by oracle/Makefile into
 * oracle/_ref/libttref.so, only where the reference checkout is present
 * (this container; never the GPU box).  No reference source is copied into the
 * repository: the reference file is #included from where it lies (REF_TU is
 * given on the compiler command line) with its main() renamed, which gives
 * this wrapper direct access to the reference's own file-scope globals
 * (fs, start, vbox, ttboxes: serial_new/sweep-tt-multistart.c:62-66) and to its
 * own sweepXYZ (:198-256).  The wrapper fills those globals exactly as the
 * reference main() does (:112-128 star, :137-147 starts) and exposes
 * sweepXYZ, so golden vectors are produced by reference code, not by a
 * restatement.
 */
#define main ttref_reference_main
#include REF_TU
#undef main

#include <string.h>

/* Limits compiled into the reference (:42,:44). */
int ttref_fsmax(void) { return FSMAX; }
int ttref_startmax(void) { return STARTMAX; }

static int g_nx, g_ny, g_nz, g_numstart, g_starsize;

void ttref_teardown(void)
{
    int s;
    for (s = 0; s < g_numstart; s++) boxfree(&ttboxes[s]);
    if (vbox.box.flat) vboxfree(&vbox);
    g_numstart = 0;
}

/* v: nx*ny*nz floats in [x][y][z] order.  offs: starsize triples (i,j,k).
 * starts: numstart triples.  Returns 1 on success, 0 on failure. */
int ttref_setup(int nx, int ny, int nz, const float *v,
                int starsize, const int *offs, float delta,
                int numstart, const int *starts)
{
    int i, s;
    if (starsize > FSMAX || numstart > STARTMAX) return 0;
    ttref_teardown();
    vboxinit(&vbox);
    if (!vboxalloc(&vbox, 1, 1, 1, nx, ny, nz)) return 0;
    memcpy(vbox.box.flat, v, (size_t)nx * ny * nz * sizeof(float));
    g_nx = nx; g_ny = ny; g_nz = nz;

    /* star preparation, the statements of :120-128 */
    for (i = 0; i < starsize; i++) {
        fs[i].i = offs[3 * i + 0];
        fs[i].j = offs[3 * i + 1];
        fs[i].k = offs[3 * i + 2];
        fs[i].d = sqrt(fs[i].i * fs[i].i + fs[i].j * fs[i].j + fs[i].k * fs[i].k);
        fs[i].d = delta * fs[i].d;
    }
    g_starsize = starsize;

    /* start preparation, the statements of :137-147 */
    for (s = 0; s < numstart; s++) {
        if (!boxalloc(&ttboxes[s], nx, ny, nz)) return 0;
        boxsetall(ttboxes[s], INFINITY);
        boxput(ttboxes[s], starts[3 * s + 0], starts[3 * s + 1], starts[3 * s + 2], 0);
        start[s].i = starts[3 * s + 0];
        start[s].j = starts[3 * s + 1];
        start[s].k = starts[3 * s + 2];
    }
    g_numstart = numstart;
    return 1;
}

/* One call of the reference's own sweepXYZ (:198) for start s. */
int ttref_sweep(int s, int starstart, int starstop)
{
    return sweepXYZ(g_nx, g_ny, g_nz, s, starstart, starstop);
}

/* The reference call site (:160): sweepXYZ(nx, ny, nz, s, 0, starsize-1). */
int ttref_sweep_default(int s)
{
    return sweepXYZ(g_nx, g_ny, g_nz, s, 0, g_starsize - 1);
}

float *ttref_tt(int s) { return ttboxes[s].flat; }
float ttref_fs_d(int l) { return fs[l].d; }

/* Reference VBOX I/O (include/velocityboxfiler.h) for file-format pinning. */
int ttref_store_vbox(const char *filename, int ox, int oy, int oz,
                     int nx, int ny, int nz, const float *v)
{
    struct VELOCITYBOX vb;
    int ok;
    vboxinit(&vb);
    if (!vboxalloc(&vb, ox, oy, oz, nx, ny, nz)) return 0;
    memcpy(vb.box.flat, v, (size_t)nx * ny * nz * sizeof(float));
    ok = vbfilestorebinary(filename, vb);
    vboxfree(&vb);
    return ok;
}

/* Loads with the reference reader; copies up to `cap` floats to out; writes
 * origin and dims to hdr[6].  Returns 1 on success. */
int ttref_load_vbox(const char *filename, int *hdr, float *out, long cap)
{
    struct VELOCITYBOX vb;
    size_t n;
    vboxinit(&vb);
    if (!vbfileloadbinary(&vb, filename)) return 0;
    hdr[0] = vb.min.x; hdr[1] = vb.min.y; hdr[2] = vb.min.z;
    hdr[3] = vb.box.size.x; hdr[4] = vb.box.size.y; hdr[5] = vb.box.size.z;
    n = boxvolume(vb.box);
    if ((long)n > cap) n = (size_t)cap;
    memcpy(out, vb.box.flat, n * sizeof(float));
    vboxfree(&vb);
    return 1;
}

int ttref_text_to_vbox(const char *textfile, const char *vboxfile)
{
    struct VELOCITYBOX vb;
    int ok;
    if (!vbfileloadtext(&vb, textfile)) return 0;
    ok = vbfilestorebinary(vboxfile, vb);
    vboxfree(&vb);
    return ok;
}
