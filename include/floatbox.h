/* floatbox.h - flat [x][y][z] float volume (z is the stride-1 axis).
 *
 * Kept surface of the reference's include/floatbox.h: struct FLOATBOX (:73-77)
 * and boxinit (:84-99), boxvolume (:102-110), boxalloc (:113-136), boxfree
 * (:139-149), boxindex (:152-161), boxget (:164-173), boxput (:176-186),
 * boxsetall (:189-199), boxfprint (:201-222).  Same names, argument order,
 * return conventions (0 = failure) and struct layout (48 bytes on LP64:
 * sx@0 sy@8 sz@16 size@24 flat@40) so host code written against the reference
 * header compiles unchanged.  Written fresh; `static inline` so that several
 * translation units may include it.
 *
 * The HIP library never sees this struct: only `flat`, nx, ny, nz cross the
 * C ABI (include/ttsweep.h).
 */
#ifndef TTSWEEP_FLOATBOX_H
#define TTSWEEP_FLOATBOX_H

#include "point3d.h"

#include <stdio.h>
#include <stdlib.h>

struct FLOATBOX {
    size_t sx, sy, sz;      /* strides (in floats) of the x, y, z axes in `flat` */
    struct POINT3D size;    /* nx, ny, nz */
    float *flat;            /* nx*ny*nz values, index = x*sx + y*sy + z*sz */
};

/* Put the box into the empty state (no storage, zero extent). NULL-safe. */
static inline void boxinit(struct FLOATBOX *box)
{
    if (!box) return;
    box->sx = box->sy = box->sz = 0;
    point3dset(&box->size, 0, 0, 0);
    box->flat = NULL;
}

/* Number of cells. */
static inline size_t boxvolume(struct FLOATBOX box)
{
    return (size_t)box.size.x * (size_t)box.size.y * (size_t)box.size.z;
}

/* Allocate (uninitialised) storage for nx*ny*nz floats and set strides.
 * Returns non-zero on success, 0 when the allocation fails.
 * (The reference forms the byte count from an `int` product, floatbox.h:122;
 * this version uses size_t so >2^31-byte boxes such as 1024x1024x512 work.) */
static inline int boxalloc(struct FLOATBOX *box, int nx, int ny, int nz)
{
    size_t cells = (size_t)nx * (size_t)ny * (size_t)nz;
    float *mem = (float *)malloc(cells * sizeof(float));
    if (!mem) return 0;
    box->sx = (size_t)ny * (size_t)nz;
    box->sy = (size_t)nz;
    box->sz = 1;
    point3dset(&box->size, nx, ny, nz);
    box->flat = mem;
    return 1;
}

/* Release storage; extent becomes (0,0,0), strides are left alone. NULL-safe. */
static inline void boxfree(struct FLOATBOX *box)
{
    if (!box) return;
    free(box->flat);
    box->flat = NULL;
    point3dset(&box->size, 0, 0, 0);
}

/* Flat index of (x,y,z); no bounds check. */
static inline size_t boxindex(struct FLOATBOX box, int x, int y, int z)
{
    return (size_t)x * box.sx + (size_t)y * box.sy + (size_t)z * box.sz;
}

static inline float boxget(struct FLOATBOX box, int x, int y, int z)
{
    return box.flat[boxindex(box, x, y, z)];
}

static inline void boxput(struct FLOATBOX box, int x, int y, int z, float val)
{
    box.flat[boxindex(box, x, y, z)] = val;
}

/* Fill every cell with `val`. No-op on an empty box. */
static inline void boxsetall(const struct FLOATBOX box, float val)
{
    size_t n, i;
    if (!box.flat) return;
    n = boxvolume(box);
    for (i = 0; i < n; i++) box.flat[i] = val;
}

/* Human-readable metadata dump, same text as the reference's boxfprint. */
static inline void boxfprint(FILE *stream, const char *prefix,
                             const char *indent, struct FLOATBOX box)
{
    if (!stream) stream = stdout;
    if (!prefix) prefix = "";
    if (!indent) indent = "  ";
    fprintf(stream, "%sFLOATBOX {\n", prefix);
    fprintf(stream, "%s%sstrides: (%zu, %zu, %zu)\n", prefix, indent,
            box.sx, box.sy, box.sz);
    fprintf(stream, "%s%ssize: (%d, %d, %d)\n", prefix, indent,
            box.size.x, box.size.y, box.size.z);
    fprintf(stream, "%s%sflat: %p\n", prefix, indent, (void *)box.flat);
    fprintf(stream, "%s}\n", prefix);
}

#endif /* TTSWEEP_FLOATBOX_H */
