/* velocitybox.h - a FLOATBOX of velocities plus its global min/max corner.
 *
 * Kept surface of the reference's include/velocitybox.h: struct VELOCITYBOX
 * (:40-43), vboxinit (:50-62), vboxalloc (:65-82), vboxfree (:85-93),
 * vboxfprint (:96-126).  Same names, arguments, return conventions and layout
 * (72 bytes on LP64: min@0 max@12 box@24).  Written fresh, `static inline`.
 */
#ifndef TTSWEEP_VELOCITYBOX_H
#define TTSWEEP_VELOCITYBOX_H

#include "floatbox.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

struct VELOCITYBOX {
    struct POINT3D min, max;   /* global coordinates of the two extreme corners */
    struct FLOATBOX box;       /* dimensions + velocity samples */
};

static inline void vboxinit(struct VELOCITYBOX *vbox)
{
    if (!vbox) return;
    point3dset(&vbox->min, 0, 0, 0);
    point3dset(&vbox->max, 0, 0, 0);
    boxinit(&vbox->box);
}

/* Allocate an nx*ny*nz volume whose least corner sits at (ox,oy,oz).
 * Non-zero on success, 0 on allocation failure. Release with vboxfree. */
static inline int vboxalloc(struct VELOCITYBOX *vbox,
                            const int ox, const int oy, const int oz,
                            const int nx, const int ny, const int nz)
{
    if (!boxalloc(&vbox->box, nx, ny, nz)) return 0;
    point3dset(&vbox->min, ox, oy, oz);
    point3dset(&vbox->max, ox + nx - 1, oy + ny - 1, oz + nz - 1);
    return 1;
}

static inline void vboxfree(struct VELOCITYBOX *vbox)
{
    if (!vbox) return;
    boxfree(&vbox->box);
}

/* Metadata dump; the nested FLOATBOX is printed one indent level deeper. */
static inline void vboxfprint(FILE *stream, const char *prefix,
                              const char *indent, struct VELOCITYBOX vbox)
{
    size_t lp, li;
    char *inner;
    if (!stream) stream = stdout;
    if (!prefix) prefix = "";
    if (!indent) indent = "  ";

    fprintf(stream, "%sVELOCITYBOX {\n", prefix);
    fprintf(stream, "%s%sminimum corner: (%d, %d, %d)\n", prefix, indent,
            vbox.min.x, vbox.min.y, vbox.min.z);
    fprintf(stream, "%s%smaximum corner: (%d, %d, %d)\n", prefix, indent,
            vbox.max.x, vbox.max.y, vbox.max.z);

    lp = strlen(prefix);
    li = strlen(indent);
    inner = (char *)malloc(lp + li + 1);
    if (inner) {
        memcpy(inner, prefix, lp);
        memcpy(inner + lp, indent, li + 1);
        boxfprint(stream, inner, indent, vbox.box);
        free(inner);
    } else {
        boxfprint(stream, prefix, indent, vbox.box);
    }
    fprintf(stream, "%s}\n", prefix);
}

#endif /* TTSWEEP_VELOCITYBOX_H */
