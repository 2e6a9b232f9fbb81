/* point3d.h - integer 3D coordinate used by FLOATBOX / VELOCITYBOX.
 *
 * Kept surface of the reference's include/point3d.h (struct POINT3D :25-27,
 * point3dset :34-43).  Same struct layout (three ints, 12 bytes) and the same
 * function name/arguments; written fresh for this project.  Unlike the
 * reference header the function is `static inline`, so the header can be
 * included from more than one translation unit.
 */
#ifndef TTSWEEP_POINT3D_H
#define TTSWEEP_POINT3D_H

#include <stddef.h>

struct POINT3D {
    int x, y, z;
};

static inline void point3dset(struct POINT3D *pt, int x, int y, int z)
{
    pt->x = x;
    pt->y = y;
    pt->z = z;
}

#endif /* TTSWEEP_POINT3D_H */
