/* vconvert.c - text velocity model ("x,y,z,velocity" per line) -> .vbox file.
 *
 * Same command line and messages as the reference's tools/vconvert.c (:15-34),
 * built on the kept header surface (include/velocityboxfiler.h).
 *
 *      vconvert <in:oldfile.txt> <out:newfile.vbox>
 */
#include "velocityboxfiler.h"

#include <stdio.h>

int main(int argc, char *argv[])
{
    struct VELOCITYBOX vbox;

    if (argc != 3) {
        printf("vconvert: velocity file converter\n");
        printf("usage: %s <in:oldfile.txt> <out:newfile.vbox>\n", argv[0]);
        return 0;
    }
    printf("reading old velocity model %s...", argv[1]); fflush(stdout);
    if (!vbfileloadtext(&vbox, argv[1])) return 1;
    printf(" done.\n"); fflush(stdout);

    printf("writing new velocity model %s...", argv[2]); fflush(stdout);
    if (!vbfilestorebinary(argv[2], vbox)) return 1;
    printf(" done.\n"); fflush(stdout);

    vboxfree(&vbox);
    return 0;
}
