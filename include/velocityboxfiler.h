/* velocityboxfiler.h - text and VBOX-binary readers/writers for VELOCITYBOX.
 *
 * Kept surface of the reference's include/velocityboxfiler.h:
 *   struct VBOXOPENFILE (:64-71), union VBOX4BYTES (:78-83),
 *   vbfileloadtext (:90-219), vbfilechecksum (:240-252),
 *   vbfilestorebinary (:309-452), vbfileopenbinary (:510-616),
 *   vbfileclosebinary (:619-627), vbfileloadbinary (:630-737),
 *   vbfileloadbinarysubset (:740-864).
 * Same names, arguments, "0 = failure" convention, stderr messages of the same
 * shape, and the same on-disk bytes (formats/VBOXFORMAT.txt:29-46 as the CODE
 * implements it, see the checksum note below).  Written fresh: the binary
 * paths move whole buffers with one fread/fwrite instead of one call per
 * 4 bytes (the reference needs ~20 s for a 1024x1024x512 file that way), and
 * everything is `static inline`.
 *
 * VBOX layout (little-endian): 'v','b','o','x' | int32 ox,oy,oz | int32
 * nx,ny,nz | float32[nx*ny*nz] in [x][y][z] order | uint32 checksum.
 * File size = 32 + 4*nx*ny*nz bytes.
 *
 * Checksum: the format text says "cyclic sum of all previous data as uint32",
 * but the reference code adds, for each 4-byte word, its bytes as *signed*
 * chars shifted into place (velocityboxfiler.h:79,248-251), so every byte
 * >= 0x80 is sign-extended before the shift.  Files written by the reference
 * carry that sum, so this implementation reproduces it exactly:
 *   term(w) = w - 0x100*[b0>=0x80] - 0x10000*[b1>=0x80] - 0x1000000*[b2>=0x80]
 * (mod 2^32; b3's extension shifts out).  The sum covers the magic, the six
 * header ints and all floats.
 *
 * Big-endian hosts (byte-swapping paths of the reference, :401-446, :578-596,
 * :694-720) are out of scope: MI355X hosts are little-endian.  Opening a file
 * on a big-endian host fails with a message instead of mis-reading it.
 */
#ifndef TTSWEEP_VELOCITYBOXFILER_H
#define TTSWEEP_VELOCITYBOXFILER_H

#include "velocitybox.h"

#include <stdint.h>
#include <stdio.h>
#include <string.h>

struct VBOXOPENFILE {
    FILE *file;             /* non-NULL while the file is open */
    long datapos;           /* byte offset of the first float */
    int is_little_endian;   /* host byte order detected from the magic */
    struct POINT3D min, dims;
    uint32_t checksum;      /* running checksum over the 28 header bytes */
    const char *filename;
};

union VBOX4BYTES {
    int8_t c4[4];
    int32_t i32;
    uint32_t u32;
    float f32;
};

#define VBFILE_MAGIC_LE 0x786f6276u   /* "vbox" read as a little-endian u32 */

/* Reverse the four bytes of a word. */
static inline union VBOX4BYTES vbfilereversebytes(const union VBOX4BYTES in)
{
    union VBOX4BYTES out;
    out.c4[0] = in.c4[3];
    out.c4[1] = in.c4[2];
    out.c4[2] = in.c4[1];
    out.c4[3] = in.c4[0];
    return out;
}

/* Add one word to the running checksum (signed-byte rule, see header note). */
static inline void vbfilechecksum(uint32_t *checksum, union VBOX4BYTES fb)
{
    const uint32_t w = fb.u32;
    const uint32_t corr = ((w & 0x00000080u) << 1)    /* b0 >= 0x80 : 0x100     */
                        + ((w & 0x00008000u) << 1)    /* b1 >= 0x80 : 0x10000   */
                        + ((w & 0x00800000u) << 1);   /* b2 >= 0x80 : 0x1000000 */
    *checksum += w - corr;
}

/* Checksum of `count` consecutive words (same rule, for bulk buffers).
 * Note: the three corrections must look at the ORIGINAL bytes of the word. */
static inline uint32_t vbfilechecksumwords(uint32_t checksum,
                                           const uint32_t *words, size_t count)
{
    size_t i;
    for (i = 0; i < count; i++) {
        uint32_t w = words[i];
        uint32_t corr = ((w & 0x00000080u) << 1) + ((w & 0x00008000u) << 1)
                      + ((w & 0x00800000u) << 1);
        checksum += w - corr;
    }
    return checksum;
}

/* ------------------------------------------------------------------------ */
/* text format:  "x,y,z,velocity" per line, z fastest                        */
/* ------------------------------------------------------------------------ */

/* Load a text velocity file.  The first line's coordinates become the origin,
 * the LAST line's coordinates give the far corner (found by seeking close to
 * the end of file and walking back to a newline), and the values are stored
 * in file order; per-line coordinates after the first are parsed but not
 * used for placement (as in the reference, :196-213).
 * Non-zero on success, 0 on failure (message on stderr). */
static inline int vbfileloadtext(struct VELOCITYBOX *vbox, const char *filename)
{
    const char *fn = "vbfileloadtext";
    FILE *in;
    int ox, oy, oz, lx, ly, lz, nx, ny, nz;
    size_t total, line;

    vboxinit(vbox);

    in = fopen(filename, "r");
    if (!in) {
        fprintf(stderr, "%s: error opening file %s\n", fn, filename);
        return 0;
    }
    if (fscanf(in, "%d,%d,%d", &ox, &oy, &oz) != 3) {
        fprintf(stderr, "%s: error reading first line from file %s\n", fn, filename);
        fclose(in);
        return 0;
    }

    /* the shortest possible line is "1,1,1,0.0": start that far from the end */
    if (fseek(in, -(long)strlen("1,1,1,0.0"), SEEK_END) != 0) {
        fprintf(stderr, "%s: error seeking to estimated last line in file %s\n",
                fn, filename);
        fclose(in);
        return 0;
    }
    for (;;) {
        int c = fgetc(in);
        if (c == '\n' || c == '\r') break;          /* cursor now at line start */
        if (c == EOF || fseek(in, -2, SEEK_CUR) != 0) {
            fprintf(stderr, "%s: error scanning for last line in file %s\n",
                    fn, filename);
            fclose(in);
            return 0;
        }
    }
    if (fscanf(in, "%d,%d,%d", &lx, &ly, &lz) != 3) {
        fprintf(stderr, "%s: error reading last line from file %s\n", fn, filename);
        fclose(in);
        return 0;
    }
    nx = lx - ox + 1;
    ny = ly - oy + 1;
    nz = lz - oz + 1;
    if (nx <= 0 || ny <= 0 || nz <= 0) {
        fprintf(stderr, "%s: nonsense coordinates in file %s\n", fn, filename);
        fclose(in);
        return 0;
    }
    if (!vboxalloc(vbox, ox, oy, oz, nx, ny, nz)) {
        fprintf(stderr, "%s: unable to allocate memory for a VELOCITYBOX with"
                "dimension: %d x %d x %d\n", fn, nx, ny, nz);
        fclose(in);
        return 0;
    }

    fseek(in, 0, SEEK_SET);
    total = boxvolume(vbox->box);
    for (line = 0; line < total; line++) {
        int x, y, z;
        float vel;
        if (fscanf(in, "%d,%d,%d,%f\n", &x, &y, &z, &vel) != 4) {
            fprintf(stderr, "%s: I am confused by line %zu in %s\n",
                    fn, line + 1, filename);
            vboxfree(vbox);
            fclose(in);
            return 0;
        }
        vbox->box.flat[line] = vel;
    }
    fclose(in);
    return 1;
}

/* ------------------------------------------------------------------------ */
/* VBOX binary format                                                        */
/* ------------------------------------------------------------------------ */

static inline int vbfile_host_is_little_endian(void)
{
    union VBOX4BYTES m;
    m.c4[0] = (int8_t)'v'; m.c4[1] = (int8_t)'b';
    m.c4[2] = (int8_t)'o'; m.c4[3] = (int8_t)'x';
    return m.u32 == VBFILE_MAGIC_LE;
}

/* Write `vbox` as a VBOX file.  Non-zero on success, 0 on failure. */
static inline int vbfilestorebinary(const char *filename, struct VELOCITYBOX vbox)
{
    const char *fn = "vbfilestorebinary";
    FILE *out;
    uint32_t header[7];
    uint32_t checksum;
    size_t count;

    if (!vbox.box.flat) {
        fprintf(stderr, "%s: provided vbox is empty\n", fn);
        return 0;
    }
    if (!vbfile_host_is_little_endian()) {
        fprintf(stderr, "%s: big-endian hosts are not supported\n", fn);
        return 0;
    }
    out = fopen(filename, "wb");
    if (!out) {
        fprintf(stderr, "%s: error creating file %s\n", fn, filename);
        return 0;
    }

    header[0] = VBFILE_MAGIC_LE;
    header[1] = (uint32_t)vbox.min.x;
    header[2] = (uint32_t)vbox.min.y;
    header[3] = (uint32_t)vbox.min.z;
    header[4] = (uint32_t)vbox.box.size.x;
    header[5] = (uint32_t)vbox.box.size.y;
    header[6] = (uint32_t)vbox.box.size.z;
    count = boxvolume(vbox.box);

    checksum = vbfilechecksumwords(0, header, 7);
    checksum = vbfilechecksumwords(checksum, (const uint32_t *)(const void *)vbox.box.flat, count);

    if (fwrite(header, 4, 7, out) != 7
        || fwrite(vbox.box.flat, 4, count, out) != count
        || fwrite(&checksum, 4, 1, out) != 1) {
        fprintf(stderr, "%s: error writing to file %s at position %ld\n",
                fn, filename, ftell(out));
        fclose(out);
        return 0;
    }
    fclose(out);
    return 1;
}

/* Open a VBOX file and read its 28-byte header.  On success the file stays
 * open (close it with vbfileclosebinary) and the cursor is at the first float.
 * Non-zero on success, 0 on failure. */
static inline int vbfileopenbinary(struct VBOXOPENFILE *vbfile, const char *filename)
{
    const char *fn = "vbfileopenbinary";
    FILE *in;
    uint32_t header[7];

    vbfile->file = NULL;

    in = fopen(filename, "rb");
    if (!in) {
        fprintf(stderr, "%s: error opening file %s\n", fn, filename);
        return 0;
    }
    if (fread(header, 1, 4, in) != 4 || memcmp(header, "vbox", 4) != 0) {
        fprintf(stderr, "%s: input file %s is not a vbox binary file, or is corrupted\n",
                fn, filename);
        fclose(in);
        return 0;
    }
    vbfile->is_little_endian = (header[0] == VBFILE_MAGIC_LE);
    if (!vbfile->is_little_endian) {
        fprintf(stderr, "%s: big-endian hosts are not supported (%s)\n", fn, filename);
        fclose(in);
        return 0;
    }
    if (fread(header + 1, 4, 6, in) != 6) {
        fprintf(stderr, "%s: error reading header in %s: suspect corruption\n",
                fn, filename);
        fclose(in);
        return 0;
    }

    vbfile->file = in;
    vbfile->datapos = ftell(in);
    point3dset(&vbfile->min, (int32_t)header[1], (int32_t)header[2], (int32_t)header[3]);
    point3dset(&vbfile->dims, (int32_t)header[4], (int32_t)header[5], (int32_t)header[6]);
    vbfile->checksum = vbfilechecksumwords(0, header, 7);
    vbfile->filename = filename;
    return 1;
}

static inline void vbfileclosebinary(struct VBOXOPENFILE *vbfile)
{
    if (!vbfile || !vbfile->file) return;
    fclose(vbfile->file);
    vbfile->file = NULL;
}

/* Load a whole VBOX file, verifying its checksum.
 * Non-zero on success, 0 on failure (vbox is left freed). */
static inline int vbfileloadbinary(struct VELOCITYBOX *vbox, const char *filename)
{
    const char *fn = "vbfileloadbinary";
    struct VBOXOPENFILE vbfile;
    size_t count, got;
    uint32_t stored;

    if (!vbox) return 0;
    if (!vbfileopenbinary(&vbfile, filename)) return 0;

    if (vbfile.dims.x <= 0 || vbfile.dims.y <= 0 || vbfile.dims.z <= 0
        || !vboxalloc(vbox, vbfile.min.x, vbfile.min.y, vbfile.min.z,
                      vbfile.dims.x, vbfile.dims.y, vbfile.dims.z)) {
        fprintf(stderr, "%s: unable to allocate memory for a VELOCITYBOX with"
                "dimension: %d x %d x %d\n", fn,
                vbfile.dims.x, vbfile.dims.y, vbfile.dims.z);
        vbfileclosebinary(&vbfile);
        return 0;
    }

    count = boxvolume(vbox->box);
    got = fread(vbox->box.flat, 4, count, vbfile.file);
    if (got != count) {
        fprintf(stderr, "%s: error reading value at byte position %zu in %s\n",
                fn, (size_t)vbfile.datapos + 4 * got, filename);
        vbfileclosebinary(&vbfile);
        vboxfree(vbox);
        return 0;
    }
    if (fread(&stored, 4, 1, vbfile.file) != 1) {
        fprintf(stderr, "%s: error reading stored checksum value from %s\n",
                fn, filename);
        vbfileclosebinary(&vbfile);
        vboxfree(vbox);
        return 0;
    }
    vbfileclosebinary(&vbfile);

    vbfile.checksum = vbfilechecksumwords(vbfile.checksum,
                          (const uint32_t *)(const void *)vbox->box.flat, count);
    if (vbfile.checksum != stored) {
        fprintf(stderr, "%s: checksum mismatch in input file %s: suspect corruption\n",
                fn, filename);
        vboxfree(vbox);
        return 0;
    }
    return 1;
}

/* Load the sub-volume of nx*ny*nz cells whose corner is (ox,oy,oz) from an
 * already-open file (the file stays open).  No checksum is verified.
 *
 * Coordinates follow the reference exactly (:760-827): the bounds test treats
 * (ox,oy,oz) as GLOBAL coordinates (it must not lie below the file's origin),
 * while the file position is computed from (ox,oy,oz) as given, i.e. as a
 * 0-based index into the stored array.  examples/example_velocityboxfiler.c
 * (:80-104) relies on that pairing: it reads back index (x,y,z) of the full
 * volume at (x-ox, y-oy, z-oz) of the subset.
 * Non-zero on success, 0 on failure. */
static inline int vbfileloadbinarysubset(struct VELOCITYBOX *vbox,
                                         const int ox, const int oy, const int oz,
                                         const int nx, const int ny, const int nz,
                                         const struct VBOXOPENFILE vbfile)
{
    const char *fn = "vbfileloadbinarysubset";
    long sz, sy, sx;
    int x, y;

    if (!vbox) return 0;
    if (!vbfile.file) {
        fprintf(stderr, "%s: error: source file parameter is not open\n", fn);
        return 0;
    }
    if (ox < vbfile.min.x || oy < vbfile.min.y || oz < vbfile.min.z
        || nx > vbfile.dims.x - (ox - vbfile.min.x)
        || ny > vbfile.dims.y - (oy - vbfile.min.y)
        || nz > vbfile.dims.z - (oz - vbfile.min.z)) {
        fprintf(stderr, "%s: error: file %s doesn't contain the requested subset!\n"
                "file: (%d,%d,%d) to (%d,%d,%d)\n"
                "requested subset: (%d,%d,%d) to (%d,%d,%d)\n",
                fn, vbfile.filename,
                vbfile.min.x, vbfile.min.y, vbfile.min.z,
                vbfile.min.x + vbfile.dims.x - 1,
                vbfile.min.y + vbfile.dims.y - 1,
                vbfile.min.z + vbfile.dims.z - 1,
                ox, oy, oz, ox + nx - 1, oy + ny - 1, oz + nz - 1);
        return 0;
    }
    if (nx <= 0 || ny <= 0 || nz <= 0 || !vboxalloc(vbox, ox, oy, oz, nx, ny, nz)) {
        fprintf(stderr, "%s: unable to allocate memory for a VELOCITYBOX with"
                "dimension: %d x %d x %d\n", fn, nx, ny, nz);
        return 0;
    }

    sz = (long)sizeof(float);
    sy = (long)vbfile.dims.z * sz;
    sx = (long)vbfile.dims.y * sy;
    for (x = 0; x < nx; x++) {
        for (y = 0; y < ny; y++) {
            float *strip = vbox->box.flat + boxindex(vbox->box, x, y, 0);
            long pos = vbfile.datapos + (long)(x + ox) * sx + (long)(y + oy) * sy
                     + (long)oz * sz;
            if (fseek(vbfile.file, pos, SEEK_SET) != 0
                || fread(strip, 4, (size_t)nz, vbfile.file) != (size_t)nz) {
                fprintf(stderr, "%s: error reading value at byte position %zu in %s\n",
                        fn, (size_t)ftell(vbfile.file), vbfile.filename);
                vboxfree(vbox);
                return 0;
            }
        }
    }
    return 1;
}

#endif /* TTSWEEP_VELOCITYBOXFILER_H */
