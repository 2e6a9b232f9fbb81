// ttsweep_dev.h - structures shared by the host API and the HIP kernels.
//
// Device data layout (DESIGN.md "Data layout in HBM").  The library never
// computes on the caller's FLOATBOX arrays directly: it keeps padded copies in
// which every cell has a full halo of R = max|offset| cells on every side, so
// the relaxation loops need no bounds tests:
//   * travel-time halo cells hold +INFINITY  -> a candidate through them is
//     +INFINITY and never wins the min;
//   * velocity halo cells hold 0             -> delay stays finite, no NaN.
// "Device axes" (a, b, c) are a permutation of the user's (x, y, z); c is the
// stride-1 axis of the padded arrays.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ttsweep {

struct DevLayout {
    int n[3];           // extents along device axes a, b, c (interior)
    int lo[3];          // halo cells in front of the interior, per axis
    int p[3];           // padded extents
    long long s0, s1;   // strides (floats) of axes a and b; axis c has stride 1
    long long cells;    // padded volume = p0*p1*p2
    int perm[3];        // device axis d is user axis perm[d] (0=x, 1=y, 2=z)
    int un[3];          // user extents nx, ny, nz
};

__host__ __device__ inline long long dev_index(const DevLayout &L, int a, int b, int c)
{
    return (long long)(a + L.lo[0]) * L.s0 + (long long)(b + L.lo[1]) * L.s1 + (c + L.lo[2]);
}

// flags of a pull entry (same meaning as ttsweep_pull_entry.flags)
enum : int {
    PULL_FWD = 1,   // edge exists as (centre = this cell, offset +e): dead if this cell is the start
    PULL_REV = 2,   // edge exists as (centre = neighbour, offset -e): dead if the neighbour is the start
};

// One pull entry for the per-cell kernel: neighbour = cell + delta (floats).
struct CellEntry {
    int delta;
    float h;
    int flags;
    int pad_;
};

// Per-start device record.
struct StartDesc {
    float *T;               // padded travel-time volume of this start
    long long sidx;         // padded linear index of the start cell
    int sa, sb, sc;         // start cell, device-axis interior coordinates
    int pad_;
};

} // namespace ttsweep
