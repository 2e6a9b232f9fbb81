// pullstar.h - host-side construction of the pull form of a forward star.
//
// The reference relaxes, for every cell c and every star entry l in
// [starstart, starstop), the undirected edge {c, c + off[l]} (both directions,
// serial_new/sweep-tt-multistart.c:206-249) unless c is the start (:219-221).
// The GPU kernels instead let every cell PULL from its neighbours.  Neighbour
// o = c + e of cell c is reachable through
//   * a forward entry  (e = +off[l]): the edge is centred on c -> dead iff c is the start;
//   * a reverse entry  (e = -off[l]): the edge is centred on o -> dead iff o is the start.
// Entries with equal offset and equal length are merged (flags OR-ed).  For a
// point-symmetric star every offset ends up with both flags except the last
// star entry off[S-1] (excluded by the exclusive bound at :160): +off[S-1] is
// reverse-only and -off[S-1] is forward-only, which makes exactly one edge,
// {start, start - off[S-1]}, dead.
#pragma once

#include <vector>

#include "../../include/ttsweep.h"

namespace ttsweep {

// Unique pull entries, sorted by (di, dj, dk).  Zero offsets are dropped (an
// edge from a cell to itself can never improve it).
std::vector<ttsweep_pull_entry> build_pull_star(const ttsweep_fs *fs, int starstart,
                                                int starstop);

// max |component| over the entries (0 for an empty star)
int pull_star_radius(const std::vector<ttsweep_pull_entry> &pull);

} // namespace ttsweep
