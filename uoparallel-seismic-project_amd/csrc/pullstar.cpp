// pullstar.cpp - see pullstar.h
#include "pullstar.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <map>
#include <tuple>

namespace ttsweep {

std::vector<ttsweep_pull_entry> build_pull_star(const ttsweep_fs *fs, int starstart,
                                                int starstop)
{
    // key: offset + bit pattern of the length, so parallel edges of different
    // length (possible only with a hand-made fs[]) stay separate
    std::map<std::tuple<int, int, int, uint32_t>, int> merged;
    for (int l = starstart; l < starstop; l++) {
        const ttsweep_fs &f = fs[l];
        if (f.i == 0 && f.j == 0 && f.k == 0) continue;
        const float h = f.d * 0.5f;     // exact; delay = h * (v[c] + v[o])
        uint32_t hb;
        std::memcpy(&hb, &h, 4);
        merged[std::make_tuple(f.i, f.j, f.k, hb)] |= 1;      // forward: centre = c
        merged[std::make_tuple(-f.i, -f.j, -f.k, hb)] |= 2;   // reverse: centre = o
    }
    std::vector<ttsweep_pull_entry> out;
    out.reserve(merged.size());
    for (const auto &kv : merged) {
        ttsweep_pull_entry e;
        e.di = std::get<0>(kv.first);
        e.dj = std::get<1>(kv.first);
        e.dk = std::get<2>(kv.first);
        const uint32_t hb = std::get<3>(kv.first);
        std::memcpy(&e.h, &hb, 4);
        e.flags = kv.second;
        out.push_back(e);
    }
    return out;
}

int pull_star_radius(const std::vector<ttsweep_pull_entry> &pull)
{
    int r = 0;
    for (const auto &e : pull)
        r = std::max(r, std::max(std::abs(e.di), std::max(std::abs(e.dj), std::abs(e.dk))));
    return r;
}

} // namespace ttsweep
