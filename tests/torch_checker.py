"""TEST INFRASTRUCTURE: an every-cell fixed-point check in plain PyTorch (eager element-wise
ops on the device; no code shared with libttsweep.so).  It restates the store conditions of
serial_new/sweep-tt-multistart.c:219-249 for a whole box at once:

  open_edges   (cell c, entry l < L) pairs, c != start, c + off[l] inside the grid, through which
               one more reference sweep would still store: exactly one of T[c], T[o] infinite,
               or delay + T[o] < T[c], or delay + T[c] < T[o], with
               delay = fl(fl(d_l * fl(v[c] + v[o])) / 2)   (:216; eager ops round separately)
  unsupported  cells (other than the start) whose finite travel time is below every candidate
               their live edges offer (no store of :222-223 / :246-247 can have produced it)

Both zero <=> the box is the fixed point the reference's loop :151-170 converges to (all
delays >= 0).  Used where no CPU run is feasible (512x512x256, 1024x1024x512): the library's own
device validator then is not the only witness.
"""
import numpy as np
import torch


def _slices(n, e):
    """Index ranges of the centre cells c (and of o = c + e) along one axis of length n."""
    lo, hi = max(0, -e), min(n, n - e)
    return slice(lo, hi), slice(lo + e, hi + e)


def fixed_point_counts(v: torch.Tensor, T: torch.Tensor, fs: np.ndarray, start, starstart: int = 0,
                       starstop: int | None = None):
    """(open_edges, cells_infinite, cells_unsupported) of box T (device tensors [nx,ny,nz])."""
    if starstop is None:
        starstop = len(fs) - 1              # the reference call site, :160
    nx, ny, nz = T.shape
    si, sj, sk = (int(x) for x in start)
    inf = float("inf")
    open_edges = 0
    best = torch.full_like(T, inf)          # smallest candidate offered to each cell
    half = torch.tensor(0.5, dtype=torch.float32, device=T.device)
    for l in range(starstart, starstop):
        e = (int(fs["i"][l]), int(fs["j"][l]), int(fs["k"][l]))
        if e == (0, 0, 0):
            continue
        d = torch.tensor(float(fs["d"][l]), dtype=torch.float32, device=T.device)
        (ci, oi), (cj, oj), (ck, ok) = _slices(nx, e[0]), _slices(ny, e[1]), _slices(nz, e[2])
        if ci.start >= ci.stop or cj.start >= cj.stop or ck.start >= ck.stop:
            continue
        Tc, To = T[ci, cj, ck], T[oi, oj, ok]
        delay = (d * (v[ci, cj, ck] + v[oi, oj, ok])) * half      # three separately rounded ops
        cand_c = delay + To                 # what c is offered through this edge
        cand_o = delay + Tc                 # what o is offered
        opened = (cand_c < Tc) | (cand_o < To) | (torch.isinf(Tc) != torch.isinf(To))
        # edges centred on the start are never relaxed (:219-221)
        at = (si - ci.start, sj - cj.start, sk - ck.start)
        centre_is_start = all(0 <= a < n for a, n in zip(at, Tc.shape))
        if centre_is_start:
            opened[at] = False
            cand_c = cand_c.clone(); cand_c[at] = inf
            cand_o = cand_o.clone(); cand_o[at] = inf
        open_edges += int(opened.sum().item())
        torch.minimum(best[ci, cj, ck], cand_c, out=best[ci, cj, ck])
        torch.minimum(best[oi, oj, ok], cand_o, out=best[oi, oj, ok])
    unsupported = (T < best) & torch.isfinite(T)
    unsupported[si, sj, sk] = False
    return open_edges, int(torch.isinf(T).sum().item()), int(unsupported.sum().item())
