"""Host-side helpers of the slab decomposition along k: slab ranges, particle ownership,
stacked weak-scaling scenes.  Pure index arithmetic, no computation of the simulation itself."""
import numpy as np


MIN_SLAB_PLANES = 8   # ceil(cfl_number = 5) + 3: the widest halo a slab must be able to send from its own planes


def slab_ranges(K, nranks, min_planes=MIN_SLAB_PLANES):
    """Split K cell planes into nranks contiguous slabs [k_begin, k_end), as evenly as possible.  With more than
    one rank every slab must hold at least `min_planes` planes (the library rejects thinner ones, include/flipv.h)."""
    if nranks < 1 or nranks > K:
        raise ValueError("need 1 <= nranks <= K")
    if nranks > 32:
        raise ValueError("at most 32 ranks per communicator")
    if nranks > 1 and K // nranks < min_planes:
        raise ValueError("slabs of %d planes are thinner than the widest halo (%d planes)" % (K // nranks, min_planes))
    base, rem = divmod(K, nranks)
    out, k = [], 0
    for r in range(nranks):
        n = base + (1 if r < rem else 0)
        out.append((k, k + n))
        k += n
    return out


def axis_cuts(n, parts, align=1, min_cells=MIN_SLAB_PLANES):
    """Cut n cells into `parts` contiguous ranges as evenly as the alignment allows (cuts along i must be multiples of 8:
    include/flipv.h); every range must hold at least min_cells cells when there is more than one."""
    if parts < 1:
        raise ValueError("need at least one part")
    cuts = [0]
    for p in range(1, parts):
        c = int(round(n * p / parts / align)) * align
        cuts.append(c)
    cuts.append(n)
    out = [(cuts[q], cuts[q + 1]) for q in range(parts)]
    for lo, hi in out:
        if hi <= lo or (parts > 1 and hi - lo < min_cells):
            raise ValueError("cannot cut %d cells into %d parts of at least %d cells (alignment %d)" % (n, parts, min_cells, align))
    return out


def block_boxes(I, J, K, dims, min_cells=MIN_SLAB_PLANES):
    """The boxes (lo, hi) of a dims[0] x dims[1] x dims[2] block decomposition in rank order (x fastest,
    rank = x + dims[0] * (y + dims[1] * z)): a tensor product of axis cuts, cuts along i multiples of 8."""
    if dims[0] * dims[1] * dims[2] > 32:
        raise ValueError("at most 32 ranks per communicator")
    cx, cy, cz = axis_cuts(I, dims[0], 8, min_cells), axis_cuts(J, dims[1], 1, min_cells), axis_cuts(K, dims[2], 1, min_cells)
    return [((x[0], y[0], z[0]), (x[1], y[1], z[1])) for z in cz for y in cy for x in cx]


def box_owner(particles, dx, boxes, dims):
    """rank owning each particle: the block that holds its cell (floor(p / dx) in fp64, reference grid3d.h:60-65);
    particles outside the domain go to the end blocks of that axis."""
    p = np.asarray(particles)[:, :3].astype(np.float64) * (1.0 / float(np.float32(dx)))
    cell = np.floor(p).astype(np.int64)
    co = []
    for a in range(3):
        stride = 1 if a == 0 else (dims[0] if a == 1 else dims[0] * dims[1])
        starts = np.array(sorted({b[0][a] for b in boxes})[1:], np.int64)
        co.append(np.searchsorted(starts, cell[:, a], side="right") * stride)
    return co[0] + co[1] + co[2]


def split_particles_boxes(particles, dx, boxes, dims):
    owner = box_owner(particles, dx, boxes, dims)
    return [np.ascontiguousarray(np.asarray(particles)[owner == r]) for r in range(len(boxes))]


def particle_owner(particles, dx, ranges):
    """rank owning each particle = slab containing the k index of its cell, floor(z / dx) in fp64 like
    Grid3d::positionToGridIndex (reference grid3d.h:60-65); out-of-domain particles go to the end ranks."""
    z = np.asarray(particles)[:, 2].astype(np.float64)
    k = np.floor(z * (1.0 / float(np.float32(dx)))).astype(np.int64)
    starts = np.array([r[0] for r in ranges[1:]], np.int64)
    return np.searchsorted(starts, k, side="right")


def split_particles(particles, dx, ranges):
    owner = particle_owner(particles, dx, ranges)
    return [np.ascontiguousarray(np.asarray(particles)[owner == r]) for r in range(len(ranges))]


def gather_owned(ranks_grids, ranges, K):
    """Assemble a global grid (numpy (depth, h, w)) from the planes each rank owns; the last rank also owns the closing
    plane of W faces / nodes when the array is K+1 deep."""
    out = np.array(ranks_grids[0], copy=True)
    depth = out.shape[0]
    for r, (g, (k0, k1)) in enumerate(zip(ranks_grids, ranges)):
        hi = depth if r == len(ranges) - 1 else k1
        out[k0:hi] = g[k0:hi]
    return out


def stack_scene(solid_nodes, particles, copies, K, dx):
    """Weak-scaling scene: `copies` instances of one closed K-deep scene stacked along k.  Node plane m*K is taken from
    the bottom plane of copy m (both candidates are inside the solid for a closed container)."""
    solid_nodes = np.asarray(solid_nodes)
    body = solid_nodes[:K]
    g = np.concatenate([body] * copies + [solid_nodes[K:K + 1]], axis=0)
    height = float(np.float32(dx)) * K
    parts = []
    for m in range(copies):
        p = np.array(particles, np.float32, copy=True)
        p[:, 2] = (p[:, 2].astype(np.float64) + m * height).astype(np.float32)
        parts.append(p)
    return np.ascontiguousarray(g), parts
