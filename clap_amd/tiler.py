"""Host-side layout of an entity forest for the single-launch tiled update kernel.

The reference keeps entities on linked lists in creation order (model.h:334,222,377)
and tolerates a one-frame lag when a child is listed before its parent
(model.c:1911-1922).  The device layout instead groups whole subtrees into TILES:

    tile  = consecutive 64-entity rows, one hierarchy level per row
    row r = slots [64 r, 64 r + 64); a child in row r has its parent in row r - 1

so one wavefront walks one tile top-down and a parent matrix never leaves registers
(include/clapgpu.h, clapgpu_entities_update_tiles).  Slots not used by an entity are
padding (flags == 0).  Parents precede children in slot order, so the converged
(level-ordered) result is what both the oracle and the kernel compute.
"""
import numpy as np

WAVE = 64


def forest_structure(parent):
    """depth[i], tree[i] (dense tree id) and the per-tree, per-level widths."""
    parent = np.asarray(parent, np.int64)
    n = parent.shape[0]
    idx = np.arange(n)
    has_p = parent >= 0
    safe_p = np.where(has_p, parent, 0)
    depth = np.zeros(n, np.int64)
    while True:                                       # relax until stable: max_depth iterations
        nd = np.where(has_p, depth[safe_p] + 1, 0)
        if np.array_equal(nd, depth):
            break
        depth = nd
        if depth.max(initial=0) > n:
            raise ValueError("parent array has a cycle")
    root = np.where(has_p, parent, idx)
    while True:                                       # pointer jumping
        nr = root[root]
        if np.array_equal(nr, root):
            break
        root = nr
    roots, tree = np.unique(root, return_inverse=True)
    n_trees = roots.shape[0]
    max_d = int(depth.max(initial=0)) + 1
    widths = np.bincount(tree * max_d + depth, minlength=n_trees * max_d).reshape(n_trees, max_d)
    return depth, tree, widths


def pack_trees(widths, wave=WAVE):
    """Next-fit packing of whole trees into tiles: every level of a tile holds <= `wave` entities.
    Returns tile_of_tree.  Raises if a single tree is wider than a wavefront at some level."""
    n_trees, max_d = widths.shape
    if n_trees == 0:
        return np.zeros(0, np.int64)
    if widths.max() > wave:
        raise NotImplementedError("a tree is wider than 64 at some level: use the per-level path "
                                  "(clapgpu_entities_update) for this forest")
    if np.all(widths == widths[0]):                   # identical trees (e.g. chains): closed form
        per_tile = int(wave // widths[0].max())
        return np.arange(n_trees) // per_tile
    tile_of = np.empty(n_trees, np.int64)
    fill = np.zeros(max_d, np.int64)
    t = 0
    for k in range(n_trees):
        w = widths[k]
        if np.any(fill + w > wave):
            t += 1
            fill[:] = 0
        fill += w
        tile_of[k] = t
    return tile_of


def tile_forest(parent):
    """Slot assignment for clapgpu_entities_update_tiles.

    Returns dict(slot_of[n], orig_of[n_slots], n_slots, tile_row_start[n_tiles+1] (uint32),
    n_tiles, fill = n / n_slots)."""
    parent = np.asarray(parent, np.int64)
    n = parent.shape[0]
    depth, tree, widths = forest_structure(parent)
    tile_of_tree = pack_trees(widths)
    tile = tile_of_tree[tree]
    n_tiles = int(tile.max(initial=-1)) + 1
    rows_of_tile = np.zeros(n_tiles, np.int64)
    np.maximum.at(rows_of_tile, tile, depth + 1)
    tile_row_start = np.concatenate([[0], np.cumsum(rows_of_tile)])
    order = np.lexsort((np.arange(n), depth, tile))   # by tile, then level, then original order
    t_s, d_s = tile[order], depth[order]
    new_group = np.ones(n, bool)
    new_group[1:] = (t_s[1:] != t_s[:-1]) | (d_s[1:] != d_s[:-1])
    group_start = np.maximum.accumulate(np.where(new_group, np.arange(n), 0))
    lane = np.arange(n) - group_start
    assert lane.max(initial=0) < WAVE
    slot_sorted = (tile_row_start[t_s] + d_s) * WAVE + lane
    slot_of = np.empty(n, np.int64)
    slot_of[order] = slot_sorted
    n_slots = int(tile_row_start[-1]) * WAVE
    orig_of = np.full(n_slots, -1, np.int64)
    orig_of[slot_of] = np.arange(n)
    return dict(slot_of=slot_of, orig_of=orig_of, n_slots=n_slots,
                tile_row_start=tile_row_start.astype(np.uint32), n_tiles=n_tiles,
                fill=n / max(n_slots, 1))


def apply_layout(scene, slot_of, n_slots):
    """Scatter an entity scene (clap_amd.synth dict) into `n_slots` slots; padding slots are
    dead identity entities.  Parents are re-indexed."""
    n = int(scene["n"])
    slot_of = np.asarray(slot_of, np.int64)
    orig_of = np.full(n_slots, -1, np.int64)
    orig_of[slot_of] = np.arange(n)
    pad = orig_of < 0

    def scatter(a, fill=0):
        out = np.full((n_slots,) + a.shape[1:], fill, a.dtype)
        out[slot_of] = a
        return out

    out = dict(scene)
    out["n"] = n_slots
    out["n_real"] = int(scene.get("n_real", n))
    out["pos_scale"] = scatter(scene["pos_scale"])
    out["pos_scale"][pad, 3] = 1.0
    out["rot"] = scatter(scene["rot"])
    out["rot"][pad, 3] = 1.0
    par = np.asarray(scene["parent"], np.int64)
    out["parent"] = scatter(np.where(par >= 0, slot_of[np.maximum(par, 0)], -1).astype(np.int32), -1)
    for k in ("model", "flags", "seqs"):
        out[k] = scatter(scene[k])
    out["slot_of"] = slot_of
    out["orig_of"] = orig_of
    out.pop("level_start", None)
    return out


def tiled_scene(scene):
    """scene (any parent order) -> (scene in tile layout, tiling dict)."""
    tl = tile_forest(scene["parent"])
    out = apply_layout(scene, tl["slot_of"], tl["n_slots"])
    out["tile_row_start"] = tl["tile_row_start"]
    return out, tl
