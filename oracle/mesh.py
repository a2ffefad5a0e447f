"""CPU ORACLE -- test infrastructure, NOT product code (see oracle/__init__.py).

Mesh post-processing steps right behind the hot path, restated in numpy from the reference's call sites:
  * delete_invalid_verts           common/marching_cubes_util.py:38-52
  * largest connected component    eval.py:497-503 and :536-540 (igl.adjacency_matrix + igl.connected_components + np.argmax(cc_sizes))
  * hole removal                   eval.py:529-548 (value threshold -> delete_invalid_verts -> largest component -> delete_invalid_verts)
Pinning: libigl is absent from this image ("parity unpinned" for igl's component NUMBERING); its published algorithm -- breadth-first search
started from every not-yet-visited vertex in ascending index, components numbered in that order -- is restated literally below
(`connected_components_bfs`) and cross-checked against scipy.sparse.csgraph.connected_components (tests/test_oracle_mesh.py).  Only the
numbering's ORDER matters downstream: np.argmax(cc_sizes) takes the first of several largest components.
"""
import numpy as np


def delete_invalid_verts(verts, faces, is_vert_on_surface):
    """common/marching_cubes_util.py:38-52"""
    valid = is_vert_on_surface[faces].all(axis=1) if len(faces) else np.zeros(0, dtype=bool)
    raw = faces[valid]
    used = np.unique(raw.flatten())
    remap = np.zeros(len(verts), dtype=faces.dtype)
    remap[used] = np.arange(len(used))
    return verts[used], remap[raw].reshape(-1, 3)


def connected_components_bfs(faces, n):
    """igl.connected_components(igl.adjacency_matrix(faces)) -> (num_cc, cc_idxs [n], cc_sizes): BFS from every unvisited vertex in ascending
    index.  Pure Python: small meshes only."""
    adj = [[] for _ in range(n)]
    for a, b, c in np.asarray(faces).tolist():
        adj[a] += [b, c]; adj[b] += [a, c]; adj[c] += [a, b]
    idx = np.full(n, -1, dtype=np.int64)
    sizes = []
    for s in range(n):
        if idx[s] >= 0:
            continue
        cid, queue, k = len(sizes), [s], 0
        idx[s] = cid
        while k < len(queue):
            v = queue[k]; k += 1
            for w in adj[v]:
                if idx[w] < 0:
                    idx[w] = cid
                    queue.append(w)
        sizes.append(len(queue))
    return len(sizes), idx, np.asarray(sizes, dtype=np.int64)


def connected_components(faces, n):
    """the same through scipy's csgraph, renumbered in the order of each component's lowest vertex (any size)"""
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import connected_components as cc
    f = np.asarray(faces, dtype=np.int64).reshape(-1, 3)
    rows = np.concatenate([f[:, 0], f[:, 1], f[:, 2]])
    cols = np.concatenate([f[:, 1], f[:, 2], f[:, 0]])
    g = coo_matrix((np.ones(len(rows), dtype=np.int8), (rows, cols)), shape=(n, n))
    num, lab = cc(g, directed=False)
    first = np.full(num, n, dtype=np.int64)
    np.minimum.at(first, lab, np.arange(n))
    order = np.argsort(first, kind="stable")
    rank = np.empty(num, dtype=np.int64)
    rank[order] = np.arange(num)
    idx = rank[lab]
    return num, idx, np.bincount(idx, minlength=num).astype(np.int64)


def largest_component_mask(faces, n=None):
    """eval.py:497-503: is_cc_vert over the n = faces.max() + 1 vertices igl.adjacency_matrix sees (or a given n >= that)"""
    faces = np.asarray(faces)
    n = int(faces.max()) + 1 if n is None else int(n)
    _, idx, sizes = connected_components(faces, n)
    return idx == np.argmax(sizes)


def remove_holes(verts, faces, pred_value, value_threshold):
    """eval.py:529-548 -> (cc_verts, cc_faces)"""
    on = pred_value > value_threshold
    v1, f1 = delete_invalid_verts(verts, faces, on)
    is_cc = largest_component_mask(f1, len(v1))
    return delete_invalid_verts(v1, f1, is_cc)
