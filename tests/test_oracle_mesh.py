"""CPU: the oracle's restatement of the reference's mesh post-processing (eval.py:497-548) -- the literal BFS form of libigl's connected
components against scipy's csgraph on random meshes, known answers for the numbering / tie rule, delete_invalid_verts on a hand case."""
import numpy as np
import pytest

from oracle import mesh as OM


def _random_faces(rng, n, f):
    return rng.integers(0, n, size=(f, 3)).astype(np.int32)


@pytest.mark.parametrize("n,f,seed", [(12, 6, 0), (60, 25, 1), (300, 90, 2), (1000, 1500, 3), (50, 0, 4)])
def test_bfs_numbering_equals_scipy_renumbered(n, f, seed):
    rng = np.random.default_rng(seed)
    faces = _random_faces(rng, n, f)
    a, b = OM.connected_components_bfs(faces, n), OM.connected_components(faces, n)
    assert a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    assert a[2].sum() == n and a[1][0] == 0                                 # vertex 0 is always in component 0


def test_known_components_and_first_largest_wins():
    # two triangles {2,3,4} and {0,5,6} + the isolated vertex 1: components in the order of their lowest vertex: {0,5,6}, {1}, {2,3,4}
    faces = np.array([[2, 3, 4], [5, 0, 6]], dtype=np.int32)
    num, idx, sizes = OM.connected_components_bfs(faces, 7)
    assert num == 3 and idx.tolist() == [0, 1, 2, 2, 2, 0, 0] and sizes.tolist() == [3, 1, 3]
    assert OM.largest_component_mask(faces).tolist() == [True, False, False, False, False, True, True]     # tie: the first (argmax)
    # a bigger component elsewhere takes over
    faces2 = np.vstack([faces, [[2, 4, 7]]]).astype(np.int32)
    assert OM.largest_component_mask(faces2).tolist() == [False, False, True, True, True, False, False, True]


def test_remove_holes_on_a_strip():
    # a strip of 4 triangles 0-1-2, 1-2-3, 2-3-4, 3-4-5 and a far triangle 6-7-8; vertex 2 predicted off the surface:
    # only faces without vertex 2 survive: {3,4,5} and {6,7,8} -> tie -> the one with the lowest (renumbered) vertex: old 3,4,5
    verts = np.arange(27, dtype=np.float64).reshape(9, 3)
    faces = np.array([[0, 1, 2], [1, 2, 3], [2, 3, 4], [3, 4, 5], [6, 7, 8]], dtype=np.int32)
    value = np.ones(9); value[2] = 0.0
    v, f = OM.remove_holes(verts, faces, value, 0.5)
    assert np.array_equal(v, verts[[3, 4, 5]]) and f.tolist() == [[0, 1, 2]]
    v1, f1 = OM.delete_invalid_verts(verts, faces, value > 0.5)
    assert np.array_equal(v1, verts[[3, 4, 5, 6, 7, 8]]) and f1.tolist() == [[0, 1, 2], [3, 4, 5]]
