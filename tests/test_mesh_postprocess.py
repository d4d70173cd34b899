"""Known-answer tests for the CPU mesh post-process, restating tests/test_marching_cubes.cpp:12-271
(combine / removeDuplicateFacesTriangle / removeDuplicateVerticesTriangle) against the oracle's versions."""
import ctypes as C

import numpy as np


def _dedup_vertices(oracle, V, F, eps):
    V = np.ascontiguousarray(V, np.float64).reshape(-1, 3)
    F = np.ascontiguousarray(F, np.int32).reshape(-1, 3)
    oracle.orc_remove_duplicate_vertices.restype = C.c_int64
    oracle.orc_remove_duplicate_vertices.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]
    oV = np.zeros((max(len(V), 1), 3), np.float64)
    oF = np.zeros((max(len(F), 1), 3), np.int32)
    m = np.zeros(max(len(V), 1), np.int32)
    nu = oracle.orc_remove_duplicate_vertices(V.ctypes.data, len(V), F.ctypes.data, len(F), eps, oV.ctypes.data, oF.ctypes.data, m.ctypes.data)
    return oV[:nu], oF[: len(F)], m[: len(V)]


def _dedup_faces(oracle, F):
    F = np.ascontiguousarray(F, np.int32).reshape(-1, 3)
    oracle.orc_remove_duplicate_faces.restype = C.c_int64
    oracle.orc_remove_duplicate_faces.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
    o = np.zeros((max(len(F), 1), 3), np.int32)
    n = oracle.orc_remove_duplicate_faces(F.ctypes.data, len(F), o.ctypes.data)
    return o[:n]


def test_remove_duplicate_faces(oracle):
    # REMOVE_DUPL_FACES.duplicates / no_duplicates / empty
    assert len(_dedup_faces(oracle, [[0, 1, 2], [1, 2, 3], [0, 1, 2], [1, 2, 3]])) == 2
    u = _dedup_faces(oracle, [[0, 1, 2], [1, 2, 3], [0, 2, 3]])
    assert np.array_equal(u, [[0, 1, 2], [1, 2, 3], [0, 2, 3]])  # keeps first occurrences, in order
    assert _dedup_faces(oracle, np.zeros((0, 3), np.int32)).shape == (0, 3)


def test_remove_duplicate_vertices_basic(oracle):
    # REMOVE_DUPL_VERTICES.basic_zero / basic_nonzero
    V = [[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 0], [1, 0, 0], [0, 1, 0]]
    F = [[0, 1, 2], [1, 3, 2]]
    for eps in (0.0, 1e-12):
        uV, uF, m = _dedup_vertices(oracle, V, F, eps)
        assert uV.shape == (3, 3) and len(m) == 6 and uF.shape == (2, 3)
        assert np.allclose(uV, np.array(V, float)[:3], atol=1e-12)
        assert np.allclose(uV[m], np.array(V, float), atol=1e-12)
        assert np.array_equal(uF, m[np.array(F)])


def test_remove_duplicate_vertices_edge_cases(oracle):
    # REMOVE_DUPL_VERTICES.empty_vertices / no_duplicates
    uV, uF, m = _dedup_vertices(oracle, np.zeros((0, 3)), np.zeros((0, 3), np.int32), 1e-12)
    assert uV.shape == (0, 3) and len(m) == 0
    V = [[0, 0, 0], [1, 0, 0], [0, 1, 0]]
    uV, uF, m = _dedup_vertices(oracle, V, [[0, 1, 2]], 1e-12)
    assert np.array_equal(m, [0, 1, 2]) and np.allclose(uV, V) and np.array_equal(uF, [[0, 1, 2]])


def test_combine_is_index_offset(oracle):
    # COMBINE.simple_combine: concatenating meshes offsets the second mesh's face indices by the vertex count.
    # processTriangles builds exactly that (face i = (3i, 3i+1, 3i+2) over the concatenated soup).
    V1 = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [1, 1, 0]], float)
    F1 = np.array([[0, 1, 2], [1, 3, 2]], np.int32)
    V2 = V1 + [0, 0, 1]
    F2 = F1.copy()
    V = np.vstack([V1, V2])
    F = np.vstack([F1, F2 + len(V1)])
    uV, uF, m = _dedup_vertices(oracle, V, F, 0.0)
    assert len(uV) == 8 and np.array_equal(uF, F)
