"""CPU: pins against REFERENCE CODE compiled here (oracle/_ref/libref_nanoflann.so = the reference's vendored
nanoflann.hpp + KDTreeVectorOfVectorsAdaptor.h behind oracle/ref_nanoflann_shim.cpp; recipe: `make -C oracle _ref`).

(i)  RSCManager::VanillaKDNNSearch (RadarScancontext.cpp:225-248): the mirror's candidate lists equal the reference
     tree's over a node sequence, including the every-50th-call rebuild and the zero-filled tail.
(ii) FLANN-lineage single kd-tree with L2_Simple on float 2-D points -- what PCL's KdTreeFLANN<PointXY> runs for
     MapPointNormal's radius search (pointnormal.cpp:291) and GetClosestIdx (:238-254) -- against the oracle's restatement
     (SURVEY App. B.2 / B.3): strict `<` radius rule, float accumulation x then y, sorted result order, 1-NN incl. ties.
"""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "oracle", "_ref", "libref_nanoflann.so")


def _ref():
    if not os.path.exists(SO) and os.path.isdir("/root/reference/place_recognition_radar"):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "_ref"])
    if not os.path.exists(SO):
        pytest.skip("oracle/_ref/libref_nanoflann.so not built and /root/reference absent")
    L = C.CDLL(SO)
    L.ref_keytree_build.restype = C.c_void_p
    L.ref_keytree_build.argtypes = [C.c_void_p, C.c_int64, C.c_int32]
    L.ref_keytree_free.argtypes = [C.c_void_p]
    L.ref_keytree_knn.restype = C.c_int64
    L.ref_keytree_knn.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
    L.ref_tree2f_build.restype = C.c_void_p
    L.ref_tree2f_build.argtypes = [C.c_void_p, C.c_int64, C.c_int32]
    L.ref_tree2f_free.argtypes = [C.c_void_p]
    L.ref_tree2f_radius.restype = C.c_int64
    L.ref_tree2f_radius.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p]
    L.ref_tree2f_knn.restype = C.c_int64
    L.ref_tree2f_knn.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
    return L


class RefKeyTree:
    def __init__(self, L, keys):
        self.L = L
        k = np.ascontiguousarray(keys, np.float32)
        self.h = L.ref_keytree_build(k.ctypes.data, k.shape[0], k.shape[1])
        assert self.h

    def knn(self, q, k):
        q = np.ascontiguousarray(q, np.float32)
        idx = np.zeros(k, np.uint64)
        d2 = np.zeros(k, np.float32)
        n = self.L.ref_keytree_knn(self.h, q.ctypes.data, k, idx.ctypes.data, d2.ctypes.data)
        return int(n), idx.astype(np.int64), d2

    def __del__(self):
        self.L.ref_keytree_free(self.h)


class RefTree2f:
    def __init__(self, L, xy, leaf=15):          # pcl::KdTreeFLANN builds KDTreeSingleIndexParams(15)
        self.L = L
        p = np.ascontiguousarray(xy, np.float32)
        self.n = p.shape[0]
        self.h = L.ref_tree2f_build(p.ctypes.data, p.shape[0], leaf)
        assert self.h

    def radius(self, q, r2, sorted_=True):
        q = np.ascontiguousarray(q, np.float32)
        idx = np.zeros(self.n, np.uint64)
        d2 = np.zeros(self.n, np.float32)
        n = self.L.ref_tree2f_radius(self.h, q.ctypes.data, np.float32(r2), int(sorted_), self.n, idx.ctypes.data, d2.ctypes.data)
        return idx[:n].astype(np.int64), d2[:n]

    def knn(self, q, k=1):
        q = np.ascontiguousarray(q, np.float32)
        idx = np.zeros(k, np.uint64)
        d2 = np.zeros(k, np.float32)
        n = self.L.ref_tree2f_knn(self.h, q.ctypes.data, k, idx.ctypes.data, d2.ctypes.data)
        return idx[:n].astype(np.int64), d2[:n]

    def __del__(self):
        self.L.ref_tree2f_free(self.h)


# ---------------------------------------------------------------------------------------------------------------
# (i) ring-key retrieval
# ---------------------------------------------------------------------------------------------------------------
def _lap_keys(n_nodes, seed):
    """Ring keys of a synthetic lap: oracle descriptors of clouds that drift smoothly and revisit the start."""
    from oracle import pyoracle as O
    rng = np.random.default_rng(seed)
    base = np.zeros((2500, 4), np.float32)
    r = rng.uniform(2, 78, base.shape[0]); a = rng.uniform(0, 2 * np.pi, base.shape[0])
    base[:, 0], base[:, 1], base[:, 3] = r * np.cos(a), r * np.sin(a), rng.integers(60, 256, base.shape[0])
    keys = []
    for t in range(n_nodes):
        ph = 2 * np.pi * t / n_nodes
        c = base.copy()
        c[:, 0] += np.float32(12 * np.cos(ph) - 12)
        c[:, 1] += np.float32(12 * np.sin(ph))
        c[:, 3] = np.clip(c[:, 3] + rng.integers(-3, 4, c.shape[0]), 0, 255)
        rk, _ = O.sc_keys(O.sc_descriptor(c))
        keys.append(np.asarray(rk, np.float64).astype(np.float32))
    return keys


def _reference_protocol(L, keys_so_far, num_exclude, state, query, K):
    """RadarScancontext.cpp:227-247 with the tree and the search being the reference's own code."""
    if state["counter"] % 50 == 0:
        n = len(keys_so_far) - num_exclude
        state["tree"] = RefKeyTree(L, np.asarray(keys_so_far[:n], np.float32))
    state["counter"] += 1
    n, idx, d2 = state["tree"].knn(query, K)
    return n, [int(i) for i in idx], d2


@pytest.mark.parametrize("seed", [0, 1])
def test_vanilla_kdnn_search_matches_reference_tree(seed):
    from tbv_slam_public_amd import api
    L = _ref()
    keys = _lap_keys(130, seed)
    rng = np.random.default_rng(100 + seed)
    mgr = api.RSCManager(odometry_coupled_closure=False, augment_sc=True)
    state = {"counter": 0, "tree": None}
    K = mgr.NUM_CANDIDATES_FROM_TREE
    n_checked = n_tail = 0
    for t, key in enumerate(keys):
        mgr.polarcontext_invkeys_mat_.append(key)
        mgr.NUM_EXCLUDE_RECENT = 2 if t < 3 else int(rng.integers(3, 9))       # what the odometry walk would set (:181-199)
        if len(mgr.polarcontext_invkeys_mat_) < mgr.NUM_EXCLUDE_RECENT + 1:     # detectLoopClosureID's early return (:288)
            continue
        queries = [key] + [key + rng.normal(0, 0.02, key.shape).astype(np.float32) for _ in range(4)]   # current + 4 augments
        for q in queries:
            mine = mgr._vanilla_nn_search(q)
            n, ref_idx, ref_d2 = _reference_protocol(L, mgr.polarcontext_invkeys_mat_, mgr.NUM_EXCLUDE_RECENT, state, q, K)
            assert len(mine) == K
            # found neighbours: the same nodes in the same order, unless two distances are EQUAL floats (then only the
            # set is defined: tree-visiting order there, index order here)
            d_mine = api.RSCManager._l2_adaptor(np.asarray(mgr._tree_keys), q)
            assert np.array_equal(d_mine[mine[:n]], ref_d2[:n]), (t, d_mine[mine[:n]], ref_d2[:n])
            if len(set(ref_d2[:n].tolist())) == n:
                assert mine[:n] == ref_idx[:n], (t, mine, ref_idx)
            else:
                assert sorted(mine[:n]) == sorted(ref_idx[:n])
            assert mine[n:] == ref_idx[n:] == [0] * (K - n)                     # the zero-initialised tail shows through
            n_tail += K - n
            n_checked += 1
    assert n_checked > 500 and n_tail > 0
    assert state["counter"] > 100                                              # the rebuild period was crossed


def test_kdnn_exact_ties_are_sets_only():
    """Duplicate keys: the tree returns both, in its own visiting order; the mirror in index order."""
    L = _ref()
    rng = np.random.default_rng(5)
    keys = rng.random((60, 40)).astype(np.float32)
    keys[17] = keys[3]; keys[44] = keys[3]
    tree = RefKeyTree(L, keys)
    n, idx, d2 = tree.knn(keys[3], 5)
    assert n == 5 and set(idx[:3].tolist()) == {3, 17, 44} and np.all(d2[:3] == 0)


# ---------------------------------------------------------------------------------------------------------------
# (ii) float 2-D, L2_Simple
# ---------------------------------------------------------------------------------------------------------------
def _scan_cloud(seed):
    from oracle import pyoracle as O
    from tbv_slam_public_amd import synth
    imgs, _, _ = synth.scene_v1(seed, 2)
    sr, si, sc = O.kstrongest(imgs[1], 40, 60)
    return O.kstrongest_cloud(sr, si, sc, 0.0438, 2.5)


def _l2_simple(q, pts):
    dx = np.float32(q[0]) - pts[:, 0]
    dy = np.float32(q[1]) - pts[:, 1]
    d = np.zeros(pts.shape[0], np.float32)
    d = d + dx * dx
    d = d + dy * dy
    return d


@pytest.mark.parametrize("seed", [0, 3])
def test_radius_search_sets_and_order(seed):
    """Every voxel centroid's radius search over the filtered cloud: the oracle's neighbour rule (orc_surface_points:
    float d accumulated x then y, d < float(radius * radius), stable sort by d) against the reference tree."""
    from oracle import pyoracle as O
    L = _ref()
    cloud = _scan_cloud(seed)
    pts = np.ascontiguousarray(cloud[:, :2], np.float32)
    tree = RefTree2f(L, pts)
    radius = np.float32(3.0)
    r2 = np.float32(np.float64(radius) * np.float64(radius))
    _, cen = O.surface_points(cloud, radius, return_centroids=True)
    n_tie_order = n_ge6 = 0
    for v in range(cen.shape[0]):
        idx, d2 = tree.radius(cen[v], r2)
        d = _l2_simple(cen[v], pts)
        mine = np.nonzero(d < r2)[0]
        assert set(idx.tolist()) == set(mine.tolist()), v                      # strict <, float arithmetic: same set
        assert np.array_equal(d2, d[idx])                                       # the same float distances, bit for bit
        assert np.all(np.diff(d2) >= 0)                                         # sorted ascending
        order = mine[np.argsort(d[mine], kind="stable")]                        # the oracle's order: (distance, input index)
        if not np.array_equal(order, idx):
            # only ever inside groups of EQUAL distances (std::sort is not stable)
            assert np.array_equal(d[order], d[idx])
            n_tie_order += 1
        n_ge6 += len(mine) >= 6
    assert n_ge6 > 100
    print("voxels", cen.shape[0], "with a different order among equal distances:", n_tie_order)


def test_radius_boundary_is_strict():
    L = _ref()
    pts = np.array([[3.0, 0.0], [0.0, 3.0], [2.9999998, 0.0], [1.0, 1.0], [-3.0, 0.0], [0, 0], [0.5, 0.5]], np.float32)
    tree = RefTree2f(L, pts)
    idx, d2 = tree.radius(np.zeros(2, np.float32), np.float32(9.0))
    assert set(idx.tolist()) == {2, 3, 5, 6}                                    # d == r^2 is outside
    assert np.array_equal(np.sort(idx), np.nonzero(_l2_simple((0, 0), pts) < np.float32(9.0))[0])


@pytest.mark.parametrize("seed", [0, 3])
def test_nearest_neighbour_matches_oracle_closest_idx(seed):
    """GetClosestIdx (pointnormal.cpp:238-254): nearestKSearch(1) on the float means, accepted iff d2 < d * d."""
    from oracle import pyoracle as O
    L = _ref()
    cloud = _scan_cloud(seed)
    cells = O.surface_points(cloud, 3.0, weight_intensity=True)
    means_f = np.ascontiguousarray(cells["mean"], np.float64).astype(np.float32)
    tree = RefTree2f(L, means_f)
    rng = np.random.default_rng(seed)
    th, t = 0.02, np.array([0.7, -0.4])
    Rm = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
    q = cells["mean"] @ Rm.T + t                                                # a moved copy of the scan as queries
    q = np.concatenate([q, rng.uniform(-60, 60, (300, 2))])
    for d in (2.0, 4.0):
        mine = O.closest_idx(cells, q, d)
        for i in range(q.shape[0]):
            idx, d2 = tree.knn(q[i].astype(np.float32), 1)
            ref = int(idx[0]) if float(d2[0]) < d * d else -1                   # :249-252
            if ref != mine[i]:
                # equal float distances: the tree answers in visiting order, the oracle with the lowest index
                dd = _l2_simple(q[i].astype(np.float32), means_f)
                assert ref >= 0 and mine[i] >= 0 and dd[ref] == dd[mine[i]], (i, ref, mine[i])


def test_nearest_neighbour_ties():
    """Deliberate ties: where two targets are exactly equidistant the reference tree's answer depends on its split order;
    the oracle (and the GPU matcher) take the lowest index.  This test documents which way the reference goes on a few
    symmetric layouts and asserts only what is defined: the returned distance is the minimum."""
    L = _ref()
    pts = np.array([[1, 0], [-1, 0], [0, 1], [0, -1], [5, 5], [5, -5], [-5, 5], [-5, -5], [1, 0]], np.float32)
    tree = RefTree2f(L, pts)
    idx, d2 = tree.knn(np.zeros(2, np.float32), 1)
    assert d2[0] == 1.0 and int(idx[0]) in (0, 1, 2, 3, 8)
    print("tie at the origin among {0,1,2,3,8}: reference returns", int(idx[0]), "(lowest-index rule gives 0)")


def test_tie_rule_on_oracle_cells():
    """Which way does the one FLANN-lineage tree of this image go on the oracle's OWN cell sets?  (VERDICT r05 #5.)

    The oracle (and the HIP matcher, bit for bit) returns the LOWEST cell index among equidistant targets; the reference asks
    pcl::KdTreeFLANN<PointXY>::nearestKSearch(k = 1) (pointnormal.cpp:249), i.e. FLANN 1.9.1's KDTreeSingleIndex with
    KDTreeSingleIndexParams(15) (pointnormal.cpp:151-162) -- absent here.  Its header-only descendant, the reference's vendored
    nanoflann (same middleSplit, same leaf scan with a strict `<` against the worst distance), is compiled as oracle/_ref.
    Over the scene pairs of tests/test_oracle_order_robustness.py (CFEAR-3: P2P, Huber 0.1, weight option 4):

      * the keyframe's float means go into Tree2f with leaf size 15; the source's means, moved by the start pose AND by the
        registered pose, are the queries -- the searches Register runs in its first and in its last association pass;
      * a query whose two nearest targets are equidistant in float is a TIE; counted: how often the tree's 1-NN index differs
        from the oracle's lowest-index answer (the distances always agree: asserted);
      * every group of cells with bit-identical float means gets the tree's winner moved to the group's lowest index (a
        permutation of the target's cells: nothing else of the problem changes) and the pair is registered again -- how far the
        pose moves when the oracle is given the tree's tie rule.

    Measured (this container, 200 pairs, 134 892 queries): 8 802 queries (6.5 %) have an exact tie for the nearest target, and the
    tree answers 4 313 of them (49 % of the ties, 3.2 % of all queries, on 196 of 200 pairs) with another index than the lowest
    -- it takes whichever duplicate its leaf holds first, a coin toss against the cell index.  Giving the oracle the tree's
    winners moves the registered pose by: median 1.9e-16 m, 90 % of the pairs <= 1e-6 m, 96 % <= 1e-4 m, worst 3.3e-3 m / 1.3e-4
    rad.  So against a FLANN-lineage tie rule ~4 % of CFEAR-3 registrations differ by more than the 1e-4 m budget -- the same
    figure as flipping EVERY tie (test_oracle_order_robustness.py), because half the ties flip and the worst pairs are the same.
    The disagreement is far above 1 % of pairs, which the round-5 review asked to be REPORTED rather than built: reproducing a
    leaf order on the GPU needs the reference's tree (build order, split rule, leaf size), and only a reference-written golden
    (tools/ref_golden, `tie_*` arrays) can say whether FLANN 1.9.1 orders its leaves like nanoflann.
    """
    from oracle import pyoracle as O
    from tbv_slam_public_amd import synth
    L = _ref()
    par = O.reg_params(cost="P2P", loss="Huber", loss_limit=0.1, weight_opt=4)
    n_pairs = int(os.environ.get("CFEAR_TIE_PAIRS", "200"))
    queries = ties = differ = pairs_with_diff = 0
    dpos, drot = [], []
    for seed in range(1000, 1000 + n_pairs):
        imgs, gt, _ = synth.scene_v1(seed, 2)
        cells = []
        for f in range(2):
            sr, si, sc = O.kstrongest(imgs[f], 40, 60)
            cells.append(O.surface_points(O.kstrongest_cloud(sr, si, sc, 0.0438, 2.5), 3.0, 1.0, (0, 0), True))
        c, s = np.cos(gt[0][2]), np.sin(gt[0][2])
        d = gt[1][:2] - gt[0][:2]
        guess = np.array([c * d[0] + s * d[1], -s * d[0] + c * d[1], gt[1][2] - gt[0][2]]) + [0.3, -0.2, 0.01]
        poses = np.array([[0.0, 0.0, 0.0], guess])
        ok, p_can, r_can = O.register(cells, poses, par)
        assert ok
        tgt = cells[0]
        mf = np.ascontiguousarray(tgt["mean"], np.float64).astype(np.float32)
        tree = RefTree2f(L, mf, leaf=15)
        pair_diff = 0
        for pose in (guess, p_can[-1]):
            cs, sn = np.cos(pose[2]), np.sin(pose[2])
            q = cells[1]["mean"] @ np.array([[cs, -sn], [sn, cs]]).T + pose[:2]
            mine = O.closest_idx(tgt, q, 1e9)                        # the oracle's 1-NN (no radius cut: the tie rule alone)
            for i in range(q.shape[0]):
                qf = q[i].astype(np.float32)
                idx, d2 = tree.knn(qf, 2)
                dd = _l2_simple(qf, mf)
                assert d2[0] == dd[mine[i]]                          # same nearest DISTANCE, always
                queries += 1
                if d2.shape[0] > 1 and d2[1] == d2[0]:
                    ties += 1
                    one, _ = tree.knn(qf, 1)
                    if int(one[0]) != int(mine[i]):
                        differ += 1
                        pair_diff += 1
        pairs_with_diff += pair_diff > 0
        # the tree's winner of every duplicate group -> the group's lowest index
        order = np.arange(len(tgt))
        _, inv, cnt = np.unique(mf, axis=0, return_inverse=True, return_counts=True)
        for g in np.nonzero(cnt > 1)[0]:
            members = np.nonzero(inv == g)[0]                        # ascending cell indices
            win, _ = tree.knn(mf[members[0]], 1)
            w = int(win[0])
            assert w in members
            if w != members[0]:
                order[members[0]], order[w] = order[w], order[members[0]]
        ok2, p_tree, r_tree = O.register([tgt[order].copy(), cells[1]], poses, par)
        assert ok2
        dpos.append(np.abs(p_can[-1, :2] - p_tree[-1, :2]).max())
        drot.append(abs(p_can[-1, 2] - p_tree[-1, 2]))
    dpos, drot = np.array(dpos), np.array(drot)
    print("tie rule: %d pairs, %d queries, %d ties (%.2f %%), tree != lowest index on %d (%.1f %% of ties, %.2f %% of queries, %d pairs)"
          % (n_pairs, queries, ties, 100.0 * ties / queries, differ, 100.0 * differ / max(ties, 1), 100.0 * differ / queries, pairs_with_diff))
    print("pose moved by the tree's tie rule: median %.1e m, <=1e-6: %.0f %%, <=1e-4: %.0f %%, worst %.1e m / %.1e rad"
          % (np.median(dpos), 100 * (dpos <= 1e-6).mean(), 100 * (dpos <= 1e-4).mean(), dpos.max(), drot.max()))
    assert ties > 0 and differ <= ties
    assert dpos.max() <= 2e-2 and drot.max() <= 1e-3                 # inside the all-ties-flipped bound (3.3e-3 m measured there)
    assert np.median(dpos) <= 1e-9
