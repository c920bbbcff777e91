"""simple_graph.sgh (SaveSimpleGraph / LoadSimpleGraph, types.cpp:103-130) through cfear_graph_save / cfear_graph_load: host
code only.  No reference-produced .sgh exists (the reference's own test refers to a file that is not committed), so the
Boost 1.71 binary-archive layout is UNPINNED against a real file; tested here: the round trip is lossless, and the bytes
of small hand-checked cases follow the layout rules restated in csrc/graph.hip."""
import os
import struct

import numpy as np
import pytest

from tbv_slam_public_amd import _lib as L
from tbv_slam_public_amd import api


def _cells(n, rng):
    c = np.zeros(n, L.CELL_DTYPE)
    c["mean"] = rng.normal(0, 30, (n, 2))
    c["normal"] = rng.normal(0, 1, (n, 2))
    c["cov"] = rng.uniform(0.01, 1, (n, 4))
    c["cov"][:, 2] = c["cov"][:, 1]
    c["scale"], c["avg_intensity"] = rng.uniform(0, 5, n), rng.uniform(0, 100, n)
    c["lambda_min"], c["lambda_max"] = rng.uniform(0.01, 0.1, n), rng.uniform(0.1, 2, n)
    c["nsamples"] = rng.integers(6, 500, n)
    return c


def _node(i, rng, n_pts=50, n_cells=20):
    cloud = rng.normal(0, 40, (n_pts, 4)).astype(np.float32)
    cloud[:, 2] = 0
    return dict(T=(1.5 * i, -0.2 * i, 0.1 * i), Tgt=(1.5 * i + 0.01, 0.0, 0.1 * i), has_Tgt=i % 2, idx=i, stamp=1547120000000000000 + i,
                motion=np.array([[1, 0, 0, 2.5], [0, 1, 0, 0.1], [0, 0, 1, 0], [0, 0, 0, 1.0]]),
                cloud_peaks=dict(xyzi=cloud[: n_pts // 3], stamp=77 + i, seq=i, frame_id="sensor_est"),
                cloud_nopeaks=dict(xyzi=cloud, stamp=78 + i, seq=i, frame_id="sensor_est"),
                cells=_cells(n_cells, rng), radius=3.0, weight_intensity=1,
                constraints=[] if i == 0 else [dict(id_begin=i, id_end=i - 1, t_be=(-1.5, 0.2, -0.1), information=np.diag([100.0, 100, 0, 0, 0, 1e4]),
                                                   type=0, quality={"sc-sim": 0.25, "alignment_quality": 0.9, "odom-bounds": 0.1}, info="odom")])


def test_round_trip_is_lossless(tmp_path):
    rng = np.random.default_rng(0)
    nodes = [_node(i, rng) for i in range(5)]
    nodes[3]["cloud_peaks"] = None                               # a null shared_ptr
    nodes[4]["cells"] = None                                     # cloud_normal_ == NULL
    nodes[2]["input_is_nopeaks"] = False                         # a MapPointNormal built from another cloud object
    nodes[2]["normal_input"] = rng.normal(0, 1, (7, 4)).astype(np.float32)
    path = str(tmp_path / "simple_graph.sgh")
    api.SaveSimpleGraph(path, nodes)
    back = api.LoadSimpleGraph(path)
    assert len(back) == 5
    for a, b in zip(nodes, back):
        assert (b["idx"], b["stamp"], b["has_Tgt"]) == (a["idx"], a["stamp"], bool(a["has_Tgt"]))
        np.testing.assert_allclose(b["T_xyt"], a["T"], atol=1e-15)
        np.testing.assert_array_equal(b["motion"], a["motion"])
        for key in ("cloud_peaks", "cloud_nopeaks"):
            if a[key] is None:
                assert b[key] is None
            else:
                np.testing.assert_array_equal(b[key]["xyzi"], a[key]["xyzi"])
                assert (b[key]["stamp"], b[key]["seq"], b[key]["frame_id"]) == (a[key]["stamp"], a[key]["seq"], a[key]["frame_id"])
        if a["cells"] is None:
            assert b["cells"] is None
        else:
            for name in ("mean", "normal", "cov", "scale", "avg_intensity", "lambda_min", "lambda_max", "nsamples"):
                np.testing.assert_array_equal(b["cells"][name], a["cells"][name], err_msg=name)
            assert b["radius"] == 3.0 and b["weight_intensity"] and b["input_is_nopeaks"] == a.get("input_is_nopeaks", True)
        assert len(b["constraints"]) == len(a["constraints"])
        for ca, cb in zip(a["constraints"], b["constraints"]):
            assert (cb["id_begin"], cb["id_end"], cb["type"], cb["info"], cb["quality"]) == (ca["id_begin"], ca["id_end"], 0, "odom", ca["quality"])
            np.testing.assert_array_equal(cb["information"], ca["information"])
    np.testing.assert_array_equal(back[2]["normal_input"]["xyzi"], nodes[2]["normal_input"])
    # saving what was loaded reproduces the file byte for byte
    again = str(tmp_path / "again.sgh")
    for nd in back:
        nd["T"], nd["Tgt"] = nd["T"], nd["Tgt"]
    api.SaveSimpleGraph(again, back)
    assert open(path, "rb").read() == open(again, "rb").read()


def test_archive_header_and_first_records_follow_boost_layout(tmp_path):
    """Hand-checked bytes: header = u64 22 + "serialization::archive" + u16 17 + the native type sizes 04 08 04 08 and
    int32 1 (basic_binary_oprimitive::init, which binary_oarchive_impl::init calls after the signature); the vector's class info (tracking 0,
    version 0), count u64, item_version u32; the pair's and RadarScan's class info; Pose3d + Vector3d class info, then
    three doubles."""
    path = str(tmp_path / "one.sgh")
    api.SaveSimpleGraph(path, [dict(T=(1.0, 2.0, 0.0), idx=7, stamp=9, cloud_peaks=None, cloud_nopeaks=None, cells=None)])
    b = open(path, "rb").read()
    o = 0
    assert struct.unpack_from("<Q", b, o)[0] == 22 and b[8:30] == b"serialization::archive"
    o = 30
    assert struct.unpack_from("<H", b, o)[0] == 17
    o += 2
    assert b[o:o + 4] == bytes([4, 8, 4, 8]) and struct.unpack_from("<i", b, o + 4)[0] == 1   # sizeof int/long/float/double, endian marker
    o += 8
    for _ in range(1):                                            # simple_graph (vector): tracking, version
        assert b[o] == 0 and struct.unpack_from("<I", b, o + 1)[0] == 0
        o += 5
    assert struct.unpack_from("<Q", b, o)[0] == 1 and struct.unpack_from("<I", b, o + 8)[0] == 0   # count, item_version
    o += 12
    for _ in range(4):                                            # pair, RadarScan, Pose3d, Vector3d: class info each
        assert b[o] == 0 and struct.unpack_from("<I", b, o + 1)[0] == 0
        o += 5
    assert struct.unpack_from("<3d", b, o) == (1.0, 2.0, 0.0)
    o += 24
    assert b[o] == 0 and struct.unpack_from("<I", b, o + 1)[0] == 0      # Quaterniond class info
    o += 5
    assert struct.unpack_from("<4d", b, o) == (0.0, 0.0, 0.0, 1.0)       # x, y, z, w
    o += 32
    # Tgt: Pose3d / Vector3d / Quaterniond are initialised -> data only (3 + 4 doubles)
    o += 56
    assert b[o] == 0                                                     # has_Tgt_
    assert struct.unpack_from("<I", b, o + 1)[0] == 7 and struct.unpack_from("<Q", b, o + 5)[0] == 9
    o += 13
    o += 5 + 128                                                         # Affine3d class info + 16 doubles
    # three null shared_ptrs: shared_ptr<PointCloud> class info (tracking 0, version 1) once, then class_id -1 each
    assert b[o] == 0 and struct.unpack_from("<I", b, o + 1)[0] == 1
    assert struct.unpack_from("<h", b, o + 5)[0] == -1 and struct.unpack_from("<h", b, o + 7)[0] == -1
    o += 9
    assert b[o] == 0 and struct.unpack_from("<I", b, o + 1)[0] == 1      # shared_ptr<MapPointNormal> class info
    assert struct.unpack_from("<h", b, o + 5)[0] == -1
    o += 7
    assert b[o] == 0 and struct.unpack_from("<I", b, o + 1)[0] == 0      # vector<Constraint3d> class info
    assert struct.unpack_from("<Q", b, o + 5)[0] == 0
    assert len(b) == o + 5 + 12


def test_shared_cloud_is_stored_once(tmp_path):
    """cloud_normal_->input_ is the cloud_nopeaks_ object in every node the fuser builds (odometrykeyframefuser.cpp:161,
    244): the tracked pointer is written once and referenced by object id -- the file grows by the cloud only once."""
    rng = np.random.default_rng(1)
    nd = _node(0, rng, n_pts=1000)
    shared = str(tmp_path / "a.sgh")
    api.SaveSimpleGraph(shared, [nd])
    nd2 = dict(nd, input_is_nopeaks=False, normal_input=np.array(nd["cloud_nopeaks"]["xyzi"]))    # equal content, another object
    twice = str(tmp_path / "b.sgh")
    api.SaveSimpleGraph(twice, [nd2])
    per_point = 5 * 0 + 8 + 16 + 4                                # count + float[4] + intensity
    assert os.path.getsize(twice) - os.path.getsize(shared) >= 1000 * per_point
    assert api.LoadSimpleGraph(shared)[0]["input_is_nopeaks"] and not api.LoadSimpleGraph(twice)[0]["input_is_nopeaks"]


def test_one_buffer_in_several_slots_is_one_tracked_object(tmp_path):
    """Boost writes an object reached through several shared_ptrs once and refers back to it by object id
    (basic_oarchive tracking).  The caller's buffer is the identity here: one array as cloud_peaks_ AND cloud_nopeaks_ of a
    node, and again as the peaks cloud of the NEXT node, is stored once; the loader resolves every back-reference --
    across slots and across nodes -- to the stored cloud."""
    rng = np.random.default_rng(3)
    a, b = _node(0, rng, n_pts=700), _node(1, rng, n_pts=300)
    one = a["cloud_nopeaks"]                                    # dict(xyzi = ONE array, ...)
    copy = lambda: dict(one, xyzi=np.array(one["xyzi"]))
    sep = str(tmp_path / "separate.sgh")
    api.SaveSimpleGraph(sep, [dict(a, cloud_peaks=copy()), dict(b, cloud_peaks=copy())])
    shr = str(tmp_path / "shared.sgh")
    api.SaveSimpleGraph(shr, [dict(a, cloud_peaks=one), dict(b, cloud_peaks=one)])   # peaks = nopeaks of node 0 = peaks of node 1
    per_point = 8 + 16 + 4
    assert os.path.getsize(sep) - os.path.getsize(shr) >= 2 * 700 * per_point
    for path in (sep, shr):
        g = api.LoadSimpleGraph(path)
        for nd in g:
            np.testing.assert_array_equal(nd["cloud_peaks"]["xyzi"], one["xyzi"])
        np.testing.assert_array_equal(g[0]["cloud_nopeaks"]["xyzi"], one["xyzi"])
        np.testing.assert_array_equal(g[1]["cloud_nopeaks"]["xyzi"], b["cloud_nopeaks"]["xyzi"])
        assert g[0]["input_is_nopeaks"] and g[1]["input_is_nopeaks"]


def test_small_nodes_pass_the_size_bounds(tmp_path):
    """The file-size bounds must be true LOWER bounds: 50 pose-only nodes (273 bytes each) and 50 nodes with one-point clouds
    and a single cell load again (round 3's bound of 400 bytes per node rejected both with CFEAR_ERR_FORMAT)."""
    rng = np.random.default_rng(5)
    bare = [dict(_node(i, rng), cloud_peaks=None, cloud_nopeaks=None, cells=None, constraints=[]) for i in range(50)]
    p1 = str(tmp_path / "bare.sgh")
    api.SaveSimpleGraph(p1, bare)
    assert os.path.getsize(p1) < 50 * 400
    back = api.LoadSimpleGraph(p1)
    assert [b["idx"] for b in back] == list(range(50)) and all(b["cells"] is None and b["cloud_peaks"] is None for b in back)
    tiny = [dict(_node(i, rng, n_pts=3, n_cells=1), constraints=[]) for i in range(50)]
    p2 = str(tmp_path / "tiny.sgh")
    api.SaveSimpleGraph(p2, tiny)
    back = api.LoadSimpleGraph(p2)
    for a, b in zip(tiny, back):
        np.testing.assert_array_equal(b["cloud_nopeaks"]["xyzi"], a["cloud_nopeaks"]["xyzi"])
        np.testing.assert_array_equal(b["cells"]["mean"], a["cells"]["mean"])


def test_one_buffer_under_two_headers_is_two_objects(tmp_path):
    """The save-side identity of a cloud is (buffer, size, stamp, seq, frame_id): the same array handed in with another
    header is another pcl::PointCloud object and must come back with ITS header, not the first one's."""
    rng = np.random.default_rng(6)
    a, b = _node(0, rng, n_pts=40), _node(1, rng, n_pts=40)
    arr = a["cloud_nopeaks"]["xyzi"]
    b = dict(b, cloud_nopeaks=dict(xyzi=arr, stamp=999, seq=41, frame_id="other"), cloud_peaks=None)
    path = str(tmp_path / "hdr.sgh")
    api.SaveSimpleGraph(path, [a, b])
    g = api.LoadSimpleGraph(path)
    assert (g[0]["cloud_nopeaks"]["stamp"], g[0]["cloud_nopeaks"]["frame_id"]) == (a["cloud_nopeaks"]["stamp"], "sensor_est")
    assert (g[1]["cloud_nopeaks"]["stamp"], g[1]["cloud_nopeaks"]["seq"], g[1]["cloud_nopeaks"]["frame_id"]) == (999, 41, "other")
    np.testing.assert_array_equal(g[1]["cloud_nopeaks"]["xyzi"], arr)


def test_sizes_in_the_file_are_bounded_by_the_file(tmp_path):
    """Every count the archive dictates (nodes, points, cells, constraints) is checked against what is left of the file
    before anything is allocated: a 70-byte file that announces 16 million nodes is a format error, not a multi-GB
    allocation."""
    import struct
    rng = np.random.default_rng(4)
    good = str(tmp_path / "g.sgh")
    api.SaveSimpleGraph(good, [_node(0, rng)])
    data = bytearray(open(good, "rb").read())
    sig = 8 + len("serialization::archive") + 2 + 8           # string, version, native sizes + endianness marker
    off = sig + 5                                                # class info of the vector (tracking byte + version)
    assert struct.unpack_from("<Q", data, off)[0] == 1          # the node count
    struct.pack_into("<Q", data, off, (1 << 24) - 1)
    bad = tmp_path / "huge_count.sgh"
    bad.write_bytes(bytes(data[:off + 12 + 40]))
    with pytest.raises(L.CfearError) as e:
        api.LoadSimpleGraph(str(bad))
    assert e.value.status == L.ERR_FORMAT


def test_bad_files_are_status_codes(tmp_path):
    with pytest.raises(L.CfearError) as e:
        api.LoadSimpleGraph(str(tmp_path / "missing.sgh"))
    assert e.value.status == L.ERR_IO
    p = tmp_path / "garbage.sgh"
    p.write_bytes(b"\x16\0\0\0\0\0\0\0not an archive at all......")
    with pytest.raises(L.CfearError) as e:
        api.LoadSimpleGraph(str(p))
    assert e.value.status == L.ERR_FORMAT
    rng = np.random.default_rng(2)
    good = str(tmp_path / "g.sgh")
    api.SaveSimpleGraph(good, [_node(i, rng) for i in range(2)])
    data = open(good, "rb").read()
    (tmp_path / "cut.sgh").write_bytes(data[: len(data) // 2])
    with pytest.raises(L.CfearError) as e:
        api.LoadSimpleGraph(str(tmp_path / "cut.sgh"))
    assert e.value.status == L.ERR_FORMAT


def test_reference_written_graph_loads():
    """A simple_graph.sgh written by the reference's own SaveSimpleGraph (tools/ref_golden: three RadarScan nodes from the golden
    clouds).  Not committed yet -- no reference build exists in this image -- so this skips and the archive layout stays
    pinned to Boost's documented format only."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_simple_graph.sgh")
    if not os.path.exists(path):
        pytest.skip("no ref_simple_graph.sgh: run tools/ref_golden in the reference's image (tools/ref_golden/README.md)")
    reg = np.load(os.path.join(os.path.dirname(path), "registration.npz"))
    g = api.LoadSimpleGraph(path)
    assert len(g) == 3
    for i, nd in enumerate(g):
        np.testing.assert_allclose(nd["T_xyt"], reg["poses"][i], atol=1e-12)
        assert nd["cloud_nopeaks"]["xyzi"].shape[0] == reg["cloud%d" % i].shape[0] and nd["cells"] is not None
        assert len(nd["constraints"]) == (1 if i else 0)


def test_planar_pose_quaternion_convention():
    for th in (0.0, 0.3, -2.0, 3.0, np.pi, -3.1):
        p, q = api.pose3d_from_xyt((1.0, -2.0, th))
        assert q[3] >= 0 and abs(np.linalg.norm(q) - 1) < 1e-15 and q[0] == 0 and q[1] == 0
        ang = 2 * np.arctan2(q[2], q[3])
        assert abs(np.angle(np.exp(1j * (ang - th)))) < 1e-15
