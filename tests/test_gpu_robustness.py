"""GPU: non-finite and absurd inputs must come back as status codes or flagged results -- never a hang, a crash or an
exception other than CfearError (the reference exits or asserts on several of these, pointnormal.cpp:72-75)."""
import numpy as np
import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(120)]


@pytest.fixture(scope="module")
def data():
    from oracle import pyoracle as O
    from tbv_slam_public_amd import synth
    imgs, _, _ = synth.scene_v1(21, 1)
    sr, si, sc = O.kstrongest(imgs[0], 40, 60)
    cloud = O.kstrongest_cloud(sr, si, sc, 0.0438, 2.5)
    return cloud, O.surface_points(cloud, 3.0, 1.0, (0, 0), True)


def test_surface_points_with_non_finite_or_huge_points(data):
    from tbv_slam_public_amd import api, _lib as L
    cloud, cells = data
    bad = cloud.copy()
    bad[10, 0] = np.nan                                   # one NaN point: dropped with its voxel, the rest survives
    assert abs(api.MapPointNormal(bad, 3.0, (0, 0), True).GetSize() - len(cells)) < 20
    for poison in (np.inf, 1e30):
        bad = cloud.copy()
        bad[10, 0] = poison
        with pytest.raises(L.CfearError) as e:
            api.MapPointNormal(bad, 3.0, (0, 0), True)
        assert e.value.status == L.ERR_CAPACITY            # the voxel grid would not fit: reported, not attempted


def test_registration_with_non_finite_poses_and_cells(data):
    from tbv_slam_public_amd import api
    _, cells = data
    m0 = api.MapPointNormal(cells=cells)
    c2 = cells.copy()
    c2["mean"][5] = np.nan
    m1 = api.MapPointNormal(cells=c2)
    reg = api.n_scan_normal_reg("P2L")
    for pose in (np.nan, np.inf):
        ok, _, _ = reg.Register([m0, m0], np.array([[0, 0, 0], [pose, 0, 0]]))
        assert not ok                                      # no correspondences -> too few residuals
        assert not reg.GetCost([m0, m0], np.array([[0, 0, 0], [pose, 0, 0]]))[0]
    for scans in ([m1, m0], [m0, m1]):                     # a NaN cell never matches anything; the rest registers
        ok, T, _ = reg.Register(scans, np.array([[0, 0, 0], [0.1, 0, 0]]))
        assert ok and np.isfinite(T).all() and np.abs(T[1]).max() < 0.05
    assert m0.GetClosestIdx(np.array([[np.nan, 0.0], [1e300, 0.0]]), 2.0).tolist() == [-1, -1]


def test_alignment_quality_and_descriptors_with_non_finite_input(data):
    from tbv_slam_public_amd import api, _lib as L
    cloud, cells = data
    pk = cloud[:1500]
    q = api.CorAlRadarQuality(pk, (0, 0, 0), pk, (np.nan, 0, 0))
    assert q.GetQualityMeasure() == [0.0, 0.0, 0.0] and not q.valid_
    bp = pk.copy()
    bp[3, 1] = np.nan
    assert np.isfinite(api.CorAlRadarQuality(pk, (0, 0, 0), bp, (0.1, 0, 0)).GetQualityMeasure()).all()
    huge = pk.copy()
    huge[:, 0] *= 1e20
    with pytest.raises(L.CfearError):
        api.CorAlRadarQuality(pk, (0, 0, 0), huge, (0.1, 0, 0))
    assert np.isfinite(api.sc_descriptors([bp])[0]).all()
    m0 = api.MapPointNormal(cells=cells)
    r = api.verify_loop_candidates([dict(from_scan=m0, to_scan=m0, from_peaks=pk, to_peaks=pk, from_pose=(0, 0, 0),
                                         t_be_guess=(np.nan, 0, 0), sc_sim=0.1, odom_bounds=0.0, group=0)])
    # the registration fails, t_be stays Identity and is still scored (loopclosure.cpp:351-352) -- here the two nodes
    # are the same scan, so Identity happens to be a perfect alignment
    assert r["reg_ok"][0] == 0 and (r["t_be"][0] == 0).all() and np.isfinite(r["probability"][0])


def test_two_contexts_on_two_host_threads(data):
    """The reference runs odometry and loop closure on two threads (tbv_slam.cpp); the header promises that calls on
    different contexts are independent.  Two threads, each with its own context and stream, hammer registration and
    CorAl batches concurrently; every result must equal the single-threaded one."""
    import threading
    from tbv_slam_public_amd import api
    cloud, cells = data
    pk = cloud[:1800]
    rng = np.random.default_rng(0)
    guesses = rng.normal(0, [0.4, 0.4, 0.02], (64, 3))

    def work(ctx):
        scan = api.MapPointNormal(cells=cells, ctx=ctx)
        reg = api.n_scan_normal_reg("P2L", ctx=ctx)
        reg.SetParameters(4, 10)
        out = []
        for rep in range(6):
            r = reg.RegisterBatch([([scan, scan], np.array([[0.0, 0.0, 0.0], g])) for g in guesses])
            q, _ = api.coral_quality_batch([(pk, (0, 0, 0), pk, g, (0, 0, 0)) for g in guesses[:16]], ctx=ctx)
            out.append((r["pose"].copy(), r["outer_iters"].copy(), q["joint"].copy()))
        return out
    ref = work(api.Context(0))
    results, errors = {}, []

    def run(name):
        try:
            results[name] = work(api.Context(0))
        except Exception as e:                            # noqa: BLE001
            errors.append(e)
    threads = [threading.Thread(target=run, args=(n,)) for n in ("a", "b")]
    for t in threads:
        t.start()
    for t in threads:
        t.join(100)
    assert not errors and set(results) == {"a", "b"}
    for name in ("a", "b"):
        for (p, it, j), (p0, it0, j0) in zip(results[name], ref):
            np.testing.assert_array_equal(p, p0)
            np.testing.assert_array_equal(it, it0)
            np.testing.assert_array_equal(j, j0)


def test_verification_with_empty_peak_clouds(data):
    """Nodes built with the CA-CFAR filter have no peaks cloud (radar_driver.cpp:52-56).  The reference would assert in
    CorAl (AlignmentQuality.cpp:118); here the CorAl features of such a candidate are {0, 0, 0}, the job status says
    why, and the rest of the batch is unaffected."""
    from tbv_slam_public_amd import api
    cloud, cells = data
    pk = cloud[:1500]
    m0 = api.MapPointNormal(cells=cells)
    empty = np.zeros((0, 4), np.float32)
    base = dict(from_scan=m0, to_scan=m0, from_pose=(0, 0, 0), t_be_guess=(0.1, 0.0, 0.0), sc_sim=0.1, odom_bounds=0.0)
    r = api.verify_loop_candidates([dict(base, from_peaks=pk, to_peaks=pk, group=0),
                                    dict(base, from_peaks=empty, to_peaks=pk, group=1),
                                    dict(base, from_peaks=pk, to_peaks=empty, group=2)])
    assert (r["reg_ok"] == 1).all()
    assert r["accepted"][0] == 1 and r["coral"][0][2] > 0.5
    assert (r["coral"][1:] == 0).all() and (r["cfear"][1:, 0] > 0).all()
    q, _ = api.coral_quality_batch([(pk, (0, 0, 0), empty, (0, 0, 0), (0, 0, 0))])
    assert q["status"][0] != 0 and q["valid"][0] == 0


def test_context_options_and_stream_handle(data):
    """cfear_ctx_set_option / _get_option / _get_stream: the test hooks that replaced the CFEAR_* environment switches.  Unknown
    options and values outside their range are argument errors and leave the option unchanged; a context created on a torch
    stream reports that stream's handle, and dist.py's ordering test (Context.shares_torch_stream) is true exactly there."""
    import torch
    from tbv_slam_public_amd import api
    from tbv_slam_public_amd import _lib as L
    ctx = api.Context(0)
    assert ctx.get_option(L.OPT_FUSED_DECODE) == 1 and ctx.get_option(L.OPT_MATCHER_LDS_KB) == 0
    ctx.set_option(L.OPT_MATCHER_LDS_KB, 52)
    assert ctx.get_option(L.OPT_MATCHER_LDS_KB) == 52
    for opt, val in ((L.OPT_MATCHER_LDS_KB, 4), (L.OPT_MATCHER_LDS_KB, 161), (L.OPT_MATCHER_WAVES, 3), (L.OPT_FUSED_DECODE, 2), (L.OPT_COUNT, 0), (-1, 0)):
        with pytest.raises(L.CfearError) as e:
            ctx.set_option(opt, val)
        assert e.value.status == L.ERR_INVALID_ARGUMENT
    assert ctx.get_option(L.OPT_MATCHER_LDS_KB) == 52 and ctx.get_option(L.OPT_MATCHER_WAVES) == 0
    assert ctx.stream_handle() != 0 and not ctx.shares_torch_stream()      # a private stream: nothing of torch's is ordered with it
    ctx.close()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        shared = api.Context(0, stream=side.cuda_stream)
        assert shared.stream_handle() == side.cuda_stream and shared.shares_torch_stream()
    assert not shared.shares_torch_stream()                                # torch's current stream is the default one again
    shared.close()
