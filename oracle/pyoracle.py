"""ctypes binding of the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never by the product package (tbv_slam_public_amd).  PARITY UNPINNED, see
oracle/cfear_oracle.h.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class OrcCell(C.Structure):
    _fields_ = [("mean", C.c_double * 2), ("normal", C.c_double * 2), ("cov", C.c_double * 4),
                ("scale", C.c_double), ("avg_intensity", C.c_double),
                ("lambda_min", C.c_double), ("lambda_max", C.c_double),
                ("nsamples", C.c_int32), ("pad", C.c_int32)]


CELL_DTYPE = np.dtype([("mean", "<f8", (2,)), ("normal", "<f8", (2,)), ("cov", "<f8", (4,)),
                       ("scale", "<f8"), ("avg_intensity", "<f8"), ("lambda_min", "<f8"),
                       ("lambda_max", "<f8"), ("nsamples", "<i4"), ("pad", "<i4")])
assert CELL_DTYPE.itemsize == C.sizeof(OrcCell) == 104

COST = {"P2P": 0, "P2L": 1, "P2D": 2}
LOSS = {"None": 0, "Huber": 1, "Cauchy": 2, "SoftLOne": 3, "Combined": 4, "Tukey": 5}


class OrcRegParams(C.Structure):
    _fields_ = [("cost", C.c_int32), ("loss", C.c_int32), ("loss_limit", C.c_double),
                ("weight_opt", C.c_int32), ("max_outer", C.c_int32), ("max_inner", C.c_int32),
                ("min_outer", C.c_int32), ("radius", C.c_double), ("cov_scale", C.c_double),
                ("regularization", C.c_double), ("score_tolerance", C.c_double),
                ("first_itr", C.c_int32), ("pad", C.c_int32)]


class OrcRegResult(C.Structure):
    _fields_ = [("pose", C.c_double * 3), ("score", C.c_double), ("final_cost", C.c_double),
                ("num_residuals", C.c_int32), ("outer_iters", C.c_int32),
                ("lm_iters", C.c_int32), ("status", C.c_int32)]


class OrcFuserParams(C.Structure):
    _fields_ = [("reg", OrcRegParams), ("res", C.c_float), ("submap_scan_size", C.c_int32),
                ("weight_intensity", C.c_int32), ("use_guess", C.c_int32),
                ("compensate", C.c_int32), ("radar_ccw", C.c_int32), ("use_keyframe", C.c_int32),
                ("min_keyframe_dist", C.c_double), ("min_keyframe_rot_deg", C.c_double),
                ("downsample_factor", C.c_double),
                ("estimate_cov_by_sampling", C.c_int32), ("cov_samples_per_axis", C.c_int32),
                ("cov_xy_range", C.c_double), ("cov_yaw_range", C.c_double), ("cov_scaler", C.c_double)]


def reg_params(cost="P2L", loss="Huber", loss_limit=0.1, weight_opt=0, max_outer=8, max_inner=20,
               min_outer=3, radius=2.0, cov_scale=1.0, regularization=0.01, first_itr=0):
    """n_scan_normal_reg defaults (n_scan_normal.h:35,72-75; registration.h:117-122)."""
    p = OrcRegParams()
    p.cost = COST[cost] if isinstance(cost, str) else int(cost)
    p.loss = LOSS[loss] if isinstance(loss, str) else int(loss)
    p.loss_limit = loss_limit
    p.weight_opt = weight_opt
    p.max_outer, p.max_inner, p.min_outer = max_outer, max_inner, min_outer
    p.radius, p.cov_scale, p.regularization = radius, cov_scale, regularization
    p.score_tolerance = 1e-5
    p.first_itr = first_itr
    return p


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "cfear_oracle.cpp")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B" if force else "-s"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        u8p, i32p, f32p, f64p = (C.POINTER(C.c_uint8), C.POINTER(C.c_int32), C.POINTER(C.c_float),
                                 C.POINTER(C.c_double))
        L.orc_kstrongest.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, i32p, u8p, i32p]
        L.orc_peaks.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_int, i32p, i32p, u8p]
        L.orc_kstrongest_cloud.argtypes = [C.c_int, C.c_int, i32p, u8p, i32p, u8p, C.c_float, C.c_float, f32p]
        L.orc_cacfar.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float,
                                 C.c_float, C.c_float, C.c_double, f32p, i32p, C.c_int]
        L.orc_kstrongest_legacy.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double,
                                            f32p, C.c_int]
        L.orc_compensate.argtypes = [f32p, C.c_int, f64p, C.c_int]
        L.orc_compensate.restype = None
        L.orc_surface_points.argtypes = [f32p, C.c_int, C.c_float, C.c_double, f64p, C.c_int,
                                         C.c_void_p, C.c_int, f32p, i32p]
        pp = C.POINTER(C.c_void_p)
        L.orc_register.argtypes = [pp, i32p, C.c_int, f64p, C.POINTER(OrcRegParams), C.POINTER(OrcRegResult)]
        L.orc_get_cost.argtypes = [pp, i32p, C.c_int, f64p, C.POINTER(OrcRegParams), f64p, f64p, i32p, f64p]
        L.orc_sc_descriptor.argtypes = [f32p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, C.c_double, C.c_double,
                                        C.c_double, f64p]
        L.orc_sc_descriptor.restype = None
        L.orc_sc_keys.argtypes = [f64p, C.c_int, C.c_int, f64p, f64p]
        L.orc_sc_keys.restype = None
        L.orc_sc_distance.argtypes = [f64p, f64p, C.c_int, C.c_int, C.c_double, i32p]
        L.orc_sc_distance.restype = C.c_double
        L.orc_coral_quality.argtypes = [f32p, C.c_int, f32p, C.c_int, f64p, f64p, f64p, C.c_double, C.c_int, f64p, f64p]
        L.orc_cov_by_sampling.argtypes = [pp, i32p, C.c_int, f64p, C.POINTER(OrcRegParams), C.c_double, C.c_int32,
                                          C.c_double, C.c_double, C.c_int32, C.c_double, f64p, f64p]
        L.orc_fuser_run_sequence.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                             C.c_float, C.c_float, f64p]
        L.orc_closest_idx.argtypes = [C.c_void_p, C.c_int, f64p, C.c_int, C.c_double, i32p]
        L.orc_closest_idx.restype = None
        L.orc_associate.argtypes = [pp, i32p, C.c_int, f64p, C.POINTER(OrcRegParams), C.c_int, i32p, f64p, C.c_int]
        L.orc_normal_eq.argtypes = [pp, i32p, C.c_int, f64p, C.POINTER(OrcRegParams), C.c_int, f64p,
                                    f64p, f64p, f64p, i32p]
        L.orc_lm_trace.argtypes = [pp, i32p, C.c_int, f64p, C.POINTER(OrcRegParams), C.c_int, C.c_int, f64p, f64p, C.c_int,
                                   f64p, i32p]
        L.orc_fuser_create.argtypes = [C.POINTER(OrcFuserParams)]
        L.orc_fuser_create.restype = C.c_void_p
        L.orc_fuser_destroy.argtypes = [C.c_void_p]
        L.orc_fuser_destroy.restype = None
        L.orc_fuser_stage_times.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
        L.orc_fuser_stage_times.restype = None
        L.orc_fuser_process.argtypes = [C.c_void_p, f32p, C.c_int, f64p, i32p]
        L.orc_fuser_last_cov.argtypes = [C.c_void_p, f64p, i32p]
        L.orc_fuser_last_cov.restype = None
        _LIB = L
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def kstrongest(img, k, z_min):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    rows, cols = img.shape
    sr = np.empty((rows, k), np.int32)
    si = np.empty((rows, k), np.uint8)
    sc = np.empty(rows, np.int32)
    rc = lib().orc_kstrongest(_p(img, C.c_uint8), rows, cols, cols, k, int(z_min), _p(sr, C.c_int32),
                              _p(si, C.c_uint8), _p(sc, C.c_int32))
    assert rc == 0
    return sr, si, sc


def peaks(img, k, sel_range, sel_count):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    rows, cols = img.shape
    pk = np.empty((rows, k), np.uint8)
    lib().orc_peaks(_p(img, C.c_uint8), rows, cols, cols, k, _p(sel_range, C.c_int32),
                    _p(sel_count, C.c_int32), _p(pk, C.c_uint8))
    return pk


def kstrongest_cloud(sel_range, sel_intensity, sel_count, range_res, min_distance, mask=None):
    rows, k = sel_range.shape
    out = np.empty((rows * k, 4), np.float32)
    m = _p(mask, C.c_uint8) if mask is not None else None
    n = lib().orc_kstrongest_cloud(rows, k, _p(sel_range, C.c_int32), _p(sel_intensity, C.c_uint8),
                                   _p(sel_count, C.c_int32), m, np.float32(range_res),
                                   np.float32(min_distance), _p(out, C.c_float))
    return out[:n].copy()


def kstrongest_legacy(img, k, z_min, range_res, min_distance):
    """k_strongest_filter (radar_filters.cpp:40-78) -> float32 [n, 4]."""
    img = np.ascontiguousarray(img, np.uint8)
    rows, cols = img.shape
    out = np.empty((rows * max(k, 1), 4), np.float32)
    n = lib().orc_kstrongest_legacy(_p(img, C.c_uint8), rows, cols, cols, int(k), float(z_min), float(range_res),
                                    float(min_distance), _p(out, C.c_float), out.shape[0])
    assert n >= 0
    return out[:n].copy()


def cacfar(img, window, guard, pfa, range_res, z_min, min_distance, max_distance=400.0):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    rows, cols = img.shape
    cap = rows * cols
    out = np.empty((cap, 4), np.float32)
    rc = np.empty((cap, 2), np.int32)
    n = lib().orc_cacfar(_p(img, C.c_uint8), rows, cols, cols, window, guard, np.float32(pfa),
                         np.float32(range_res), np.float32(z_min), np.float32(min_distance),
                         float(max_distance), _p(out, C.c_float), _p(rc, C.c_int32), cap)
    assert n >= 0
    return out[:n].copy(), rc[:n].copy()


def compensate(xyzi, mot, ccw):
    out = np.ascontiguousarray(xyzi, dtype=np.float32).copy()
    m = np.asarray(mot, np.float64).copy()
    lib().orc_compensate(_p(out, C.c_float), out.shape[0], _p(m, C.c_double), int(ccw))
    return out


def surface_points(xyzi, radius, downsample_factor=1.0, origin=(0.0, 0.0), weight_intensity=False,
                   return_centroids=False):
    xyzi = np.ascontiguousarray(xyzi, dtype=np.float32)
    n = xyzi.shape[0]
    cells = np.zeros(max(n, 1), CELL_DTYPE)
    cen = np.empty((max(n, 1), 2), np.float32)
    nv = np.zeros(1, np.int32)
    o = np.asarray(origin, np.float64).copy()
    nc = lib().orc_surface_points(_p(xyzi, C.c_float), n, np.float32(radius), float(downsample_factor),
                                  _p(o, C.c_double), int(weight_intensity), cells.ctypes.data, cells.shape[0],
                                  _p(cen, C.c_float), _p(nv, C.c_int32))
    assert nc >= 0
    if return_centroids:
        return cells[:nc].copy(), cen[:nv[0]].copy()
    return cells[:nc].copy()


def _scan_args(scans):
    scans = [np.ascontiguousarray(s, dtype=CELL_DTYPE) for s in scans]
    ptrs = (C.c_void_p * len(scans))(*[s.ctypes.data for s in scans])
    n = np.array([s.shape[0] for s in scans], np.int32)
    return scans, ptrs, n


def register(scans, poses, par):
    keep, ptrs, n = _scan_args(scans)
    p = np.ascontiguousarray(poses, dtype=np.float64).copy()
    res = OrcRegResult()
    ok = lib().orc_register(ptrs, _p(n, C.c_int32), len(keep), _p(p, C.c_double), C.byref(par), C.byref(res))
    return bool(ok), p, res


def get_cost(scans, poses, par):
    keep, ptrs, n = _scan_args(scans)
    p = np.ascontiguousarray(poses, dtype=np.float64).copy()
    cap = 2 * int(n.sum()) + 2
    r = np.empty(cap, np.float64)
    cost = C.c_double()
    score = C.c_double()
    nres = C.c_int32()
    ok = lib().orc_get_cost(ptrs, _p(n, C.c_int32), len(keep), _p(p, C.c_double), C.byref(par),
                            C.byref(cost), _p(r, C.c_double), C.byref(nres), C.byref(score))
    return bool(ok), cost.value, r[:nres.value].copy(), score.value


def cov_by_sampling(scans, poses, par, final_cost, num_residuals, xy_range=0.4, yaw_range=0.0043625,
                    samples_per_axis=3, covariance_scaler=4.0):
    """odometrykeyframefuser.cpp:261-380.  par.first_itr must be the leftover itr_ of the Register call.
    -> (success, cov 6x6, samples [n^3, 4])."""
    keep, ptrs, n = _scan_args(scans)
    p = np.ascontiguousarray(poses, dtype=np.float64).copy()
    cov = np.zeros((6, 6), np.float64)
    smp = np.zeros((samples_per_axis ** 3, 4), np.float64)
    ok = lib().orc_cov_by_sampling(ptrs, _p(n, C.c_int32), len(keep), _p(p, C.c_double), C.byref(par),
                                   float(final_cost), int(num_residuals), float(xy_range), float(yaw_range),
                                   int(samples_per_axis), float(covariance_scaler), _p(cov, C.c_double),
                                   _p(smp, C.c_double))
    return bool(ok), cov, smp


def sc_descriptor(xyzi, num_ring=40, num_sector=120, max_radius=80.0, desc_function="sum", desc_divider=1000.0,
                  no_point=0.0, shift_y=0.0):
    """RSCManager::MakeRadarCloudContext (RadarScancontext.cpp:59-131) -> desc [num_ring, num_sector]."""
    c = np.ascontiguousarray(xyzi, dtype=np.float32)
    desc = np.zeros((num_ring, num_sector), np.float64)
    lib().orc_sc_descriptor(_p(c, C.c_float), c.shape[0], num_ring, num_sector, float(max_radius),
                            0 if desc_function == "sum" else 1, float(desc_divider), float(no_point), float(shift_y),
                            _p(desc, C.c_double))
    return desc


def sc_keys(desc):
    """(ring key [R], sector key [S]) (Scancontext.cpp:239-268)."""
    d = np.ascontiguousarray(desc, dtype=np.float64)
    rk, sk = np.zeros(d.shape[0]), np.zeros(d.shape[1])
    lib().orc_sc_keys(_p(d, C.c_double), d.shape[0], d.shape[1], _p(rk, C.c_double), _p(sk, C.c_double))
    return rk, sk


def sc_distance(sc1, sc2, search_ratio=0.1):
    """distanceBtnScanContext (Scancontext.cpp:157-189) -> (distance, argmin column shift of sc2)."""
    a = np.ascontiguousarray(sc1, dtype=np.float64)
    b = np.ascontiguousarray(sc2, dtype=np.float64)
    sh = C.c_int32()
    d = lib().orc_sc_distance(_p(a, C.c_double), _p(b, C.c_double), a.shape[0], a.shape[1], float(search_ratio),
                              C.byref(sh))
    return d, sh.value


def coral_quality(ref_xyzi, src_xyzi, ref_pose, src_pose, offset=(0.0, 0.0, 0.0), radius=1.0,
                  weight_res_intensity=False):
    """CorAlRadarQuality (AlignmentQuality.cpp:8-230) as alignmentinterface.cpp:437-456 calls it.
    -> (valid, quality [joint, sep, overlap], per_point [n_src + n_ref, 3] (joint_res, sep_res, valid))."""
    r = np.ascontiguousarray(ref_xyzi, dtype=np.float32)
    s = np.ascontiguousarray(src_xyzi, dtype=np.float32)
    rp = np.ascontiguousarray(ref_pose, dtype=np.float64)
    sp = np.ascontiguousarray(src_pose, dtype=np.float64)
    of = np.ascontiguousarray(offset, dtype=np.float64)
    q = np.zeros(3, np.float64)
    pp = np.zeros((r.shape[0] + s.shape[0], 3), np.float64)
    ok = lib().orc_coral_quality(_p(r, C.c_float), r.shape[0], _p(s, C.c_float), s.shape[0], _p(rp, C.c_double),
                                 _p(sp, C.c_double), _p(of, C.c_double), float(radius), int(weight_res_intensity),
                                 _p(q, C.c_double), _p(pp, C.c_double))
    return bool(ok), q, pp


def xyt_compose(a, b):
    c, s = np.cos(a[2]), np.sin(a[2])
    return np.array([c * b[0] - s * b[1] + a[0], s * b[0] + c * b[1] + a[1], a[2] + b[2]], np.float64)


def xyt_inverse(a):
    c, s = np.cos(a[2]), np.sin(a[2])
    return np.array([-(c * a[0] + s * a[1]), s * a[0] - c * a[1], -a[2]], np.float64)


ALIGN_MODEL = (-8.42595, (-15.2287, 7.47573, -0.0680198, -1.74182, 0.0945444, 0.022217))   # trained_alignment_classifier.txt
LOOP_MODEL = (2.67958289, (-2.89398535, -9.40230684, 0.23891265))                         # loopclosure.cpp:224-232


def verify_by_odometry(rel_xyt, odom_sigma_error=0.03, verify_via_odometry=True):
    """loopclosure::VerifyByOdometry (tbv_slam/src/tbv_slam/loopclosure.cpp:776-808)."""
    if not verify_via_odometry:
        return 1.0
    T, trav = np.zeros(3), 0.0
    for d in np.asarray(rel_xyt, np.float64).reshape(-1, 3):
        trav += float(np.hypot(d[0], d[1]))
        T = xyt_compose(T, d)
    error = max(float(np.hypot(T[0], T[1])) - 5.0, 0.0)
    with np.errstate(invalid="ignore", divide="ignore"):
        rel = np.float64(error) / np.float64(trav)
        return float(1.0 - np.exp(-rel * rel / (2.0 * odom_sigma_error * odom_sigma_error)))


def verify_loop_candidate(from_cells, from_peaks, from_pose, to_cells, to_peaks, t_be_guess, sc_sim, odom_bounds,
                          align_model=ALIGN_MODEL, loop_model=LOOP_MODEL, use_covariance_sampling=False,
                          verification_disabled=False):
    """One candidate through RegisterLoopCandidate + VerifyLoopCandidate (loopclosure.cpp:320-384; Register :35-97;
    PredAlignment alignmentinterface.cpp:349-367; VerificationModel loopclosure.cpp:220-238).  -> dict."""
    from_pose = np.asarray(from_pose, np.float64)
    Tto = xyt_compose(from_pose, np.asarray(t_be_guess, np.float64))
    par = reg_params("P2L", "Huber", 0.1, 0, 4, 10)                      # :56-57
    ok, p, res = register([to_cells, from_cells], np.stack([Tto, from_pose]), par)
    out = {"reg_ok": ok, "cov_sampled": False}
    if ok:
        Trev = p[1]
        inv = xyt_inverse(Trev)
        out["t_be"] = xyt_compose(inv, Tto)                              # :91
        cov = np.diag([0.01, 0.01, 0.0, 0.0, 0.0, 1e-4])                 # n_scan_normal.cpp:171
        if use_covariance_sampling:                                      # :62-71, ranges :108-112
            par.first_itr = res.outer_iters
            cok, csamp, _ = cov_by_sampling([to_cells, from_cells], np.stack([Tto, Trev]), par, res.final_cost,
                                            res.num_residuals, 0.4, 0.0044, 3, 4.0)
            if cok:
                cov, out["cov_sampled"] = csamp, True
        c, s = np.cos(inv[2]), np.sin(inv[2])
        R = np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]])
        cov = cov.copy()
        cov[:3, :3] = R @ cov[:3, :3] @ R.T                              # :93
        out["cov"] = cov
    else:
        out["t_be"] = np.zeros(3)                                        # :351-352 initial values survive
        out["cov"] = np.eye(6)
    Tto2 = xyt_compose(from_pose, out["t_be"])                           # :367-368
    _, q, _ = coral_quality(from_peaks, to_peaks, from_pose, Tto2)       # ref = current = from
    out["coral"] = q
    qpar = reg_params("P2L", "Huber", 0.3)
    gok, cost, r, _ = get_cost([from_cells, to_cells], np.stack([from_pose, Tto2]), qpar)
    out["cfear"] = np.array([cost, float(r.shape[0]), (len(from_cells) + len(to_cells)) / 2.0]) if gok else np.zeros(3)
    out["alignment_quality"] = float(align_model[0] + np.dot(align_model[1], np.concatenate([out["coral"], out["cfear"]])))
    z = loop_model[0] + np.dot(loop_model[1], [odom_bounds, sc_sim, out["alignment_quality"]])
    out["probability"] = 0.0 if verification_disabled else float(1.0 / (1.0 + np.exp(-z)))
    return out


def apply_constraints(prob, group, model_threshold=0.8, all_candidates=True):
    """loopclosure::ApplyConstratins (loopclosure.cpp:261-274) per query group -> accepted flags."""
    prob, group = np.asarray(prob, np.float64), np.asarray(group)
    acc = np.zeros(prob.shape[0], bool)
    for g in np.unique(group):
        idx = np.nonzero(group == g)[0]
        order = idx[np.argsort(-prob[idx], kind="stable")]
        use = order if all_candidates else order[:1]
        acc[use] = prob[use] > model_threshold
    return acc


def closest_idx(cells, queries_xy, d):
    """MapPointNormal::GetClosestIdx (pointnormal.cpp:238-254) for a batch of points -> int32 [n] (-1 = none)."""
    c = np.ascontiguousarray(cells, dtype=CELL_DTYPE)
    q = np.ascontiguousarray(queries_xy, dtype=np.float64).reshape(-1, 2)
    out = np.zeros(q.shape[0], np.int32)
    lib().orc_closest_idx(c.ctypes.data_as(C.c_void_p), c.shape[0], _p(q, C.c_double), q.shape[0], C.c_double(d),
                          _p(out, C.c_int32))
    return out


def associate(scans, poses, par, itr):
    keep, ptrs, n = _scan_args(scans)
    p = np.ascontiguousarray(poses, dtype=np.float64).copy()
    cap = int(n[-1]) * max(len(keep) - 1, 1) + 1
    pairs = np.empty((cap, 3), np.int32)
    w = np.empty(cap, np.float64)
    m = lib().orc_associate(ptrs, _p(n, C.c_int32), len(keep), _p(p, C.c_double), C.byref(par), int(itr),
                            _p(pairs, C.c_int32), _p(w, C.c_double), cap)
    assert m >= 0
    return pairs[:m].copy(), w[:m].copy()


def lm_trace(scans, poses, par, itr, max_iter):
    """One ceres::Solve of the oracle on the associations built at `poses` -> (x, trace [n, 4] = cost,
    relative_decrease, successful, radius per summary iteration, final_cost, usable)."""
    keep, ptrs, ns = _scan_args(scans)
    poses = np.ascontiguousarray(poses, np.float64)
    x = np.zeros(3)
    trace = np.zeros((max_iter + 8, 4))
    fc = C.c_double()
    us = C.c_int32()
    n = lib().orc_lm_trace(ptrs, _p(ns, C.c_int32), len(keep), _p(poses, C.c_double), C.byref(par), int(itr), int(max_iter),
                           _p(x, C.c_double), _p(trace, C.c_double), trace.shape[0], C.byref(fc), C.byref(us))
    return x, trace[:n].copy(), fc.value, bool(us.value)


def normal_eq(scans, poses, par, itr, x):
    keep, ptrs, n = _scan_args(scans)
    p = np.ascontiguousarray(poses, dtype=np.float64).copy()
    xx = np.asarray(x, np.float64).copy()
    H = np.empty(9, np.float64)
    g = np.empty(3, np.float64)
    cost = C.c_double()
    nres = C.c_int32()
    lib().orc_normal_eq(ptrs, _p(n, C.c_int32), len(keep), _p(p, C.c_double), C.byref(par), int(itr),
                        _p(xx, C.c_double), _p(H, C.c_double), _p(g, C.c_double), C.byref(cost), C.byref(nres))
    return H.reshape(3, 3), g, cost.value, nres.value


class Fuser:
    """OdometryKeyframeFuser restatement (odometrykeyframefuser.cpp:143-259)."""

    def __init__(self, reg, res=3.0, submap_scan_size=4, weight_intensity=True, use_guess=True,
                 compensate=True, radar_ccw=False, use_keyframe=True, min_keyframe_dist=1.5,
                 min_keyframe_rot_deg=5.0, downsample_factor=1.0, estimate_cov_by_sampling=False,
                 cov_xy_range=0.4, cov_yaw_range=0.0043625, cov_samples_per_axis=3, cov_scaler=4.0):
        p = OrcFuserParams()
        p.reg = reg
        p.res = res
        p.submap_scan_size = submap_scan_size
        p.weight_intensity, p.use_guess, p.compensate = int(weight_intensity), int(use_guess), int(compensate)
        p.radar_ccw, p.use_keyframe = int(radar_ccw), int(use_keyframe)
        p.min_keyframe_dist, p.min_keyframe_rot_deg = min_keyframe_dist, min_keyframe_rot_deg
        p.downsample_factor = downsample_factor
        p.estimate_cov_by_sampling, p.cov_samples_per_axis = int(estimate_cov_by_sampling), int(cov_samples_per_axis)
        p.cov_xy_range, p.cov_yaw_range, p.cov_scaler = cov_xy_range, cov_yaw_range, cov_scaler
        self._h = lib().orc_fuser_create(C.byref(p))

    def process(self, xyzi):
        x = np.ascontiguousarray(xyzi, dtype=np.float32).copy()
        pose = np.empty(3, np.float64)
        info = np.zeros(4, np.int32)
        rc = lib().orc_fuser_process(self._h, _p(x, C.c_float), x.shape[0], _p(pose, C.c_double),
                                     _p(info, C.c_int32))
        assert rc == 0
        return pose, info

    def run_sequence(self, imgs, k, z_min, range_res, min_distance):
        """k-strongest filter + fuser over uint8 images [n, rows, cols] in ONE native call (no Python per frame; the
        GIL is released for its whole duration) -> poses [n, 3]."""
        im = np.ascontiguousarray(imgs, dtype=np.uint8)
        poses = np.zeros((im.shape[0], 3), np.float64)
        rc = lib().orc_fuser_run_sequence(self._h, im.ctypes.data_as(C.c_void_p), im.shape[0], im.shape[1], im.shape[2],
                                          int(k), int(z_min), C.c_float(range_res), C.c_float(min_distance),
                                          _p(poses, C.c_double))
        assert rc == 0
        return poses

    def stage_times(self):
        """{stage: seconds} accumulated so far, keys as the reference's `timing` names, + frames covered."""
        sec = np.zeros(4, np.float64)
        n = C.c_int64()
        lib().orc_fuser_stage_times(self._h, _p(sec, C.c_double), C.byref(n))
        return dict(Filtering=float(sec[0]), compensate=float(sec[1]), build_normals=float(sec[2]), register=float(sec[3])), int(n.value)

    def last_cov(self):
        """cov_current after the last frame -> (cov 6x6, sampled flag)."""
        cov = np.zeros((6, 6), np.float64)
        flag = C.c_int32()
        lib().orc_fuser_last_cov(self._h, _p(cov, C.c_double), C.byref(flag))
        return cov, bool(flag.value)

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                lib().orc_fuser_destroy(self._h)
                self._h = None
        except Exception:                        # interpreter shutdown: the module globals may already be gone
            pass
