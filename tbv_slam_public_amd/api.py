"""Host-side mirror of the reference's `cfear_radarodometry` class API over the C-ABI.

Names, argument meaning and error behaviour follow namespace CFEAR_Radarodometry:
  radarDriver            radar_driver.h:32-118        (filters: radar_filters.h / cfar.h)
  MapPointNormal         pointnormal.h:110-243
  n_scan_normal_reg      n_scan_normal.h:27-85 (+ Registration, registration.h:68-133)
  OdometryKeyframeFuser  odometrykeyframefuser.h:72-249  (batched over independent streams here)
with ROS / PCL / Eigen types replaced by NumPy arrays (host) or torch CUDA tensors (device):
  pcl::PointCloud<PointXYZI>  ->  float32 [n,4] (x, y, z, intensity)
  Eigen::Affine3d (planar)    ->  float64 (x, y, theta)  [Affine3dToVectorXYeZ, utils.cpp:115-122]
Everything computes in libcfear_hip.so on the GPU; there is no CPU path in this package.
"""
import ctypes as C
import os

import numpy as np

from . import _lib as L


def _is_torch(x):
    return type(x).__module__.startswith("torch")


def _ptr(x):
    """(address, keepalive) of a NumPy array (host) or torch tensor (device)."""
    if x is None:
        return None, None
    if _is_torch(x):
        assert x.is_contiguous()
        return x.data_ptr(), x
    assert x.flags["C_CONTIGUOUS"]
    return x.ctypes.data, x


class Context:
    """cfear_ctx: one per host thread / HIP stream.  `stream` = a hipStream_t handle to enqueue on; None or 0 (which is
    also what torch reports for its default stream) gives the context a PRIVATE non-blocking stream: work torch has queued
    that produces this context's inputs must then be synchronised by the caller (default_context() passes hipStreamLegacy
    for torch's default stream instead, which orders the two)."""

    default_options = {}      # {L.OPT_*: value} applied to every new context (measurement scripts: bench.py --ctx-option)

    def __init__(self, device=0, stream=None):
        self._lib = L.lib()
        h = C.c_void_p()
        rc = self._lib.cfear_ctx_create(int(device), C.c_void_p(stream) if stream else None, C.byref(h))
        if rc != L.OK:
            raise L.CfearError(rc, self._lib.cfear_status_string(rc).decode() +
                               " (libcfear_hip needs an MI355X; there is no CPU fallback)")
        self.h = h
        self.device = device
        for opt, val in Context.default_options.items():
            self.set_option(opt, val)

    def check(self, rc, allowed=()):
        if rc != L.OK and rc not in allowed:
            raise L.CfearError(rc, self._lib.cfear_last_error(self.h).decode())
        return rc

    def synchronize(self):
        self.check(self._lib.cfear_ctx_synchronize(self.h))

    def stream_handle(self):
        """The hipStream_t this context enqueues on, as an int (the caller's stream, or the private one)."""
        s = C.c_void_p()
        self.check(self._lib.cfear_ctx_get_stream(self.h, C.byref(s)))
        return int(s.value or 0)

    def shares_torch_stream(self):
        """True when work enqueued on torch's CURRENT stream is ordered with this context's kernels without a host
        synchronisation: the two are the same HIP stream (torch's default stream reports 0 -- a context given 0 made a
        private stream, so 0 never counts as shared)."""
        import torch
        ts = int(torch.cuda.current_stream().cuda_stream)
        return ts != 0 and ts == self.stream_handle()

    def set_option(self, option, value):
        """cfear_ctx_set_option: the test / measurement hooks of include/cfear_hip.h (L.OPT_*)."""
        self.check(self._lib.cfear_ctx_set_option(self.h, int(option), int(value)))

    def get_option(self, option):
        v = C.c_int64()
        self.check(self._lib.cfear_ctx_get_option(self.h, int(option), C.byref(v)))
        return int(v.value)

    def profile_enable(self, on=True):
        """True / 1: every kernel family; 2: only the polar filter's row kernels; False / 0: off."""
        self.check(self._lib.cfear_ctx_profile_enable(self.h, int(on)))

    def profile_read(self, reset=True):
        """{kernel family: (total_ms, launches)} measured with hipEvents on the context's stream."""
        cap = 32
        names = (C.c_char_p * cap)()
        ms = (C.c_double * cap)()
        cnt = (C.c_int64 * cap)()
        n = self._lib.cfear_ctx_profile_read(self.h, names, ms, cnt, cap, int(reset))
        return {names[i].decode(): (ms[i], cnt[i]) for i in range(min(n, cap))}

    def close(self):
        if getattr(self, "h", None):
            self._lib.cfear_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_DEFAULT_CTX = None


def default_context():
    """The process-wide context on device 0.  It enqueues on torch's current stream, so kernels launched through
    the mirror are ordered with torch work on the tensors they read and write (device-pointer calls are
    asynchronous, include/cfear_hip.h)."""
    global _DEFAULT_CTX
    if _DEFAULT_CTX is None:
        stream = None
        try:
            import torch
            if torch.cuda.is_available():
                # torch's default stream is the null stream (handle 0): address it as hipStreamLegacy (1)
                stream = torch.cuda.current_stream(0).cuda_stream or 1
        except ImportError:
            pass
        _DEFAULT_CTX = Context(0, stream)
    return _DEFAULT_CTX


# ------------------------------------------------------------------------------------------------
# radarDriver
# ------------------------------------------------------------------------------------------------
class radarDriverParameters:
    """radarDriver::Parameters (radar_driver.h:35-84); float members are float32 like the reference."""

    def __init__(self, z_min=60.0, range_res=0.0438, azimuths=400, k_strongest=12, nb_guard_cells=20,
                 window_size=10, false_alarm_rate=0.01, min_distance=2.5, max_distance=200.0,
                 dataset="oxford", filter_type="kstrong"):
        self.z_min, self.range_res, self.azimuths, self.k_strongest = z_min, range_res, azimuths, k_strongest
        self.nb_guard_cells, self.window_size, self.false_alarm_rate = nb_guard_cells, window_size, false_alarm_rate
        self.min_distance, self.max_distance = min_distance, max_distance
        self.dataset, self.filter_type = dataset, filter_type


def _desc(img):
    if img.ndim == 2:
        rows, cols = img.shape
        batch = 1
    else:
        batch, rows, cols = img.shape
    d = L.PolarDesc()
    d.rows, d.cols, d.stride, d.batch = rows, cols, cols, batch
    d.batch_stride = rows * cols
    return d, batch, rows, cols


def polar_rotate_ccw(img, ctx=None):
    """cv::rotate(ROTATE_90_COUNTERCLOCKWISE) of radarDriver::Callback (radar_driver.cpp:74-90) for a uint8 image
    [bins, azimuths] or a batch [b, bins, azimuths] -> [azimuths, bins] / [b, azimuths, bins] (NumPy -> NumPy,
    torch CUDA -> torch CUDA); out[i, j] = img[j, azimuths - 1 - i]."""
    ctx = ctx or default_context()
    d, batch, rows, cols = _desc(img)
    shape = (cols, rows) if img.ndim == 2 else (batch, cols, rows)
    if _is_torch(img):
        import torch
        img = img.contiguous()
        out = torch.empty(shape, dtype=torch.uint8, device=img.device)
    else:
        img = np.ascontiguousarray(img, dtype=np.uint8)
        out = np.empty(shape, np.uint8)
    ctx.check(ctx._lib.cfear_polar_rotate_ccw(ctx.h, _ptr(img)[0], C.byref(d), _ptr(out)[0], rows, rows * cols))
    return out


def filter_kstrongest(img, k, z_min, range_res, min_distance, want_peaks=False, ctx=None):
    """StructuredKStrongest (radar_filters.cpp:198-337) for a uint8 image [rows, cols] or a batch
    [b, rows, cols] (NumPy -> NumPy results, torch CUDA tensor -> torch CUDA results).
    Returns dict(sel_range, sel_intensity, sel_count, is_peak, xyzi, n_points, xyzi_peaks, n_peaks)."""
    ctx = ctx or default_context()
    d, batch, rows, cols = _desc(img)
    par = L.KStrongParams(int(k), float(z_min), float(range_res), float(min_distance), int(bool(want_peaks)))
    if _is_torch(img):
        import torch
        dev = img.device
        mk = lambda shape, dt: torch.empty(shape, dtype=dt, device=dev)
        res = dict(sel_range=mk((batch, rows, k), torch.int32), sel_intensity=mk((batch, rows, k), torch.uint8),
                   sel_count=mk((batch, rows), torch.int32), xyzi=mk((batch, rows * k, 4), torch.float32),
                   n_points=mk((batch,), torch.int32))
        if want_peaks:
            res.update(is_peak=mk((batch, rows, k), torch.uint8), xyzi_peaks=mk((batch, rows * k, 4), torch.float32),
                       n_peaks=mk((batch,), torch.int32))
    else:
        img = np.ascontiguousarray(img, dtype=np.uint8)
        res = dict(sel_range=np.empty((batch, rows, k), np.int32), sel_intensity=np.empty((batch, rows, k), np.uint8),
                   sel_count=np.empty((batch, rows), np.int32), xyzi=np.empty((batch, rows * k, 4), np.float32),
                   n_points=np.empty((batch,), np.int32))
        if want_peaks:
            res.update(is_peak=np.empty((batch, rows, k), np.uint8),
                       xyzi_peaks=np.empty((batch, rows * k, 4), np.float32), n_peaks=np.empty((batch,), np.int32))
    out = L.KStrongOut()
    for name in ("sel_range", "sel_intensity", "sel_count", "is_peak", "xyzi", "n_points", "xyzi_peaks", "n_peaks"):
        setattr(out, name, _ptr(res.get(name))[0])
    p, _keep = _ptr(img)
    ctx.check(ctx._lib.cfear_filter_kstrongest(ctx.h, p, C.byref(d), C.byref(par), C.byref(out)))
    return res


def filter_kstrongest_rowkeys(img, k, z_min, range_res, min_distance, bins_major=False, two_pass=False, tile_sweep=False, route=0, ctx=None):
    """cfear_filter_kstrongest_rowkeys: the batched odometry's filter stage on its own, for a torch CUDA uint8 image
    [rows, cols] or batch [b, rows, cols]; bins_major: the images are [range bins][azimuths] and are decoded (rotated
    counter-clockwise, radar_driver.cpp:74-90) by the sweep itself (two_pass: by the rotation kernel first; tile_sweep:
    every 16-column tile through the LDS transposition instead of the candidate lists; route 1 / 2: the lists in global
    memory / one workgroup per image whatever the batch size).
    Returns (row_keys uint32-as-int32 [b, azimuths, k], row_counts int32 [b, azimuths, 2]) as CUDA tensors."""
    import torch
    ctx = ctx or default_context()
    d, batch, rows, cols = _desc(img)
    assert img.dtype == torch.uint8 and img.stride(-1) == 1      # a view with a row pitch / batch stride is fine
    d.stride = img.stride(-2)
    d.batch_stride = img.stride(0) if img.ndim == 3 else rows * d.stride
    az = cols if bins_major else rows
    par = L.KStrongParams(int(k), float(z_min), float(range_res), float(min_distance), 0)
    keys = torch.zeros((batch, az, k), dtype=torch.int32, device=img.device)
    cnt = torch.zeros((batch, az, 2), dtype=torch.int32, device=img.device)
    p = img.data_ptr()
    flags = (1 if bins_major else 0) | (2 if two_pass else 0) | (4 if tile_sweep else 0) | (int(route) << 4)
    ctx.check(ctx._lib.cfear_filter_kstrongest_rowkeys(ctx.h, p, C.byref(d), C.byref(par), flags, _ptr(keys)[0], _ptr(cnt)[0]))
    return keys, cnt


def filter_cacfar(img, window_size, nb_guard_cells, false_alarm_rate, range_res, z_min, min_distance,
                  max_distance=400.0, cap_points=None, want_mask=False, ctx=None):
    """AzimuthCACFAR::getFilteredPointCloud (cfar.cpp:35-71).  Returns dict(xyzi, n_points[, det_mask])."""
    ctx = ctx or default_context()
    d, batch, rows, cols = _desc(img)
    cap = int(cap_points or rows * cols)
    par = L.CacfarParams(int(window_size), int(nb_guard_cells), float(false_alarm_rate), float(range_res),
                         float(z_min), float(min_distance), float(max_distance))
    if _is_torch(img):
        import torch
        xyzi = torch.empty((batch, cap, 4), dtype=torch.float32, device=img.device)
        npts = torch.empty((batch,), dtype=torch.int32, device=img.device)
        mask = torch.empty((batch, rows, cols), dtype=torch.uint8, device=img.device) if want_mask else None
    else:
        img = np.ascontiguousarray(img, dtype=np.uint8)
        xyzi = np.empty((batch, cap, 4), np.float32)
        npts = np.empty((batch,), np.int32)
        mask = np.empty((batch, rows, cols), np.uint8) if want_mask else None
    ctx.check(ctx._lib.cfear_filter_cacfar(ctx.h, _ptr(img)[0], C.byref(d), C.byref(par), _ptr(xyzi)[0],
                                           _ptr(npts)[0], cap, _ptr(mask)[0]))
    res = dict(xyzi=xyzi, n_points=npts)
    if want_mask:
        res["det_mask"] = mask
    return res


def k_strongest_filter(img, k_strongest, z_min, range_res, min_distance, ctx=None):
    """The legacy k_strongest_filter / InsertStrongestK (radar_filters.cpp:25-78; CorAl's kstrongRadar).  img: uint8
    [rows, cols] or [batch, rows, cols] (NumPy or torch CUDA).  Returns dict(xyzi [batch, rows * k, 4], n_points)."""
    ctx = ctx or default_context()
    d, batch, rows, cols = _desc(img)
    cap = rows * int(k_strongest)
    if _is_torch(img):
        import torch
        xyzi = torch.empty((batch, cap, 4), dtype=torch.float32, device=img.device)
        npts = torch.empty((batch,), dtype=torch.int32, device=img.device)
    else:
        img = np.ascontiguousarray(img, dtype=np.uint8)
        xyzi = np.empty((batch, cap, 4), np.float32)
        npts = np.empty((batch,), np.int32)
    ctx.check(ctx._lib.cfear_filter_kstrongest_legacy(ctx.h, _ptr(img)[0], C.byref(d), int(k_strongest), float(z_min), float(range_res),
                                                      float(min_distance), _ptr(xyzi)[0], _ptr(npts)[0], cap))
    return dict(xyzi=xyzi, n_points=npts)


class radarDriver:
    """radarDriver (radar_driver.cpp): CallbackOffline(image) -> (cloud, cloud_peaks)."""

    def __init__(self, pars=None, ctx=None):
        self.par = pars or radarDriverParameters()
        self.ctx = ctx or default_context()
        self.cv_polar_image = None

    def CallbackOffline(self, radar_image_polar):
        img = radar_image_polar
        if self.par.dataset != "oxford":
            # Callback (radar_driver.cpp:74-90): MONO8 + rotate 90 deg CCW so rows = azimuth
            img = polar_rotate_ccw(img, self.ctx)
        self.cv_polar_image = img
        p = self.par
        if p.filter_type == "CA-CFAR":                                      # radar_driver.cpp:52-56
            r = filter_cacfar(img, p.window_size, p.nb_guard_cells, p.false_alarm_rate, p.range_res, p.z_min,
                              p.min_distance, 400.0, ctx=self.ctx)
            n = int(r["n_points"][0])
            return r["xyzi"][0, :n], r["xyzi"][0, :0]
        r = filter_kstrongest(img, p.k_strongest, p.z_min, p.range_res, p.min_distance, True, ctx=self.ctx)
        n, m = int(r["n_points"][0]), int(r["n_peaks"][0])
        return r["xyzi"][0, :n], r["xyzi_peaks"][0, :m]


# ------------------------------------------------------------------------------------------------
# Compensate / MapPointNormal
# ------------------------------------------------------------------------------------------------
def Compensate(cloud, mot, ccw, ctx=None):
    """Compensate(cloud, mot, ccw) (utils.cpp:96-107): in place on `cloud` (float32 [n,4])."""
    ctx = ctx or default_context()
    m = (C.c_double * 3)(*[float(v) for v in mot])
    ctx.check(ctx._lib.cfear_compensate(ctx.h, _ptr(cloud)[0], int(cloud.shape[0]), m, int(bool(ccw))))
    return cloud


class MapPointNormal:
    """MapPointNormal (pointnormal.h:110): device-resident oriented surface points of one scan."""

    downsample_factor = 1.0      # static member, pointnormal.cpp:5

    def __init__(self, cld=None, radius=3.0, origin=(0.0, 0.0), weight_intensity=False, raw=False, ctx=None,
                 cells=None, compensate=None, ccw=False):
        self.ctx = ctx or default_context()
        self._h = C.c_void_p()
        lib = self.ctx._lib
        if cells is not None:
            cells = np.ascontiguousarray(cells, dtype=L.CELL_DTYPE)
            self.ctx.check(lib.cfear_scan_from_cells(self.ctx.h, cells.ctypes.data, int(cells.shape[0]), C.byref(self._h)))
            return
        if raw:
            # GetIdentityCell per point (pointnormal.cpp:76-82, pointnormal.h:56,78-80)
            pts = np.asarray(cld if not _is_torch(cld) else cld.cpu().numpy())
            c = np.zeros(pts.shape[0], L.CELL_DTYPE)
            c["mean"] = pts[:, :2]
            c["normal"] = (1.0, 0.0)
            c["cov"] = (0.1, 0.0, 0.0, 0.1)
            c["scale"], c["avg_intensity"], c["lambda_min"], c["lambda_max"], c["nsamples"] = 1.0, 1.0, 1.0, 1.0, 1
            self.ctx.check(lib.cfear_scan_from_cells(self.ctx.h, c.ctypes.data, int(c.shape[0]), C.byref(self._h)))
            return
        fp = L.FeatureParams()
        fp.radius = float(radius)
        fp.downsample_factor = float(MapPointNormal.downsample_factor)
        fp.origin[0], fp.origin[1] = float(origin[0]), float(origin[1])
        fp.weight_intensity = int(bool(weight_intensity))
        fp.compensate = int(compensate is not None)
        if compensate is not None:
            fp.mot[0], fp.mot[1], fp.mot[2] = [float(v) for v in compensate]
        fp.ccw = int(bool(ccw))
        n = int(cld.shape[0])
        self.ctx.check(lib.cfear_scan_create(self.ctx.h, _ptr(cld)[0], n, C.byref(fp), C.byref(self._h)))

    @classmethod
    def _from_handle(cls, h, ctx):
        """Wraps a cfear_scan* the library handed out (cfear_odometry_get_scan); the wrapper owns it."""
        m = cls.__new__(cls)
        m.ctx, m._h = ctx, h
        return m

    def GetSize(self):
        return self.ctx._lib.cfear_scan_size(self._h)

    def GetClosestIdx(self, p, d):
        """GetClosestIdx(p, d) (pointnormal.cpp:238-254): [index of the nearest cell mean] or [] beyond d.
        p may also be an array [n, 2] of points (NumPy or torch CUDA float64): -> int32 [n], -1 where none."""
        single = not _is_torch(p) and np.ndim(p) == 1
        q = p if _is_torch(p) else np.ascontiguousarray(np.atleast_2d(p), dtype=np.float64)
        n = int(q.shape[0])
        if _is_torch(q):
            import torch
            q = q.contiguous()
            out = torch.empty(n, dtype=torch.int32, device=q.device)
        else:
            out = np.empty(n, np.int32)
        self.ctx.check(self.ctx._lib.cfear_scan_closest_idx(self._h, _ptr(q)[0], n, float(d), _ptr(out)[0]))
        if single:
            return [int(out[0])] if out[0] >= 0 else []
        return out

    def GetCells(self):
        n = self.GetSize()
        out = np.zeros(max(n, 1), L.CELL_DTYPE)
        rc = self.ctx._lib.cfear_scan_get_cells(self._h, out.ctypes.data, out.shape[0])
        if rc < 0:
            self.ctx.check(rc)
        return out[:n]

    def close(self):
        if getattr(self, "_h", None):
            if getattr(self.ctx, "h", None):              # the context may already be gone at interpreter exit: its
                self.ctx._lib.cfear_scan_destroy(self._h)   # objects died with it, destroying them again would be a use after free
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ------------------------------------------------------------------------------------------------
# n_scan_normal_reg
# ------------------------------------------------------------------------------------------------
class ScanTable:
    """The device views of a set of scans, uploaded once (cfear_scan_table_create): what a loop-closure thread keeps for the
    graph nodes' cloud_normal_ (types.h:119-122) so that a candidate is two indices and two poses."""

    def __init__(self, scans, ctx=None):
        self.ctx = ctx or scans[0].ctx
        self._scans = list(scans)                                  # (the table holds its own reference on every scan too)
        hs = (C.c_void_p * len(scans))(*[s._h for s in scans])
        self._h = C.c_void_p()
        self.ctx.check(self.ctx._lib.cfear_scan_table_create(self.ctx.h, hs, len(scans), C.byref(self._h)))

    def __len__(self):
        return int(self.ctx._lib.cfear_scan_table_size(self._h))

    @staticmethod
    def candidates(targets, sources, source_xyt, target_xyt=None):
        """CANDIDATE_DTYPE array from index arrays and [n][3] poses (target pose: the origin unless given)."""
        n = len(targets)
        c = np.zeros(n, L.CANDIDATE_DTYPE)
        c["target"], c["source"] = targets, sources
        c["source_xyt"] = np.asarray(source_xyt, dtype=np.float64).reshape(n, 3)
        if target_xyt is not None:
            c["target_xyt"] = np.asarray(target_xyt, dtype=np.float64).reshape(n, 3)
        return c

    def close(self):
        if self._h:
            self.ctx._lib.cfear_scan_table_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class RcclComm:
    """cfear_rccl_comm made by the library itself (cfear_rccl_unique_id / cfear_rccl_comm_init): `exchange(b)` hands rank 0's
    128-byte id to every rank -- any broadcast will do (bench.py uses torch.distributed); world 1 needs none."""

    def __init__(self, ctx, world=1, rank=0, exchange=None):
        self.ctx = ctx
        lib = ctx._lib
        uid = C.create_string_buffer(128)
        if rank == 0:
            rc = lib.cfear_rccl_unique_id(uid)
            if rc != L.OK:
                raise L.CfearError(rc, "librccl.so / ncclGetUniqueId not available")
        if world > 1:
            raw = exchange(bytes(uid.raw))
            assert len(raw) == 128
            uid = C.create_string_buffer(raw, 128)
        self.c = L.RcclComm()
        ctx.check(lib.cfear_rccl_comm_init(ctx.h, uid, int(world), int(rank), C.byref(self.c)))
        self.world, self.rank = int(world), int(rank)

    def close(self):
        if getattr(self, "c", None) is not None and self.c.nccl_comm:
            self.ctx._lib.cfear_rccl_comm_destroy(C.byref(self.c))

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class CandidatePipe:
    """cfear_candidate_pipe: up to `depth` sharded candidate steps in flight (submit never waits, collect waits on one event);
    every rank hands in the FULL candidate list and gets all n records back in candidate order."""

    def __init__(self, reg, table, max_candidates, comm=None, rank=0, world=1, depth=2, graph=False, timing=False):
        self.reg, self.table, self.ctx, self.comm = reg, table, reg.ctx, comm
        self._h = C.c_void_p()
        self.ctx.check(self.ctx._lib.cfear_candidate_pipe_create(self.ctx.h, table._h, int(max_candidates), int(rank), int(world),
                                                                 C.byref(comm.c) if comm is not None else None, int(depth),
                                                                 (L.PIPE_GRAPH if graph else 0) | (L.PIPE_TIMING if timing else 0),
                                                                 C.byref(self._h)))
        self.depth = int(depth)

    def stats(self):
        """{exchange_ms: all_gather + read-back summed over the collected steps (timing=True), steps, graph_slots}"""
        ms, n, g = C.c_double(), C.c_int64(), C.c_int32()
        self.ctx.check(self.ctx._lib.cfear_candidate_pipe_stats(self._h, C.byref(ms), C.byref(n), C.byref(g)))
        return {"exchange_ms": float(ms.value), "steps": int(n.value), "graph_slots": int(g.value)}

    def submit(self, cands):
        cands = np.ascontiguousarray(cands, dtype=L.CANDIDATE_DTYPE)
        t = C.c_int64()
        self.ctx.check(self.ctx._lib.cfear_candidate_pipe_submit(self._h, C.c_void_p(cands.ctypes.data), cands.shape[0],
                                                                 C.byref(self.reg.par), C.byref(t)))
        return (int(t.value), cands.shape[0])

    def collect(self, ticket, out=None):
        t, n = ticket
        out = np.empty(n, L.RESULT_DTYPE) if out is None else out
        self.ctx.check(self.ctx._lib.cfear_candidate_pipe_collect(self._h, t, C.c_void_p(out.ctypes.data)))
        return out

    def close(self):
        if getattr(self, "_h", None):
            self.ctx._lib.cfear_candidate_pipe_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class n_scan_normal_reg:
    """n_scan_normal_reg(cost, loss=Huber, loss_limit=0.1, opt=Uniform) (n_scan_normal.h:35)."""

    def __init__(self, cost="P2L", loss="Huber", loss_limit=0.1, opt=0, ctx=None):
        self.ctx = ctx or default_context()
        self.par = L.RegParams()
        self.ctx._lib.cfear_reg_params_default(C.byref(self.par))
        self.par.cost = L.COST[cost] if isinstance(cost, str) else int(cost)
        self.par.loss = L.LOSS[loss] if isinstance(loss, str) else int(loss)
        self.par.loss_limit = float(loss_limit)
        self.par.weight_opt = int(opt)
        self.summary_ = None
        self.score_ = 0.0

    def SetParameters(self, max_itr_association, max_itr_solver):          # n_scan_normal.cpp:15-19
        self.par.max_itr_association = int(max_itr_association)
        self.par.max_itr_solver = int(max_itr_solver)

    def SetD2dPar(self, cov_scale, regularization):                        # n_scan_normal.h:59
        self.par.cov_scale, self.par.regularization = float(cov_scale), float(regularization)

    def _handles(self, scans):
        return (C.c_void_p * len(scans))(*[s._h for s in scans])

    def Register(self, scans, Tsrc):
        """Register(scans, Tsrc, reg_cov) (n_scan_normal.cpp:82-185): Tsrc float64 [n,3] (x,y,theta);
        returns (success, Tsrc_out, reg_cov) with reg_cov the reference's constant diagonal."""
        p = np.ascontiguousarray(Tsrc, dtype=np.float64).copy()
        res = L.RegResult()
        rc = self.ctx._lib.cfear_register(self.ctx.h, self._handles(scans), len(scans),
                                          p.ctypes.data_as(C.POINTER(C.c_double)), C.byref(self.par), C.byref(res))
        self.ctx.check(rc, allowed=(L.ERR_TOO_FEW_RESIDUALS, L.ERR_SOLVER))
        self.summary_ = res
        self.score_ = res.score
        self.par.itr = res.outer_iters                                     # itr_ is left behind for GetCost
        cov = np.diag([0.1 * 0.1, 0.1 * 0.1, 0, 0, 0, 0.01 * 0.01])        # n_scan_normal.cpp:171-175
        return rc == L.OK, p, cov

    def PrepareBatch(self, jobs):
        """Marshals a list of (scans, Tsrc) once; the result can be passed to RegisterBatch repeatedly."""
        n = len(jobs)
        arr = (L.RegJob * n)()
        keep = []
        for i, (scans, T) in enumerate(jobs):
            hs = self._handles(scans)
            p = np.ascontiguousarray(T, dtype=np.float64)
            keep.append((hs, p, scans))
            arr[i].scans = C.cast(hs, C.POINTER(C.c_void_p))
            arr[i].n_scans = len(scans)
            arr[i].poses_xyt = p.ctypes.data_as(C.POINTER(C.c_double))
        return (arr, n, keep)

    def RegisterBatch(self, jobs):
        """jobs: list of (scans, Tsrc) or a PrepareBatch result.  One launch; returns a RESULT_DTYPE array
        (loop-closure candidate batches, tbv_slam/src/tbv_slam/loopclosure.cpp:35-97 per candidate)."""
        arr, n, _keep = jobs if isinstance(jobs, tuple) else self.PrepareBatch(jobs)
        out = np.zeros(n, L.RESULT_DTYPE)
        if n:
            self.ctx.check(self.ctx._lib.cfear_register_batch(self.ctx.h, arr, n, C.byref(self.par), out.ctypes.data))
        return out

    def RegisterBatchInto(self, jobs, device_ptr):
        """As RegisterBatch, but the records stay on the GPU: device_ptr = a device buffer of n * 72 bytes; the launch is
        enqueued on the context's stream and NOT synchronised (cfear_register_batch with a device pointer for results).
        Returns n."""
        arr, n, _keep = jobs if isinstance(jobs, tuple) else self.PrepareBatch(jobs)
        if n:
            self.ctx.check(self.ctx._lib.cfear_register_batch(self.ctx.h, arr, n, C.byref(self.par), C.c_void_p(int(device_ptr))))
        return n

    def RegisterCandidates(self, table, cands, device_ptr=None):
        """Candidate pairs among the scans of a ScanTable (cfear_register_candidates): cands = a CANDIDATE_DTYPE array
        (ScanTable.candidates builds one) -- 56 bytes per candidate cross PCIe instead of a marshalled job record.  Returns a
        RESULT_DTYPE array, or, with device_ptr (a device buffer of n * 72 bytes), leaves the records there without
        synchronising and returns n."""
        cands = np.ascontiguousarray(cands, dtype=L.CANDIDATE_DTYPE)
        n = cands.shape[0]
        out = None if device_ptr is not None else np.zeros(n, L.RESULT_DTYPE)
        if n:
            dst = C.c_void_p(int(device_ptr)) if device_ptr is not None else C.c_void_p(out.ctypes.data)
            self.ctx.check(self.ctx._lib.cfear_register_candidates(self.ctx.h, table._h, C.c_void_p(cands.ctypes.data), n,
                                                                   C.byref(self.par), dst))
        return n if device_ptr is not None else out

    def GetCost(self, scans, Tsrc):
        """GetCost (n_scan_normal.cpp:186-211) -> (success, score(cost), residuals)."""
        p = np.ascontiguousarray(Tsrc, dtype=np.float64)
        cap = 2 * sum(s.GetSize() for s in scans) + 2
        r = np.empty(cap, np.float64)
        cost, score, nres = C.c_double(), C.c_double(), C.c_int32()
        rc = self.ctx._lib.cfear_get_cost(self.ctx.h, self._handles(scans), len(scans),
                                          p.ctypes.data_as(C.POINTER(C.c_double)), C.byref(self.par), C.byref(cost),
                                          r.ctypes.data_as(C.POINTER(C.c_double)), cap, C.byref(nres), C.byref(score))
        self.ctx.check(rc, allowed=(L.ERR_TOO_FEW_RESIDUALS,))
        self.score_ = score.value
        return rc == L.OK, cost.value, r[:nres.value].copy()

    def getScore(self):
        return self.score_

    def GetCostBatch(self, jobs):
        """GetCost for a list of (scans, Tsrc) in one launch -> RESULT_DTYPE array (final_cost = robust cost,
        score, num_residuals, status).  The radius follows self.par.itr like GetCost."""
        arr, n, _keep = jobs if isinstance(jobs, tuple) else self.PrepareBatch(jobs)
        out = np.zeros(n, L.RESULT_DTYPE)
        if n:
            self.ctx.check(self.ctx._lib.cfear_get_cost_batch(self.ctx.h, arr, n, C.byref(self.par), out.ctypes.data))
        return out

    def GetCovarianceScaler(self):
        """GetCovarianceScaler (n_scan_normal.cpp:433-439) of the last Register -> (ok, scale)."""
        r = self.summary_
        if r is None or r.num_residuals - 3 == 0:
            return False, 1.0
        return True, r.final_cost / (r.num_residuals - 3)

    @staticmethod
    def sampling_params(xy_range=0.4, yaw_range=0.0043625, samples_per_axis=3, covariance_scaler=4.0):
        """OdometryKeyframeFuser::Parameters cov_sampling_* (odometrykeyframefuser.h:107-110); the loop-closure
        copy uses xy_range=0.4, yaw_range=0.0044 (loopclosure.cpp:108-112)."""
        sp = L.CovSamplingParams()
        sp.xy_range, sp.yaw_range = float(xy_range), float(yaw_range)
        sp.samples_per_axis, sp.covariance_scaler = int(samples_per_axis), float(covariance_scaler)
        return sp

    def approximateCovarianceBySampling(self, scans, T_vek, reg_result=None, sampling=None, want_samples=False):
        """OdometryKeyframeFuser::approximateCovarianceBySampling (odometrykeyframefuser.cpp:261-380) /
        loopclosure::approximateCovarianceBySampling (loopclosure.cpp:99-208).  T_vek: poses after Register;
        reg_result: that Register's summary (defaults to this object's last one).
        Returns (success, cov 6x6[, samples [n^3,4]])."""
        res = reg_result if reg_result is not None else self.summary_
        if res is None:
            raise ValueError("approximateCovarianceBySampling needs the result of a Register call")
        if not isinstance(res, L.RegResult):
            r = L.RegResult()
            for k in ("score", "final_cost", "num_residuals", "outer_iters", "lm_iters", "status"):
                setattr(r, k, res[k].item() if hasattr(res[k], "item") else res[k])
            res = r
        sp = sampling or self.sampling_params()
        p = np.ascontiguousarray(T_vek, dtype=np.float64)
        cov = np.zeros((6, 6), np.float64)
        m = sp.samples_per_axis ** 3
        smp = np.zeros((m, 4), np.float64)
        ok = C.c_int32()
        rc = self.ctx._lib.cfear_covariance_by_sampling(
            self.ctx.h, self._handles(scans), len(scans), p.ctypes.data_as(C.POINTER(C.c_double)), C.byref(self.par),
            C.byref(res), C.byref(sp), cov.ctypes.data_as(C.POINTER(C.c_double)),
            smp.ctypes.data_as(C.POINTER(C.c_double)), C.byref(ok))
        self.ctx.check(rc)
        return (bool(ok.value), cov, smp) if want_samples else (bool(ok.value), cov)

    def approximateCovarianceBySamplingBatch(self, jobs, reg_results, sampling=None):
        """Batch form: jobs as for RegisterBatch (poses AFTER registration), reg_results the RESULT_DTYPE array
        RegisterBatch returned.  -> (success [n] bool, cov [n,6,6])."""
        arr, n, _keep = jobs if isinstance(jobs, tuple) else self.PrepareBatch(jobs)
        sp = sampling or self.sampling_params()
        regs = np.ascontiguousarray(reg_results, dtype=L.RESULT_DTYPE)
        cov = np.zeros((n, 6, 6), np.float64)
        ok = np.zeros(n, np.int32)
        if n:
            self.ctx.check(self.ctx._lib.cfear_covariance_by_sampling_batch(
                self.ctx.h, arr, n, C.byref(self.par), regs.ctypes.data, C.byref(sp), cov.ctypes.data, None,
                ok.ctypes.data))
        return ok.astype(bool), cov


class CeresCost:
    """cfear_cost: Ceres-compatible evaluation of one association set (n_scan_normal.cpp:264-318)."""

    def __init__(self, reg, scans, Tsrc, itr=1):
        self.ctx = reg.ctx
        p = np.ascontiguousarray(Tsrc, dtype=np.float64)
        self._h = C.c_void_p()
        self.rpb = 1 if reg.par.cost == L.P2L else 2
        self.ctx.check(self.ctx._lib.cfear_cost_prepare(self.ctx.h, reg._handles(scans), len(scans),
                                                        p.ctypes.data_as(C.POINTER(C.c_double)), C.byref(reg.par),
                                                        int(itr), C.byref(self._h)))

    def blocks(self):
        n = self.ctx._lib.cfear_cost_num_blocks(self._h)
        pairs = np.empty((max(n, 1), 3), np.int32)
        w = np.empty(max(n, 1), np.float64)
        self.ctx._lib.cfear_cost_get_blocks(self._h, pairs.ctypes.data, w.ctypes.data)
        return pairs[:n], w[:n]

    def evaluate(self, x):
        n = self.ctx._lib.cfear_cost_num_residuals(self._h)
        r = np.empty(max(n, 1), np.float64)
        J = np.empty((max(n, 1), 3), np.float64)
        xx = (C.c_double * 3)(*[float(v) for v in x])
        self.ctx.check(self.ctx._lib.cfear_cost_evaluate(self._h, xx, r.ctypes.data, J.ctypes.data))
        return r[:n], J[:n]

    def normal_eq(self, x):
        xx = (C.c_double * 3)(*[float(v) for v in x])
        H = (C.c_double * 9)()
        g = (C.c_double * 3)()
        cost = C.c_double()
        self.ctx.check(self.ctx._lib.cfear_cost_normal_eq(self._h, xx, H, g, C.byref(cost)))
        return np.array(H).reshape(3, 3), np.array(g), cost.value

    def close(self):
        if getattr(self, "_h", None):
            if getattr(self.ctx, "h", None):              # the context may already be gone at interpreter exit: its
                self.ctx._lib.cfear_cost_destroy(self._h)   # objects died with it, destroying them again would be a use after free
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ------------------------------------------------------------------------------------------------
# batched radarDriver + OdometryKeyframeFuser
# ------------------------------------------------------------------------------------------------
# ------------------------------------------------------------------------------------------------
# CorAl alignment quality
# ------------------------------------------------------------------------------------------------
class CorAlRadarQuality:
    """CorAlRadarQuality(ref, src, par, Toffset) (coral_alignment_quality AlignmentQuality.cpp:99-230) over
    peak clouds, as ScanLearningInterface::getCorAlQualityMeasure builds it (alignmentinterface.cpp:437-456).

    ref / src: float32 [n, 4] clouds (NumPy or torch CUDA) in their sensor frames; poses (x, y, theta).
    GetQualityMeasure() -> [joint, sep, overlap]; .valid_ as in the reference."""

    def __init__(self, ref_cloud, ref_pose, src_cloud, src_pose, Toffset=(0.0, 0.0, 0.0), radius=1.0,
                 weight_res_intensity=False, want_per_point=False, ctx=None):
        out, pp = coral_quality_batch([(ref_cloud, ref_pose, src_cloud, src_pose, Toffset)], radius,
                                      weight_res_intensity, want_per_point, ctx)
        r = out[0]
        self.quality_ = [float(r["joint"]), float(r["sep"]), float(r["overlap"])]
        self.valid_ = bool(r["valid"])
        self.count_valid = int(r["count_valid"])
        self.per_point = pp[0] if want_per_point else None

    def GetQualityMeasure(self):
        return list(self.quality_)


def _compose_xyt(a, b):
    """Eigen Affine3d product of two planar poses (x, y, theta): a * b."""
    c, s = np.cos(a[2]), np.sin(a[2])
    return np.array([c * b[0] - s * b[1] + a[0], s * b[0] + c * b[1] + a[1], a[2] + b[2]], np.float64)


def cfear_quality_batch(jobs, method="P2L", ctx=None):
    """CFEARQuality (AlignmentQuality.cpp:330-354) for a list of (ref_scan, ref_pose, src_scan, src_pose, Toffset):
    a fresh n_scan_normal_reg(method, Huber, 0.3).GetCost on {ref, src} at {T_ref, T_src * Toffset}, one launch.
    -> float64 [n, 3] quality_ = {cost, #residuals, (N_src + N_ref) / 2} ({0, 0, 0} where GetCost fails)."""
    reg = n_scan_normal_reg(method, "Huber", 0.3, 0, ctx=ctx)
    gj = [([rs, ss], np.stack([np.asarray(rp, np.float64), _compose_xyt(np.asarray(sp, np.float64), np.asarray(off, np.float64))]))
          for rs, rp, ss, sp, off in jobs]
    out = reg.GetCostBatch(gj)
    q = np.zeros((len(jobs), 3), np.float64)
    for i, ((rs, _rp, ss, _sp, _off), r) in enumerate(zip(jobs, out)):
        if r["status"] == L.OK:
            q[i] = [r["final_cost"], r["num_residuals"], (ss.GetSize() + rs.GetSize()) / 2.0]
    return q


def coral_quality_batch(jobs, radius=1.0, weight_res_intensity=False, want_per_point=False, ctx=None):
    """jobs: list of (ref_cloud, ref_pose, src_cloud, src_pose, Toffset).  One launch.
    -> (CORAL_RESULT_DTYPE array, per-point list or None)."""
    ctx = ctx or default_context()
    n = len(jobs)
    arr = (L.CoralJob * n)()
    keep, sizes = [], []
    for i, (rc, rp, sc, sp, off) in enumerate(jobs):
        pr, nr, kr = _cloud_ptr(rc)
        ps, ns, ks = _cloud_ptr(sc)
        keep += [kr, ks]
        arr[i].ref_xyzi, arr[i].src_xyzi, arr[i].n_ref, arr[i].n_src = pr, ps, nr, ns
        for k in range(3):
            arr[i].ref_pose[k], arr[i].src_pose[k], arr[i].offset[k] = float(rp[k]), float(sp[k]), float(off[k])
        sizes.append(nr + ns)
    par = L.CoralParams()
    ctx._lib.cfear_coral_params_default(C.byref(par))
    par.radius, par.weight_res_intensity = float(radius), int(weight_res_intensity)
    out = np.zeros(n, L.CORAL_RESULT_DTYPE)
    pp = np.zeros((sum(sizes), 3), np.float64) if want_per_point else None
    if n:
        ctx.check(ctx._lib.cfear_coral_quality_batch(ctx.h, arr, n, C.byref(par), out.ctypes.data,
                                                     pp.ctypes.data if want_per_point else None))
    if not want_per_point:
        return out, None
    offs = np.cumsum([0] + sizes)
    return out, [pp[offs[i]:offs[i + 1]] for i in range(n)]


# ------------------------------------------------------------------------------------------------
# loop-candidate verification (tbv_slam loopclosure + alignment_checker ScanLearningInterface)
# ------------------------------------------------------------------------------------------------
ODOM_BOUNDS, SC_SIM, CFEAR_COST, CORAL_COST, COMBINED_COST = "odom-bounds", "sc-sim", "CFEAR", "coral", "alignment_quality"


def verify_params(ctx=None, **kw):
    """cfear_verify_params with loopclosure::BaseParameters' defaults (tbv_slam/include/tbv_slam/loopclosure.h:117-138);
    align_coef / loop_coef accept sequences (6 and 3 values)."""
    ctx = ctx or default_context()
    p = L.VerifyParams()
    ctx._lib.cfear_verify_params_default(C.byref(p))
    for k, v in kw.items():
        if k in ("align_coef", "loop_coef"):
            arr = getattr(p, k)
            assert len(v) == len(arr), k
            for i, x in enumerate(v):
                arr[i] = float(x)
        elif k in ("coral_radius",):
            p.coral.radius = float(v)
        else:
            assert hasattr(p, k), k
            setattr(p, k, type(getattr(p, k))(v))
    return p


def VerifyByOdometry(rel_xyt, odom_sigma_error=0.03, verify_via_odometry=True, ctx=None):
    """loopclosure::VerifyByOdometry (loopclosure.cpp:776-808): rel_xyt [n, 3] = RelativeMotion(i, i+1) for
    i = to .. from-1 -> similarity (quality["odom-bounds"])."""
    ctx = ctx or default_context()
    r = np.ascontiguousarray(rel_xyt, dtype=np.float64).reshape(-1, 3)
    out = C.c_double()
    ctx.check(ctx._lib.cfear_verify_by_odometry(r.ctypes.data, int(r.shape[0]), float(odom_sigma_error),
                                               int(bool(verify_via_odometry)), C.byref(out)))
    return out.value


def verify_loop_candidates(cands, par=None, ctx=None, device_ptr=None):
    """RegisterLoopCandidate + VerifyLoopCandidate + ApplyConstratins (loopclosure.cpp:320-384, 261-274) for a batch.
    cands: list of dicts with keys from_scan, to_scan (MapPointNormal), from_peaks, to_peaks (float32 [n, 4], NumPy
    or torch CUDA), from_pose (x, y, theta), t_be_guess, sc_sim, odom_bounds, group -- or a prepare_verify_batch
    result.  -> VERIFY_RESULT_DTYPE array; with device_ptr (a device buffer of n * 480 bytes) the records stay there
    WITHOUT the selection (accepted = rank = 0: verify_apply_constraints over the gathered list) and n is returned."""
    ctx = ctx or default_context()
    par = par or verify_params(ctx)
    arr, n, _keep = cands if isinstance(cands, tuple) else prepare_verify_batch(cands)
    out = None if device_ptr is not None else np.zeros(n, L.VERIFY_RESULT_DTYPE)
    if n:
        dst = C.c_void_p(int(device_ptr)) if device_ptr is not None else C.c_void_p(out.ctypes.data)
        ctx.check(ctx._lib.cfear_verify_loop_candidates(ctx.h, arr, n, C.byref(par), dst))
    return n if device_ptr is not None else out


def verify_apply_constraints(results, groups, par):
    """cfear_verify_apply_constraints: ApplyConstratins (loopclosure.cpp:261-274) over gathered records, in place."""
    g = np.ascontiguousarray(groups, dtype=np.int32)
    assert results.flags["C_CONTIGUOUS"] and results.dtype == L.VERIFY_RESULT_DTYPE and g.shape[0] == results.shape[0]
    rc = L.lib().cfear_verify_apply_constraints(C.c_void_p(g.ctypes.data), int(g.shape[0]), C.byref(par), C.c_void_p(results.ctypes.data))
    if rc != L.OK:
        raise L.CfearError(rc, "cfear_verify_apply_constraints")
    return results


def prepare_verify_batch(cands):
    """Marshals a candidate list once; the result can be passed to verify_loop_candidates repeatedly."""
    n = len(cands)
    arr = (L.VerifyJob * n)()
    keep = []
    for i, c in enumerate(cands):
        pf, nf, kf = _cloud_ptr(c["from_peaks"])
        pt, nt, kt = _cloud_ptr(c["to_peaks"])
        keep += [kf, kt]
        j = arr[i]
        j.from_scan, j.to_scan = c["from_scan"]._h, c["to_scan"]._h
        j.from_peaks, j.to_peaks, j.n_from, j.n_to = pf, pt, nf, nt
        for k in range(3):
            j.from_pose[k] = float(c["from_pose"][k])
            j.t_be_guess[k] = float(c.get("t_be_guess", (0.0, 0.0, 0.0))[k])
        j.sc_sim, j.odom_bounds = float(c.get("sc_sim", 0.0)), float(c.get("odom_bounds", 0.0))
        j.group = int(c.get("group", 0))
    return (arr, n, keep + [cands])


class LogisticRegression:
    """PythonClassifierInterface + LogisticRegression (alignmentinterface.cpp:14-279): training rows, the two text
    formats, predict_linear.  fit() is sklearn's LogisticRegression(class_weight="balanced", max_iter=1000), the
    call the reference makes through pybind11 (:192-222)."""

    def __init__(self):
        self.X_ = np.zeros((0, 0))
        self.y_ = np.zeros((0,))
        self.coef_ = None
        self.intercept_ = 0.0
        self.is_fit_ = False

    def IsFit(self):
        return self.is_fit_

    def AddDataPoint(self, X_i, y_i):                                   # :51-67
        X_i = np.atleast_2d(np.asarray(X_i, np.float64))
        y_i = np.atleast_1d(np.asarray(y_i, np.float64))
        self.X_ = X_i if self.X_.shape[0] == 0 else np.vstack([self.X_, X_i])
        self.y_ = y_i if self.y_.shape[0] == 0 else np.concatenate([self.y_, y_i])

    def DataValid(self):                                                # :175-187
        return (self.X_.shape[0] == self.y_.shape[0] and self.y_.shape[0] >= 1 and np.isfinite(self.X_).all()
                and np.isfinite(self.y_).all())

    def fit(self):
        if not self.DataValid():
            raise ValueError("training data invalid")                   # the reference calls exit(0) (:194-196)
        from sklearn.linear_model import LogisticRegression as SkLR
        clf = SkLR(class_weight="balanced", max_iter=1000).fit(self.X_, self.y_)
        self.coef_ = np.asarray(clf.coef_[0], np.float64).copy()
        self.intercept_ = float(clf.intercept_[0])
        self.is_fit_ = True

    def LoadData(self, path):                                           # :103-134: "y,x0,x1,..." per line
        rows = [ln.strip().split(",") for ln in open(path) if ln.strip()]
        if rows:
            a = np.array(rows, dtype=np.float64)
            self.y_, self.X_ = a[:, 0].copy(), a[:, 1:].copy()

    def SaveData(self, path):                                           # :152-173, default ostream precision (%g)
        with open(path, "w") as f:
            for y, x in zip(self.y_, self.X_):
                f.write(",".join(["%g" % y] + ["%g" % v for v in x]) + "\n")

    def LoadCoefficients(self, path):                                   # :224-253: "intercept,c0,c1,..."
        for ln in open(path):
            v = [float(t) for t in ln.strip().split(",") if t]
            if v:
                self.intercept_, self.coef_ = v[0], np.array(v[1:], np.float64)
        self.is_fit_ = True

    def SaveCoefficients(self, path):                                   # :255-269
        with open(path, "w") as f:
            f.write(",".join(["%g" % self.intercept_] + ["%g" % c for c in self.coef_]) + "\n")

    def predict_linear(self, X):                                        # :271-279
        return np.atleast_2d(np.asarray(X, np.float64)) @ self.coef_ + self.intercept_

    def predict_proba(self, X):                                         # :21-33: P(y = 1); zeros when not fitted
        if not self.is_fit_:
            return np.zeros(np.atleast_2d(X).shape[0])
        return 1.0 / (1.0 + np.exp(-self.predict_linear(X)))


class ScanLearningInterface:
    """ScanLearningInterface (alignment_checker/alignmentinterface.h:96-213, alignmentinterface.cpp:288-510).
    A scan is a dict {"T": (x, y, theta), "cldPeaks": float32 [n, 4], "CFEAR": MapPointNormal} (s_scan, :103-109).
    Every CorAl / CFEAR quality evaluation of one call -- the 13 perturbations of AddTrainingData -- is one launch."""

    range_error_, min_dist_btw_scans_ = 0.5, 0.5                         # alignmentinterface.h:196-197
    small_th_err, medium_th_err, large_th_err = 0.5 * np.pi / 180.0, 2 * np.pi / 180.0, 15 * np.pi / 180.0

    def __init__(self, combined=True, ctx=None):
        self.ctx = ctx or default_context()
        self.combined_ = combined
        self.cfear_class, self.coral_class, self.combined_class = LogisticRegression(), LogisticRegression(), LogisticRegression()
        self.frame_ = 0
        self.prev_ = None
        e = self.range_error_                                            # CreatePerturbations (:479-495)
        self.vek_perturbation_ = [(0.0, 0.0, 0.0)]
        for m, th in ((1, self.small_th_err), (2, self.medium_th_err), (4, self.large_th_err)):
            self.vek_perturbation_ += [(m * e, 0.0, th), (0.0, m * e, th), (-m * e, 0.0, th), (0.0, -m * e, th)]

    def _quality(self, current, prev, offsets):
        """X_CorAl, X_CFEAR [len(offsets), 3] for ref = current, src = prev * Toffset (getCorAlQualityMeasure :437-456,
        getCFEARQualityMeasure :459-478)."""
        cj = [(current["cldPeaks"], current["T"], prev["cldPeaks"], prev["T"], o) for o in offsets]
        qj = [(current["CFEAR"], current["T"], prev["CFEAR"], prev["T"], o) for o in offsets]
        co, _ = coral_quality_batch(cj, 1.0, False, False, self.ctx)
        X_coral = np.stack([co["joint"], co["sep"], co["overlap"]], 1).astype(np.float64)
        X_cfear = cfear_quality_batch(qj, "P2L", self.ctx)
        return X_coral, X_cfear

    def AddTrainingData(self, current):                                  # :296-347
        first = self.frame_ == 0
        self.frame_ += 1
        if first:
            self.prev_ = current
            return
        d = np.hypot(current["T"][0] - self.prev_["T"][0], current["T"][1] - self.prev_["T"][1])
        if d < self.min_dist_btw_scans_:
            return
        Xc, Xf = self._quality(current, self.prev_, self.vek_perturbation_)
        for verr, xc, xf in zip(self.vek_perturbation_, Xc, Xf):
            y = float(sum(abs(v) for v in verr) < 0.0001)
            if self.combined_:
                self.combined_class.AddDataPoint(np.concatenate([xc, xf]), y)
            else:
                self.coral_class.AddDataPoint(xc, y)
                self.cfear_class.AddDataPoint(xf, y)
        self.prev_ = current

    def PredAlignment(self, current, prev, quality=None):                # :349-367
        """-> (quality dict, X_CorAl, X_CFEAR); `valid` of the reference is valid1 && valid2 of two locals that
        are never written, i.e. always false, and unused by its callers."""
        quality = {} if quality is None else quality
        Xc, Xf = self._quality(current, prev, [(0.0, 0.0, 0.0)])
        if self.combined_:
            quality[COMBINED_COST] = float(self.combined_class.predict_linear(np.concatenate([Xc[0], Xf[0]]))[0])
        else:
            quality[CORAL_COST] = float(self.coral_class.predict_proba(Xc)[0])
            quality[CFEAR_COST] = float(self.cfear_class.predict_proba(Xf)[0])
        return quality, Xc, Xf

    def _files(self, d, combined_name, coral_name, cfear_name):
        d = str(d)
        if self.combined_:
            return [(self.combined_class, d + combined_name)]
        return [(self.coral_class, d + coral_name), (self.cfear_class, d + cfear_name)]

    def LoadData(self, d):                                               # :376-383
        for clf, f in self._files(d, "/combined.txt", "/CorAl.txt", "/CFEAR.txt"):
            clf.LoadData(f)

    def SaveData(self, d):                                               # :386-393
        for clf, f in self._files(d, "/combined.txt", "/CorAl.txt", "/CFEAR.txt"):
            clf.SaveData(f)

    def LoadCoefficients(self, d):                                       # :396-403 (dir is concatenated without "/")
        for clf, f in self._files(d, "trained_alignment_classifier.txt", "trained_alignment_classifier_CorAl.txt",
                                  "trained_alignment_classifier_CFEAR.txt"):
            clf.LoadCoefficients(f)

    def SaveCoefficients(self, d):                                       # :405-412
        for clf, f in self._files(d, "/trained_alignment_classifier.txt", "/trained_alignment_classifier_CorAl.txt",
                                  "/trained_alignment_classifier_CFEAR.txt"):
            clf.SaveCoefficients(f)

    def FitModels(self, model="LogisticRegression"):                     # :423-434
        for clf, _ in self._files("", "", "", ""):
            clf.fit()

    def verify_params(self, **kw):
        """cfear_verify_params carrying this interface's combined classifier."""
        assert self.combined_ and self.combined_class.IsFit()
        return verify_params(self.ctx, align_intercept=self.combined_class.intercept_,
                             align_coef=list(self.combined_class.coef_), **kw)


def _cloud_ptr(cloud):
    """float32 [n, 4] NumPy array or torch CUDA tensor -> (pointer, n, keep-alive)."""
    if isinstance(cloud, np.ndarray):
        c = np.ascontiguousarray(cloud, dtype=np.float32)
        assert c.ndim == 2 and c.shape[1] == 4
        return c.ctypes.data, c.shape[0], c
    assert cloud.dim() == 2 and cloud.shape[1] == 4 and cloud.is_contiguous()
    return cloud.data_ptr(), int(cloud.shape[0]), cloud


# ------------------------------------------------------------------------------------------------
# radar Scan Context (loop-candidate generation; place_recognition_radar)
# ------------------------------------------------------------------------------------------------
def sc_params(**kw):
    """cfear_sc_params with TBV's defaults (40 x 120, 80 m, search ratio 0.1, sum / 1000)."""
    p = L.ScParams()
    L.lib().cfear_sc_params_default(C.byref(p))
    for k, v in kw.items():
        if k == "desc_function" and isinstance(v, str):
            v = {"sum": 0, "max": 1}[v]
        if not hasattr(p, k):
            raise KeyError(k)
        setattr(p, k, v)
    return p


def sc_descriptors(clouds, par=None, shifts_y=(0.0,), ctx=None, device_out=False):
    """MakeRadarCloudContext (RadarScancontext.cpp:59-131) for a list of clouds and lateral shifts.
    -> (desc [n, A, R, S], ringkey [n, A, R], sectorkey [n, A, S]); with device_out the descriptors stay in HBM
    (torch CUDA float64 tensor, accepted by sc_distance_batch) and only the keys come back to the host."""
    ctx = ctx or default_context()
    par = par or sc_params()
    n, A = len(clouds), len(shifts_y)
    arr = (L.ScCloud * max(n, 1))()
    keep = []
    for i, c in enumerate(clouds):
        ptr, m, k = _cloud_ptr(c)
        keep.append(k)
        arr[i].xyzi, arr[i].n = ptr, m
    R, S = par.num_ring, par.num_sector
    if device_out:
        import torch
        desc = torch.zeros((n, A, R, S), dtype=torch.float64, device="cuda:%d" % ctx.device)
    else:
        desc = np.zeros((n, A, R, S), np.float64)
    rk = np.zeros((n, A, R), np.float64)
    sk = np.zeros((n, A, S), np.float64)
    sh = (C.c_double * A)(*[float(v) for v in shifts_y])
    ctx.check(ctx._lib.cfear_sc_descriptors(ctx.h, arr, n, C.byref(par), sh, A, _ptr(desc)[0], rk.ctypes.data,
                                            sk.ctypes.data))
    return desc, rk, sk


def sc_distance_batch(desc_q, desc_c, pairs, par=None, ctx=None):
    """distanceBtnScanContext (Scancontext.cpp:157-189) for pairs (query index, candidate index).
    desc_q [nq, R, S], desc_c [nc, R, S] float64 -> (dist [n_pairs], argmin_shift [n_pairs])."""
    ctx = ctx or default_context()
    par = par or sc_params()
    q = desc_q.contiguous() if _is_torch(desc_q) else np.ascontiguousarray(desc_q, dtype=np.float64)
    c = desc_c.contiguous() if _is_torch(desc_c) else np.ascontiguousarray(desc_c, dtype=np.float64)
    pr = np.ascontiguousarray(pairs, dtype=np.int32).reshape(-1, 2)
    dist = np.zeros(pr.shape[0], np.float64)
    shift = np.zeros(pr.shape[0], np.int32)
    if pr.shape[0]:
        ctx.check(ctx._lib.cfear_sc_distance_batch(ctx.h, _ptr(q)[0], q.shape[0], _ptr(c)[0], c.shape[0],
                                                   pr.ctypes.data, pr.shape[0], C.byref(par), dist.ctypes.data,
                                                   shift.ctypes.data))
    return dist, shift


class RSCManager:
    """RSCManager (place_recognition_radar RadarScancontext.{h,cpp}) for cloud descriptors: the descriptor
    database, the recent-node exclusion, the odometry similarity and the candidate ranking are host policy
    restated here; descriptors and scan-context distances run on the GPU."""

    DISTANCE_EXCLUDE_RECENT = 10.0                    # Scancontext.h:108
    AUGMENTS_Y = (-2.0, 2.0, -4.0, 4.0)               # RadarScancontext.cpp:164

    def __init__(self, par=None, num_candidates_from_tree=10, n_candidates=3, odom_sigma_error=0.05,
                 odometry_coupled_closure=True, augment_sc=True, ctx=None):
        self.ctx = ctx
        self.par = par or sc_params()
        self.NUM_CANDIDATES_FROM_TREE = int(num_candidates_from_tree)
        self.N_candidates = int(n_candidates)
        self.odom_sigma_error = float(odom_sigma_error)
        self.odometry_coupled_closure = bool(odometry_coupled_closure)
        self.augment_sc = bool(augment_sc)
        self.polarcontexts_ = []                      # [R, S] float64 per node (device tensors with the GPU hooks)
        self.polarcontext_invkeys_mat_ = []           # float32 ring keys (eig2stdvec), [R] per node
        self.odom_poses_ = []                         # (x, y, theta)
        self.odom_similarity = np.zeros(0)
        self.NUM_EXCLUDE_RECENT = 0
        self.current_and_augments_ = []               # (desc, ring key float32, (tx, ty, theta) of the augmentation)

    # the two device operations (tests substitute the CPU oracle here to check the host policy around them)
    def _descriptors(self, clouds, shifts):
        return sc_descriptors(clouds, self.par, shifts, self.ctx, device_out=True)   # the database lives in HBM

    def _distances(self, desc_q, desc_c, pairs):
        return sc_distance_batch(desc_q, desc_c, pairs, self.par, self.ctx)

    def makeAndSaveScancontextAndKeysRadarCloud(self, cloud, Todom):
        """RadarScancontext.cpp:156-180 (+ :133-146, :181-225)."""
        shifts = (0.0,) + (self.AUGMENTS_Y if self.augment_sc else ())
        desc, rk, _ = self._descriptors([cloud], shifts)
        self.polarcontexts_.append(desc[0, 0])
        self.polarcontext_invkeys_mat_.append(rk[0, 0].astype(np.float32))
        self.current_and_augments_ = [(desc[0, k], rk[0, k].astype(np.float32), (0.0, float(shifts[k]), 0.0))
                                      for k in range(len(shifts))]
        self._exclude_and_update_likelihood(np.asarray(Todom, np.float64))

    def _exclude_and_update_likelihood(self, Todom):                       # RadarScancontext.cpp:181-222
        self.odom_poses_.append(Todom)
        P = np.asarray(self.odom_poses_)
        if len(P) <= 2:
            self.NUM_EXCLUDE_RECENT = 2
        else:
            distance, n_ex, prev = 0.0, 0, P[-1]
            i = len(P) - 1
            while i >= 0 and distance < self.DISTANCE_EXCLUDE_RECENT:
                distance = distance + float(np.hypot(*(P[i][:2] - prev[:2])))   # |(Tprev^-1 T_i).translation()|
                prev = P[i]
                n_ex += 1
                i -= 1
            self.NUM_EXCLUDE_RECENT = n_ex
        idx_current = len(P) - 1
        self.odom_similarity = np.zeros(idx_current)
        tprev = Todom[:2].copy()
        trav = 0.0
        for i in range(idx_current - 1, -1, -1):
            t_i = P[i][:2]
            trav += float(np.hypot(*(tprev - t_i)))
            tprev = t_i
            est = float(np.hypot(*(Todom[:2] - t_i)))
            error = max(est - 5.0, 0.0)
            with np.errstate(divide="ignore", invalid="ignore"):
                rel = np.float64(error) / np.float64(trav)
            prob = np.exp(-rel * rel / (2 * self.odom_sigma_error * self.odom_sigma_error))
            self.odom_similarity[i] = 1.0 - prob

    def _odometry_nn_search(self, curr_key):                                # RadarScancontext.cpp:259-284
        key = np.append(curr_key.astype(np.float32), np.float32(0.0))
        idx_current = len(self.polarcontext_invkeys_mat_) - 1
        cands = []
        for idx in range(0, max(idx_current - 1 - self.NUM_EXCLUDE_RECENT, 0)):
            other = np.append(self.polarcontext_invkeys_mat_[idx], np.float32(10 * self.odom_similarity[idx]))
            l2 = np.float32(0.0)                      # `float l2`, err in double (L2norm, :250-257)
            for a_, b_ in zip(key, other):
                err = np.float64(a_ - b_)             # float subtraction promoted to double
                l2 = np.float32(np.float64(l2) + err * err)
            cands.append((float(l2), idx))
        cands.sort()                                  # lower_bound insertion == sort by (distance, index)
        return [i for _, i in cands[:self.NUM_CANDIDATES_FROM_TREE]]

    TREE_MAKING_PERIOD_ = 50                          # Scancontext.h:116

    @staticmethod
    def _l2_adaptor(K, q):
        """nanoflann::L2_Adaptor::evalMetric (the reference's vendored nanoflann.hpp:383-408) in float: groups of four
        squared differences are added among themselves first, then to the running sum; the tail one by one."""
        K = np.asarray(K, np.float32)
        e = K - np.asarray(q, np.float32)[None]
        e2 = e * e
        d = np.zeros(K.shape[0], np.float32)
        g4 = (K.shape[1] // 4) * 4
        for c in range(0, g4, 4):
            d = d + (((e2[:, c] + e2[:, c + 1]) + e2[:, c + 2]) + e2[:, c + 3])
        for c in range(g4, K.shape[1]):
            d = d + e2[:, c]
        return d

    def _vanilla_nn_search(self, curr_key):                                 # RadarScancontext.cpp:225-248
        """The ring-key kd-tree retrieval with the reference's bookkeeping: the tree is rebuilt only on every
        TREE_MAKING_PERIOD_-th CALL (one call per augmentation of every node) from the keys older than the recent-node
        exclusion AT THAT MOMENT, and a tree with fewer points than NUM_CANDIDATES_FROM_TREE leaves the rest of the
        zero-initialised index vector in place (node 0 is then proposed again).  The search itself is exact (nanoflann,
        eps = 0), so a linear scan with the tree's own metric arithmetic returns the same neighbours; equal distances
        come back in index order here, in tree-visiting order there (tests/test_ref_nanoflann.py)."""
        if getattr(self, "_tree_counter", None) is None:
            self._tree_counter, self._tree_keys = 0, np.zeros((0, 0), np.float32)
        if self._tree_counter % self.TREE_MAKING_PERIOD_ == 0:
            n = len(self.polarcontext_invkeys_mat_) - self.NUM_EXCLUDE_RECENT
            self._tree_keys = np.asarray(self.polarcontext_invkeys_mat_[:max(n, 0)], np.float32).copy()
        self._tree_counter += 1
        K = self._tree_keys
        out = [0] * self.NUM_CANDIDATES_FROM_TREE
        if K.shape[0] > 0:
            d = self._l2_adaptor(K, curr_key)
            order = np.argsort(d, kind="stable")[:self.NUM_CANDIDATES_FROM_TREE]
            for j, i in enumerate(order):
                out[j] = int(i)
        return out

    def detectLoopClosureID(self):
        """RadarScancontext.cpp:286-345 -> list of candidates dict(min_dist, min_dist_sc, min_dist_odom,
        yaw_diff_rad, nn_idx, argmin_shift, Taug), closest first."""
        if len(self.polarcontext_invkeys_mat_) < self.NUM_EXCLUDE_RECENT + 1:
            return []
        jobs = []                                     # (query k, candidate idx) in the reference's visiting order
        for k, (_d, rk, _T) in enumerate(self.current_and_augments_):
            idxs = self._odometry_nn_search(rk) if self.odometry_coupled_closure else self._vanilla_nn_search(rk)
            jobs += [(k, i) for i in idxs]
        if not jobs:
            return []
        def stack(xs):
            if _is_torch(xs[0]):
                import torch
                return torch.stack(xs)
            return np.stack(xs)
        qd = stack([d for d, _, _ in self.current_and_augments_])
        uniq = sorted({i for _, i in jobs})
        pos = {i: p for p, i in enumerate(uniq)}
        cd = stack([self.polarcontexts_[i] for i in uniq])
        dist, shift = self._distances(qd, cd, [(k, pos[i]) for k, i in jobs])
        unit = 360.0 / float(self.par.num_sector)
        similar = []
        for (k, i), d_sc, sh in zip(jobs, dist, shift):
            d_odom = float(self.odom_similarity[i]) if self.odometry_coupled_closure else 0.0
            d_tot = float(d_sc) + d_odom if self.odometry_coupled_closure else float(d_sc)
            similar.append(dict(min_dist=d_tot, min_dist_sc=float(d_sc), min_dist_odom=d_odom,
                                # deg2rad(float) (RadarScancontext.cpp:6-9): float degrees, double product, float result
                                yaw_diff_rad=float(np.float32(float(np.float32(sh * unit)) * np.pi / 180.0)), nn_idx=int(i),
                                argmin_shift=int(sh), Taug=self.current_and_augments_[k][2]))
            similar.sort(key=lambda c: c["min_dist"])     # std::sort + erase of the worst (:317-320)
            if len(similar) > self.N_candidates:
                similar.pop()
        return similar


class RSCManagerNative:
    """The same manager as a library object (cfear_sc_manager_*): database in HBM, policy in the library's C++.
    Same two calls as RSCManager; what a C++ host uses (include/cfear_hip.hpp)."""

    def __init__(self, par=None, num_candidates_from_tree=10, n_candidates=3, odom_sigma_error=0.05,
                 odometry_coupled_closure=True, augment_sc=True, ctx=None):
        self.ctx = ctx or default_context()
        p = L.ScManagerParams()
        self.ctx._lib.cfear_sc_manager_params_default(C.byref(p))
        if par is not None:
            p.sc = par
        p.num_candidates_from_tree, p.n_candidates = int(num_candidates_from_tree), int(n_candidates)
        p.odom_sigma_error = float(odom_sigma_error)
        p.odometry_coupled_closure, p.augment_sc = int(odometry_coupled_closure), int(augment_sc)
        self.par = p
        self._h = C.c_void_p()
        self.ctx.check(self.ctx._lib.cfear_sc_manager_create(self.ctx.h, C.byref(p), C.byref(self._h)))

    def makeAndSaveScancontextAndKeysRadarCloud(self, cloud, Todom):
        ptr, n, _keep = _cloud_ptr(cloud)
        T = (C.c_double * 3)(*[float(v) for v in Todom])
        self.ctx.check(self.ctx._lib.cfear_sc_manager_add(self._h, ptr, n, T))

    def detectLoopClosureID(self):
        out = np.zeros(max(int(self.par.n_candidates), 1), L.SC_CANDIDATE_DTYPE)
        n = C.c_int32()
        self.ctx.check(self.ctx._lib.cfear_sc_manager_detect(self._h, out.ctypes.data, out.shape[0], C.byref(n)))
        return [dict(min_dist=float(c["min_dist"]), min_dist_sc=float(c["min_dist_sc"]), min_dist_odom=float(c["min_dist_odom"]),
                     yaw_diff_rad=float(c["yaw_diff_rad"]), nn_idx=int(c["nn_idx"]), argmin_shift=int(c["argmin_shift"]),
                     Taug=tuple(float(v) for v in c["Taug"])) for c in out[:n.value]]

    def size(self):
        return self.ctx._lib.cfear_sc_manager_size(self._h)

    def close(self):
        if getattr(self, "_h", None):
            if getattr(self.ctx, "h", None):              # the context may already be gone at interpreter exit: its
                self.ctx._lib.cfear_sc_manager_destroy(self._h)   # objects died with it, destroying them again would be a use after free
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def odometry_params(**kw):
    """cfear_odometry_params with the CFEAR-3 / Oxford preset; keyword overrides use the C field names
    (nested: kstrong_k_strongest, cacfar_window_size, reg_cost, cov_sampling_xy_range, ...)."""
    p = L.OdometryParams()
    L.lib().cfear_odometry_params_default(C.byref(p))
    return _override_odometry_params(p, kw)


def _override_odometry_params(p, kw):
    for k, v in kw.items():
        for prefix, sub in (("kstrong_", p.kstrong), ("cacfar_", p.cacfar), ("reg_", p.reg),
                            ("cov_sampling_", p.cov_sampling)):
            if k.startswith(prefix) and hasattr(sub, k[len(prefix):]):
                setattr(sub, k[len(prefix):], v)
                break
        else:
            if not hasattr(p, k):
                raise KeyError(k)
            setattr(p, k, v)
    return p


class EvalTrajectory:
    """The KITTI-style trajectory text of the reference's evaluation (EvalTrajectory::Write, eval_trajectory.cpp:169-183;
    MatToString, types.cpp:64-73): one pose per line, the top three rows of the 4 x 4 matrix, std::fixed (6 decimals)."""

    @staticmethod
    def MatToString(pose):
        x, y, th = (float(v) for v in pose)
        c, s = np.cos(th), np.sin(th)
        m = (c, -s, 0.0, x, s, c, 0.0, y, 0.0, 0.0, 1.0, 0.0)
        return " ".join("%f" % v for v in m)

    @staticmethod
    def Write(path, poses):
        with open(path, "w") as f:
            for p in np.asarray(poses, np.float64).reshape(-1, 3):
                f.write(EvalTrajectory.MatToString(p) + "\n")

    @staticmethod
    def Read(path):
        """-> poses [n, 3] (x, y, theta) of a planar trajectory file."""
        a = np.loadtxt(path, dtype=np.float64, ndmin=2)
        assert a.shape[1] == 12
        return np.stack([a[:, 3], a[:, 7], np.arctan2(a[:, 4], a[:, 0])], 1)


PRESETS = {"CFEAR-1": 1, "CFEAR-2": 2, "CFEAR-3": 3, "CFEAR-3-s10": 4}
DATASETS = {"oxford": 0, "mulran": 1, "kvarntorp": 2, "volvo": 3}


def odometry_preset(preset="CFEAR-3", dataset="oxford", **kw):
    """cfear_odometry_params of one of the reference's shipped configurations (launch/oxford/eval/params/baseline/
    oxford_cfear-{1,2,3,3-s10}) on one of its sensor setups (tbv_slam/script/*/run_tbv_simple.sh); keyword overrides
    as in odometry_params.  Non-Oxford datasets expect [range bins][azimuths] images (rotate_ccw)."""
    p = L.OdometryParams()
    rc = L.lib().cfear_odometry_params_preset(C.byref(p), PRESETS[preset], DATASETS[dataset.lower()])
    if rc != L.OK:
        raise L.CfearError(rc, "unknown preset / dataset")
    return _override_odometry_params(p, kw)


class OdometryKeyframeFuser:
    """n_streams independent radarDriver + OdometryKeyframeFuser pairs advanced one frame per call
    (radar_driver.cpp:163-176 + odometrykeyframefuser.cpp:143-259), everything on the GPU."""

    def __init__(self, n_streams, rows, cols, par=None, ctx=None):
        """rows x cols: the layout of the images passed to process() -- azimuths x range bins, or range bins x
        azimuths when par.rotate_ccw is set (non-Oxford drivers, radar_driver.cpp:74-90)."""
        self.ctx = ctx or default_context()
        self.par = par or odometry_params()
        d = L.PolarDesc()
        d.rows, d.cols, d.stride, d.batch = rows, cols, cols, n_streams
        d.batch_stride = rows * cols
        self.n_streams = n_streams
        self._h = C.c_void_p()
        self.ctx.check(self.ctx._lib.cfear_odometry_create(self.ctx.h, n_streams, C.byref(d), C.byref(self.par),
                                                           C.byref(self._h)))
        self._info = np.zeros(n_streams, L.FRAMEINFO_DTYPE)

    def process(self, polar, polar_next=None):
        """polar: uint8 [n_streams, rows, cols] (NumPy or torch CUDA).  Returns a FRAMEINFO_DTYPE array.
        polar_next (optional): the NEXT frame's images; their filter is enqueued behind this frame's
        kernels so the GPU works while the host applies this frame's keyframe policy.  The next call
        must pass that same buffer as `polar`."""
        self._keep = (polar, polar_next)
        # per-stream failures (an empty sweep, a capacity overflow) are results, not exceptions: the other streams
        # have advanced and their info is filled in -- inspect info["reg_status"]
        self._info[:] = 0
        return self._done(self.ctx._lib.cfear_odometry_process_prefetch(self._h, _ptr(polar)[0], _ptr(polar_next)[0],
                                                                        self._info.ctypes.data))

    _PER_STREAM = (L.ERR_EMPTY_CLOUD, L.ERR_CAPACITY)

    def _done(self, rc):
        """A status that some stream reports as its own (info.reg_status) is that stream's result; anything else is a
        failed call."""
        if rc in self._PER_STREAM and (self._info["reg_status"] == rc).any():
            return self._info.copy()
        self.ctx.check(rc)
        return self._info.copy()

    def process_offsets(self, base, offsets, offsets_next=None):
        """The same step for sweeps that do not sit at a constant stride: stream b's image starts `offsets[b]` bytes
        into the device buffer `base` (a ring of frames, one buffer per sequence).  offsets / offsets_next: int64
        [n_streams].  offsets_next prefetches the next frame's filter; pass the same values as `offsets` next time."""
        self._keep = (base,)
        o = np.ascontiguousarray(offsets, np.int64)
        on = None if offsets_next is None else np.ascontiguousarray(offsets_next, np.int64)
        assert o.shape == (self.n_streams,) and (on is None or on.shape == o.shape)
        self._info[:] = 0
        return self._done(self.ctx._lib.cfear_odometry_process_offsets(self._h, _ptr(base)[0], o.ctypes.data,
                                                                       None if on is None else on.ctypes.data,
                                                                       self._info.ctypes.data))

    def discard_prefetch(self):
        """Forget a prefetched filter output (call before REUSING an image buffer for different content)."""
        self.ctx.check(self.ctx._lib.cfear_odometry_discard_prefetch(self._h))

    def process_clouds(self, clouds, peaks=None):
        """pointcloudCallback(cloud, cloud_peaks, ...) (odometrykeyframefuser.cpp:413-426) for every stream: the caller's
        driver has filtered the sweeps already.  clouds / peaks: one float32 [n, 4] array (NumPy or torch CUDA) per
        stream; peaks are kept only with par.keep_nodes.  Returns a FRAMEINFO_DTYPE array."""
        def pack(lst):
            arr = (L.ScCloud * self.n_streams)()
            keep = []
            for i, c in enumerate(lst):
                ptr, n, k = _cloud_ptr(c)
                arr[i].xyzi, arr[i].n = ptr, n
                keep.append(k)
            return arr, keep
        assert len(clouds) == self.n_streams and (peaks is None or len(peaks) == self.n_streams)
        ca, k1 = pack(clouds)
        pa, k2 = pack(peaks) if peaks is not None else (None, None)
        self._info[:] = 0
        return self._done(self.ctx._lib.cfear_odometry_process_clouds(self._h, ca, pa, self._info.ctypes.data))

    def node(self, stream, device=False):
        """The RadarScan of `stream`'s last processed frame (scan_, odometrykeyframefuser.cpp:172, 244; types.h:119-122)
        -> dict(scan=MapPointNormal copy of cloud_normal_, cloud=cloud_nopeaks_, peaks=cloud_peaks_ (par.keep_nodes)).
        Clouds are NumPy arrays, or torch CUDA tensors with device=True.  Valid until the next process()."""
        lib = self.ctx._lib
        h = C.c_void_p()
        self.ctx.check(lib.cfear_odometry_get_scan(self._h, int(stream), C.byref(h)))
        out = {"scan": MapPointNormal._from_handle(h, self.ctx)}
        for name, fn in (("cloud", lib.cfear_odometry_get_cloud), ("peaks", lib.cfear_odometry_get_peaks)):
            if name == "peaks" and not self.par.keep_nodes:
                continue
            n = C.c_int32()
            self.ctx.check(fn(self._h, int(stream), None, 0, C.byref(n)))
            if device:
                import torch
                buf = torch.empty((n.value, 4), dtype=torch.float32, device="cuda:%d" % self.ctx.device)
            else:
                buf = np.empty((n.value, 4), np.float32)
            if n.value:
                self.ctx.check(fn(self._h, int(stream), _ptr(buf)[0], n.value, C.byref(n)))
            out[name] = buf
        return out

    def constraint(self, stream):
        """OdometryKeyframeFuser::AddToGraph (odometrykeyframefuser.cpp:428-445): the odometry Constraint3d from the keyframe
        the last frame added to the keyframe before it -> dict (SaveSimpleGraph layout), or None if the last frame added
        no keyframe (or the first one)."""
        c = L.GraphConstraint()
        rc = self.ctx._lib.cfear_odometry_get_constraint(self._h, int(stream), C.byref(c))
        if rc == L.ERR_INVALID_ARGUMENT:
            return None
        self.ctx.check(rc)
        return dict(id_begin=int(c.id_begin), id_end=int(c.id_end), t_be=np.array(list(c.t_be.p) + list(c.t_be.q)),
                    information=np.array(c.information).reshape(6, 6), type=int(c.type), quality={}, info="")

    def covariance(self):
        """cov_current of every stream after the last frame -> (cov [n_streams,6,6], sampled [n_streams] bool):
        Register's constant diagonal, Identity after a failed registration, or the sampled covariance when
        par.estimate_cov_by_sampling is set (odometrykeyframefuser.cpp:196, 203-208)."""
        cov = np.zeros((self.n_streams, 6, 6), np.float64)
        flag = np.zeros(self.n_streams, np.int32)
        self.ctx.check(self.ctx._lib.cfear_odometry_get_covariance(self._h, cov.ctypes.data, flag.ctypes.data))
        return cov, flag.astype(bool)

    def close(self):
        if getattr(self, "_h", None):
            if getattr(self.ctx, "h", None):              # the context may already be gone at interpreter exit: its
                self.ctx._lib.cfear_odometry_destroy(self._h)   # objects died with it, destroying them again would be a use after free
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- simple_graph.sgh (types.cpp:103-130): host-only, no GPU ---------------------------------------------------
def pose3d_from_xyt(xyt):
    """PoseEigToCeres of a planar pose -> (p [3], q [4] = x, y, z, w)."""
    p = L.Pose3d()
    L.lib().cfear_pose3d_from_xyt((C.c_double * 3)(*[float(v) for v in xyt]), C.byref(p))
    return np.array(p.p), np.array(p.q)


def _pose3d(v):
    p = L.Pose3d()
    if v is None:
        p.q[3] = 1.0
    elif len(v) == 3:
        L.lib().cfear_pose3d_from_xyt((C.c_double * 3)(*[float(x) for x in v]), C.byref(p))
    else:
        for i in range(3):
            p.p[i] = float(v[i])
        for i in range(4):
            p.q[i] = float(v[3 + i])
    return p


def _graph_cloud(c, keep):
    g = L.GraphCloud()
    if c is None:
        g.n = -1
        return g
    xyzi = c if not isinstance(c, dict) else c["xyzi"]
    a = np.ascontiguousarray(xyzi, np.float32).reshape(-1, 4)
    keep.append(a)
    g.xyzi, g.n = a.ctypes.data, a.shape[0]
    if isinstance(c, dict):
        g.seq, g.stamp = int(c.get("seq", 0)), int(c.get("stamp", 0))
        fid = c.get("frame_id", "").encode()
        keep.append(fid)
        g.frame_id = fid
    return g


def SaveSimpleGraph(path, nodes):
    """SaveSimpleGraph (types.cpp:103-113).  nodes: list of dicts with keys T, Tgt (xyt or p+q 7-vectors), has_Tgt, idx,
    stamp, motion (4x4), cloud_peaks, cloud_nopeaks (float [n, 4] or dict(xyzi, stamp, seq, frame_id) or None), cells
    (CELL_DTYPE array or None), radius, weight_intensity, input_is_nopeaks (default True), normal_input, constraints
    (list of dicts: id_begin, id_end, t_be, information [6, 6], type, quality {str: float}, info)."""
    keep = []
    arr = (L.GraphNode * len(nodes))()
    for i, nd in enumerate(nodes):
        g = arr[i]
        g.T, g.Tgt = _pose3d(nd.get("T")), _pose3d(nd.get("Tgt"))
        g.has_Tgt, g.idx, g.stamp = int(nd.get("has_Tgt", 0)), int(nd.get("idx", i)), int(nd.get("stamp", 0))
        m = np.asarray(nd.get("motion", np.eye(4)), np.float64).reshape(4, 4)
        for k, v in enumerate(m.T.reshape(-1)):                       # Affine3d::data() is column-major
            g.motion[k] = float(v)
        g.cloud_peaks = _graph_cloud(nd.get("cloud_peaks"), keep)
        g.cloud_nopeaks = _graph_cloud(nd.get("cloud_nopeaks"), keep)
        g.normal_input = _graph_cloud(nd.get("normal_input"), keep)
        cells = nd.get("cells")
        g.has_normal = int(cells is not None)
        g.input_is_nopeaks = int(nd.get("input_is_nopeaks", True))
        if cells is not None:
            ca = np.ascontiguousarray(cells, L.CELL_DTYPE)
            keep.append(ca)
            g.cells, g.n_cells = ca.ctypes.data, ca.shape[0]
        g.radius, g.weight_intensity = float(nd.get("radius", 0.0)), int(nd.get("weight_intensity", 0))
        cons = nd.get("constraints", [])
        carr = (L.GraphConstraint * max(len(cons), 1))()
        keep.append(carr)
        for j, c in enumerate(cons):
            carr[j].id_begin, carr[j].id_end = int(c["id_begin"]), int(c["id_end"])
            carr[j].t_be = _pose3d(c.get("t_be"))
            info = np.asarray(c.get("information", np.eye(6)), np.float64).reshape(-1)
            for k in range(36):
                carr[j].information[k] = float(info[k])
            carr[j].type = int(c.get("type", 0))
            q = c.get("quality", {})
            keys = sorted(q)                                          # std::map order
            ka = (C.c_char_p * max(len(keys), 1))(*[k.encode() for k in keys])
            va = (C.c_double * max(len(keys), 1))(*[float(q[k]) for k in keys])
            keep += [ka, va]
            carr[j].n_quality, carr[j].quality_keys, carr[j].quality_values = len(keys), ka, va
            ib = c.get("info", "").encode()
            keep.append(ib)
            carr[j].info = ib
        g.constraints, g.n_constraints = carr, len(cons)
    rc = L.lib().cfear_graph_save(os.fsencode(path), arr, len(nodes))
    if rc != L.OK:
        raise L.CfearError(rc, "cfear_graph_save(%s)" % path)


def LoadSimpleGraph(path):
    """LoadSimpleGraph (types.cpp:115-130) -> list of node dicts (the SaveSimpleGraph layout; poses as p+q 7-vectors and
    'T_xyt' as (x, y, theta); 'scan' is NOT built here -- feed node['cells'] to MapPointNormal(cells=...) on a GPU box)."""
    lib = L.lib()
    h = C.c_void_p()
    rc = lib.cfear_graph_load(os.fsencode(path), C.byref(h))
    if rc != L.OK:
        raise L.CfearError(rc, "cfear_graph_load(%s)" % path)
    try:
        out = []
        for i in range(lib.cfear_graph_size(h)):
            g = L.GraphNode()
            assert lib.cfear_graph_node_at(h, i, C.byref(g)) == L.OK

            def pose(p):
                return np.array(list(p.p) + list(p.q))

            def cloud(c):
                if c.n < 0:
                    return None
                a = np.ctypeslib.as_array(C.cast(c.xyzi, C.POINTER(C.c_float)), (c.n, 4)).copy() if c.n else np.zeros((0, 4), np.float32)
                return dict(xyzi=a, stamp=int(c.stamp), seq=int(c.seq), frame_id=(c.frame_id or b"").decode())
            xyt = (C.c_double * 3)()
            lib.cfear_pose3d_to_xyt(C.byref(g.T), xyt)
            nd = dict(T=pose(g.T), T_xyt=np.array(xyt), Tgt=pose(g.Tgt), has_Tgt=bool(g.has_Tgt), idx=int(g.idx), stamp=int(g.stamp),
                      motion=np.array(g.motion).reshape(4, 4).T.copy(), cloud_peaks=cloud(g.cloud_peaks),
                      cloud_nopeaks=cloud(g.cloud_nopeaks), normal_input=cloud(g.normal_input), input_is_nopeaks=bool(g.input_is_nopeaks),
                      cells=None, radius=float(g.radius), weight_intensity=bool(g.weight_intensity), constraints=[])
            if g.has_normal:
                nd["cells"] = (np.frombuffer(C.string_at(g.cells, g.n_cells * L.CELL_DTYPE.itemsize), L.CELL_DTYPE).copy()
                               if g.n_cells else np.zeros(0, L.CELL_DTYPE))
            for j in range(g.n_constraints):
                c = g.constraints[j]
                nd["constraints"].append(dict(
                    id_begin=int(c.id_begin), id_end=int(c.id_end), t_be=pose(c.t_be), information=np.array(c.information).reshape(6, 6),
                    type=int(c.type), quality={c.quality_keys[k].decode(): float(c.quality_values[k]) for k in range(c.n_quality)},
                    info=(c.info or b"").decode()))
            out.append(nd)
        return out
    finally:
        lib.cfear_graph_destroy(h)


def pose_graph_optimize(poses, ids, constraints, **par):
    """CeresLeastSquares::Solve (tbv_slam/src/tbv_slam/ceresoptimizer.cpp:13-62) over the nodes `poses` ([n, 7] = p, q(x, y, z, w),
    or [n, 3] planar (x, y, theta)) with node ids `ids` (ascending) and `constraints` (SaveSimpleGraph's constraint dicts:
    id_begin, id_end, t_be, information, type).  Keyword overrides: the cfear_pgo_params fields.  Host code, no GPU.
    Returns (poses [n, 7], summary dict)."""
    lib = L.lib()
    poses = np.asarray(poses, np.float64)
    n = poses.shape[0]
    arr = (L.Pose3d * n)()
    for i in range(n):
        arr[i] = _pose3d(poses[i])
    ida = np.ascontiguousarray(ids, np.uint64)
    carr = (L.GraphConstraint * max(len(constraints), 1))()
    for j, c in enumerate(constraints):
        carr[j].id_begin, carr[j].id_end = int(c["id_begin"]), int(c["id_end"])
        carr[j].t_be = _pose3d(c.get("t_be"))
        info = np.asarray(c.get("information", np.eye(6)), np.float64).reshape(-1)
        for k in range(36):
            carr[j].information[k] = float(info[k])
        carr[j].type = int(c.get("type", 0))
    p = L.PgoParams()
    lib.cfear_pgo_params_default(C.byref(p))
    for k, v in par.items():
        setattr(p, k, type(getattr(p, k))(v))
    s = L.PgoSummary()
    rc = lib.cfear_pgo_solve(arr, ida.ctypes.data, n, carr, len(constraints), C.byref(p), C.byref(s))
    if rc != L.OK:
        raise L.CfearError(rc, "cfear_pgo_solve")
    out = np.array([list(a.p) + list(a.q) for a in arr])
    return out, dict(initial_cost=s.initial_cost, final_cost=s.final_cost, iterations=s.iterations, usable=bool(s.usable),
                     num_residual_blocks=s.num_residual_blocks, linear_iterations=s.linear_iterations)
