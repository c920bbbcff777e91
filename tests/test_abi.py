"""CPU: the C-ABI library loads and exports every symbol include/cfear_hip.h declares; struct
layouts of the ctypes binding match the header; without a GPU the product fails loudly."""
import ctypes as C
import os
import re

import pytest

from tbv_slam_public_amd import _lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_so_exists_and_exports_header_symbols():
    lib = L.lib()
    hdr = open(os.path.join(ROOT, "include", "cfear_hip.h")).read()
    declared = set(re.findall(r"\b(cfear_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"cfear_ctx", "cfear_scan", "cfear_cost", "cfear_odometry"}
    assert declared == set(L.EXPORTS), declared ^ set(L.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.cfear_abi_version() == 1


def test_struct_sizes():
    assert C.sizeof(L.Cell) == 104
    assert C.sizeof(L.RegResult) == 72
    assert C.sizeof(L.FrameInfo) == 56
    assert C.sizeof(L.PolarDesc) == 24
    assert C.sizeof(L.RegParams) == 72
    assert C.sizeof(L.CovSamplingParams) == 32
    assert C.sizeof(L.CoralParams) == 16
    assert C.sizeof(L.ScParams) == 48 and C.sizeof(L.ScCloud) == 16
    assert C.sizeof(L.CoralJob) == 96
    assert C.sizeof(L.CoralResult) == 40 == L.CORAL_RESULT_DTYPE.itemsize
    assert C.sizeof(L.VerifyParams) == 160 and C.sizeof(L.VerifyJob) == 112
    assert C.sizeof(L.VerifyResult) == 480 == L.VERIFY_RESULT_DTYPE.itemsize
    assert C.sizeof(L.ScManagerParams) == 88 and L.SC_CANDIDATE_DTYPE.itemsize == 64


def test_defaults_follow_reference():
    lib = L.lib()
    p = L.RegParams()
    lib.cfear_reg_params_default(C.byref(p))
    # n_scan_normal.h:72-75, registration.h:117-122
    assert (p.cost, p.loss, p.loss_limit, p.weight_opt) == (L.P2L, 1, 0.1, 0)
    assert (p.max_itr_association, p.max_itr_solver, p.min_itr, p.radius) == (8, 20, 3, 2.0)
    o = L.OdometryParams()
    lib.cfear_odometry_params_default(C.byref(o))
    # CFEAR-3 preset
    assert (o.kstrong.k_strongest, o.kstrong.z_min, o.res, o.submap_scan_size) == (40, 60.0, 3.0, 4)
    # cost-sampling defaults (odometrykeyframefuser.h:104-110)
    assert o.estimate_cov_by_sampling == 0
    assert (o.cov_sampling.xy_range, o.cov_sampling.yaw_range, o.cov_sampling.samples_per_axis,
            o.cov_sampling.covariance_scaler) == (0.4, 0.0043625, 3, 4.0)
    assert (o.reg.cost, o.reg.weight_opt, o.weight_intensity) == (L.P2P, 4, 1)
    assert lib.cfear_status_string(-4) == b"too few residuals"


def test_no_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    h = C.c_void_p()
    rc = L.lib().cfear_ctx_create(0, None, C.byref(h))
    assert rc == L.ERR_NO_DEVICE and not h.value
    from tbv_slam_public_amd import api
    with pytest.raises(L.CfearError):
        api.Context(0)


def test_presets_follow_the_reference_launch_files():
    """cfear_odometry_params_preset is pure host code: the shipped CFEAR-1/2/3/3-s10 configurations
    (cfear_radarodometry/launch/oxford/eval/params/baseline/oxford_cfear-*:13-26) on the four sensor setups
    (tbv_slam/script/*/run_tbv_simple.sh)."""
    from tbv_slam_public_amd import api
    want = {"CFEAR-1": (L.COST["P2L"], 1, 3.5, 12, L.LOSS["Huber"], 0, 1.0),     # EVALUATION_regularization="1" (:23)
            "CFEAR-2": (L.COST["P2L"], 3, 3.5, 12, L.LOSS["Huber"], 0, 1.0),
            "CFEAR-3": (L.COST["P2P"], 4, 3.0, 40, L.LOSS["Huber"], 1, 1.0),
            "CFEAR-3-s10": (L.COST["P2P"], 10, 3.0, 40, L.LOSS["Cauchy"], 1, 0.1)}
    for name, (cost, s, res, k, loss, wi, regu) in want.items():
        p = api.odometry_preset(name)
        assert (p.reg.cost, p.submap_scan_size, p.res, p.kstrong.k_strongest, p.reg.loss, p.weight_intensity) == (cost, s, res, k, loss, wi)
        assert p.reg.regularization == regu and p.reg.loss_limit == 0.1 and p.reg.weight_opt == 4 and p.kstrong.z_min == 60.0
        assert p.min_keyframe_dist == 1.5 and p.min_keyframe_rot_deg == 5.0 and p.compensate == 1 and p.use_guess == 1
    for ds, (rr, ccw, rot) in {"oxford": (0.0438, 0, 0), "Mulran": (0.0595238, 1, 1), "kvarntorp": (0.175, 1, 1), "Volvo": (0.175, 1, 1)}.items():
        p = api.odometry_preset("CFEAR-3", ds)
        assert abs(p.kstrong.range_res - rr) < 1e-7 and p.cacfar.range_res == p.kstrong.range_res
        assert (p.radar_ccw, p.rotate_ccw) == (ccw, rot)
    same = api.odometry_params(reg_regularization=1.0)     # the struct default keeps OdometryKeyframeFuser::Parameters' 0.0
    assert bytes(api.odometry_preset("CFEAR-3", "oxford")) == bytes(same)
    assert api.odometry_preset("CFEAR-2", submap_scan_size=2).submap_scan_size == 2
    p = L.OdometryParams()
    assert L.lib().cfear_odometry_params_preset(C.byref(p), 9, 0) == L.ERR_INVALID_ARGUMENT
    assert L.lib().cfear_odometry_params_preset(C.byref(p), 1, 7) == L.ERR_INVALID_ARGUMENT
