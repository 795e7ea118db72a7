"""CPU model of the RT-2D integer bulk pass (cartographer_amd/csrc/rt_2d.hip, Rt2DBulkKernel
/ Rt2DExactKernel): the quantised integer sums bound every candidate's reference score, and
the finalist rule (weighted upper bound >= best weighted lower bound) always keeps the
candidate the reference returns.  The oracle supplies the discretised scans and the
reference's own per-candidate scores; numpy restates the device arithmetic of the bounds."""
import math

import numpy as np
import pytest

Q_SHIFT = 5
SLACK = 1e-4


def _model(oracle, cells, lim, init, scan, lin, ang, wt, wr):
    res = lim["resolution"]
    ny, nx = cells.shape
    # SearchParameters on the cloud pre-rotated by the initial yaw (real_time_..._2d.cc:123-130)
    c, s = math.cos(init[2]), math.sin(init[2])
    pre = scan.copy()
    pre[:, 0] = np.float32(c) * scan[:, 0] - np.float32(s) * scan[:, 1]
    pre[:, 1] = np.float32(s) * scan[:, 0] + np.float32(c) * scan[:, 1]
    sp = oracle.search_parameters(lin, ang, pre, res)
    na, step, nl = (sp["num_angular_perturbations"], sp["angular_perturbation_step_size"],
                    sp["num_linear_perturbations"])
    cellsxy = oracle.discretize_scans(scan, init[2], na, step, res, lim["max_x"], lim["max_y"],
                                      nx, ny, init[0], init[1])          # [S][N][2]
    S, N = cellsxy.shape[:2]
    side = 2 * nl + 1
    v = (cells & 0x7fff).astype(np.int64)
    u = np.where(v == 0, 0, 32767 - v)
    q = u >> Q_SHIFT
    pad = 2 * nl + 2
    qp = np.zeros((ny + 2 * pad, nx + 2 * pad), np.int64)
    qp[pad:pad + ny, pad:pad + nx] = q
    k_scale = float((np.float32(0.9) - (np.float32(1) - np.float32(0.9))) / np.float32(32766.0))
    lo = np.empty((S, side, side))
    for si in range(S):
        ix = np.clip(cellsxy[si, :, 0], -(nl + 1), nx + nl) + pad
        iy = np.clip(cellsxy[si, :, 1], -(nl + 1), ny + nl) + pad
        for a, dx in enumerate(range(-nl, nl + 1)):
            for b, dy in enumerate(range(-nl, nl + 1)):
                Q = qp[iy + dy, ix + dx].sum()
                lo[si, a, b] = 0.1 + k_scale * (Q * (1 << Q_SHIFT)) / N
    hi = lo + k_scale * ((1 << Q_SHIFT) - 1)
    dxs = np.arange(-nl, nl + 1)
    w = np.empty((S, side, side), np.float32)
    for si in range(S):
        theta = np.float32((si - na) * step)
        cx = (-dxs[None, :] * np.float32(res)).astype(np.float32)      # -dy * res
        cy = (-dxs[:, None] * np.float32(res)).astype(np.float32)      # -dx * res
        t = np.sqrt(cx * cx + cy * cy) * np.float32(wt) + np.abs(theta) * np.float32(wr)
        w[si] = np.exp(-(t * t)).astype(np.float32)
    lb = (lo - SLACK).astype(np.float32) * w * np.float32(1 - 1e-5)
    ub = (hi + SLACK).astype(np.float32) * w * np.float32(1 + 1e-5)
    return lo, hi, lb, ub, (S, side)


@pytest.mark.parametrize("seed,size,beams,lin,ang", [
    (42, 200, 1000, 0.3, 7.0), (7, 200, 400, 0.3, 7.0), (11, 120, 300, 0.2, 4.0),
    (3, 160, 250, 0.15, 10.0)])
def test_finalists_contain_reference_winner(oracle, synth, seed, size, beams, lin, ang):
    cells, lim, world = synth.make_submap(seed, size, size, 0.05, 20, 600, 5.0, 0.01)
    pose = world.free_pose(seed + 100, 0.5)
    scan = world.scan(pose, beams, 5.0, 0.01, 7)
    init = [pose[0] + 0.12, pose[1] - 0.08, pose[2] + math.radians(3.0)]
    ref = oracle.rt2d_match(cells, lim["resolution"], lim["max_x"], lim["max_y"], init, scan, lin,
                            math.radians(ang), 0.1, 0.1, want_scores=True)
    lo, hi, lb, ub, (S, side) = _model(oracle, cells, lim, init, scan, lin, math.radians(ang),
                                       0.1, 0.1)
    weighted = ref["scores"].reshape(S, side, side)      # generation order: scan, x, y
    # The reference's weighted score lies inside the weighted bounds of every candidate.
    assert np.all(weighted >= lb - 1e-7) and np.all(weighted <= ub + 1e-7)
    finalists = ub >= lb.max()
    best = np.unravel_index(np.argmax(weighted), weighted.shape)      # first maximum
    assert finalists[best]
    assert np.float32(weighted[best]) == np.float32(ref["score"])
    # The bulk pass leaves a handful of candidates for the exact f32 chain.
    assert finalists.sum() <= 64, finalists.sum()


def test_flat_landscape_keeps_everything(oracle, synth):
    """An all-unknown grid: every candidate scores 0.1 x weight; the finalist rule must keep
    (at least) the centre candidate, which the reference returns."""
    cells = np.zeros((60, 60), np.uint16)
    lim = dict(resolution=0.05, max_x=1.5, max_y=1.5, num_x_cells=60, num_y_cells=60)
    rng = np.random.default_rng(0)
    scan = np.zeros((50, 3), np.float32)
    scan[:, :2] = rng.uniform(-1.0, 1.0, (50, 2))
    init = [0.0, 0.0, 0.1]
    ref = oracle.rt2d_match(cells, 0.05, 1.5, 1.5, init, scan, 0.1, math.radians(2.0), 0.1, 0.1,
                            want_scores=True)
    lo, hi, lb, ub, (S, side) = _model(oracle, cells, lim, init, scan, 0.1, math.radians(2.0),
                                       0.1, 0.1)
    weighted = ref["scores"].reshape(S, side, side)
    best = np.unravel_index(np.argmax(weighted), weighted.shape)
    assert (ub >= lb.max())[best]
