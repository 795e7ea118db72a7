"""RT-2D block bounds, on the CPU: how many 2 x 2 blocks survive on bench.py's C1 worlds (to
calibrate against the device's statistics) and how many 4 x 4 blocks WOULD, with the same 5-bit
ceil(max / 1057) bytes and the finish's weights / slack.  numpy, float64 cells (statistics only).
   python tools/probes/rt2d_block_levels.py [worlds]"""
import math
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from cartographer_amd import synth  # noqa: E402

worlds = int(sys.argv[1]) if len(sys.argv) > 1 else 24
GRID = int(sys.argv[2]) if len(sys.argv) > 2 else 200
res, lin, ang, wt, wr = 0.05, 0.3, math.radians(7.0), 0.1, 0.1
nl = math.ceil(lin / res)
side = 2 * nl + 1
UNIT = 1057
out = []
for k in range(worlds):
    cells, lim, world = synth.make_submap(42 + k, GRID, GRID, res, 30, 1000, 5.0, 0.01)
    pose = world.free_pose(1234, 0.5)
    scan = bench.c1_scan(world, pose, 1000, 5.0, 7)
    init = (pose[0] + 0.12, pose[1] - 0.08, pose[2] + math.radians(3.0))
    n = len(scan)
    u = np.where((cells & 32767) > 0, 32767 - (cells & 32767).astype(np.int64), 0)   # [y][x] rows = grid y index
    ny, nx = u.shape
    pad = side + 8
    U = np.zeros((ny + 2 * pad, nx + 2 * pad), np.int64)
    U[pad:pad + ny, pad:pad + nx] = u
    r = np.hypot(scan[:, 0], scan[:, 1]).max()
    step = (1 - 1e-3) * math.acos(1 - res * res / (2 * r * r))
    na = math.ceil(ang / step)
    S = 2 * na + 1
    # pooled images for every alignment: m2[Y, X] = max U[Y..Y+1, X..X+1], m4 likewise
    def pooled(w):
        m = np.zeros_like(U)
        for dy in range(w):
            for dx in range(w):
                m = np.maximum(m, np.roll(np.roll(U, -dy, 0), -dx, 1))
        return (m + UNIT - 1) // UNIT
    m2, m4 = pooled(2), pooled(4)
    exact = np.zeros((S, side, side), np.int64)
    b2 = np.zeros((S, nl + 1, nl + 1), np.int64)
    nb4 = (side + 3) // 4
    b4 = np.zeros((S, nb4, nb4), np.int64)
    for s in range(S):
        th = init[2] + (s - na) * step
        c, sn = math.cos(th), math.sin(th)
        wx = init[0] + c * scan[:, 0] - sn * scan[:, 1]
        wy = init[1] + sn * scan[:, 0] + c * scan[:, 1]
        # cell index as the reference: x index from the map's y (limits.max - point) / res
        ix = np.rint((lim["max_y"] - wy) / res - 0.5).astype(np.int64)
        iy = np.rint((lim["max_x"] - wx) / res - 0.5).astype(np.int64)
        X0, Y0 = ix - nl + pad, iy - nl + pad              # window start; U[y][x] with y <- iy, x <- ix
        ok = (X0 >= 0) & (Y0 >= 0) & (X0 + side + 4 < U.shape[1]) & (Y0 + side + 4 < U.shape[0])
        X0, Y0 = X0[ok], Y0[ok]
        for dy in range(side):
            for dx in range(side):
                exact[s, dx, dy] = U[Y0 + dy, X0 + dx].sum()
        for j in range(nl + 1):
            for kk in range(nl + 1):
                b2[s, kk, j] = m2[Y0 + 2 * j, X0 + 2 * kk].sum()
        for j in range(nb4):
            for kk in range(nb4):
                b4[s, kk, j] = m4[Y0 + 4 * j, X0 + 4 * kk].sum()
    kscale = 0.8 / 32766.0
    slack = 1e-5
    def weight(s, dxi, dyi):
        t = math.hypot((dxi - nl) * res, (dyi - nl) * res) * wt + abs((s - na) * step) * wr
        return math.exp(-t * t)
    W = np.array([[[weight(s, dx, dy) for dy in range(side)] for dx in range(side)] for s in range(S)])
    score = (0.1 + kscale * exact / n) * W
    best = score.max()
    def blocks(b, w, nb):
        ub = np.zeros(b.shape)
        for s in range(S):
            for kk in range(nb):
                for j in range(nb):
                    wm = W[s, w * kk:min(side, w * kk + w), w * j:min(side, w * j + w)].max()
                    ub[s, kk, j] = (0.1 + kscale * UNIT * b[s, kk, j] / n + slack) * wm
        return ub
    ub2, ub4 = blocks(b2, 2, nl + 1), blocks(b4, 4, nb4)
    # the kernel's lower bound: the best candidate of the best block (by upper bound)
    def lb_of(ub, w):
        s, kk, j = np.unravel_index(np.argmax(ub), ub.shape)
        return score[s, w * kk:w * kk + w, w * j:w * j + w].max() * (1 - 2e-5)
    lb2, lb4 = lb_of(ub2, 2), lb_of(ub4, 4)
    # the lower bound from the K best blocks (by upper bound) instead of the best one
    order = np.argsort(-ub4.ravel())
    topk = []
    for K in (1, 2, 3, 4):
        lbk = 0.0
        for e in order[:K]:
            s_, kk_, j_ = np.unravel_index(e, ub4.shape)
            lbk = max(lbk, score[s_, 4 * kk_:4 * kk_ + 4, 4 * j_:4 * j_ + 4].max() * (1 - 2e-5))
        topk.append(int((ub4 >= lbk).sum()))
    print("   4x4 survivors with the lower bound from the best 1 / 2 / 3 / 4 blocks:", topk, flush=True)
    surv2 = int((ub2 >= lb2).sum())
    surv4 = int((ub4 >= lb4).sum())
    # with the TRUE best as the bound (a second pass after the survivors' own candidates are known)
    surv4_true = int((ub4 >= best * (1 - 2e-5)).sum())
    # two-level: 2 x 2 blocks inside surviving 4 x 4 blocks that reach the bound
    inside = 0
    for s in range(S):
        for kk in range(nl + 1):
            for j in range(nl + 1):
                if ub4[s, kk // 2, j // 2] >= lb4 and ub2[s, kk, j] >= lb4:
                    inside += 1
    out.append((S, surv2, surv4, surv4_true, inside, float(best)))
    print(f"world {k}: rotations {S}, best {best:.4f}; 2x2 blocks {ub2.size} surviving {surv2}; "
          f"4x4 blocks {ub4.size} surviving {surv4} (true best as bound: {surv4_true}); 2x2 blocks "
          f"inside surviving 4x4 that reach the bound: {inside}", flush=True)
a = np.array([o[:5] for o in out], float)
print("mean surviving 2x2 %.1f (median %.0f, max %.0f); 4x4 %.1f (median %.0f, max %.0f); 2x2 inside %.1f"
      % (a[:, 1].mean(), np.median(a[:, 1]), a[:, 1].max(), a[:, 2].mean(), np.median(a[:, 2]), a[:, 2].max(), a[:, 4].mean()))
