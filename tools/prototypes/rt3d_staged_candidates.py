"""Statistics only: how many of C4's candidates that survive the 2x2x2 group bound are still
alive after a fraction of the points (UB_c = U_g - U_g(Q) + s_c(Q))."""
import os, sys, math, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cartographer_amd import synth
from scipy.ndimage import maximum_filter

t0 = time.time()
grid, world = synth.make_submap_3d(42, 0.1, (15.0, 15.0, 7.5), 8, 32, 512)
vox = grid.voxels()
print("voxel dtype", vox.dtype, len(vox))
pos = world.free_position(77, 0.5)
cloud = world.scan(pos, 0.3, 64, 1024, seed=9).astype(np.float64)
print("cloud", cloud.shape)
SUB = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rng = np.random.default_rng(0)
perm = rng.permutation(len(cloud))
pts = cloud[perm[: len(cloud) // SUB]]
N = len(pts)
res = 0.1
names = vox.dtype.names
xs, ys, zs, val = (vox[names[0]].astype(np.int64), vox[names[1]].astype(np.int64),
                   vox[names[2]].astype(np.int64), vox[names[3]].astype(np.int64))
pad = 16
lo = np.array([xs.min(), ys.min(), zs.min()]) - pad
dims = np.array([xs.max(), ys.max(), zs.max()]) - lo + pad + 1
B = np.zeros(dims, np.uint8)
u = np.maximum(val & 32767, 1) - 1
B[xs - lo[0], ys - lo[1], zs - lo[2]] = (u >> 7).astype(np.uint8)
D = maximum_filter(B, size=3, mode="constant")
print("brick", B.shape, "nonzero", (B > 0).mean())

# search space (reference: real_time_correlative_scan_matcher_3d.cc:116-160)
L = int(math.ceil(0.5 / res))
maxr = np.linalg.norm(cloud, axis=1).max()
step = math.acos(1 - res * res / (2 * maxr * maxr))
aw = int(math.ceil(math.radians(2.0) / step))
print("L", L, "angular window", aw, "step deg", math.degrees(step))
def quat_from_aa(v):
    a = np.linalg.norm(v)
    if a < 1e-12: return np.array([1.0, 0, 0, 0])
    return np.concatenate([[math.cos(a / 2)], math.sin(a / 2) * v / a])
def qmul(a, b):
    w1, x1, y1, z1 = a; w2, x2, y2, z2 = b
    return np.array([w1*w2-x1*x2-y1*y2-z1*z2, w1*x2+x1*w2+y1*z2-z1*y2, w1*y2-x1*z2+y1*w2+z1*x2, w1*z2+x1*y2-y1*x2+z1*w2])
def rotm(q):
    w, x, y, z = q
    return np.array([[1-2*(y*y+z*z), 2*(x*y-z*w), 2*(x*z+y*w)], [2*(x*y+z*w), 1-2*(x*x+z*z), 2*(y*z-x*w)], [2*(x*z-y*w), 2*(y*z+x*w), 1-2*(x*x+y*y)]])
c, s = math.cos(0.31 / 2), math.sin(0.31 / 2)
q0 = np.array([c, 0, 0, s]); t0v = pos + np.array([0.07, -0.04, 0.02])
R0 = rotm(q0)
rots, angs = [], []
for z in range(-aw, aw + 1):
    for y in range(-aw, aw + 1):
        for x in range(-aw, aw + 1):
            v = np.array([x, y, z]) * step
            rots.append(rotm(quat_from_aa(v))); angs.append(np.linalg.norm(v))
rots = np.array(rots); angs = np.array(angs); R = len(rots)
side = 2 * L + 1
tr = np.array([[x, y, z] for z in range(-L, L + 1) for y in range(-L, L + 1) for x in range(-L, L + 1)]) * res
T = len(tr)
gpa = (side + 1) // 2
groups, gmembers, gdist = [], [], []
for gz in range(gpa):
    for gy in range(gpa):
        for gx in range(gpa):
            mem = [((2*gz+c_)*side + (2*gy+b_))*side + (2*gx+a_) for c_ in range(min(2, side-2*gz)) for b_ in range(min(2, side-2*gy)) for a_ in range(min(2, side-2*gx))]
            gmembers.append(mem)
            groups.append(tr[mem].mean(axis=0)); gdist.append(np.linalg.norm(tr[mem], axis=1).min())
groups = np.array(groups); G = len(groups); gdist = np.array(gdist)
print("R", R, "T", T, "G", G, "N", N)
wt = wr = 0.1
kscale = 0.8 / 32766 * 128

# point subsets: quarter / half by random split
quarter = np.arange(N) < N // 4
half = np.arange(N) < N // 2
eighth = np.arange(N) < N // 8

def lookup(brick, P):   # P (..., 3) metres in map frame
    idx = np.rint(P / res).astype(np.int64) - lo
    np.clip(idx, 0, dims - 1, out=idx)
    return brick[idx[..., 0], idx[..., 1], idx[..., 2]]

Ug = np.zeros((R, G)); Ug8 = np.zeros((R, G)); Ug4 = np.zeros((R, G)); Ug2 = np.zeros((R, G))
rp_all = []
gt = groups @ R0.T + t0v          # (G,3)
for r in range(R):
    rp = pts @ (R0 @ rots[r]).T    # (N,3) rotated into map frame (without translation)
    v = lookup(D, rp[None, :, :] + gt[:, None, :]).astype(np.int64)   # (G,N)
    Ug[r] = v.sum(1); Ug8[r] = v[:, eighth].sum(1); Ug4[r] = v[:, quarter].sum(1); Ug2[r] = v[:, half].sum(1)
    if r % 200 == 0: print("group pass", r, time.time() - t0, flush=True)
wgt_g = np.exp(-((gdist[None, :] * wt + angs[:, None] * wr) ** 2))
UBg = (0.1 + kscale * (Ug + 0.99 * N) / N) * wgt_g
# pass 1: members of groups within 3 % of the best UB
def exact_members(sel):
    out = {}
    tt = tr @ R0.T + t0v
    for r in np.unique(sel[0]):
        gs = sel[1][sel[0] == r]
        mem = np.concatenate([gmembers[g] for g in gs])
        grp = np.concatenate([[g] * len(gmembers[g]) for g in gs])
        rp = pts @ (R0 @ rots[r]).T
        v = lookup(B, rp[None, :, :] + tt[mem][:, None, :]).astype(np.int64)
        out[r] = (mem, grp, v.sum(1), v[:, eighth].sum(1), v[:, quarter].sum(1), v[:, half].sum(1))
    return out
top = np.nonzero(UBg >= UBg.max() * 0.97)
e1 = exact_members(top)
wdist = np.linalg.norm(tr, axis=1)
b = 0.0
for r, (mem, grp, s, s8, s4, s2) in e1.items():
    lbv = (0.1 + kscale * s / N) * np.exp(-((wdist[mem] * wt + angs[r] * wr) ** 2))
    b = max(b, lbv.max())
print("pass 1: groups", len(top[0]), "lower bound", b, "best group UB", UBg.max())
surv = np.nonzero(UBg >= b)
print("pass 2: surviving groups", len(surv[0]), "of", R * G, "-> candidates ~", sum(len(gmembers[g]) for g in surv[1]))
e2 = exact_members(surv)
tot = alive8 = alive4 = alive2 = fin = 0
for r, (mem, grp, s, s8, s4, s2) in e2.items():
    w = np.exp(-((wdist[mem] * wt + angs[r] * wr) ** 2))
    def ub(sq, uq, frac):
        q = Ug[r, grp] - uq[r, grp] + sq
        return (0.1 + kscale * (q + 0.99 * N) / N) * w
    tot += len(mem)
    a8 = ub(s8, Ug8, 8) >= b
    a4 = a8 & (ub(s4, Ug4, 4) >= b)
    a2 = a4 & (ub(s2, Ug2, 2) >= b)
    alive8 += a8.sum(); alive4 += a4.sum(); alive2 += a2.sum()
    fin += ((0.1 + kscale * (s + 0.99 * N) / N) * w >= b).sum()
print(f"survivors {tot}: alive after 1/8 {alive8} ({alive8/tot:.3f}), after 1/4 {alive4} ({alive4/tot:.3f}), after 1/2 {alive2} ({alive2/tot:.3f}), finalists {fin}")
work = 1/8 + alive8/tot*(1/8) + alive4/tot*(1/4) + alive2/tot*(1/2)
print("relative work of a 1/8,1/4,1/2,1 schedule:", work, " 1/4,1:", 0.25 + (alive4 if False else 0))
a4only = 0
for r, (mem, grp, s, s8, s4, s2) in e2.items():
    w = np.exp(-((wdist[mem] * wt + angs[r] * wr) ** 2))
    q = Ug[r, grp] - Ug4[r, grp] + s4
    a4only += ((0.1 + kscale * (q + 0.99 * N) / N) * w >= b).sum()
print("alive after 1/4 alone", a4only, a4only / tot, "-> work of a 1/4,1 schedule", 0.25 + 0.75 * a4only / tot)
print("time", time.time() - t0)
