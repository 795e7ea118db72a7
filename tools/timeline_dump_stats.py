"""Raw timeline dump (CMX_TIMELINE_DUMP) of the RT-2D bound kernel: do slow items go with the
match (data) or with the place they ran (XCC, SE, CU)?"""
import sys
import collections
import numpy as np
rows = []
name = None
for line in open(sys.argv[1]):
    if line.startswith("#"):
        name = line.split()[1]
        continue
    if name != "Rt2DBoundKernel":
        continue
    v = [int(x) for x in line.split()]
    rows.append(v)
a = np.array(rows, dtype=np.int64)
blk, st = a[:, 0], a[:, 1:]
dur = (st[:, 2] - st[:, 0]) * 0.01            # start -> end of phase A (us)
stage = (st[:, 1] - st[:, 0]) * 0.01
item = st[:, 14]
hw = st[:, 15] & 0xffffffff
xcc = (st[:, 15] >> 32) & 0xf
cu = (hw >> 8) & 0xf
sh = (hw >> 12) & 0x1
se = (hw >> 13) & 0x7
wg = blk // 4
idx = blk % 4
print("items", len(a), "phase A+staging: median %.1f p90 %.1f max %.1f" % (np.median(dur), np.percentile(dur, 90), dur.max()))
print("by item index of the workgroup:", [(i, int((idx == i).sum()), round(float(np.median(dur[idx == i])), 1)) for i in range(4) if (idx == i).any()])
for key, lab in ((xcc, "xcc"), (se, "se"), (cu, "cu")):
    print(lab, [(int(k), int((key == k).sum()), round(float(np.median(dur[key == k])), 1), round(float(dur[key == k].max()), 1)) for k in np.unique(key)])
# same place = (xcc, se, sh, cu): how many items ran there, their durations
place = xcc * 100000 + se * 1000 + sh * 100 + cu
cnt = collections.Counter(place.tolist())
print("places", len(cnt), "items per place:", collections.Counter(cnt.values()))
slow = dur > 1.5 * np.median(dur)
print("slow items", int(slow.sum()), "on places", len(set(place[slow].tolist())))
sp = collections.Counter(place[slow].tolist())
print("slow per place:", collections.Counter(sp.values()))
# per world (item % 256): same scans/grids recur four times in the 1024 batch
w = item % 256
bw = collections.defaultdict(list)
for k, d in zip(w.tolist(), dur.tolist()):
    bw[k].append(d)
spread = [max(v) - min(v) for v in bw.values() if len(v) > 1]
print("same world, spread of durations: median %.1f max %.1f" % (np.median(spread), max(spread)))
wmean = np.array([np.mean(v) for v in bw.values()])
print("per-world mean: min %.1f median %.1f max %.1f" % (wmean.min(), np.median(wmean), wmean.max()))
