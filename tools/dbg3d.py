import sys, math
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
from cartographer_amd import synth, scan_matching_3d as sm3
from oracle import pyoracle as orc
from test_oracle_reference_pins_3d import FAST3D_CLOUD, fast3d_fixture
from test_gpu_3d import REF_OPTIONS
g = fast3d_fixture(synth, [0.1, -0.2, 0.3], 0.05)
vox = g.voxels()
hist = np.zeros(10, np.float32)
print("voxels", len(vox), "grid_size", g.grid_size, flush=True)
gm = sm3.FastCorrelativeScanMatcher3D(0.05, vox, g.grid_size, 0.05, vox, hist, **REF_OPTIONS)
print("created", flush=True)
data = sm3.TrajectoryNodeData(FAST3D_CLOUD, FAST3D_CLOUD, hist)
got = gm.match(sm3.Rigid3d(), sm3.Rigid3d(), data, 0.1)
print(got, gm.last_stats)
