import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from nntoolkitcore_amd import capi
L = capi.load()
P = lambda a: a.ctypes.data_as(capi.fp)
r = np.random.default_rng(0)
u = lambda *s, sc=1.0: r.uniform(-sc, sc, s).astype(np.float32)
B, T, n_in, H = 64, 200, 128, 512
acts = L.LSTMActivationsCreateDefault(H)
cfg = L.LSTMConfigCreate(n_in, H, True, T, True, acts)
tc = capi.ConvTrainingConfig(B)
h = L.LSTMCreateForTraining(cfg, tc)
w = L.LSTMGetWeights(h).contents
for ptr, n, sc in ((w.W, n_in * 4 * H, n_in ** -0.5), (w.U, H * 4 * H, H ** -0.5)):
    a = u(n, sc=sc); C.memmove(ptr, a.ctypes.data, a.nbytes)
x, y, d = u(B, T, n_in), np.empty((B, T, H), np.float32), u(B, T, H)
g = L.LSTMGradientCreate(cfg, tc)
for _ in range(3):
    L.LSTMApplyTrainingBatch(h, P(x), P(y)); L.LSTMCalculateGradient(h, g, P(d))
L.nntk_hip_synchronize()
t0 = time.perf_counter()
for _ in range(5):
    L.LSTMApplyTrainingBatch(h, P(x), P(y))
t1 = time.perf_counter()
for _ in range(5):
    L.LSTMCalculateGradient(h, g, P(d))
t2 = time.perf_counter()
print("fwd %.2f ms  grad %.2f ms" % ((t1 - t0) / 5 * 1e3, (t2 - t1) / 5 * 1e3))
