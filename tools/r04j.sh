mkdir -p gpurun_out/r04j
python - <<'PY' > gpurun_out/r04j/stagger.log 2>&1
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from nntoolkitcore_amd import capi, layers as NL
B, T, K, N = 512, 996, 512, 1000
torch.cuda.set_device(0); capi.load(); NL.use_torch_stream()
r = np.random.default_rng(1)
tdd = NL.TimeDistributedDense(T, K, N)
tdd.set_weights(r.uniform(-0.04, 0.04, (K, N)).astype(np.float32), r.uniform(-0.1, 0.1, N).astype(np.float32))
x = torch.rand(B, T, K, device="cuda") - 0.5
x3 = NL.frag3_pack_device(x)
out = torch.empty(B, T, N, device="cuda")
def timeit(fn, n=10):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for rnd in range(3):
    for st in (0, 250, 500, 1000, 2000):
        capi.set_option("dense_stagger", st)
        print("stagger %4d ticks/phase: %.3f ms" % (st, timeit(lambda: NL.tdd_apply_device_frag3(tdd, x3, B, out=out))))
PY
cat gpurun_out/r04j/stagger.log
