"""`python bench.py --gpus N` launches itself (VERDICT r01 "Next round" #1).

CPU part (runs anywhere): the launcher -- parent starts N fresh rank processes before touching any GPU,
rendezvous on 127.0.0.1, every rank reports in (`ranks_seen`), non-zero exit when a rank fails -- rehearsed
with `--dry-run` (gloo, no GPU work, no metric printed), bare and under torch.distributed.run.

GPU part (`-m gpu`): world_size 2 through the REAL per-rank device pipeline.  With >= 2 GPUs visible the ranks
use RCCL ("nccl"), one GPU each; on a 1-GPU box both ranks share the card and the (off-data-path) collectives
run over gloo -- the data path (per-shard HIP kernels through the C ABI) is the same code either way.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env=None, timeout=600):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run(args, cwd=ROOT, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)


def _json_lines(stdout):
    return [json.loads(l) for l in stdout.splitlines() if l.lstrip().startswith("{")]


@pytest.mark.parametrize("n", [2, 3])
def test_bench_launches_itself_dry_run(n):
    r = _run([sys.executable, BENCH, "--gpus", str(n), "--dry-run"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout                      # ONE line, from rank 0 only
    assert lines[0] == {"dry_run": True, "n_gpus": n, "ranks_seen": list(range(n)), "backend": "gloo"}
    assert "metric" not in lines[0] and "value" not in lines[0]      # a rehearsal never prints a number


def test_bench_dry_run_under_torch_distributed_run():
    # the driver's own invocation form for N > 1
    r = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
              "--master-addr", "127.0.0.1", "--master-port", "29641", BENCH, "--gpus", "2", "--dry-run"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1 and lines[0]["ranks_seen"] == [0, 1]


def test_bench_gpus_must_match_world_size():
    r = _run([sys.executable, BENCH, "--gpus", "4", "--dry-run"], env={"WORLD_SIZE": "2", "RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr


def test_bench_launcher_propagates_a_rank_failure():
    # without a GPU every non-dry rank dies on its first assert; the parent must report it, not hang or print a number
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a box without a GPU")
    r = _run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"], timeout=300)
    assert r.returncode != 0
    assert _json_lines(r.stdout) == []
    assert "exited with code" in r.stderr


def test_parent_never_imports_torch_before_forking_ranks():
    # the launcher branch must run before `import torch` (a parent that has initialised HIP must not spawn/exec)
    src = open(BENCH).read()
    main = src[src.index("def main():"):]
    assert main.index("self_launch(a)") < main.index("import torch")
    head = src[:src.index("def parse():")]
    assert "import torch" not in head


# ------------------------------------------------------------------ GPU ---

def _backend_env():
    import torch
    return {} if torch.cuda.device_count() >= 2 else {"NNTK_BENCH_BACKEND": "gloo"}


@pytest.mark.gpu
def test_bench_gpus_2_runs_the_device_pipeline_on_two_ranks(gpu):
    args = [sys.executable, BENCH, "--gpus", "2", "--batch-per-gpu", "64", "--frames", "40", "--steps", "2", "--warmup", "1",
            "--no-cpu-baseline"]
    r = _run(args, env=_backend_env())
    assert r.returncode == 0, r.stderr[-3000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout
    out = lines[0]
    assert out["n_gpus"] == 2 and out["config"]["ranks_seen"] == [0, 1] and out["scaling"] == "weak"
    assert out["config"]["global_batch"] == 128 and out["value"] > 0
    assert out["roofline"]["kernel"].startswith(("rec_", "lstm_rr_kernel", "gru_rr_kernel"))
    assert out["rank_ms_per_step"]["min"] > 0 and len(out["rank_ms_per_step"]["per_rank"]) == 2
    # the N=1 line keeps its shape
    r1 = _run([sys.executable, BENCH, "--batch-per-gpu", "64", "--frames", "40", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"])
    assert r1.returncode == 0, r1.stderr[-3000:]
    one = _json_lines(r1.stdout)[0]
    assert one["n_gpus"] == 1 and one["config"]["ranks_seen"] == [0]
    assert set(one) == set(out)


_WORKER = r"""
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, %(root)r)
import bench
from nntoolkitcore_amd import capi, layers as NL
from nntoolkitcore_amd.sharding import broadcast_weights, shard_range
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
ndev = torch.cuda.device_count()
backend = "nccl" if ndev >= world else "gloo"
local = rank %% ndev
torch.cuda.set_device(local)
if backend == "nccl":
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
else:
    dist.init_process_group("gloo", rank=rank, world_size=world)
L = capi.load(); assert L.nntk_hip_set_device(local) == 0; NL.use_torch_stream()
parts = bench.make_weights("stack", 11)
flat = bench.pack(parts) if rank == 0 else np.full_like(bench.pack(parts), np.nan)      # only rank 0 has them
w = bench.unpack(broadcast_weights(flat, torch, dist), parts)
B, frames = 6, 24
N = 240 + 160 * frames
audio = (0.1 * np.random.default_rng(3).standard_normal((B, N))).astype(np.float32)     # same on every rank
lo, hi = shard_range(B, world, rank)
spec = NL.Spectrogram(512, 400, 240, N); T = spec.out_shape[0]
conv = NL.Conv1d(257, 128, 5, 1, T); Tc = conv.out_shape[0]
bn = NL.BatchNorm(128, 1e-3, Tc); relu = NL.Activation("relu", Tc * 128, 1.0)
lstm = NL.LSTM(128, 512, True, Tc, v2=True); tdd = NL.TimeDistributedDense(Tc, 512, 1000)
conv.set_weights(w["conv_W"], w["conv_b"]); bn.set_weights(w["bn_gamma"], w["bn_beta"], w["bn_mean"], w["bn_var"])
lstm.set_weights(w["lstm_W"], w["lstm_U"], w["lstm_bi"], w["lstm_bh"]); tdd.set_weights(w["tdd_W"], w["tdd_b"])
def run(x):
    y = tdd.apply_device(lstm.apply_device(conv.apply_device(spec.apply_device(torch.from_numpy(x).cuda()), bn=bn, act=relu)))
    torch.cuda.synchronize(); assert L.nntk_hip_synchronize() == 0
    return y.cpu().numpy()
mine = run(audio[lo:hi])                    # this rank's shard on this rank's GPU: no data-path collective
np.save(os.path.join(%(out)r, "shard%%d.npy" %% rank), mine)
if rank == 0:
    np.save(os.path.join(%(out)r, "whole.npy"), run(audio))
    np.save(os.path.join(%(out)r, "audio.npy"), audio)
    np.save(os.path.join(%(out)r, "flat.npy"), bench.pack(w))
dist.barrier(device_ids=[local]) if backend == "nccl" else dist.barrier()
dist.destroy_process_group()
"""


@pytest.mark.gpu
def test_two_rank_device_shards_equal_single_process_and_oracle(gpu, tmp_path):
    import bench
    import oracle as O
    from nntoolkitcore_amd.sharding import shard_range
    script = tmp_path / "worker.py"
    script.write_text(_WORKER % {"root": ROOT, "out": str(tmp_path)})
    env = dict(os.environ, WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29653", HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(o[-2000:] for o in outs)
    whole = np.load(tmp_path / "whole.npy")
    shards = np.concatenate([np.load(tmp_path / ("shard%d.npy" % r)) for r in range(2)], axis=0)
    assert shard_range(6, 2, 0) == (0, 3)
    assert np.array_equal(shards, whole)                       # bit-identical to the one-process run
    # and both equal the oracle on the broadcast weights
    audio = np.load(tmp_path / "audio.npy")
    parts = bench.make_weights("stack", 11)
    flat = np.load(tmp_path / "flat.npy")
    assert np.array_equal(flat, bench.pack(parts))             # rank 0's weights arrived intact
    w = parts
    s = O.spectrogram(audio, O.window("hann", 400), 512, 240)
    c = O.activation(O.ACT_RELU, O.batch_norm(O.conv1d(s, w["conv_W"], w["conv_b"], 1), w["bn_gamma"], w["bn_beta"],
                                              w["bn_mean"], w["bn_var"], 1e-3))
    ref = O.time_distributed_dense(O.lstm(c, w["lstm_W"], w["lstm_U"], w["lstm_bi"], w["lstm_bh"], v2=True), w["tdd_W"], w["tdd_b"])
    err = float(np.abs(shards - ref).max())
    print("two-rank device shards vs oracle: max abs err %.3e" % err)
    assert err < 1e-4
