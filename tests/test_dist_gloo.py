"""N>1 path on CPU: world_size-2 gloo processes shard a batch, compute their
slice of the loss with the GLOBAL normalisation and all-reduce -- the sum must
equal the single-process loss of the whole batch.  The per-shard arithmetic
is done by the CPU oracle here (this is a test; the product path is the GPU)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _inputs(B, H, W):
    rng = np.random.default_rng(42)
    im1 = rng.uniform(size=(B, H, W, 3)).astype(np.float32)
    im2 = rng.uniform(size=(B, H, W, 3)).astype(np.float32)
    mask = (rng.uniform(size=(B, H, W, 1)) > 0.2).astype(np.float32)
    return im1, im2, mask


def _worker(rank, world, port, B, H, W, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import oracle
    from unflow_amd import dist as udist
    r, w = udist.init_process_group("gloo")
    assert (r, w) == (rank, world)
    im1, im2, mask = _inputs(B, H, W)
    s, e = udist.shard_range(B, rank, world)
    # per-shard census loss, normalised by the GLOBAL pixel count:
    # loss_local(B_local*H*W) * B_local / B
    local = oracle.ternary_loss(im1[s:e], im2[s:e], mask[s:e], 2) * (e - s) / B if e > s else 0.0
    losses = udist.reduce_losses({"ternary": torch.tensor(local, dtype=torch.float64),
                                  "count": torch.tensor(float(e - s), dtype=torch.float64)})
    buf = udist.FlatGradBuffer(1000, "cpu")
    buf.buffer += rank + 1
    buf.all_reduce()
    if rank == 0:
        out.put((float(losses["ternary"]), float(losses["count"]), float(buf.buffer[0])))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("B", [4, 3])   # 3: ragged split over 2 ranks
def test_sharded_loss_equals_global(B):
    import oracle
    oracle.build()
    H, W, world = 12, 16, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + B
    procs = [ctx.Process(target=_worker, args=(r, world, port, B, H, W, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    im1, im2, mask = _inputs(B, H, W)
    want = oracle.ternary_loss(im1, im2, mask, 2)
    np.testing.assert_allclose(got[0], want, rtol=1e-6)
    assert got[1] == B and got[2] == 3.0   # buffer: (0+1) + (1+1)


def test_shard_range_partitions():
    from unflow_amd.dist import shard_range
    for B in (0, 1, 7, 32, 256):
        for world in (1, 2, 3, 8):
            spans = [shard_range(B, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == B
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [e - s for s, e in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(8, 2, 2)


def _run_bench(*flags, env_extra=None):
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(env_extra or {})
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(flags), capture_output=True,
                         text=True, timeout=300, env=env)
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    return out, [json.loads(ln) for ln in lines]


def test_bench_gpus_flag_spawns_ranks():
    """`bench.py --gpus 2` with no launcher environment starts two fresh ranks itself (the parent
    never touches the GPU), they rendezvous and rank 0 prints ONE line with n_gpus = 2.  --dry
    skips the GPU work so the launch path runs here (gloo)."""
    out, lines = _run_bench("--gpus", "2", "--backend", "gloo", "--dry")
    assert out.returncode == 0, out.stderr
    assert len(lines) == 1 and lines[0]["n_gpus"] == 2 and lines[0]["gpus_requested"] == 2
    assert lines[0]["allreduce_check"] == 3.0
    out3, lines3 = _run_bench("--gpus", "3", "--backend", "gloo", "--dry")
    assert out3.returncode == 0 and lines3[0]["n_gpus"] == 3


def test_bench_eight_rank_launch_path_dry():
    """The driver's scaling run ends at `--gpus 8`: the 8-rank launch, rendezvous on 127.0.0.1 and all-reduce are
    rehearsed here (gloo, --dry: no GPU work), both self-spawned and under torch.distributed.run's environment."""
    out, lines = _run_bench("--gpus", "8", "--backend", "gloo", "--dry")
    assert out.returncode == 0, out.stderr[-2000:]
    assert len(lines) == 1 and lines[0]["n_gpus"] == 8 and lines[0]["allreduce_check"] == 36.0      # 1 + 2 + ... + 8


def test_bench_under_a_launcher_takes_ranks_from_env():
    """Under torch.distributed.run (WORLD_SIZE set) bench.py does not spawn: one process, one rank."""
    out, lines = _run_bench("--gpus", "2", "--dry", env_extra=dict(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"))
    assert out.returncode == 0 and len(lines) == 1 and lines[0]["n_gpus"] == 1 and lines[0]["gpus_requested"] == 2


def test_bench_launcher_reports_a_failing_rank():
    """A rank that dies takes the launch down with a non-zero status (no hang, no silent success)."""
    out, lines = _run_bench("--gpus", "2", "--backend", "gloo", "--dry", env_extra={"UNFLOW_BENCH_FAIL_RANK": "1"})
    assert out.returncode != 0 and not lines


@pytest.mark.gpu
def test_bench_two_ranks_on_the_gpu_rehearsal():
    """The N = 2 path of bench.py with real GPU work: two ranks (here both on cuda:0 -- a rehearsal, flagged
    as such in the line -- over gloo, because RCCL refuses two ranks on one device), the barrier /
    max-over-ranks timing, the all-reduce of loss and gradient buffer, one line from rank 0 with the
    whole-job value."""
    out, lines = _run_bench("--gpus", "2", "--backend", "gloo", "--steps", "2", "--warmup", "1", "--batch", "4",
                            "--no-cpu-baseline", "--no-extras", env_extra={"UNFLOW_BENCH_SHARE_GPU": "1"})
    assert out.returncode == 0, out.stderr[-2000:]
    assert len(lines) == 1
    line = lines[0]
    assert line["n_gpus"] == 2 and line["config"]["world_size"] == 2 and line["config"]["global_batch"] == 8
    assert line["config"]["rehearsal_all_ranks_on_one_gpu"] is True
    assert line["value"] > 0 and abs(line["value"] - 8 / (line["ms_per_step"] * 1e-3)) <= 1e-6 * line["value"]
    assert "cpu_baseline" not in line and line["roofline"]["bound"] == "mfma"
    # the gradient buffer's all-reduce starts behind the second correlation_grad (the earliest point at which a
    # trainer has those gradients), overlaps the loss leg only and is waited for at the end of the step
    order = line["config"]["step_order"]
    assert order == ["corr_fwd", "corr_fwd", "corr_bwd", "corr_bwd", "allreduce_start", "loss_leg", "loss_allreduce",
                     "allreduce_wait"], order
    assert line["ms_per_step_serialized"] > 0 and line["allreduce_ms"] > 0


def test_launcher_counts_gpus_without_hip(monkeypatch):
    """The launcher parent must not initialise HIP before it forks the ranks: it counts the GPUs from the
    visibility variables or the KFD topology, never through torch.cuda."""
    sys.path.insert(0, ROOT)
    import bench
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,2,5")
    assert bench.visible_gpu_count() == 3
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "")
    assert bench.visible_gpu_count() == 0
    monkeypatch.delenv("HIP_VISIBLE_DEVICES")
    monkeypatch.delenv("ROCR_VISIBLE_DEVICES", raising=False)
    monkeypatch.delenv("CUDA_VISIBLE_DEVICES", raising=False)
    assert bench.visible_gpu_count() in (None, 0) or bench.visible_gpu_count() > 0    # KFD topology or unreadable
    import inspect
    src = inspect.getsource(bench.spawn_ranks) + inspect.getsource(bench.visible_gpu_count)
    assert "import torch" not in src and "device_count" not in src
