"""world_size-2 gloo rehearsal of the N>1 path: images are sharded by rank, each rank produces its
own proposals (here with the CPU oracle standing in for the GPU path), and the final gather
reassembles the batch in global image order -- identical to a single-process run."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_IMAGES, P = 5, 40
INFO_HW, FSZ = (96, 160), [(6, 10)]
PARAMS = dict(num_pre_nms=300, num_post_nms=P, nms_threshold=0.7, proposal_min_size=1e-3)


def image_inputs(index):
    rng = np.random.default_rng(1234 + index)
    n = 6 * 10 * 9
    cls = (((rng.permutation(n).astype(np.float32) / np.float32(n)) - np.float32(0.5)) * np.float32(8))
    cls = cls.reshape(6, 10, 9).transpose(2, 0, 1)[None]
    reg = rng.standard_normal((1, 36, 6, 10)).astype(np.float32) * np.float32(0.3)
    return np.ascontiguousarray(cls), reg


def proposals_for(indices):
    from oracle import oracle as orc
    props = np.zeros((len(indices), P, 4), np.float32)
    counts = np.zeros((len(indices),), np.int32)
    for i, idx in enumerate(indices):
        cls, reg = image_inputs(idx)
        p = orc.rpn_proposals((1,) + INFO_HW, [cls], [reg], PARAMS)[0]
        props[i, :len(p)] = p
        counts[i] = len(p)
    return props, counts


def worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from objectdetection_amd.sharding import gather_proposals, shard_range
    lo, hi = shard_range(rank, world, N_IMAGES)
    props, counts = proposals_for(range(lo, hi))
    all_p, all_c = gather_proposals(torch.from_numpy(props), torch.from_numpy(counts), dist)
    # timing protocol of bench.py: barrier + max over ranks
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.barrier()
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), p=all_p.numpy(), c=all_c.numpy(), t=t.numpy(), span=[lo, hi])
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shard_and_gather(tmp_path, oracle):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    ref_p, ref_c = proposals_for(range(N_IMAGES))
    spans = []
    for r in range(2):
        z = np.load(tmp_path / ("rank%d.npz" % r))
        assert np.array_equal(z["c"], ref_c) and np.array_equal(z["p"], ref_p)
        assert float(z["t"][0]) == 2.0
        spans.append(tuple(z["span"]))
    assert spans == [(0, 3), (3, 5)]
    assert ref_c.sum() > 0


def _run_bench(cmd, timeout=300, env=None):
    import json
    import subprocess
    r = subprocess.run(cmd, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout,
                       env=dict(os.environ, **env) if env else None)
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
    return r.returncode, [json.loads(l) for l in lines], r.stderr.decode()


def test_bench_gpus_flag_starts_that_many_ranks():
    """`python bench.py --gpus 2` (no launcher around it) must run two ranks: rehearsed on CPU with
    --dry-run (gloo, no kernel).  n_gpus is what the process group reports, the gather ran."""
    rc, lines, err = _run_bench([sys.executable, "bench.py", "--gpus", "2", "--steps", "2", "--batch", "3", "--dry-run"])
    assert rc == 0, err
    assert len(lines) == 1, "exactly one JSON line (rank 0's)"
    assert lines[0]["n_gpus"] == 2 and lines[0]["gather_ms"] is not None and lines[0]["dry_run"] is True


def test_bench_under_torch_distributed_run():
    """The driver's command shape: torch.distributed.run creates the ranks, --gpus N only names them."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    rc, lines, err = _run_bench([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                                 "--master-addr", "127.0.0.1", "--master-port", str(port), "bench.py", "--gpus", "2",
                                 "--steps", "2", "--batch", "2", "--dry-run"])
    assert rc == 0, err
    assert len(lines) == 1 and lines[0]["n_gpus"] == 2 and lines[0]["gather_ms"] is not None


def test_bench_launcher_reports_a_failed_rank():
    rc, lines, err = _run_bench([sys.executable, "bench.py", "--gpus", "2", "--batch", "-1", "--dry-run"])
    assert rc != 0 and not lines
    assert "ranks exited with" in err


def test_bench_launcher_does_not_wait_for_ranks_stuck_behind_a_dead_one():
    """Only rank 1 dies (before the first collective); rank 0 sits in the barrier waiting for it.  The launcher
    polls every child, kills the survivors and exits non-zero well before any collective timeout."""
    import time
    t0 = time.monotonic()
    rc, lines, err = _run_bench([sys.executable, "bench.py", "--gpus", "2", "--steps", "2", "--batch", "3", "--dry-run"],
                                timeout=120, env={"BENCH_TEST_FAIL_RANK": "1"})
    assert rc != 0 and not lines
    assert "ranks exited with" in err and "3" in err
    assert time.monotonic() - t0 < 90


@pytest.mark.gpu
def test_bench_two_ranks_share_one_gpu_for_rehearsal():
    """The real kernels under the rank plumbing with a single GPU at hand: BENCH_REHEARSE_ON_ONE_GPU=1 puts both ranks
    on GPU 0 and swaps RCCL for gloo.  Each rank takes its own image shard, the bracket is barrier + max over ranks, the
    proposals of both ranks are gathered; the line says that it is a rehearsal, not a measurement."""
    rc, lines, err = _run_bench([sys.executable, "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "4",
                                 "--no-extras", "--cpu-sample", "0"], env={"BENCH_REHEARSE_ON_ONE_GPU": "1"})
    assert rc == 0, err
    assert len(lines) == 1
    line = lines[0]
    assert line["n_gpus"] == 2 and line["gather_ms"] is not None and line["value"] > 0
    assert "REHEARSAL" in line["data"] and line["config"]["batch_per_gpu"] == 4
