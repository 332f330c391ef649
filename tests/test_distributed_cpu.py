"""CPU, world_size 2, gloo: the two exchange steps of the multi-GPU path
(all-gather of per-shard embeddings; all-gather + (distance,id) merge of
per-shard top-k).  The per-shard scans are produced by the oracle here — on the
GPU box the same functions carry the HIP path's outputs over RCCL."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import knn_oracle
        from video_quierer_amd.distributed import all_gather_rows, shard_range, sharded_topk
        rng = np.random.default_rng(100)
        n, q, k, d = 1003, 37, 10, 64                      # ragged on purpose
        rows = knn_oracle.normalize_rows(rng.standard_normal((n, d)).astype(np.float32))
        rows[500] = rows[3]                                # an exact tie across the shard boundary
        qs = knn_oracle.normalize_rows(rng.standard_normal((q, d)).astype(np.float32))
        qs[0] = rows[3]
        lo, hi = shard_range(n, rank, world)
        # ingest exchange: every rank ends up with all rows in frame order
        gathered = all_gather_rows(torch.from_numpy(rows[lo:hi]))
        assert gathered.shape == (n, d) and np.array_equal(gathered.numpy(), rows)
        # equal-shard fast path
        eq = all_gather_rows(torch.full((4, 3), float(rank)))
        assert eq.shape == (8, 3) and eq[:4].eq(0).all() and eq[4:].eq(1).all()
        # search exchange
        lid, ld = knn_oracle.topk(rows[lo:hi], qs, k)
        gid, gd = sharded_topk(torch.from_numpy(lid), torch.from_numpy(ld), lo, k)
        oid, od = knn_oracle.topk(rows, qs, k)
        assert np.array_equal(gid.numpy(), oid), "merged ids differ from the single-shard oracle"
        assert np.array_equal(gd.numpy(), od)
        assert list(gid[0, :2].numpy()) == [3, 500]        # tie broken by the smaller global id
        # k larger than a shard's content: -1 / +inf padding never wins
        lid2, ld2 = knn_oracle.topk(rows[lo:lo + 3], qs, 5)
        gid2, gd2 = sharded_topk(torch.from_numpy(lid2), torch.from_numpy(ld2), lo, 5)
        assert (gid2 >= 0).all() and torch.isfinite(gd2).all()
        ret[rank] = "ok"
    except Exception as e:                                  # surfaced by the parent
        ret[rank] = f"{type(e).__name__}: {e}"
    finally:
        dist.destroy_process_group()


def test_exchange_steps_world2_gloo():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert dict(ret) == {0: "ok", 1: "ok"}, dict(ret)


def test_shard_range_partitions():
    from video_quierer_amd.distributed import shard_range
    for n in (0, 1, 7, 8, 1000, 4000):
        for w in (1, 2, 3, 8):
            parts = [shard_range(n, r, w) for r in range(w)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
            assert max(h - l for l, h in parts) - min(h - l for l, h in parts) <= 1
    # config 4: 4 videos x 1000 frames, video v on ranks {2v, 2v+1}, 500 frames each
    assert [shard_range(1000, r % 2, 2) for r in range(8)] == [(0, 500), (500, 1000)] * 4


def test_bench_launches_its_own_ranks_when_typed_bare():
    """`python bench.py --gpus N` with no launcher around it (how the driver types it): the parent starts N ranks
    under torch.distributed.run before touching a GPU, relays rank 0's ONE JSON line and the ranks' exit code.
    Rehearsed on CPU over gloo (--rehearse-cpu stands in for the encode; the exchange steps are the real ones)."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--rehearse-cpu"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1 and out["rehearsal"] is True
    assert out["config"]["gathered_rows"] == 2 * 256 and out["scaling"] == "weak" and out["value"] > 0
    # a rank that dies takes the exit code with it and no result line is invented
    p = subprocess.run(cmd, env=dict(env, VQ_BENCH_REHEARSE_FAIL_RANK="1"), capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and not [ln for ln in p.stdout.splitlines() if ln.startswith('{"metric"')]


def test_bench_relaunches_with_the_torch_exchange_when_the_native_bring_up_fails_or_hangs():
    """N > 1 with the native exchange: each rank is a supervisor (never on the GPU) + a child.  A child whose native
    communicator cannot be brought up — a clean failure on one rank, or a bring-up that never returns within its deadline —
    leaves with bench.RELAUNCH_CODE on EVERY rank, and each supervisor starts a fresh child with --exchange torch on a new
    rendezvous port; the one result line says which path ran.  Rehearsed over gloo with a stand-in communicator."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--rehearse-cpu"]

    def run(extra):
        p = subprocess.run(cmd, env=dict(env, **extra), capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stderr[-3000:]
        lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
        assert len(lines) == 1, lines
        return json.loads(lines[0]), p.stderr

    out, _ = run({})
    assert "stand-in" in out["world"]["exchange"]                                # the native path came up: no relaunch
    out, err = run({"VQ_BENCH_FAKE_NATIVE_FAIL": "1"})                            # rank 1 fails cleanly -> everyone leaves -> torch
    assert out["world"]["exchange"].startswith("torch.distributed") and "relaunched" in out["world"]["exchange"]
    assert "bring-up failed" in err and out["config"]["gathered_rows"] == 2 * 256
    out, err = run({"VQ_BENCH_FAKE_NATIVE_HANG": "0", "VQ_BENCH_COMM_DEADLINE": "4"})   # rank 0 never returns -> deadline -> torch
    assert "relaunched" in out["world"]["exchange"] and "still not done" in err
    # asked for the torch exchange from the start: no supervisor, no bring-up
    p = subprocess.run(cmd + ["--exchange", "torch"], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "relaunched" not in p.stdout
