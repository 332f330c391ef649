#!/usr/bin/env python3
"""The native exchange (vq_comm_*) with world = 2: equal and ragged all-gathers (a 0-count rank too) and the sharded search
with a shard shorter than k, against the single-index oracle answer; then the two failure paths (a local scan that fails on one
rank, a scratch allocation that fails on one rank): both ranks return an error, nobody hangs, the next call is exact.  Rank r takes GPU r when the box has two; on a ONE-GPU
box both ranks land on device 0 and RCCL refuses ('Duplicate GPU detected', measured on this pool: exit code 3) — so this is
the check to run first on a multi-GPU node, before bench.py --gpus N.
    python scripts/rccl_two_ranks.py            (starts its own two ranks)"""
import os, sys, subprocess, socket
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if "RANK" not in os.environ:
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)], env=env))
    rc = 0
    for p in procs:
        try:
            rc |= p.wait(timeout=150)
        except subprocess.TimeoutExpired:
            p.kill(); rc |= 124
    sys.exit(rc)

import numpy as np, torch, torch.distributed as dist
rank = int(os.environ["RANK"])
DEV = rank if torch.cuda.device_count() > 1 else 0
torch.cuda.set_device(DEV)
dev = torch.device("cuda", DEV)
try:
    dist.init_process_group("nccl", device_id=dev)
    t = torch.ones(4, device=dev) * (rank + 1)
    dist.all_reduce(t)
    torch.cuda.synchronize()
    print(f"rank {rank}: torch nccl all_reduce -> {t.tolist()}", flush=True)
except Exception as e:
    print(f"rank {rank}: torch.distributed nccl bring-up failed (device {DEV}): {type(e).__name__}: {str(e)[:300]}", flush=True)
    sys.exit(3)
from video_quierer_amd import _lib
from video_quierer_amd.comm import Comm
from video_quierer_amd.indexes.hnsw import OptimizedHNSWIndex
from oracle import knn_oracle
_lib.init(DEV)
comm = Comm.from_torch_distributed(DEV)
print(f"rank {rank}: native communicator up, RCCL {comm.rccl_version()}", flush=True)
st = torch.cuda.Stream()
D = 512
for counts in ([300, 300], [300, 17], [0, 40], [5, 0]):
    mine = counts[rank]
    g = torch.Generator(device=dev); g.manual_seed(100 + rank)
    local = torch.randn((max(mine, 1), D), device=dev, generator=g)
    out = torch.full((sum(counts), D), float("nan"), device=dev)
    torch.cuda.synchronize()
    comm.all_gather_rows(local.data_ptr() if mine else 0, counts, D, out.data_ptr(), st.cuda_stream)
    st.synchronize()
    parts = []
    for r, c in enumerate(counts):
        gg = torch.Generator(device=dev); gg.manual_seed(100 + r)
        parts.append(torch.randn((max(c, 1), D), device=dev, generator=gg)[:c])
    want = torch.cat(parts)
    assert torch.equal(out, want), (rank, counts)
print(f"rank {rank}: all-gather of rows ok (equal, ragged, zero-count)", flush=True)
# sharded search: 20,000 + 7 rows (the second shard shorter than k), 33 queries, k = 10
rng = np.random.default_rng(5)
rows = knn_oracle.normalize_rows(rng.standard_normal((20007, D)).astype(np.float32))
qs = knn_oracle.normalize_rows(rng.standard_normal((33, D)).astype(np.float32))
lo, hi = (0, 20000) if rank == 0 else (20000, 20007)
idx = OptimizedHNSWIndex(dimension=D, device=DEV)
idx.add_batch(list(rows[lo:hi]), list(range(hi - lo)))
q_t = torch.from_numpy(qs).to(dev)
ids = torch.empty((33, 10), dtype=torch.int32, device=dev); dd = torch.empty((33, 10), device=dev)
torch.cuda.synchronize()
comm.search_sharded(idx, q_t.data_ptr(), 33, 10, lo, ids.data_ptr(), dd.data_ptr())
idx.synchronize()
oid, od = knn_oracle.topk(rows, qs, 10)
assert np.array_equal(ids.cpu().numpy(), oid) and np.array_equal(dd.cpu().numpy(), od), rank
print(f"rank {rank}: sharded search over two ranks == the single-index oracle answer (ids and distances bit-exact)", flush=True)
comm.check()
# a rank whose LOCAL scan fails still enters the exchange: rank 1's shard refuses the fp16 mode (rows stored at 3x unit length).
# Rank 1 gets its error; rank 0's call returns, its lists are empty and comm.check() names rank 1.  Nobody hangs.
from video_quierer_amd.indexes.hnsw import MODE_FP16
bad = OptimizedHNSWIndex(dimension=D, device=DEV)
big = np.ascontiguousarray(rows[:2000] * np.float32(3.0 if rank == 1 else 1.0))
_lib.check(_lib.load().vq_index_add(bad._h, _lib.fptr(big), 2000, 0))
bad._ids = list(range(2000)); bad.element_count = 2000; bad.entry_point = 0
ids.fill_(7)
try:
    comm.search_sharded(bad, q_t.data_ptr(), 33, 10, 2000 * rank, ids.data_ptr(), dd.data_ptr(), mode=MODE_FP16)
    raised = False
except ValueError as e:
    raised = "fp16 scan needs" in str(e)
bad.synchronize()
assert raised == (rank == 1), (rank, raised)
assert bool((ids == -1).all()), rank
try:
    comm.check(); flagged = False
except _lib.VqError as e:
    flagged = "rank 1" in str(e)
assert flagged, rank
print(f"rank {rank}: a failed local scan on rank 1 voided the call on both ranks without a hang", flush=True)
bad.close()
# an allocation that fails on ONE rank is agreed on before the data collective: both ranks return an error, the next call works
os.environ["VQ_COMM_FAIL_ALLOC"] = "1" if rank == 1 else "0"
comm2 = Comm.from_torch_distributed(DEV)
del os.environ["VQ_COMM_FAIL_ALLOC"]
try:
    comm2.search_sharded(idx, q_t.data_ptr(), 33, 10, lo, ids.data_ptr(), dd.data_ptr())
    failed = False
except _lib.VqError as e:
    failed = ("scratch hipMalloc" in str(e)) if rank == 1 else ("rank 1 could not prepare" in str(e))
assert failed, rank
comm2.search_sharded(idx, q_t.data_ptr(), 33, 10, lo, ids.data_ptr(), dd.data_ptr())
idx.synchronize()
assert np.array_equal(ids.cpu().numpy(), oid) and np.array_equal(dd.cpu().numpy(), od), rank
print(f"rank {rank}: a scratch allocation failing on rank 1 made both ranks return before the collective; the retry is exact", flush=True)
comm2.close()
idx.close(); comm.close()
dist.barrier(); dist.destroy_process_group()
