"""Wall time of the device-resident ingest path: HNSWIndex.add_device (ids + kernels) and vq_index_add_device alone, 4 x 250k x 512 rows."""
import sys, time, torch
sys.path.insert(0, '.')
from video_quierer_amd.indexes.hnsw import OptimizedHNSWIndex
dev = torch.device("cuda", 0)
idx = OptimizedHNSWIndex(dimension=512)
g = torch.Generator(device=dev); g.manual_seed(1)
blks = [torch.randn((250_000, 512), device=dev, generator=g) for _ in range(4)]
torch.cuda.synchronize()
for i, b in enumerate(blks):
    t = time.perf_counter()
    idx.add_device(b.data_ptr(), 250_000, range(i * 250_000, (i + 1) * 250_000), normalize=True)
    idx.synchronize()
    print(f"add_device(normalize) block {i}: {(time.perf_counter() - t) * 1e3:.2f} ms", flush=True)
# the kernels alone: the same rows through the C ABI, no id bookkeeping
import ctypes
from video_quierer_amd import _lib
lib = _lib.load()
h = ctypes.c_void_p()
idx2 = OptimizedHNSWIndex(dimension=512)
for i, b in enumerate(blks):
    torch.cuda.synchronize()
    t = time.perf_counter()
    _lib.check(lib.vq_index_add_device(idx2._h, ctypes.c_void_p(b.data_ptr()), 250_000, 1))
    idx2.synchronize()
    print(f"vq_index_add_device(normalize=1) block {i}: {(time.perf_counter() - t) * 1e3:.2f} ms  (512 MB read + 512 MB written + the fp16 copy)", flush=True)
