import time, numpy as np, torch, ctypes
rt = torch.cuda.cudart()
n = 64 * 2**20 // 4
a = np.random.rand(n).astype(np.float32)
d = torch.empty(n, dtype=torch.float32, device="cuda")
def h2d_pageable():
    t0 = time.perf_counter(); d.copy_(torch.from_numpy(a)); torch.cuda.synchronize(); return time.perf_counter() - t0
for _ in range(2): h2d_pageable()
print("pageable H2D 64 MiB: %.2f ms" % (min(h2d_pageable() for _ in range(5)) * 1e3))
for it in range(3):
    b = np.random.rand(n).astype(np.float32)   # fresh buffer: first-touch already done by rand
    t0 = time.perf_counter(); rc = rt.cudaHostRegister(b.ctypes.data, b.nbytes, 0); t1 = time.perf_counter()
    tb = torch.from_numpy(b)
    t2 = time.perf_counter(); d.copy_(tb, non_blocking=True); torch.cuda.synchronize(); t3 = time.perf_counter()
    t4 = time.perf_counter(); rt.cudaHostUnregister(b.ctypes.data); t5 = time.perf_counter()
    print(f"register rc={rc} {1e3*(t1-t0):.2f} ms, H2D from registered {1e3*(t3-t2):.2f} ms ({b.nbytes/(t3-t2)/1e9:.1f} GB/s), unregister {1e3*(t5-t4):.2f} ms")
p = torch.empty(n, dtype=torch.float32).pin_memory()
t0 = time.perf_counter(); p.numpy()[:] = a; t1 = time.perf_counter()
print(f"cpu memcpy into pinned (1 thread): {1e3*(t1-t0):.2f} ms ({a.nbytes/(t1-t0)/1e9:.1f} GB/s)")
t2 = time.perf_counter(); d.copy_(p, non_blocking=True); torch.cuda.synchronize(); t3 = time.perf_counter()
print(f"H2D pinned: {1e3*(t3-t2):.2f} ms ({a.nbytes/(t3-t2)/1e9:.1f} GB/s)")
t2 = time.perf_counter(); p.copy_(d, non_blocking=True); torch.cuda.synchronize(); t3 = time.perf_counter()
print(f"D2H pinned: {1e3*(t3-t2):.2f} ms ({a.nbytes/(t3-t2)/1e9:.1f} GB/s)")
import threading
def par_copy(dst, src, nt):
    chunks = np.array_split(np.arange(0, n + 1, max(1, n // nt))[: nt + 1], 1)
    bounds = np.linspace(0, n, nt + 1).astype(np.int64)
    ths = [threading.Thread(target=lambda i=i: np.copyto(dst[bounds[i]:bounds[i+1]], src[bounds[i]:bounds[i+1]])) for i in range(nt)]
    t0 = time.perf_counter(); [t.start() for t in ths]; [t.join() for t in ths]; return time.perf_counter() - t0
for nt in (2, 4, 8):
    print(f"cpu memcpy into pinned ({nt} threads): {a.nbytes/par_copy(p.numpy(), a, nt)/1e9:.1f} GB/s")
