#!/usr/bin/env python3
"""How fast does the HOST read / write torch-pinned memory on this platform (hipHostMalloc: cached or not)?"""
import time, numpy as np, torch
n = 25 << 20
page = np.random.default_rng(0).integers(0, 256, n, dtype=np.uint8)
pin_t = torch.empty(n, dtype=torch.uint8).pin_memory(); pin = pin_t.numpy(); pin[:] = page
pin2_t = torch.empty(n, dtype=torch.uint8, pin_memory=True); pin2 = pin2_t.numpy(); pin2[:] = page
d = torch.from_numpy(page).cuda()
def T(f, k=5):
    f(); t = time.perf_counter()
    for _ in range(k): f()
    return (time.perf_counter() - t) / k * 1e3
for name, a in (("pageable", page), ("tensor.pin_memory()", pin), ("empty(pin_memory=True)", pin2)):
    print("%-24s host sum %.2f ms, copy out %.2f ms, array_equal %.2f ms, fill %.2f ms" % (name, T(lambda: a.sum()), T(lambda: a.copy()), T(lambda: np.array_equal(a, page)), T(lambda: a.fill(3))))
# after a device copy has landed in it
pin2_t.copy_(d, non_blocking=True); torch.cuda.synchronize()
print("after a D2H copy landed: array_equal(pinned, pageable) %.2f ms (first touch), then %.2f ms" % (T(lambda: np.array_equal(pin2, page), 1), T(lambda: np.array_equal(pin2, page))))
