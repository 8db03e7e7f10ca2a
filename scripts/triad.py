"""HBM bandwidth probes on the GPU box (report the measured stream rate next to the vendor peak):
copy (16 B/elt), scale-add triad a = b + s*c (24 B/elt) and a read-only reduction (8 B/elt) on 2 GiB arrays."""
import time, torch
n = 1 << 28  # 2 GiB per fp64 array
a = torch.empty(n, dtype=torch.float64, device="cuda")
b = torch.rand(n, dtype=torch.float64, device="cuda")
c = torch.rand(n, dtype=torch.float64, device="cuda")


def rate(fn, nbytes, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return nbytes * reps / (time.perf_counter() - t) / 1e9


print("copy   %.0f GB/s" % rate(lambda: a.copy_(b), 16 * n))
print("triad  %.0f GB/s" % rate(lambda: torch.add(b, c, alpha=1.5, out=a), 24 * n))
print("read   %.0f GB/s" % rate(lambda: b.sum(), 8 * n))
