"""What a trivially simple streaming kernel reaches at the sizes of one split level (ceiling for the
row kernel: 64 MB in + 32 MB out; for the bridge: 32 MB in + 32 MB out), back to back on one stream."""
import torch, time
def bench(fn, reps=200):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
n = 2 * (1 << 20)          # complex128 elements in 32 MB
a = torch.randn(n, dtype=torch.complex128, device="cuda")
b = torch.randn(n, dtype=torch.complex128, device="cuda")
c = torch.empty_like(a)
t = bench(lambda: torch.add(a, b, out=c)); print("add  64 MB in + 32 MB out: %.1f us  %.2f TB/s" % (t, 96 * 1.048576 / t))
t = bench(lambda: c.copy_(a)); print("copy 32 MB in + 32 MB out: %.1f us  %.2f TB/s" % (t, 64 * 1.048576 / t))
t = bench(lambda: torch.mul(a, 2.0, out=c)); print("mul  32 MB in + 32 MB out: %.1f us  %.2f TB/s" % (t, 64 * 1.048576 / t))
big = torch.randn(64 * n, dtype=torch.complex128, device="cuda"); bigc = torch.empty_like(big)
t = bench(lambda: bigc.copy_(big), 20); print("copy 2 GB in + 2 GB out: %.1f us  %.2f TB/s" % (t, 64 * 64 * 1.048576 / t))
# alternate two different kernels (as the tree does), several buffers so that nothing stays in L2
bufs = [torch.randn(n, dtype=torch.complex128, device="cuda") for _ in range(8)]
def chain():
    for i in range(7):
        torch.mul(bufs[i], 2.0, out=bufs[i + 1])
t = bench(chain, 50); print("chain of 7 dependent 32->32 MB kernels: %.1f us per kernel" % (t / 7))
