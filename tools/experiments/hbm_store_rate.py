"""What this box's HBM does for plain streaming writes and reads (torch fill_ / sum on 8.6 GB), next to the expand kernel's 4.8 TB/s
of position stores and the walk's 5.0 TB/s of record reads."""
import torch
n = int(8.6e9) // 4
x = torch.empty(n, dtype=torch.int32, device="cuda")
def timed(f, reps=5):
    f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
ms = timed(lambda: x.fill_(7))
print("fill_ of %.1f GB: %.3f ms = %.2f TB/s written" % (n * 4 / 1e9, ms, n * 4 / ms / 1e9))
ms = timed(lambda: x.zero_())
print("zero_ of %.1f GB: %.3f ms = %.2f TB/s written" % (n * 4 / 1e9, ms, n * 4 / ms / 1e9))
y = torch.empty(n * 2, dtype=torch.int32, device="cuda")
ms = timed(lambda: y.sum())
print("sum of %.1f GB: %.3f ms = %.2f TB/s read" % (n * 8 / 1e9, ms, n * 8 / ms / 1e9))
ms = timed(lambda: x.copy_(y[:n]))
print("copy of %.1f GB: %.3f ms = %.2f TB/s read + written" % (n * 4 / 1e9, ms, 2 * n * 4 / ms / 1e9))
