"""What a 67 MB write costs on this part (the stem's max-pool backward writes that much in 56 us): torch fill / copy of the same size."""
import torch
n = 64 * 16 * 16 * 16 * 64
a = torch.empty(n, device="cuda"); b = torch.randn(n, device="cuda")
def t(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
print("fill  67 MB: %.1f us" % t(lambda: a.zero_()))
print("copy  67 MB -> 67 MB: %.1f us" % t(lambda: a.copy_(b)))
print("relu  67 MB -> 67 MB: %.1f us" % t(lambda: torch.relu(b, out=a) if False else torch.clamp_min(b, 0, out=a)))
