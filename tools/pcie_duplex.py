"""Does a D2H copy overlap an H2D copy on this box?  Two streams, pinned buffers, 2 GiB each way."""
import time
import torch
dev = torch.device("cuda", 0)
n = 1 << 28  # f64: 2 GiB
hin = torch.empty(n, dtype=torch.float64).pin_memory()
hout = torch.empty(n, dtype=torch.float64).pin_memory()
din = torch.empty(n, dtype=torch.float64, device=dev)
dout = torch.ones(n, dtype=torch.float64, device=dev)
s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
def run(up, down):
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    if up:
        with torch.cuda.stream(s1):
            din.copy_(hin, non_blocking=True)
    if down:
        with torch.cuda.stream(s2):
            hout.copy_(dout, non_blocking=True)
    torch.cuda.synchronize(dev)
    return time.perf_counter() - t0
for _ in range(2):
    run(True, True)
gb = n * 8 / 1e9
tu, td, tb = run(True, False), run(False, True), run(True, True)
print(f"H2D alone {gb / tu:.1f} GB/s, D2H alone {gb / td:.1f} GB/s, both at once: {tb * 1e3:.1f} ms for 2 x {gb:.2f} GB = {2 * gb / tb:.1f} GB/s aggregate "
      f"(serial would be {(tu + td) * 1e3:.1f} ms)")
