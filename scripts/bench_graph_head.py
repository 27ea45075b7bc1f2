"""Is the ~75 us spacing of the first nodes of the step graph real?  Times, without a profiler, a small captured graph of the
same first ops (2 x to_nhwc, a 91 MB fill, 2 x cat) and the same ops launched eagerly."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import unpaired_image_generation_amd as u
from unpaired_image_generation_amd import ops
a = torch.rand(4, 3, 256, 256, device="cuda"); b = torch.rand(4, 3, 256, 256, device="cuda")
grad = torch.zeros(22_760_000, device="cuda")
def head():
    xa, xb = ops.to_nhwc(a, torch.bfloat16), ops.to_nhwc(b, torch.bfloat16)
    grad.zero_()
    x2 = torch.cat([xb, xa]); return torch.cat([x2, x2])
for _ in range(3): head()
torch.cuda.synchronize()
def t(fn, n=50):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize(); return e0.elapsed_time(e1) * 1e3 / n
print("eager   %.1f us per head" % t(head))
g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
with torch.cuda.stream(s):
    head()
    with torch.cuda.graph(g, stream=s): keep = head()
torch.cuda.synchronize()
for _ in range(3): g.replay()
print("graph   %.1f us per replay" % t(g.replay))
