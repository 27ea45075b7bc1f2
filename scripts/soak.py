"""Soak: graph-mode training with the image pools and the LR schedule switched on, N steps on synthetic batches; prints the
losses every 250 steps, checks they stay finite and that checkpoint save -> load -> step reproduces the next step exactly."""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import unpaired_image_generation_amd as u
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
torch.manual_seed(0)
m = u.CycleGAN(n_blocks=9, dtype=torch.bfloat16, use_graph=True, pool_size=50, pool_seed=3)
g = torch.Generator("cuda").manual_seed(1)
data = [(torch.rand(4, 3, 256, 256, device="cuda", generator=g) * 2 - 1, torch.rand(4, 3, 256, 256, device="cuda", generator=g) * 2 - 1) for _ in range(8)]
t0 = time.time()
for i in range(N):
    if i % 50 == 0: m.set_epoch(i // 50, n_const=N // 100, n_decay=N // 100)       # a compressed 200-"epoch" schedule
    a, b = data[i % len(data)]
    l = m.train_step(a, b, sync=(i % 250 == 249))
    if i % 250 == 249:
        assert all(v == v and abs(v) < 1e4 for v in l.values()), l
        print(i + 1, f"{(time.time() - t0) / (i + 1) * 1e3:.2f} ms/step", {k: round(v, 3) for k, v in l.items()}, "lr_scale", m.lr_scale, flush=True)
with tempfile.TemporaryDirectory() as d:
    p = os.path.join(d, "ck.pt"); m.save(p)
    l1 = m.train_step(*data[0])
    m2 = u.CycleGAN(n_blocks=9, dtype=torch.bfloat16, use_graph=True, pool_size=50, pool_seed=3); m2.load(p)
    l2 = m2.train_step(*data[0])
    print("resume exact:", all(l1[k] == l2[k] for k in l1), l1["cyc_A"], l2["cyc_A"])
