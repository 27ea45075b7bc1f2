set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_ops_gpu.py -q -m gpu -k "fused_instnorm" 2>&1 | tail -2
echo "--- force-comm (RCCL world 1)"
timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --force-comm 2>&1 | tail -1 | cut -c1-200
echo "--- 512x512 generator forward, bf16 vs fp32 path"
timeout -k 10 300 python - <<'PY'
import torch, unpaired_image_generation_amd as u
torch.manual_seed(0)
g32 = u.Generator(n_blocks=9, dtype=torch.float32)
g16 = u.Generator(n_blocks=9, dtype=torch.bfloat16); g16.load_state_dict(g32.state_dict())
x = torch.rand(2, 3, 512, 512, device="cuda") * 2 - 1
with torch.no_grad():
    a, b = g32(x), g16(x)
print("512^2 out", tuple(a.shape), "fp32 finite", bool(torch.isfinite(a).all()), "bf16-fp32 Linf", float((a - b).abs().max()))
m = u.CycleGAN(n_blocks=9, dtype=torch.bfloat16)
print("512^2 train step B=2:", m.train_step(x, x.flip(0)))
PY
