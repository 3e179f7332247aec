"""Dev tool: which Python call sites issue the ATen kernels / device copies left in a Demucs training step (torch profiler with
stacks; grouped by op and innermost remfx_amd frame).   python scripts/aten_sites.py [B]"""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from remfx_amd import ops
import bench

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda:0")
ops.set_gemm_precision("bf16")
model = bench.build_model("demucs", dev)
opt = model.configure_optimizers()["optimizer"]
data = bench.synthetic_batch(B, 0, dev)


def step():
    opt.zero_grad()
    loss = model.training_step(data, 0)
    loss.backward()
    opt.step(clip_norm=10.0)


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=False) as prof:
    step()
    torch.cuda.synchronize()
agg = collections.Counter()
for e in prof.events():
    n = e.name
    if not n.startswith("aten::") or n in ("aten::empty", "aten::empty_strided", "aten::view", "aten::as_strided", "aten::reshape",
                                           "aten::unsqueeze", "aten::squeeze", "aten::transpose", "aten::permute", "aten::select",
                                           "aten::slice", "aten::narrow", "aten::expand", "aten::detach", "aten::alias", "aten::t",
                                           "aten::_unsafe_view", "aten::view_as", "aten::stride", "aten::size", "aten::is_nonzero",
                                           "aten::empty_like", "aten::resize_", "aten::result_type", "aten::to", "aten::contiguous",
                                           "aten::unflatten", "aten::flatten", "aten::_reshape_alias", "aten::unbind", "aten::split",
                                           "aten::chunk", "aten::split_with_sizes", "aten::movedim", "aten::lift_fresh", "aten::item",
                                           "aten::_local_scalar_dense", "aten::zeros", "aten::ones", "aten::zeros_like", "aten::set_"):
        continue
    site = "?"
    for fr in (e.stack or []):
        if "remfx_amd" in fr or "bench.py" in fr:
            site = fr.split("/")[-1]
            break
    if site == "?" and e.stack:
        site = "(autograd engine)" if any("backward" in f for f in e.stack[:3]) or not any(".py" in f for f in e.stack) else e.stack[0][-60:]
    agg[(n, site)] += 1
for (n, site), c in agg.most_common(60):
    print(f"{c:5d}  {n:28s} {site}")
