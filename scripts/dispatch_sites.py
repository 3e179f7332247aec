"""Dev tool: every ATen op a Demucs training step still dispatches (forward, loss, autograd engine, optimiser), grouped by op
and by the innermost remfx_amd / bench frame that issued it -- a TorchDispatchMode sees the autograd thread too, where the
profiler's stacks come back empty (scripts/aten_sites.py).   python scripts/dispatch_sites.py [B]"""
import sys, os, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
from remfx_amd import ops
import bench

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda:0")
ops.set_gemm_precision("bf16")
model = bench.build_model("demucs", dev)
opt = model.configure_optimizers()["optimizer"]
data = bench.synthetic_batch(B, 0, dev)

VIEWS = {"view", "as_strided", "reshape", "unsqueeze", "squeeze", "transpose", "permute", "select", "slice", "narrow", "expand",
         "detach", "alias", "t", "_unsafe_view", "unflatten", "unbind", "split", "split_with_sizes", "chunk", "empty", "empty_strided",
         "empty_like", "new_empty", "new_empty_strided", "_local_scalar_dense", "is_nonzero", "record_stream", "set_", "resize_",
         "lift_fresh", "_reshape_alias", "view_as_real", "view_as_complex", "new_zeros_"}


def step():
    opt.zero_grad()
    loss = model.training_step(data, 0)
    loss.backward()
    opt.step(clip_norm=10.0)


class Log(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.agg = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.overloadpacket.__name__ if hasattr(func, "overloadpacket") else str(func)
        if name not in VIEWS:
            site = "(no python frame: autograd engine)"
            for fr in reversed(traceback.extract_stack()[:-1]):
                fn = fr.filename
                if ("remfx_amd" in fn or fn.endswith("bench.py")) and "dispatch_sites" not in fn:
                    site = f"{os.path.basename(fn)}:{fr.lineno} {fr.name}"
                    break
            numel = 0
            for a in args:
                if isinstance(a, torch.Tensor):
                    numel = a.numel()
                    break
            shp = ""
            if numel > 65536:
                for a in args:
                    if isinstance(a, torch.Tensor):
                        shp = str(tuple(a.shape))
                        break
            self.agg[(name, site, ("big " + shp) if numel > 65536 else "small")] += 1
        return func(*args, **(kwargs or {}))


for _ in range(3):
    step()
torch.cuda.synchronize()
with Log() as L:
    step()
    torch.cuda.synchronize()
tot = sum(L.agg.values())
print(f"{tot} dispatched non-view ATen ops in one step at B = {B}")
for (n, site, sz), c in L.agg.most_common(120):
    print(f"{c:5d}  {n:26s} {sz:34s} {site}")
