"""Dev tool: which call sites of one Demucs training step issue torch-native fills / adds / copies, with the element counts
(a TorchDispatchMode sees every ATen call, autograd-internal ones included; the Python stack names the remfx_amd frame)."""
import sys, os, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
import bench
from remfx_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda:0")
ops.set_gemm_precision("bf16")
model = bench.build_model("demucs", dev)
opt = model.configure_optimizers()["optimizer"]
data = bench.synthetic_batch(B, 0, dev)

WATCH = ("zero_", "fill_", "zeros", "zeros_like", "new_zeros", "add_", "add", "copy_", "clone", "_to_copy", "cat", "mul", "div",
         "contiguous", "flip", "sub", "neg", "sqrt", "mean", "sum")


class Log(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.agg = collections.defaultdict(lambda: [0, 0])

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = func.__name__.split(".")[0]
        if name in WATCH:
            t = out if isinstance(out, torch.Tensor) else (args[0] if args and isinstance(args[0], torch.Tensor) else None)
            if t is not None and t.is_cuda:
                site = "autograd-internal"
                for fr in reversed(traceback.extract_stack(limit=40)):
                    if "remfx_amd" in fr.filename or fr.filename.endswith("bench.py"):
                        site = f"{os.path.basename(fr.filename)}:{fr.lineno}"
                        break
                a = self.agg[(name, site, str(t.dtype).replace("torch.", ""))]
                a[0] += 1
                a[1] += t.numel() * t.element_size()
        return out


def step():
    opt.zero_grad()
    loss = model.training_step(data, 0)
    loss.backward()
    opt.step(clip_norm=10.0)


for _ in range(2):
    step()
torch.cuda.synchronize()
log = Log()
with log:
    step()
torch.cuda.synchronize()
rows = sorted(log.agg.items(), key=lambda kv: -kv[1][1])
print(f"torch-native elementwise calls of one step (B={B}): {sum(v[0] for _, v in rows)} calls, "
      f"{sum(v[1] for _, v in rows) / 1e9:.2f} GB of results")
for (n, s, dt), (c, b) in rows[:60]:
    print(f"{b / 1e6:10.1f} MB x{c:4d}  {n:12s} {dt:9s} {s}")
