"""Dev tool: ablation builds of one source file.  usage: python scripts/build_abl.py cl_conv RFX_CLC_DBG_BUILD 8 1 9 ...
-> remfx_amd/_C/abl/lib_<file>_<value>.so (pick one with RFX_LIBPATH_DEV)."""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from remfx_amd import _lib

name, macro, values = sys.argv[1], sys.argv[2], sys.argv[3:]
_lib.build()
out = os.path.join(_lib.LIBDIR, "abl")
os.makedirs(out, exist_ok=True)
src = os.path.join(_lib.CSRC, name + ".hip")
objs = [os.path.join(_lib.LIBDIR, os.path.basename(s)[:-4] + ".o") for s in _lib.sources() if s != src]
procs = []
for v in values:
    obj = os.path.join(out, f"{name}_{v}.o")
    procs.append((v, obj, subprocess.Popen(["/opt/rocm/bin/hipcc"] + _lib.HIPCC_FLAGS + [f"-D{macro}={v}", "-c", src, "-o", obj])))
for v, obj, p in procs:
    assert p.wait() == 0, v
    lib = os.path.join(out, f"lib_{name}_{v}.so")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs + [obj])
    os.remove(obj)
    print(lib)
