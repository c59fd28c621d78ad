import os, sys, subprocess, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from permafrost_engine_amd import build as nb
OUT = os.path.join(ROOT, "build_prof")
def build(tag, flags):
    os.makedirs(OUT, exist_ok=True)
    objs = []
    for s in nb.SOURCES:
        o = os.path.join(OUT, s[:-4] + "_%s.o" % tag)
        subprocess.check_call([nb.HIPCC] + nb.FLAGS + flags + ["-c", os.path.join(nb.CSRC, s), "-o", o])
        objs.append(o)
    lib = os.path.join(OUT, "libnavhip_%s.so" % tag)
    subprocess.check_call([nb.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
    return lib
if "--build" in sys.argv:
    from concurrent.futures import ThreadPoolExecutor
    V = {"noatomopt": ["-mllvm", "-amdgpu-atomic-optimizer-strategy=None"],
         "nodpp": ["-mllvm", "-amdgpu-dpp-combine=false"],
         "nounroll": ["-fno-unroll-loops"],
         "o2": ["-O2"],
         "nolicm": ["-mllvm", "-disable-licm-promotion"],
         "noinline": ["-fno-inline-functions"],
         "nomsched": ["-mllvm", "-enable-misched=false"]}
    with ThreadPoolExecutor(8) as ex:
        for k, r in zip(V, ex.map(lambda kv: build(*kv), V.items())): print(k, r)
    sys.exit(0)
which = sys.argv[1]
if which != "main":
    os.environ["NAVHIP_LIB"] = os.path.join(OUT, "libnavhip_%s.so" % which)
from permafrost_engine_amd import navhip
from tests import cases
ent, des, dyn, nd, stat, ns = cases.cp_problems(1, 500, 2, 2, 9.0)
ctx = navhip.NavContext(1, 1)
rows = ctx.G_ClearPath_NewVelocity(ent, des, dyn, nd, stat, ns, rows=True)
wave = ctx.G_ClearPath_NewVelocity(ent, des, dyn, nd, stat, ns)
bad = np.flatnonzero((rows.view(np.uint32) != wave.view(np.uint32)).any(1))
print(which, "rows != wave:", bad, rows[bad[:4]], wave[bad[:4]], nd[bad[:4]], ns[bad[:4]])


def run(idx):
    idx = np.asarray(idx)
    r = ctx.G_ClearPath_NewVelocity(ent[idx], des[idx], dyn[idx], nd[idx], stat[idx], ns[idx], rows=True)
    return (r.view(np.uint32) == wave[idx].view(np.uint32)).all(1)

