import os, sys, subprocess
os.environ.setdefault("OMP_NUM_THREADS", "16")
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from streamkit_amd import engine
from oracle_lib import OracleModel
path = "/tmp/synth_micro.bin"
if not os.path.exists(path): subprocess.check_call([os.path.join(ROOT, "tools", "make_synth_model"), path, "--size", "micro"])
om = OracleModel(path); gm = engine.Model(path); ctx = engine.Context(gm, 1)
rng = np.random.default_rng(0)
def check(name, kind, x):
    a = ctx.math(kind, x); b = om.math(kind, x)
    bad = np.nonzero(a.view(np.uint32) != b.view(np.uint32))[0]
    bad = [i for i in bad if not (np.isnan(a[i]) and np.isnan(b[i]))]
    print("%-22s n=%d mismatches=%d" % (name, x.size, len(bad)), [(float(x[i]), float(a[i]), float(b[i])) for i in bad[:5]])
check("expf [-90,0]", 0, np.concatenate([-rng.random(2000000).astype(np.float32) * 90, -np.exp(rng.normal(0, 3, 1000000)).astype(np.float32)]))
check("expf [-1,90]", 0, (rng.random(1000000) * 91 - 1).astype(np.float32))
check("logf", 1, np.exp(rng.normal(0, 20, 2000000)).astype(np.float32))
allf = np.arange(0, 1 << 32, 4099, dtype=np.uint64).astype(np.uint32).view(np.float32)
check("f16 round hw", 2, allf); check("f16 round sw", 3, allf)
sub = (rng.random(1000000) * 1.3e-4).astype(np.float32) * np.where(rng.random(1000000) < 0.5, -1, 1).astype(np.float32)
check("f16 round hw subnormal", 2, sub)
check("gelu", 4, (rng.normal(0, 4, 2000000)).astype(np.float32))
check("rsqrt", 5, np.exp(rng.normal(0, 3, 2000000)).astype(np.float32))
check("1/x f64", 6, (rng.random(2000000) * 1500 + 1).astype(np.float32))
check("log10 f64", 7, np.exp(rng.normal(0, 10, 2000000)).astype(np.float32))
