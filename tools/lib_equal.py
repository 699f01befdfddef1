"""Are two builds of the library bit-identical on a half-iteration?  python tools/lib_equal.py libA libB [f] [solver]
Each library runs in its own process (CUMF_ALS_LIB); the factors after one X and one Theta update on the Netflix shape
(scaled by --scale) are compared bit for bit."""
import hashlib, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, hashlib, numpy as np, torch
sys.path.insert(0, %r)
from cumf_als_amd import als, datagen
f, solver, scale = int(sys.argv[1]), sys.argv[2], float(sys.argv[3])
shp = datagen.SHAPES["netflix"]
r = datagen.synth_ratings(int(shp["m"] * scale), int(shp["n"] * scale), int(shp["nnz"] * scale * scale), 1000, seed=0, device="cuda")
eng = als.ALSEngine(r, f, shp["lam"], solver=solver)
eng.init_factors()
eng.iterate(2)
torch.cuda.synchronize()
print(hashlib.sha256(eng.XT.cpu().numpy().tobytes()).hexdigest()[:16], hashlib.sha256(eng.thetaT.cpu().numpy().tobytes()).hexdigest()[:16], als.last_kernel_name())
''' % ROOT
a, b = sys.argv[1], sys.argv[2]
f = sys.argv[3] if len(sys.argv) > 3 else "100"
solver = sys.argv[4] if len(sys.argv) > 4 else "lu"
scale = sys.argv[5] if len(sys.argv) > 5 else "0.3"
outs = []
for lib in (a, b):
    env = dict(os.environ, CUMF_ALS_LIB=os.path.join(ROOT, lib))
    o = subprocess.run([sys.executable, "-c", CHILD, f, solver, scale], env=env, capture_output=True, text=True)
    line = [l for l in o.stdout.splitlines() if l.strip()][-1] if o.stdout.strip() else o.stderr[-400:]
    outs.append(line)
    print(lib, line)
print("BIT-IDENTICAL" if outs[0].split()[:2] == outs[1].split()[:2] else "DIFFERENT")
