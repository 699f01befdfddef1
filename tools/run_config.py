"""One configuration of tools/sweep_configs.py: python tools/run_config.py netflix 200 cg [unfused] [theta_batch]"""
import sys
sys.argv_saved = sys.argv[:]
shape, f, solver = sys.argv[1], int(sys.argv[2]), sys.argv[3]
fused = not (len(sys.argv) > 4 and sys.argv[4] == "unfused")
tb = int(sys.argv[5]) if len(sys.argv) > 5 else 1
sys.argv = [sys.argv[0], "none"]
import importlib.util, os
spec = importlib.util.spec_from_file_location("sweep", os.path.join(os.path.dirname(os.path.abspath(__file__)), "sweep_configs.py"))
m = importlib.util.module_from_spec(spec)
spec.loader.exec_module(m)
m.run(shape, f, solver, fused=fused, theta_batch=tb, iters=1)
