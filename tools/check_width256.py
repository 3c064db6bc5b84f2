import sys, os
sys.path.insert(0, os.getcwd())
from tests import gpu_checks as G
for spec in G.FWD_SHAPES[-2:]:
    f = G.check_forward(spec); print(spec["name"], "fwd", {k: f"{v:.2e}" for k, v in f.items()})
    g = G.check_gradients(spec); print(spec["name"], "grad", {k: (f"{v:.2e}" if isinstance(v, float) else v) for k, v in g.items()})
