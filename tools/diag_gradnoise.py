import os, sys
sys.path.insert(0, os.getcwd())
from tests import gpu_checks as G
for mode in ("logp", "1", "0"):
    os.environ["HARL_FUSED_UPDATE"] = mode
    for spec in G.FWD_SHAPES[:2]:
        r = G.check_gradient_noise(spec)
        print("mode", mode, spec["name"], "worst ratio", round(r["gpu_over_ref32_worst"], 2))
        for k in r:
            if k.startswith("_t32/"):
                n = k[5:]
                print(f"    {n:34s} inf: t32 {r['_t32/'+n]:.2e} gpu {r['_gpu/'+n]:.2e}   rms: t32 {r['_t32rms/'+n]:.2e} gpu {r['_gpurms/'+n]:.2e}")
