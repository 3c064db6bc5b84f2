"""Print every *_excess figure (and where the worst trace entry sits) of the whole-train() golden comparisons.
    python tools/golden_excess.py [fused]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "fused":
    os.environ["HARL_FUSED_UPDATE"] = "1"
from tests import gpu_checks as G  # noqa: E402

names = [a for a in sys.argv[1:] if a != "fused"] or sorted(f[:-4] for f in os.listdir(os.path.join(os.path.dirname(G.__file__), "golden")) if f.endswith(".npz"))
for n in names:
    try:
        r = G.check_train_golden(n)
    except Exception as e:  # noqa: BLE001
        print(n, "ERR", repr(e)[:200])
        continue
    ex = {k: round(v, 3) for k, v in r.items() if k.endswith("_excess")}
    print(n, json.dumps(ex), r.get("_actor_trace_excess_at", ""), flush=True)
    if os.environ.get("TRACE_TABLE"):
        for k in ("_trace_policy_loss_ref", "_trace_policy_loss_err", "_trace_gradnorm_err"):
            for a, row in r.get(k, {}).items():
                print("     ", k, "agent", a, ":", row)
        print("      first-update rel", r.get("_actor_trace_first_update_rel"), "order", r.get("_agent_order"))
