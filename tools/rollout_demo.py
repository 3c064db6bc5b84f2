"""Print the learning curve of OnPolicyHARunner.run() on the toy environment (tests/fake_env.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import gpu_checks as G
for disc, rec in ((False, False), (True, False), (False, True)):
    print("discrete" if disc else "box", "gru" if rec else "mlp", G.check_rollout_learning(disc, rec))
