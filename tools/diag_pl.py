import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from tests import gpu_checks as G
from tests.helpers import GoldenCase, load_noise
np.set_printoptions(precision=10, linewidth=200)
for name in ("mpe_box_h128", "rnn_disc_h64_mb2"):
    case = GoldenCase(name); z = case.z; nz = load_noise(name)
    r = G.build_runner(case)
    torch.manual_seed(case.seed + 12345)
    for a_ in r.actor: a_._trace = []
    r.critic._trace = []
    cb = r.critic_buffer
    cb.compute_returns(cb.value_preds[-1].clone(), r.value_normalizer)
    r.prep_training()
    infos, cinfo = r.train()
    got = np.array([[i["policy_loss"], i["dist_entropy"], i["actor_grad_norm"], i["ratio"]] for i in infos])
    print(name, "infos got\n", got, "\n gold\n", z["actor_infos"], "\n f64\n", nz["actor_infos"], "\n sens\n", nz["sens_actor_infos"])
    gt = z["actor_trace"]; nt = nz["actor_trace"]
    for a in range(case.shapes.A):
        cum = torch.stack(r.actor[a]._trace).double().cpu().numpy()
        per = np.diff(np.concatenate([np.zeros((1, cum.shape[1])), cum]), axis=0)[:, :4]
        print(" agent", a, "policy_loss per update: got", per[:, 0], "\n   gold", gt[gt[:, 0] == a][:, 1], "\n   f64 ", nt[nt[:, 0] == a][:, 1])
