import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import torch.nn.functional as F
from tests import gpu_checks as G
from tests.gpu_checks import *
from tests.helpers import SyntheticCase
T, N = int(sys.argv[1]), int(sys.argv[2])
case = SyntheticCase("tmp", Shapes(T=T, N=N, A=1, obs_dim=18, share_obs_dim=54, act_dim=5, discrete=False, hidden_sizes=[128, 128]), 3,
                     overrides=dict(ppo_epoch=1, critic_epoch=1))
torch.manual_seed(case.seed); np.random.seed(case.seed)
r = G.build_runner(case)
a = r.actor[0]; buf = r.actor_buffer[0]; net = a.actor
cb = r.critic_buffer
cb.compute_returns(cb.value_preds[-1].clone(), r.value_normalizer)
B = T * N
adv = cb.advantages.reshape(B).contiguous()
active = buf.flat("active_masks").reshape(B)
mom = torch.zeros(3, dtype=torch.float64, device=G.DEV)
a.masked_moments(buf, adv, mom)
m = (mom[0] / mom[2]).item(); v = (mom[1] / mom[2]).item() - m * m
advn = ((adv - np.float32(m)) / (np.float32(np.sqrt(v)) + np.float32(1e-5))).contiguous()
net.fold()
buf.update_factor(torch.ones(T, N, 1, device=G.DEV))
os.environ["HARL_FUSED_UPDATE"] = "0"
a._forward_backward(buf.flat("obs"), None, B, buf.flat("actions"), None, buf.flat("action_log_probs"), advn, None, buf.factor.reshape(B), active)
torch.cuda.synchronize()
H = 128
def from_atl(t, M):
    ns = (M + 31) // 32
    x = t[:ns * 32 * H].reshape(ns, H // 8, 64, 4).cpu().numpy()
    out = np.zeros((ns * 32, H), dtype=np.float32)
    for q in range(H // 8):
        for c in range(4):
            R = 4 * q + c
            fb = 32 * (R >> 4) + (R & 3) + 8 * ((R & 15) >> 2)
            for h in (0, 1):
                out[:, fb + 4 * h] = x[:, q, 32 * h:32 * h + 32, c].reshape(-1)
    return out[:M]
dz2 = from_atl(net.dz[0], B)
x2 = from_atl(net.xh[1], B)
x1 = from_atl(net.xh[0], B)
# torch reference (fp64) with intermediate grads
p = {k: torch.from_numpy(v).double() for k, v in case.actor_sd[0].items()}
obs = buf.flat("obs").cpu().double(); act = buf.flat("actions").cpu().double(); olp = buf.flat("action_log_probs").cpu().double()
x = F.layer_norm(obs, (18,), p["base.feature_norm.weight"], p["base.feature_norm.bias"], 1e-5)
z1 = F.linear(x, p["base.mlp.fc.0.weight"], p["base.mlp.fc.0.bias"]); h1 = F.layer_norm(F.relu(z1), (H,), None, None, 1e-5)
y1 = h1 * p["base.mlp.fc.2.weight"] + p["base.mlp.fc.2.bias"]
z2 = F.linear(y1, p["base.mlp.fc.3.weight"], p["base.mlp.fc.3.bias"]); z2.requires_grad_(True)
h2 = F.layer_norm(F.relu(z2), (H,), None, None, 1e-5)
y2 = h2 * p["base.mlp.fc.5.weight"] + p["base.mlp.fc.5.bias"]
mean = F.linear(y2, p["act.action_out.fc_mean.weight"], p["act.action_out.fc_mean.bias"])
std = torch.sigmoid(p["act.action_out.log_std"] / 1.0) * 0.5
lp = -((act - mean) ** 2) / (2 * std * std) - torch.log(std) - 0.9189385332046727
imp = torch.prod(torch.exp(lp - olp), dim=-1, keepdim=True)
A_ = advn.cpu().double().reshape(B, 1); f = buf.factor.reshape(B, 1).cpu().double(); am = active.cpu().double().reshape(B, 1)
surr = torch.min(imp * A_, torch.clamp(imp, 0.8, 1.2) * A_)
loss_sum = (-(f * surr) * am).sum()        # UNSCALED sum, like the kernels
ent = (0.5 + 0.9189385332046727 + torch.log(std)).sum() * am.sum()
(loss_sum - 0.01 * ent).backward()
gz2 = z2.grad.numpy()
print("x_hat_1 max abs err", np.abs(x1 - h1.detach().numpy()).max(), " x_hat_2", np.abs(x2 - h2.detach().numpy()).max())
err = np.abs(dz2 - gz2)
print("dz2: max |ref|", np.abs(gz2).max(), "max abs err", err.max(), "col-sum rel err", np.abs(dz2.sum(0) - gz2.sum(0)).max() / np.abs(gz2.sum(0)).max())
rows = np.argsort(err.max(1))[::-1][:8]
for rr in rows:
    k = err[rr].argmax()
    print(f" row {rr} (slab {rr//32}, lane {rr%32}) feat {k}: gpu {dz2[rr,k]:.6e} ref {gz2[rr,k]:.6e}  row max|ref| {np.abs(gz2[rr]).max():.3e} adv {A_[rr,0]:.3f} imp {imp[rr,0].item():.4f}")
bad = (err.max(1) > 1e-4 * np.abs(gz2).max())
print("rows with err > 1e-4 of max:", int(bad.sum()), "of", B, " slabs:", sorted(set((np.nonzero(bad)[0] // 32).tolist()))[:40])
