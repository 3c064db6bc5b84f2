"""Where the BENCH-configuration step (tests/gpu_checks._bench_config_runs) and the oracle part ways: per update the gradient
vector of every agent (HIP vs float64 oracle next to the fp32 oracle vs float64), per parameter tensor the first update's
gradient and the final parameters, and how many samples sit within rounding distance of a PPO clip edge.

    python tools/diag_bench_parity.py [n_threads]      (GPU box; ~2 min of host time)
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import gpu_checks as G  # noqa: E402
from tests.helpers import vec_rel_err  # noqa: E402
from oracle import harl_oracle as O  # noqa: E402


def per_tensor(shapes, a, b):
    out, off = [], 0
    for name, shp in shapes:
        n = int(np.prod(shp))
        out.append((name, float(np.max(np.abs(a[off:off + n] - b[off:off + n]))), float(np.max(np.abs(b[off:off + n])))))
        off += n
    return out


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    logp = sys.argv[2] if len(sys.argv) > 2 else "recipe"
    hip, runs, shapes, meta = G._bench_config_runs(n, True, keep_grad=True, logp=logp)
    f32, f64 = runs["f32"], runs["f64"]
    A = meta["A"]
    np.set_printoptions(linewidth=200, precision=3)
    print("== per-update scalars: rows = agents, columns = epochs; |x - f64| / |f64|")
    for c, nm in enumerate(("policy_loss", "dist_entropy", "grad_norm", "ratio")):
        h = np.stack([t[:, c] for t in hip["atr"]]); o32 = np.stack([t[:, c] for t in f32["atr"]]); o64 = np.stack([t[:, c] for t in f64["atr"]])
        print(nm, "\n  hip vs f64\n", np.abs(h - o64) / np.abs(o64), "\n  f32 vs f64\n", np.abs(o32 - o64) / np.abs(o64), "\n  hip vs f32\n",
              np.abs(h - o32) / np.abs(o32), "\n  values f64\n", o64)
    print("== per-update gradient vectors, |g - g64|_inf / |g64|_inf")
    for a in range(A):
        hg, g32, g64 = hip["grads"][a], f32["grads"][a], f64["grads"][a]
        print(f"agent {a}: hip-f64", " ".join(f"{vec_rel_err(x, y):.2e}" for x, y in zip(hg, g64)),
              "| f32-f64", " ".join(f"{vec_rel_err(x, y):.2e}" for x, y in zip(g32, g64)),
              "| hip-f32", " ".join(f"{vec_rel_err(x, y):.2e}" for x, y in zip(hg, g32)))
    print("== where the hip - f32 gradient differences sit: share of the squared difference carried by the worst OUTPUT ROW of the two "
          "hidden weight matrices (a ReLU decision flipped on one heavy sample moves one row of dW_2 / one unit's column pattern)")
    off = {}
    o_ = 0
    for name, shp in shapes["actor"]:
        off[name] = (o_, shp)
        o_ += int(np.prod(shp))
    for a in range(A):
        for u in range(len(hip["grads"][a])):
            line = []
            for nm in ("base.mlp.fc.0.weight", "base.mlp.fc.3.weight", "act.action_out.fc_mean.weight"):
                o0, shp = off[nm]
                n_ = int(np.prod(shp))
                d = (hip["grads"][a][u][o0:o0 + n_] - f32["grads"][a][u][o0:o0 + n_]).reshape(shp)
                g = f32["grads"][a][u][o0:o0 + n_].reshape(shp)
                rows = (d * d).sum(1)
                cols = (d * d).sum(0)
                line.append(f"{nm.split('.')[-2]}: |d|/|g| {np.sqrt(rows.sum() / (g * g).sum()):.1e} top row {rows.max() / (rows.sum() + 1e-300):.2f} "
                            f"top col {cols.max() / (cols.sum() + 1e-300):.2f}")
            print(f"  agent {a} update {u}: " + " | ".join(line))
    print("== first update of every agent, per tensor: max|g - g64| (hip / f32) and max|g64|")
    for a in range(A):
        th = per_tensor(shapes["actor"], hip["grads"][a][0], f64["grads"][a][0])
        t3 = per_tensor(shapes["actor"], f32["grads"][a][0], f64["grads"][a][0])
        t4 = per_tensor(shapes["actor"], hip["grads"][a][0], f32["grads"][a][0])
        for (nm, eh, mx), (_, e3, _), (_, e4, _) in zip(th, t3, t4):
            print(f"  agent {a} {nm:34s} hip-f64 {eh:.2e}  f32-f64 {e3:.2e}  hip-f32 {e4:.2e}  max|g| {mx:.2e}")
    print("== final parameters, per tensor: max|p - p64| (hip / f32), max|p64 - p0|")
    for a in range(A):
        p0 = torch.cat([v.reshape(-1) for v in meta["actor_sd"][a].values()]).double().numpy()
        th = per_tensor(shapes["actor"], hip["fin"][a], f64["fin"][a])
        t3 = per_tensor(shapes["actor"], f32["fin"][a], f64["fin"][a])
        tm = per_tensor(shapes["actor"], f64["fin"][a], p0)
        t4 = per_tensor(shapes["actor"], hip["fin"][a], f32["fin"][a])
        for (nm, eh, mx), (_, e3, _), (_, mv, _), (_, e4, _) in zip(th, t3, tm, t4):
            print(f"  agent {a} {nm:34s} hip-f64 {eh:.2e}  f32-f64 {e3:.2e}  hip-f32 {e4:.2e}  moved {mv:.2e}  max|p| {mx:.2e}")
        # the worst element: its gradient history
        d = np.abs(hip["fin"][a] - f64["fin"][a])
        i = int(np.argmax(d))
        print(f"  agent {a} worst element {i}: hip {hip['fin'][a][i]:.8g} f32 {f32['fin'][a][i]:.8g} f64 {f64['fin'][a][i]:.8g} start {p0[i]:.8g}")
        print("     grads hip", " ".join(f"{g[i]:.3e}" for g in hip["grads"][a]))
        print("     grads f32", " ".join(f"{g[i]:.3e}" for g in f32["grads"][a]))
        print("     grads f64", " ".join(f"{g[i]:.3e}" for g in f64["grads"][a]))
    # ---- clip-edge neighbours of agent 0's first update (float64 forward of the initial weights)
    cfg = meta["cfg"]
    O.set_work_dtype(torch.float64)
    try:
        d = meta["abuf"][0]
        T = meta["T"]
        B = T * n
        with torch.no_grad():
            lp, _, _ = O.actor_evaluate_actions({k: v.double() for k, v in meta["actor_sd"][0].items()}, cfg,
                                                torch.from_numpy(d["obs"][:-1].reshape(B, -1)).double(),
                                                torch.from_numpy(d["actions"].reshape(B, -1)).double(), None, None)
        imp = torch.exp(lp - torch.from_numpy(d["logp"].reshape(B, -1)).double()).prod(-1)
        for m in (1e-7, 1e-6, 1e-5, 1e-4):
            near = sum(int(((imp - e).abs() < m * e).sum()) for e in (1 - cfg.clip_param, 1 + cfg.clip_param))
            print(f"agent 0, first update: {near} of {B} samples with the ratio within {m:g} (relative) of a clip edge")
        print("ratio quantiles", np.quantile(imp.numpy(), [0.01, 0.1, 0.5, 0.9, 0.99, 0.9999, 1.0]))
        # how much of the first update's gradient mass single samples carry: |d loss / d imp| x imp per sample (unclipped branch)
        adv = torch.from_numpy(np.asarray(runs["f64"]["adv"], dtype=np.float64).reshape(B)) if "adv" in runs["f64"] else None
        if adv is not None:
            advn = (adv - adv.mean()) / (adv.std(unbiased=False) + 1e-5)
            lo, hi = 1 - cfg.clip_param, 1 + cfg.clip_param
            surr1, surr2 = imp * advn, imp.clamp(lo, hi) * advn
            live = (surr1 <= surr2)  # the unclipped branch carries the gradient
            wgt = (imp * advn.abs() * live).numpy()
            srt = np.sort(wgt)[::-1]
            print(f"first update, agent 0: samples on the unclipped branch {int(live.sum())}; share of sum|w| carried by the top 1 / 10 / 100 "
                  f"samples: {srt[0] / srt.sum():.3f} {srt[:10].sum() / srt.sum():.3f} {srt[:100].sum() / srt.sum():.3f}")
    finally:
        O.set_work_dtype(torch.float32)


if __name__ == "__main__":
    main()
