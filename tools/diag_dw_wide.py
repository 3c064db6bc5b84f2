"""harl_mlp_dw_partials on a wide first layer (x0n ATL(416) x dz ATL(128)) against numpy, for several row counts."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from harl_amd import _lib  # noqa: E402
from harl_amd._lib import call, ptr, stream  # noqa: E402


def to_atl(a):  # [M, H] -> ATL image (M multiple of 32)
    M, H = a.shape
    g = a.reshape(M // 32, 32, H // 32, 4, 2, 4)           # slab, sample i, tile, q&3, h, c
    g = g.transpose(0, 2, 3, 4, 1, 5)                      # slab, tile, q&3, h, i, c
    return np.ascontiguousarray(g).reshape(-1)             # piece q = 4 tile + (q&3); lane = 32 h + i


def main():
    dev = torch.device("cuda:0")
    H, K = 128, 416
    for M in [int(x) for x in os.environ.get("DIAG_M", "4000,5000,6000,8000").split(",")]:
        Mp = (M + 31) // 32 * 32
        rng = np.random.default_rng(M)
        dz = np.zeros((Mp, H), np.float32)
        x = np.zeros((Mp, K), np.float32)
        dz[:M] = rng.standard_normal((M, H)).astype(np.float32)
        x[:M] = rng.standard_normal((M, K)).astype(np.float32)
        n_slabs = Mp // 32
        for n_wg in sorted({max(1, min(512, (n_slabs + 1) // 2)), 64, 125}):
            elems = H * K + H
            part = torch.full((n_wg * elems,), float("nan"), device=dev)
            out = torch.zeros(elems, device=dev)
            a, b = torch.from_numpy(to_atl(dz)).to(dev), torch.from_numpy(to_atl(x)).to(dev)
            call("harl_mlp_dw_partials", ptr(a), 0, 0, H, ptr(b), 0, 0, None, None, None, K, M, ptr(part), n_wg, stream())
            call("harl_reduce_partials", ptr(part), n_wg, elems, ptr(out), stream())
            torch.cuda.synchronize()
            got = out.cpu().numpy().astype(np.float64)
            ref = dz.astype(np.float64).T @ x.astype(np.float64)
            dw = got[:H * K].reshape(H, K)
            err = np.abs(dw - ref)
            tile_err = [float(err[:, 32 * t:32 * t + 32].max()) for t in range(K // 32)]
            print(f"M={M} n_slabs={n_slabs} n_wg={n_wg}: max|err| {err.max():.3e} (|ref| max {np.abs(ref).max():.1f}); db err "
                  f"{np.abs(got[H * K:] - dz.astype(np.float64).sum(0)).max():.2e}; per column tile: " + " ".join(f"{e:.0e}" for e in tile_err))


if __name__ == "__main__":
    main()
