"""k_gru_fwd_tp (two waves per slab) vs k_gru_fwd on a full-length inference pass: outputs and time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from harl_amd import _lib
from harl_amd._lib import call, ptr, stream
H = 64
for (L, m) in ((160, 512),) if os.environ.get("GRU_ONE") else ((160, 512), (160, 96), (10, 8192), (1, 512)):
    dev = "cuda:0"
    g = torch.Generator(device=dev); g.manual_seed(1)
    rn = lambda *s: torch.randn(*s, generator=g, device=dev)
    M = L * m
    xin = rn(M * H)
    mask = (torch.rand(M, generator=g, device=dev) > 0.04).float()
    h0 = 0.3 * rn(m, H)
    Wih, Whh = rn(3 * H, H) / 8, rn(3 * H, H) / 8
    bih, bhh = 0.1 * rn(3 * H), 0.1 * rn(3 * H)
    outs = {}
    for save in (1, 0):
        y = torch.zeros(M * H, device=dev); rstd = torch.zeros(M, device=dev)
        sv = [torch.zeros(M * H, device=dev) for _ in range(5)]
        hl = torch.zeros(m, H, device=dev)
        gi = torch.empty(3 * M * H, device=dev)
        args = (ptr(xin), ptr(mask), ptr(h0), ptr(Wih), ptr(bih), ptr(Whh), ptr(bhh), H, L, m, ptr(y), ptr(rstd), *[ptr(t) for t in sv],
                ptr(hl), save, ptr(gi), stream())
        call("harl_gru_fwd", *args)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            call("harl_gru_fwd", *args)
        torch.cuda.synchronize()
        outs[save] = (y.clone(), rstd.clone(), hl.clone(), (time.perf_counter() - t0) / 5 * 1e3)
    a, b = outs[1], outs[0]
    print(f"L={L} m={m}: save=1 {a[3]:.3f} ms  save=0 {b[3]:.3f} ms | y max abs diff {float((a[0]-b[0]).abs().max()):.2e} (|y| max {float(a[0].abs().max()):.2f})"
          f" rstd rel {float(((a[1]-b[1]).abs()/a[1]).max()):.2e} h_last {float((a[2]-b[2]).abs().max()):.2e}")
