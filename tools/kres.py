"""Static resource usage of the compiled kernels of one translation unit (no GPU): registers, spills, scratch, occupancy.

    python tools/kres.py mlp.hip [name-filter] [-- extra hipcc flags]

Parses hipcc's -Rpass-analysis=kernel-resource-usage remarks (one compile of harl_amd/csrc/<file> with the flags of
harl_amd/_build.py)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from harl_amd._build import CSRC, EXTRA_FLAGS
    args = sys.argv[1:]
    extra = []
    if "--" in args:
        k = args.index("--")
        args, extra = args[:k], args[k + 1:]
    src = args[0]
    filt = args[1] if len(args) > 1 else ""
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", os.path.join(CSRC, src), "-o", "/dev/null",
           "-Rpass-analysis=kernel-resource-usage"] + EXTRA_FLAGS.get(src, []) + extra
    out = subprocess.run(cmd, capture_output=True, text=True).stderr
    cur = None
    rows = {}
    for ln in out.splitlines():
        m = re.search(r"Function Name: (\S+)", ln)
        if m:
            cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
            cur = re.sub(r"\(.*", "", cur).replace("void ", "")
            rows[cur] = {}
            continue
        m = re.search(r"remark:\s+(\w[\w \[\]/]*?): (\d+)", ln)
        if m and cur:
            rows[cur][m.group(1).strip()] = int(m.group(2))
    print(f"{'kernel':70s} {'VGPR':>5s} {'AGPR':>5s} {'spill':>5s} {'scratch':>7s} {'occ':>3s} {'SGPR':>5s}")
    for k, v in rows.items():
        if filt in k:
            print(f"{k[:70]:70s} {v.get('VGPRs', 0):5d} {v.get('AGPRs', 0):5d} {v.get('VGPRs Spill', 0):5d} "
                  f"{v.get('ScratchSize [bytes/lane]', 0):7d} {v.get('Occupancy [waves/SIMD]', 0):3d} {v.get('TotalSGPRs', 0):5d}")
    if "error" in out:
        print(out[-3000:])


if __name__ == "__main__":
    main()
