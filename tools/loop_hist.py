"""Instruction histogram of the main (slab) loop of one kernel in a hipcc -S listing, by LLVM's own loop annotations
(block comments `in Loop: Header=BBn_m Depth=d` / `This Loop Header`), which stay correct when block placement moves
epilogue blocks into the loop's textual range (the widest-backward-branch heuristic of isa_census.py overcounts there).

    python tools/loop_hist.py file.s 'k_upd_fwdILi128ELi32ELi8ELb0ELb1ENS_9ActorArgs' [--top 60]
"""
import collections
import re
import sys


def main():
    path, pat = sys.argv[1], sys.argv[2]
    top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 60
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*:", l) and pat in l)
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    body = lines[start:end]
    # blocks: label line with comment
    blocks, cur = [], None
    for l in body:
        m = re.match(r"^(\.LBB\d+_\d+):\s*(;.*)?$", l)
        if m or re.match(r"^; %bb\.\d+:", l):
            cur = dict(label=m.group(1) if m else l.split(":")[0], comment=l, ins=[])
            blocks.append(cur)
            continue
        if cur is None:
            cur = dict(label="entry", comment="", ins=[])
            blocks.append(cur)
        s = l.strip()
        if s.startswith(";") and ("Loop" in s or "Depth" in s):
            cur["comment"] += " " + s
            continue
        if not s or s.startswith(";") or s.startswith("."):
            continue
        cur["ins"].append(s)
    # loop membership by header
    hdr = collections.Counter()
    for b in blocks:
        for h in re.findall(r"Header=(BB\d+_\d+) Depth=1", b["comment"]):
            hdr[h] += len(b["ins"])
    def n_mfma(h):
        return sum(1 for b in blocks if (f"Header={h} " in b["comment"] or b["label"] == ".L" + h) for i in b["ins"] if i.startswith("v_mfma"))
    main_h = max(hdr, key=lambda h: (n_mfma(h), hdr[h]))
    sel = [b for b in blocks if f"Header={main_h} " in b["comment"] or b["label"] == "." + "L" + main_h]
    c = collections.Counter()
    for b in sel:
        for s in b["ins"]:
            c[s.split()[0]] += 1
    tot = sum(c.values())
    valu = sum(v for k, v in c.items() if k.startswith("v_") and not k.startswith("v_mfma"))
    print(f"loop header {main_h}: {len(sel)} blocks, {tot} instructions; VALU {valu}, MFMA "
          f"{sum(v for k, v in c.items() if k.startswith('v_mfma'))}, LDS {sum(v for k, v in c.items() if k.startswith('ds_'))}, "
          f"waits {c['s_waitcnt']}, s_nop {c['s_nop']}, vmem {sum(v for k, v in c.items() if k.startswith(('global_', 'scratch_', 'buffer_')))}")
    for k, v in c.most_common(top):
        print(f"  {k:34s}{v}")


if __name__ == "__main__":
    main()
