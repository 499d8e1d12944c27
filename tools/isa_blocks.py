"""Per-basic-block instruction mix of one kernel in a hipcc -S listing (development aid).

usage: python tools/isa_blocks.py file.s [kernel-substring] [min_mfma]
Prints, for every basic block with at least `min_mfma` MFMAs: MFMA count, other VALU, LDS, vector-memory, scalar and
waitcnt counts -- the VALU : MFMA ratio is what bounds an f32-MFMA kernel on gfx950."""
import re
import sys


def main():
    path = sys.argv[1]
    pat = sys.argv[2] if len(sys.argv) > 2 else ""
    min_mfma = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    lines = open(path).read().split("\n")
    start = None
    for i, l in enumerate(lines):
        if re.match(r"^_Z\S*:", l) and pat in l:
            start = i
            break
    if start is None:
        sys.exit("kernel not found")
    blocks = []
    cur = {"name": "entry", "line": start}
    counts = dict(mfma=0, valu=0, lds=0, vmem=0, salu=0, wait=0, br=0)
    cur.update(counts)
    tot = dict(counts)
    for i in range(start + 1, len(lines)):
        l = lines[i].strip()
        if l.startswith(".Lfunc_end") or l.startswith("s_endpgm") and False:
            break
        m = re.match(r"^(\.LBB\S+):", l)
        if m:
            blocks.append(cur)
            cur = {"name": m.group(1), "line": i}
            cur.update(counts)
            continue
        if not l or l.startswith(";") or l.startswith("."):
            continue
        op = l.split()[0]
        k = None
        if "mfma" in op:
            k = "mfma"
        elif op.startswith("v_"):
            k = "valu"
        elif op.startswith("ds_"):
            k = "lds"
        elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
            k = "vmem"
        elif op.startswith("s_waitcnt"):
            k = "wait"
        elif op.startswith(("s_cbranch", "s_branch")):
            k = "br"
        elif op.startswith("s_"):
            k = "salu"
        if k:
            cur[k] += 1
            tot[k] += 1
    blocks.append(cur)
    print(f"{'block':>14} {'line':>7} mfma valu  lds vmem salu wait   valu/mfma")
    for b in blocks:
        if b["mfma"] >= min_mfma:
            print(f"{b['name']:>14} {b['line'] - start:7d} {b['mfma']:4d} {b['valu']:4d} {b['lds']:4d} {b['vmem']:4d} {b['salu']:4d} {b['wait']:4d}   {b['valu'] / max(1, b['mfma']):.2f}")
    print("static total", tot)


if __name__ == "__main__":
    main()
