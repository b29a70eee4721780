"""Instruction mix of the MFMA loops of a kernel, from hipcc's assembly (no GPU needed):
  hipcc --offload-arch=gfx950 -O3 ... --offload-device-only -S x.hip -o x.s ; python tools/kloop.py x.s <kernel substring>
For every loop (by its header block) that contains MFMAs: static instruction counts of the blocks that belong to the loop at
its own depth or deeper -- VALU (without MFMA), MFMA, SALU, LDS, VMEM, waits -- and the VALU opcodes by frequency.  Static
counts: both sides of a branch inside the loop are counted."""
import collections
import re
import sys


def main():
    path, pat = sys.argv[1], sys.argv[2]
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 12
    lines = open(path).read().split("\n")
    start = None
    for i, ln in enumerate(lines):
        if re.match(r"^_Z\S*:", ln) and pat in ln:
            start = i
            break
    if start is None:
        sys.exit("kernel not found")
    end = next(i for i in range(start, len(lines)) if lines[i].startswith("\t.section") or lines[i].strip() == "s_endpgm")
    # blocks: label -> (header, depth) from the "in Loop: Header=BBx_y Depth=d" / "=>This Inner Loop Header: Depth=d" comments
    cur, loops = None, collections.OrderedDict()
    parent = {}
    for ln in lines[start:end]:
        m = re.match(r"^(\.LBB\d+_\d+):\s*;?(.*)", ln)
        if m:
            label, note = m.group(1), m.group(2)
            hdrs = []
            h = re.search(r"=>This (?:Inner )?Loop Header: Depth=(\d+)", note)
            if h:
                hdrs.append((label[2:], int(h.group(1))))
            for hh, d in re.findall(r"(?:in Loop: Header=|Parent Loop )(BB\d+_\d+) Depth=(\d+)", note):
                hdrs.append((hh, int(d)))
            cur = hdrs
            continue
        if cur is None:
            continue
        s = ln.strip()
        if not s or s.startswith(";") or s.startswith("."):
            continue
        op = s.split()[0]
        for hh, d in cur:
            loops.setdefault((hh, d), collections.Counter())[op] += 1
    # the comment of a block names only its innermost loop on the label line; nested parents appear on following comment lines,
    # so re-scan including the "Parent Loop" lines that follow a label
    loops = collections.OrderedDict()
    cur = []
    i = start
    while i < end:
        ln = lines[i]
        m = re.match(r"^(\.LBB\d+_\d+):\s*;?(.*)", ln)
        if m:
            label, note = m.group(1), m.group(2)
            j = i + 1
            while j < end and lines[j].strip().startswith(";") and "Loop" in lines[j]:
                note += " " + lines[j]
                j += 1
            cur = []
            h = re.search(r"=>\s*This (?:Inner )?Loop Header: Depth=(\d+)", note)
            if h:
                cur.append((label[2:], int(h.group(1))))
            for hh, d in re.findall(r"(?:in Loop: Header=|Parent Loop )(BB\d+_\d+) Depth=(\d+)", note):
                cur.append((hh, int(d)))
            i = j
            continue
        s = ln.strip()
        if s and not s.startswith(";") and not s.startswith("."):
            op = s.split()[0]
            for key in cur:
                loops.setdefault(key, collections.Counter())[op] += 1
        i += 1
    for (hh, d), c in loops.items():
        mf = sum(v for k, v in c.items() if k.startswith("v_mfma"))
        if not mf:
            continue
        valu = sum(v for k, v in c.items() if k.startswith("v_") and not k.startswith("v_mfma"))
        salu = sum(v for k, v in c.items() if k.startswith("s_") and not k.startswith("s_waitcnt") and k != "s_nop")
        lds = sum(v for k, v in c.items() if k.startswith("ds_"))
        vmem = sum(v for k, v in c.items() if k.startswith(("global_", "buffer_", "flat_", "scratch_")))
        trans = sum(v for k, v in c.items() if k.startswith(("v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt")))
        print(f"loop {hh} depth {d}: mfma {mf}  valu {valu} ({valu / mf:.1f}/mfma, {trans} transcendental)  salu {salu}  lds {lds}  vmem {vmem}  "
              f"waitcnt {c.get('s_waitcnt', 0)}  nop {c.get('s_nop', 0)}  barrier {c.get('s_barrier', 0)}")
        vs = sorted(((v, k) for k, v in c.items() if k.startswith("v_") and not k.startswith("v_mfma")), reverse=True)[:top]
        print("   " + "  ".join(f"{k}:{v}" for v, k in vs))


if __name__ == "__main__":
    main()
