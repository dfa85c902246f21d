"""Static instruction mix per basic block of one kernel in a hipcc --save-temps .s file.

    python tools/isa_blocks.py file.s kernel_symbol_substring [min_valu]

Prints, per label: line range, VALU / SALU / LDS / VMEM / MFMA counts and the loop-nesting comment the compiler left,
so that the blocks of the generation loop can be weighted by hand (tries per phase, phases per generation).
"""
import re
import sys


def main():
    path, sym = sys.argv[1], sys.argv[2]
    minv = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and sym in l.split(":")[0] and l.rstrip().split(";")[0].strip().endswith(":"))
    end = next(i for i in range(start, len(lines)) if ".end_amdhsa_kernel" in lines[i] or lines[i].startswith(".Lfunc_end"))
    blocks = []
    cur = {"label": "entry", "start": start, "v": 0, "s": 0, "l": 0, "m": 0, "mf": 0, "note": ""}
    for i in range(start + 1, end):
        t = lines[i]
        m = re.match(r"^(\.LBB[0-9_]+):\s*(;.*)?", t)
        if m:
            cur["end"] = i - 1
            blocks.append(cur)
            cur = {"label": m.group(1), "start": i, "v": 0, "s": 0, "l": 0, "m": 0, "mf": 0, "note": (m.group(2) or "").strip()}
            continue
        s = t.strip()
        if s.startswith("v_mfma"):
            cur["mf"] += 1
        elif s.startswith("v_"):
            cur["v"] += 1
        elif s.startswith("s_"):
            cur["s"] += 1
        elif s.startswith("ds_"):
            cur["l"] += 1
        elif re.match(r"^(global|flat|buffer|scratch)_", s):
            cur["m"] += 1
        elif s.startswith(";") and "Loop" in s and not cur["note"]:
            cur["note"] = s
    cur["end"] = end
    blocks.append(cur)
    tot = 0
    for b in blocks:
        tot += b["v"]
        if b["v"] >= minv:
            print("%-14s %5d-%-5d valu %4d salu %4d lds %3d vmem %3d mfma %3d  %s" % (b["label"], b["start"] + 1 - start, b["end"] + 1 - start, b["v"], b["s"], b["l"], b["m"], b["mf"], b["note"][:70]))
    print("static VALU total", tot)


if __name__ == "__main__":
    main()
