#!/usr/bin/env python3
"""Static instruction mix of one kernel in a `hipcc -S --cuda-device-only` listing.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -o /tmp/fbk.s featurebase_amd/csrc/fbk.hip
    python scripts/isa_stats.py /tmp/fbk.s k_bsi_between_sum_partILi4 [--dump out.s]

Prints registers, LDS, scratch and the number of VALU / SALU / LDS / VMEM / branch instructions per basic
block (label), largest first: a first look at where a kernel's issue slots go, before any GPU time is spent.
"""
import re
import sys


def main():
    path, pat = sys.argv[1], sys.argv[2]
    dump = sys.argv[sys.argv.index("--dump") + 1] if "--dump" in sys.argv else None
    lines = open(path, errors="replace").read().split("\n")
    start = None
    for i, l in enumerate(lines):
        if l.endswith(":") and pat in l and l.startswith("_Z") and not l.startswith("."):
            start = i
            name = l[:-1]
            break
        m = re.match(r"^(_Z\S*%s\S*):" % re.escape(pat), l)
        if m:
            start, name = i, m.group(1)
            break
    if start is None:
        sys.exit("kernel not found")
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    body = lines[start:end]
    if dump:
        open(dump, "w").write("\n".join(l for l in body if not l.strip().startswith((".loc", ".cfi", ";"))))
    blocks, cur = [], ["entry", {}]
    for l in body[1:]:
        s = l.strip()
        if not s or s.startswith((";", ".loc", ".cfi", ".p2align", ".file")):
            continue
        if re.match(r"^\.L\w+:", s):
            blocks.append(cur)
            cur = [s.split(":")[0], {}]
            continue
        op = s.split()[0]
        if op.startswith("v_mfma") or op.startswith("v_smfma"):
            k = "mfma"
        elif op.startswith("v_"):
            k = "valu"
        elif op.startswith(("s_cbranch", "s_branch")):
            k = "branch"
        elif op.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_sleep")):
            k = "wait"
        elif op.startswith(("s_load", "s_buffer_load")):
            k = "smem"
        elif op.startswith("s_"):
            k = "salu"
        elif op.startswith("ds_"):
            k = "lds"
        elif op.startswith(("global_", "flat_", "buffer_", "scratch_")):
            k = "vmem" if not op.startswith("scratch_") else "scratch"
        else:
            continue
        cur[1][k] = cur[1].get(k, 0) + 1
    blocks.append(cur)
    tot = {}
    for _, d in blocks:
        for k, v in d.items():
            tot[k] = tot.get(k, 0) + v
    print(name)
    for l in lines[end : end + 400]:
        m = re.search(r"\.(num_vgpr|num_agpr|numbered_sgpr|private_seg_size), (\d+)", l)
        if m and name in l:
            print("  ", m.group(1), m.group(2))
    print("   static totals:", dict(sorted(tot.items())))
    blocks.sort(key=lambda b: -sum(b[1].values()))
    for lab, d in blocks[:25]:
        print(f"   {lab:14s}", dict(sorted(d.items())))


if __name__ == "__main__":
    main()
