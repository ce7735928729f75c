#!/usr/bin/env python3
"""Registers, LDS, scratch and the wavefronts per CU they allow, for every fbk kernel of a `hipcc -S --cuda-device-only` listing.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -o /tmp/fbk.s featurebase_amd/csrc/fbk.hip
    python scripts/kernel_resources.py /tmp/fbk.s > profiles/rNN_kernel_resources.txt

The numbers DESIGN.md quotes ("93 registers", "8.7 KiB of LDS per wave", "32 registers and 4 KiB") come from here.  Limits used
(MI355X_MICROARCH.md): 512 vector registers per lane and SIMD in granules of 8, 8 wavefronts per SIMD, 4 SIMDs and 160 KiB of LDS
per CU; a block's LDS is shared by its waves (workgroup size from .amdhsa / launch bounds is not in the listing: the LDS bound
is given per BLOCK, divide by the block's waves)."""
import re
import subprocess
import sys


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout.split("\n")
        return dict(zip(names, out))
    except Exception:
        return {n: n for n in names}


def main():
    text = open(sys.argv[1], errors="replace").read()
    rows = []
    for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", text, re.S):
        name, body = m.group(1), m.group(2)
        if not name.startswith("_ZN3fbk"):
            continue
        g = lambda k: int(re.search(r"\.amdhsa_%s (\d+)" % k, body).group(1))
        vg, sg, lds, scr = g("next_free_vgpr"), g("next_free_sgpr"), g("group_segment_fixed_size"), g("private_segment_fixed_size")
        waves_simd = min(8, 512 // max(8, (vg + 7) // 8 * 8))
        blocks_lds = (160 * 1024) // lds if lds else None
        rows.append((name, vg, sg, lds, scr, waves_simd, blocks_lds))
    dm = demangle([r[0] for r in rows])
    print("# vgpr sgpr lds_bytes_per_block scratch_bytes waves_per_SIMD_by_registers blocks_per_CU_by_LDS kernel")
    for name, vg, sg, lds, scr, ws, bl in sorted(rows, key=lambda r: dm[r[0]]):
        short = re.sub(r"\(.*", "", dm[name]).replace("void ", "")
        print(f"{vg:4d} {sg:4d} {lds:7d} {scr:5d} {ws:2d} {'-' if bl is None else bl:>4} {short}")


if __name__ == "__main__":
    main()
