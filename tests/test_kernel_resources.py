"""No kernel of the built library spills: a spilled vector register is a scratch (= memory) round trip per use inside loops that were
hand-tuned to stay in registers (k_icount2's item loop once compiled to 247 spilled registers — fbk_pair_kernels.hip.h).  Read from
the metadata of the gfx950 code object inside libfbk.so; CPU only.  The one known exception is listed with its size so that it cannot
grow unnoticed."""
import os
import re
import subprocess
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"
# kernel name prefix (demangled) -> bytes of scratch it may use
ALLOWED = {
    "void fbk::k_fold_scatter<3, ": 12,  # the n-way Difference fold at its 64-register cap (1024-thread blocks): 4 registers spilled outside the chunk loop
}


def kernel_metadata(lib):
    with tempfile.TemporaryDirectory(prefix="fbk_res_") as tmp:
        fat, co = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "gfx950.co")
        subprocess.check_call([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat])
        subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}", f"--targets={TARGET}", f"--output={co}"])
        notes = subprocess.check_output([f"{LLVM}/llvm-readelf", "--notes", co], text=True)
    out = []
    for blk in notes.split("  - .agpr_count:")[1:]:
        g = lambda k: re.search(r"\.%s:\s+(\S+)" % k, blk)
        name = g("name").group(1)
        out.append((name, int(g("private_segment_fixed_size").group(1)), int(g("vgpr_spill_count").group(1)), int(g("sgpr_spill_count").group(1)), int(g("vgpr_count").group(1))))
    return out


def test_no_kernel_of_this_repository_spills():
    import __graft_entry__ as g

    g.build()
    from featurebase_amd import lib as L

    kernels = [k for k in kernel_metadata(L.LIB_PATH) if k[0].startswith("_ZN3fbk")]
    assert len(kernels) >= 90, len(kernels)
    names = subprocess.run(["c++filt"], input="\n".join(k[0] for k in kernels), capture_output=True, text=True, check=True).stdout.split("\n")
    bad = []
    for (mangled, scratch, vspill, sspill, vgpr), name in zip(kernels, names):
        cap = max((v for p, v in ALLOWED.items() if name.startswith(p)), default=0)
        if scratch > cap or (cap == 0 and (vspill or sspill)):
            bad.append((name[:100], scratch, vspill, sspill, vgpr))
    assert not bad, bad
