"""CPU-only checks of the drop-in boundary: libfbk.so loads without a GPU, exports every
symbol include/fbk.h declares, the ctypes table covers exactly that set, and the
library fails loudly (no CPU fallback) when no device is present."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "fbk.h")


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return set(re.findall(r"\b(fbk_[a-z0-9_]+)\s*\(", src))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g

    g.build()
    from featurebase_amd import lib as L

    return L


def test_header_symbols_exported(lib):
    decl = declared_symbols()
    assert len(decl) >= 20
    out = subprocess.check_output(["nm", "-D", "--defined-only", lib.LIB_PATH], text=True)
    exported = set(re.findall(r" T (fbk_[a-z0-9_]+)", out))
    assert decl <= exported, f"declared but not exported: {sorted(decl - exported)}"
    assert exported <= decl, f"exported but not declared in include/fbk.h: {sorted(exported - decl)}"


def test_ctypes_table_matches_header(lib):
    assert set(lib.SIGNATURES) == declared_symbols()
    lib.load()


def test_struct_layout_matches_header(lib):
    # fbk_container_desc: u64 key, u64 off, u32 row, u32 len, i32 n, u8 type, u8 pad[3]
    assert C.sizeof(lib.ContainerDesc) == 32
    assert lib.ContainerDesc.off.offset == 8 and lib.ContainerDesc.row.offset == 16
    assert lib.ContainerDesc.n.offset == 24 and lib.ContainerDesc.type.offset == 28


def test_abi_version(lib):
    assert lib.load().fbk_abi_version() == 6


def test_no_device_fails_loudly(lib):
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from featurebase_amd.roaring import Context

    with pytest.raises(lib.FbkError) as ei:
        Context(0)
    assert ei.value.code == lib.FBK_E_NODEVICE
    assert "no CPU fallback" in str(ei.value)


def test_null_arguments_are_errors_not_crashes(lib):
    l = lib.load()
    assert l.fbk_open(0, 0, None) == lib.FBK_E_INVALID
    assert l.fbk_device_count(None) == lib.FBK_E_INVALID
    assert l.fbk_batch_info(None, None, None, None, None) == lib.FBK_E_INVALID
    assert l.fbk_close(None) == lib.FBK_OK
    assert l.fbk_last_error(None) is not None


def test_product_never_imports_oracle():
    """The product path must not route through the oracle (or any CPU fallback)."""
    pkg = os.path.join(ROOT, "featurebase_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".hpp")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "oracle" not in txt.lower(), f"{f} mentions the oracle"


def test_option_names_in_the_header_are_the_library_s():
    """The list of tuning knobs in include/fbk.h (the comment above fbk_set_option) names exactly the options the library
    registers (kOptions in fbk.hip), the three experiment-only ones aside: a knob that was removed or added without the header
    following is a documentation bug a maintainer trips over."""
    import re

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hip = open(os.path.join(root, "featurebase_amd", "csrc", "fbk.hip")).read()
    table = hip[hip.index("const OptionDesc kOptions[] = {"):]
    table = table[: table.index("};")]
    registered = set(re.findall(r'\{"([a-z_0-9]+)", &FbkOptions::', table))
    experiments = {"pair_ablate", "pair_stamp", "pair_spw", "matrix_fused_ablate", "ring_geom", "ring_nt", "ring_debug", "ring_flags"}  # (-DFBK_EXPERIMENTS builds only)
    header = open(os.path.join(root, "include", "fbk.h")).read()
    doc = header[header.index("/* Options ("): header.index("int32_t fbk_set_option")]
    words = set(re.findall(r"\b[a-z]+(?:_[a-z0-9]+)+\b", doc))
    missing = sorted((registered - experiments) - words)
    assert not missing, f"options the header does not name: {missing}"
    stale = sorted(w for w in words if (w.startswith(("pair_", "matrix_", "bsi_", "setop_", "fold_", "upload_", "query_", "dense_", "topk_")) and w not in registered))
    assert not stale, f"names in the header that are not options (any more): {stale}"
    # round 5's prune: the product library registers at most 22 options (the experiment builds' three aside)
    assert len(registered - experiments) <= 22, sorted(registered - experiments)


def test_no_exception_can_leave_an_entry_point():
    """Every `int32_t fbk_*` definition of the library is a function-try-block closed by FBK_ABI_CATCH / FBK_ABI_CATCH_GROUP
    (fbk.hip): the host side uses std::vector / std::string / std::thread, the caller is cgo, and SURVEY 8b's boundary has "no
    exceptions / aborts across it".  (fbk_abi_version and fbk_last_error allocate nothing that could throw after the first
    call and return no status.)"""
    csrc = os.path.join(ROOT, "featurebase_amd", "csrc")
    n = 0
    for f in sorted(os.listdir(csrc)):
        if not f.endswith((".hip", ".inc")):
            continue
        lines = open(os.path.join(csrc, f)).read().split("\n")
        i = 0
        while i < len(lines):
            if lines[i].startswith("int32_t fbk_") and not lines[i].startswith("int32_t fbk_abi_version") and not lines[i].rstrip().endswith(";"):
                j = i
                while not lines[j].rstrip().endswith("{"):
                    j += 1
                assert lines[j].rstrip().endswith(") try {"), f"{f}:{i + 1}: {lines[i]}"
                k = j + 1
                while not lines[k].startswith("}"):
                    k += 1
                assert lines[k].startswith("} FBK_ABI_CATCH"), f"{f}:{k + 1}: {lines[k]}"
                n += 1
                i = k
            i += 1
    assert n == len(declared_symbols()) - 2, (n, len(declared_symbols()))


def test_cgo_snippets_of_integration_md_call_the_header_s_functions():
    """Go is not installed here, so the cgo shim of INTEGRATION.md cannot be compiled; what CAN be checked is that every `C.fbk_*(...)`
    call in it names a function include/fbk.h declares and passes as many arguments as the prototype has."""
    hdr = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(?:int32_t|const char\*)\s+(fbk_[a-z0-9_]+)\s*\(([^;]*?)\)\s*;", hdr, flags=re.S):
        args = m.group(2).strip()
        protos[m.group(1)] = 0 if args in ("", "void") else args.count(",") + 1
    assert len(protos) == len(declared_symbols())
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    calls, bad = 0, []
    for m in re.finditer(r"C\.(fbk_[a-z0-9_]+)\(", md):
        j, depth, n_args, cur = m.end(), 1, 0, ""
        while depth and j < len(md):
            c = md[j]
            if c in "([{":
                depth += 1
            elif c in ")]}":
                depth -= 1
            if depth == 1 and c == ",":
                n_args, cur = n_args + 1, ""
            elif depth >= 1:
                cur += c
            j += 1
        n_args += 1 if cur.strip() else 0
        calls += 1
        if protos.get(m.group(1)) != n_args:
            bad.append((m.group(1), protos.get(m.group(1)), n_args))
    assert calls >= 20 and not bad, bad
