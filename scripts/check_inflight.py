#!/usr/bin/env python3
"""Static check of the SHIPPED device code: no instruction touches a vector register whose load is still in flight.

Some kernels of this library issue their streaming loads as `asm volatile` and count the outstanding ones by hand
(`s_waitcnt vmcnt(N)` written in the source: fbk_bsi_kernels.hip.h plane_request / plane_landed, the ring of
fbk_fold_kernels.hip.h) because the compiler's own s_waitcnt insertion serialises a software pipeline it cannot count.
The price: the compiler does not know those registers are not ready, so a register copy it decides to place between
the load and the hand-written wait (a PHI copy between two request sites, a live-range split) would move stale data —
silently.  This script makes that a BUILD-TIME failure instead of a wrong answer:

  * the gfx950 code object is taken out of featurebase_amd/csrc/libfbk.so (the file that is loaded on the GPU box) and
    disassembled with llvm-objdump;
  * per kernel a data-flow analysis over the control-flow graph keeps, for every vector register that may have a load
    outstanding, the least number of vector-memory operations issued since (they return in order on this target —
    the compiler's own s_waitcnt insertion relies on the same): a load resets its destination registers to 0 and
    ages the others, `s_waitcnt vmcnt(N)` retires everything with at least N younger operations, and any other
    instruction that reads OR writes a register still in the set is a violation (except a younger LOAD into the same
    registers: the younger one wins).  FLAT operations return out of order: after one, only vmcnt(0) retires.

    python scripts/check_inflight.py [--lib path/to/libfbk.so] [kernel name regex ...]
    python scripts/check_inflight.py --lint-serialised [regex ...]     # performance lint: loops whose waits are all vmcnt(0)

Without a regex every kernel of the library is checked (compiler-managed loads must pass as well: the check is a
model of the hardware rule, not of the source).  Exit status 1 and one line per violation if anything is found.
tests/test_isa_inflight.py runs it on every build.
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"

_reg1 = re.compile(r"(?<![\w\[])v(\d+)\b")
_regn = re.compile(r"(?<![\w])v\[(\d+):(\d+)\]")
_line = re.compile(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):")
_func = re.compile(r"^[0-9a-f]+ <(\S+)>:$")
_target = re.compile(r"<[^>+]+\+0x([0-9a-fA-F]+)>|<[^>+]+>$")


def disassemble(lib):
    with tempfile.TemporaryDirectory(prefix="fbk_isa_") as tmp:
        fat, co = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "gfx950.co")
        subprocess.check_call([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat])
        subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}", f"--targets={TARGET}", f"--output={co}"])
        return subprocess.check_output([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", co], text=True)


def vregs(text):
    out = set()
    for a, b in _regn.findall(text):
        out.update(range(int(a), int(b) + 1))
    out.update(int(x) for x in _reg1.findall(text))
    return out


def split_operands(ops):
    parts, depth, cur = [], 0, ""
    for ch in ops:
        if ch == "[":
            depth += 1
        elif ch == "]":
            depth -= 1
        if ch == "," and depth == 0:
            parts.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        parts.append(cur.strip())
    return parts


class Insn:
    __slots__ = ("addr", "op", "ops", "kind", "dest", "uses", "target", "vm_wait")

    def __init__(self, addr, op, ops, base, tail=""):
        self.addr, self.op, self.ops = addr, op, ops
        self.target, self.vm_wait, self.dest = None, None, frozenset()
        self.kind = "alu"
        parts = split_operands(ops)
        if op.startswith(("global_load_lds", "buffer_load_lds")) or (op.startswith(("buffer_load", "global_load")) and " lds" in " " + ops):
            self.kind, self.uses = "load", frozenset(vregs(ops))  # DMA into LDS: no destination registers
        elif op.startswith(("global_load", "flat_load", "buffer_load", "scratch_load")):
            self.kind = "load"
            self.dest = frozenset(vregs(parts[0])) if parts else frozenset()
            self.uses = frozenset(vregs(",".join(parts[1:])))
        elif op.startswith(("global_atomic", "flat_atomic", "buffer_atomic")):
            ret = " glc" in " " + ops or " sc0" in " " + ops
            self.kind = "load" if ret else "store"
            self.dest = frozenset(vregs(parts[0])) if ret and parts else frozenset()
            self.uses = frozenset(vregs(",".join(parts[1:] if ret else parts)))
        elif op.startswith(("global_store", "flat_store", "buffer_store", "scratch_store")):
            self.kind, self.uses = "store", frozenset(vregs(ops))
        else:
            self.uses = frozenset(vregs(ops))
            if op.startswith("s_waitcnt"):
                self.kind = "wait"
                m = re.search(r"vmcnt\((\d+)\)", ops)
                self.vm_wait = int(m.group(1)) if m else None
            elif op.startswith(("s_cbranch", "s_branch")):
                self.kind = "cbranch" if op.startswith("s_cbranch") else "branch"
                m = re.search(r"\+0x([0-9a-fA-F]+)>\s*$", tail)
                self.target = base + int(m.group(1), 16) if m else (base if re.search(r"<[^>+]+>\s*$", tail) else None)
            elif op.startswith(("s_endpgm", "s_trap")):
                self.kind = "end"
            elif op.startswith(("s_setpc", "s_swappc", "s_call")):
                self.kind = "indirect"


def parse(dis):
    funcs, name, base, body = {}, None, 0, []
    for l in dis.split("\n"):
        m = _func.match(l)
        if m:
            if name:
                funcs[name] = body
            name, base, body = m.group(1), int(l.split()[0], 16), []
            continue
        m = _line.match(l)
        if m and name:
            body.append(Insn(int(m.group(3), 16) - base, m.group(1), m.group(2), 0, l))
    if name:
        funcs[name] = body
    # long branches (beyond the 16-bit offset of s_branch): s_getpc_b64 s[a:b]; s_add_u32 sa, sa, IMM; s_addc_u32 sb, sb, IMM;
    # s_setpc_b64 s[a:b] — the target is the address after s_getpc plus IMM
    for body in funcs.values():
        for k, ins in enumerate(body):
            if ins.kind == "indirect" and k >= 3 and body[k - 3].op == "s_getpc_b64" and body[k - 2].op == "s_add_u32" and body[k - 1].op == "s_addc_u32":
                lo = int(body[k - 2].ops.split(",")[-1].strip(), 0)
                hi = int(body[k - 1].ops.split(",")[-1].strip(), 0)
                off = (hi << 32 | lo) & 0xFFFFFFFFFFFFFFFF
                if off >= 1 << 63:
                    off -= 1 << 64
                ins.kind, ins.target = "branch", body[k - 3].addr + 4 + off
    return funcs


def check_kernel(name, insns):
    """Returns a list of violation strings.

    State at a program point: for every vector register that MAY have a load outstanding, the smallest number of
    vector-memory operations that can have been issued since that load (over all paths reaching the point), plus a
    flag "a FLAT operation may be outstanding" (those also count on lgkmcnt and return out of order: only vmcnt(0)
    retires anything then, as the compiler assumes).  `s_waitcnt vmcnt(N)` retires the registers with at least N
    younger operations.  Join = union of the registers with the minimum of the counts: a monotone data-flow problem,
    iterated to its fixed point."""
    index = {ins.addr: i for i, ins in enumerate(insns)}
    n = len(insns)
    state = [None] * n  # (dict reg -> younger ops, flat flag)
    bad = {}

    def join(i, st):
        cur = state[i]
        if cur is None:
            state[i] = (dict(st[0]), st[1])
            return True
        changed = False
        d = cur[0]
        for r, y in st[0].items():
            if r not in d or y < d[r]:
                d[r] = y
                changed = True
        if st[1] and not cur[1]:
            state[i] = (d, True)
            changed = True
        return changed

    if n == 0:
        return []
    state[0] = ({}, False)
    work = [0]
    while work:
        i = work.pop()
        d, flat = state[i]
        ins = insns[i]
        touched = (ins.uses | (frozenset() if ins.kind == "load" else ins.dest)) & d.keys()  # (a younger LOAD into the same registers is fine: returns are in order)
        if touched and ins.kind != "wait":
            bad.setdefault((ins.addr, "touch"), f"{name}+{ins.addr:#x}: `{ins.op} {ins.ops}` touches v{sorted(touched)} while a load into them may be outstanding")
        out, oflat = d, flat
        if ins.kind in ("load", "store"):
            out = {r: min(y + 1, 64) for r, y in d.items()}
            for r in ins.dest:
                out[r] = 0
            oflat = flat or ins.op.startswith("flat_")
        elif ins.kind == "wait" and ins.vm_wait is not None:
            if ins.vm_wait == 0:
                out, oflat = {}, False
            elif not flat:
                out = {r: y for r, y in d.items() if y < ins.vm_wait}
        succ = []
        if ins.kind == "end":
            pass
        elif ins.kind == "indirect":
            bad.setdefault((ins.addr, "indirect"), f"{name}+{ins.addr:#x}: indirect control flow, not modelled")
        elif ins.kind in ("branch", "cbranch"):
            if ins.target is None or ins.target not in index:
                bad.setdefault((ins.addr, "target"), f"{name}+{ins.addr:#x}: branch target not found")
            else:
                succ.append(index[ins.target])
            if ins.kind == "cbranch" and i + 1 < n:
                succ.append(i + 1)
        elif i + 1 < n:
            succ.append(i + 1)
        for j in succ:
            if join(j, (out, oflat)):
                work.append(j)
    return list(bad.values())


def serialised_loops(name, insns):
    """A performance lint, not a correctness check: loops (a backward branch and the address range it closes) that issue
    vector loads into registers and in which EVERY s_waitcnt on vmcnt waits for 0 — whatever the source meant to keep in
    flight across iterations, each pass drains it.  That is how the compiler's wait insertion ends where it cannot count
    (conditional loads, a merge of "loaded before the loop" with "reloaded in the last pass"); whether it costs anything
    depends on how much arithmetic the loop has to hide (the one-pass BSI kernels: 199 -> 137 us; k_bsi_sum_slot: nothing).
    Returns (loop start, loop end, loads, waits) tuples."""
    index = {ins.addr: i for i, ins in enumerate(insns)}
    out = []
    for i, ins in enumerate(insns):
        if ins.kind in ("branch", "cbranch") and ins.target is not None and ins.target in index and ins.target <= ins.addr:
            body = insns[index[ins.target] : i + 1]
            loads = [b for b in body if b.kind == "load" and b.dest]
            waits = [b for b in body if b.kind == "wait" and b.vm_wait is not None]
            if len(loads) >= 2 and waits and all(w.vm_wait == 0 for w in waits):
                out.append((ins.target, ins.addr, len(loads), len(waits)))
    # innermost first, drop loops that contain an already reported one
    out.sort(key=lambda t: t[1] - t[0])
    kept = []
    for t in out:
        if not any(k[0] >= t[0] and k[1] <= t[1] for k in kept):
            kept.append(t)
    return kept


def main(argv):
    lib = os.path.join(ROOT, "featurebase_amd", "csrc", "libfbk.so")
    if "--lib" in argv:
        k = argv.index("--lib")
        lib = argv[k + 1]
        argv = argv[:k] + argv[k + 2:]
    lint = "--lint-serialised" in argv
    argv = [a for a in argv if a != "--lint-serialised"]
    pats = [re.compile(p) for p in argv] or [re.compile(".")]
    funcs = parse(disassemble(lib))
    if lint:
        for name, insns in funcs.items():
            if any(p.search(name) for p in pats) and "fbk" in name:
                for a, b, nl, nw in serialised_loops(name, insns):
                    print(f"{name[:90]}  loop +{a:#x}..+{b:#x}: {nl} vector loads, {nw} waits, all vmcnt(0)")
        return 0
    n, out = 0, []
    for name, insns in funcs.items():
        if not any(p.search(name) for p in pats) or not insns:
            continue
        n += 1
        out += check_kernel(name, insns)
    for l in out:
        print(l)
    print(f"{n} kernels checked, {len(out)} violations", file=sys.stderr)
    return 1 if out else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
