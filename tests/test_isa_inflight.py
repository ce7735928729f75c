"""The shipped device code must never touch a vector register whose load is still in flight.

Several kernels issue their streaming loads as `asm volatile` and count the outstanding ones by hand (the one-pass
BSI kernels, the chunk rings of k_fold_scatter / k_rows_vs_filter): the compiler does not know those registers are
not ready, so a copy it places between the load and the hand-written wait would move stale data.  On the GPU that shows
as a parity failure (it did, twice, in round 3); here it is a build-time failure: scripts/check_inflight.py takes the
gfx950 code object out of libfbk.so — the very file the GPU box loads —, disassembles it and runs a data-flow analysis
of the outstanding vector-memory operations over every kernel (compiler-managed ones included).  No GPU needed."""
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
LIB = os.path.join(ROOT, "featurebase_amd", "csrc", "libfbk.so")


@pytest.fixture(scope="module")
def kernels():
    import check_inflight as C

    if not os.path.exists(os.path.join(C.LLVM, "llvm-objdump")):
        pytest.skip("no llvm-objdump in this image")
    if not os.path.exists(LIB):
        import __graft_entry__

        __graft_entry__.build()
    return C, C.parse(C.disassemble(LIB))


@pytest.mark.timeout(600)
def test_no_kernel_touches_a_register_in_flight(kernels):
    C, funcs = kernels
    bad = []
    for name, insns in funcs.items():
        bad += C.check_kernel(name, insns)
    assert not bad, "\n".join(bad[:20])


def test_the_hand_counted_kernels_are_in_the_binary(kernels):
    """(the check above must not pass because the kernels it is there for were renamed away)"""
    _, funcs = kernels
    names = "\n".join(funcs)
    for k in ("k_bsi_range_sum_halfILb0ELi3", "k_bsi_range_sum_halfILb1ELi3", "k_bsi_between_sum_partILi8ELi3", "k_fold_scatterILi1", "k_rows_vs_filter", "k_count_matrix_fusedqILb1ELb0", "k_count_matrix_mfma"):
        assert k in names, k
    # and they do contain hand-written waits with loads in flight behind them
    for k in ("k_bsi_range_sum_halfILb0ELi3", "k_fold_scatterILi1ELi0"):
        insns = next(v for n, v in funcs.items() if k in n)
        assert any(i.kind == "wait" and i.vm_wait not in (None, 0) for i in insns), k


def test_the_checker_finds_a_planted_copy():
    """A copy of a register between its load and the wait is reported; the same code with the copy behind the wait is clean."""
    import check_inflight as C

    def kernel(copy_first):
        lines = ["0000000000001000 <planted>:"]
        body = ["global_load_dwordx4 v[4:7], v[0:1], off", "global_load_dwordx4 v[8:11], v[0:1], off offset:16"]
        body += ["v_mov_b32_e32 v12, v4", "s_waitcnt vmcnt(1)"] if copy_first else ["s_waitcnt vmcnt(1)", "v_mov_b32_e32 v12, v4"]
        body += ["s_waitcnt vmcnt(0)", "v_add_u32_e32 v13, v8, v12", "s_endpgm"]
        for k, b in enumerate(body):
            op, _, ops = b.partition(" ")
            lines.append(f"\t{op} {ops}    // {0x1000 + 4 * k:012X}: 00000000")
        return C.parse("\n".join(lines))["planted"]

    assert C.check_kernel("planted", kernel(False)) == []
    bad = C.check_kernel("planted", kernel(True))
    assert len(bad) == 1 and "v_mov_b32_e32 v12, v4" in bad[0]
    # one load too few behind the register for the count in the wait
    short = kernel(False)
    short[2].vm_wait = 2
    assert C.check_kernel("planted", short)


def test_the_checker_follows_a_software_pipeline_around_its_loop():
    """Two register sets refilled in turn with one load each: waiting for vmcnt(1) before a set is used is right (one
    younger load in flight), vmcnt(2) is one load short — found through the back-edge, not in the first pass."""
    import check_inflight as C

    def kernel(n):
        body = [
            "global_load_dwordx4 v[4:7], v[0:1], off",      # set 0
            "global_load_dwordx4 v[8:11], v[0:1], off",     # set 1
            f"s_waitcnt vmcnt({n})",                         # <- loop header (0x1008)
            "v_add_u32_e32 v12, v4, v12",
            "global_load_dwordx4 v[4:7], v[0:1], off",
            f"s_waitcnt vmcnt({n})",
            "v_add_u32_e32 v12, v8, v12",
            "global_load_dwordx4 v[8:11], v[0:1], off",
            "s_cbranch_scc1 65528",                          # back to the header
            "s_waitcnt vmcnt(0)",
            "s_endpgm",
        ]
        lines = ["0000000000001000 <pipe>:"]
        for k, b in enumerate(body):
            op, _, ops = b.partition(" ")
            tail = "  <pipe+0x8>" if op.startswith("s_cbranch") else ""
            lines.append(f"\t{op} {ops}    // {0x1000 + 4 * k:012X}: 00000000{tail}")
        return C.parse("\n".join(lines))["pipe"]

    assert C.check_kernel("pipe", kernel(1)) == []
    bad = C.check_kernel("pipe", kernel(2))
    assert bad and all("may be outstanding" in b for b in bad)
    assert C.serialised_loops("pipe", kernel(1)) == [] and C.serialised_loops("pipe", kernel(0)) != []
