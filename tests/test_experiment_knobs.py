"""The timing-experiment switches that make a kernel skip work (pair_ablate, pair_stamp, matrix_fused_ablate: WRONG results by
design) exist only in builds with -DFBK_EXPERIMENTS (scripts/ build their own variant into build_variants/).  The product
library has neither the option names nor the device branches, so no FBK_* environment variable can corrupt a count."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KNOBS = ("pair_ablate", "pair_stamp", "pair_spw", "matrix_fused_ablate", "ring_geom", "ring_nt", "ring_debug", "ring_flags")  # (ring_*: k_icount3, round 6: bit-exact and slower, an experiment)


def test_product_library_has_no_experiment_option_names():
    from featurebase_amd import lib as L

    blob = open(L.LIB_PATH, "rb").read()
    for k in KNOBS:
        assert k.encode() + b"\0" not in blob, f"option {k} is compiled into the product library"


@pytest.mark.gpu
def test_experiment_options_are_rejected(gpu_ctx):
    from featurebase_amd import lib as L

    for k in KNOBS:
        with pytest.raises(L.FbkError) as e:
            gpu_ctx.set_option(k, 1)
        assert e.value.code == L.FBK_E_INVALID, (k, e.value)
        with pytest.raises(L.FbkError):
            gpu_ctx.get_option(k)


@pytest.mark.gpu
def test_ablate_environment_variables_change_nothing():
    """A process whose environment carries FBK_PAIR_ABLATE / FBK_PAIR_STAMP / FBK_MATRIX_FUSED_ABLATE computes the same
    pair counts and the same mixed-row count matrix as the numpy reference."""
    code = r'''
import sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
from featurebase_amd.roaring import Context
import datagen as D
rng = np.random.default_rng(11)
S, R = 4, 3   # shards, rows per side and shard (rows of one shard share their container keys)
ra = [D.random_row(rng, s, p_missing=0.1) for s in range(S) for _ in range(R)]
rb = [D.random_row(rng, s, p_missing=0.1) for s in range(S) for _ in range(R)]
def words(row, s):
    w = np.zeros((16, 1024), dtype=np.uint64)
    for k, c in row.items():
        w[k - s * 16] = c.words()
    return w
wa = np.stack([words(r, i // R) for i, r in enumerate(ra)])
wb = np.stack([words(r, i // R) for i, r in enumerate(rb)])
ctx = Context(0)
A, B = ctx.upload([D.to_fbk_row(r) for r in ra]), ctx.upload([D.to_fbk_row(r) for r in rb])
idx = np.arange(S * R)
exp = np.array([int(np.bitwise_count(wa[i] & wb[i]).sum()) for i in idx], dtype=np.uint64)
for pk in (1, 2):
    ctx.set_option("pair_kernels", pk)
    got = ctx.intersection_count(A, idx, B, idx)
    assert (got == exp).all(), (pk, got, exp)
rows = idx.reshape(S, R)
em = np.zeros((R, R), dtype=np.uint64)
for s in range(S):
    for i in range(R):
        for j in range(R):
            em[i, j] += int(np.bitwise_count(wa[s * R + i] & wb[s * R + j]).sum())
for fused in (1, 0):
    ctx.set_option("matrix_fused", fused)
    m = ctx.count_matrix(A, rows, B, rows)
    assert (np.asarray(m) == em).all(), (fused, m, em)
print("ok")
''' % (ROOT, os.path.join(ROOT, "tests"))
    env = dict(os.environ, FBK_PAIR_ABLATE="255", FBK_PAIR_STAMP="3", FBK_MATRIX_FUSED_ABLATE="31")
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 0 and "ok" in p.stdout, p.stderr[-3000:]
