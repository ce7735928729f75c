"""The numpy-only input generator of the benchmark scripts (datagen.fbk_container_of_vals) picks
the same encoding, bytes and cardinality as the oracle's Container.optimize() restatement."""
import numpy as np

import datagen as D


def test_fbk_container_of_vals_matches_oracle_optimize(oracle):
    O = oracle
    for seed, (d, rs) in enumerate([(0.001, False), (0.01, False), (0.05, False), (0.0624, False), (0.0626, False), (0.3, False), (0.5, False), (0.01, True), (0.1, True), (0.4, True), (0.6, True)]):
        for rep in range(4):
            vals = D.mixed_vals_for_density(D.rng_for(77, seed, rep), d, rs)
            if vals.size == 0:
                continue
            want = D.mixed_container_for_density(D.rng_for(77, seed, rep), d, rs)  # the oracle path, same draws
            got = D.fbk_container_of_vals(vals)
            assert got.typ == want.typ and got.n == want.n and got.length == want.length
            assert np.array_equal(np.asarray(got.data).reshape(-1), np.asarray(want.data()).reshape(-1))
    # policy edges: runs == n/2 is a run, n == 4096 is a bitmap, n == 4095 spread out is an array
    pairs = np.arange(0, 8000, 4)
    v = np.sort(np.concatenate([pairs, pairs + 1]))  # 2000 runs of 2: runs == n / 2
    assert D.fbk_container_of_vals(v).typ == O.RUN
    v = np.arange(0, 4096 * 3, 3)
    assert D.fbk_container_of_vals(v).typ == O.BITMAP and D.fbk_container_of_vals(v[:-1]).typ == O.ARRAY
