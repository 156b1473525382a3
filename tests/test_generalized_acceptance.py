"""The reference's generalized-M acceptance tests (tests/generalized_problems.py) on the CPU oracle -- the checker the GPU tests of
tests/test_gpu_generalized_acceptance.py compare against has to satisfy the reference's bounds itself."""
import numpy as np
import pytest

import generalized_problems as G


@pytest.mark.parametrize("name", sorted(G.CASES))
def test_oracle_meets_the_reference_bounds(oracle, name):
    A, Md, B, params, bound, seed = G.build(name)
    out = oracle.solve(G.csr(A), B, oracle.params({**params, "rng_mode": 1, "seed": seed}), M=G.diag_csr(Md))
    assert out["ret"] == 0
    G.check_acceptance(A, Md, B, out["V"], out["T"], abs(float(out["res_hist"][-1])), int(out["trips"]), bound)
