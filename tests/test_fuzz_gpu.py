"""GPU parity: a fixed sample of the seeded random differential test (tests/fuzz_cases.py) - random shapes around the
tile edges x random feature combinations, HIP path vs the oracle.  Larger sweeps: `python tests/fuzz_cases.py --n 400`."""
import pytest

import fuzz_cases

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kind,n", [("dense", 48), ("varlen", 32), ("kvcache", 40), ("dense_long", 12), ("varlen_long", 12)])
def test_random_cases_agree_with_oracle(kind, n):
    fuzz_cases.run(kind, seed=20260928, n=n)
