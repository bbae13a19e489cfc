"""A short run of the differential fuzz driver (tests/fuzz_gpu.py) so that it stays exercised: random annotation sizes,
read modes, preset flags and overrides; rows and the records-in / records-out stream against the oracle."""
import pytest

from tests import fuzz_gpu

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [101, 202])
def test_fuzz_rounds_agree_with_the_oracle(seed):
    bad = fuzz_gpu.run(25, seed, verbose=False, read_counts=(200, 800), gene_counts=(30, 120))
    assert bad is None, bad
