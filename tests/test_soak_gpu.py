"""A short fixed-seed slice of the randomised call-sequence soak (tests/soak_parity.py) in the GPU suite; the long runs
are logged under profiles/."""
import pytest

import soak_parity


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [11, 12])
def test_random_call_sequences_match_the_oracle(hip, oracle, seed):
    res = soak_parity.soak(hip, oracle, seed, cases=30)
    assert res["cases"] == 30 and res["env_steps_compared"] > 0
