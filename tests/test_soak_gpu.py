"""A short fixed-seed slice of the randomised call-sequence soak (tests/soak_parity.py) in the GPU suite; the long runs
are logged under profiles/."""
import pytest

import soak_parity


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [11, 12])
def test_random_call_sequences_match_the_oracle(hip, oracle, seed):
    res = soak_parity.soak(hip, oracle, seed, cases=30)
    assert res["cases"] == 30 and res["env_steps_compared"] > 0


@pytest.mark.gpu
def test_the_soak_notices_a_one_ulp_difference(hip, oracle, monkeypatch):
    """Negative control: the product is handed actions that differ from the oracle's in ONE float by one ulp."""
    import numpy as np

    real_put = soak_parity.put

    def put_one_ulp_off(dst, src):
        src = np.array(src, dtype=np.float32, copy=True)
        flat = src.reshape(-1).view(np.uint32)
        flat[0] ^= 1  # lowest mantissa bit of the first action
        real_put(dst, src)

    monkeypatch.setattr(soak_parity, "put", put_one_ulp_off)
    with pytest.raises(AssertionError, match="differ"):
        soak_parity.soak(hip, oracle, 13, cases=30)
