"""CPU: the oracle (restatement) against the committed golden vectors, which
were produced by the reference's own code (tests/golden/make_golden.py)."""
import numpy as np
import pytest

from helpers import assert_slots_equal, golden_cases, oracle_env_from_case
from oracle import oracle as O

CASES = list(golden_cases())


def test_fixture_is_complete():
    assert len(CASES) == 80
    controls = {c["control"] for _, c, _ in CASES}
    assert controls == {0x01, 0x03, 0x07, 0x0F, 0x11, 0x13, 0x17, 0x1F}
    tot = sum(e["status"].size for _, _, e in CASES)
    fin = sum(int(np.count_nonzero(e["status"] == 1)) for _, _, e in CASES)
    blk = sum(int(np.count_nonzero(e["status"] == 2)) for _, _, e in CASES)
    assert tot > 30000 and fin > 2500 and blk > 2500, (tot, fin, blk)


@pytest.mark.parametrize("name,case,exp", CASES, ids=[c[0] for c in CASES])
def test_oracle_reproduces_reference(name, case, exp):
    got = O.expand(oracle_env_from_case(case), case["nodes"], threads=1)
    # the port and the reference run the same libm on the same host: everything bit-exact
    assert_slots_equal(got, exp, cost_rtol=0.0, what=name)
