"""GPU box: the tests that PIN the oracle to the reference's recorded MATLAB/quadprog answers (tests/test_oracle_golden.py), collected a
second time under the `gpu` marker -- the driver's round-end run is `pytest -m gpu`, which deselects the CPU suite: with this module its
record (GPUTEST_rNN.json) carries the oracle pin itself, on the box whose host cores also run the oracle as the checker of every parity test
and as bench.py's cpu_baseline.  No GPU work happens here; the functions are the CPU suite's own."""
import pytest

from test_oracle_golden import (test_model_matrices_bit_exact, test_step14_known_answers,  # noqa: F401
                                test_step2_closed_loop_known_answers)

pytestmark = pytest.mark.gpu
