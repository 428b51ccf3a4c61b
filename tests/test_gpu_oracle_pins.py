"""The oracle's pins that need nothing of /root/reference, once more under the `gpu` marker: the driver's GPU suite then RE-PINS the checker it is about to use on that
box -- KAT-1 (src/vmisknn/mod.rs:229-310), the heap-order known answers (mod.rs:313-411), the committed golden fixture of the reference's example
(tests/golden/example_golden.npz), the fast builder == the literal prepare_hashmap, the restricted builder, literal == canonical on unique timestamps.  The same
functions run unmarked in tests/test_oracle_pins.py on the CPU suite."""
import pytest

import test_oracle_pins as P

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("fast", [False, True])
def test_kat1_should_train_and_predict(fast):
    P.test_kat1_should_train_and_predict(fast)


def test_heap_order_known_answers():
    P.test_heap_order_known_answers()


def test_panic_inputs_are_reported():
    P.test_panic_inputs_are_reported()


def test_fast_index_builder_equals_literal_prepare_hashmap():
    P.test_fast_index_builder_equals_literal_prepare_hashmap()


def test_literal_is_a_valid_instance_of_canonical_when_timestamps_are_unique():
    P.test_literal_is_a_valid_instance_of_canonical_when_timestamps_are_unique()


def test_canonical_topn_is_sorted_and_excludes_current_item():
    P.test_canonical_topn_is_sorted_and_excludes_current_item()


def test_golden_fixture_is_reproduced_by_the_oracle():
    P.test_golden_fixture_is_reproduced_by_the_oracle()


def test_restricted_parallel_builder_equals_prepare_hashmap():
    P.test_restricted_parallel_builder_equals_prepare_hashmap()
