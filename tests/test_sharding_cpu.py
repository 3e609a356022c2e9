"""CPU: work-item assignment of the batch registration (BASELINE.json configs[4])."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("icp_sharding_t", os.path.join(ROOT, "icp-proposal_amd", "sharding.py"))
sharding = importlib.util.module_from_spec(spec)
spec.loader.exec_module(sharding)


@pytest.mark.parametrize("T,C,W", [(10, 10, 8), (10, 10, 2), (10, 10, 1), (3, 3, 2), (5, 7, 8), (1, 10, 8), (100, 1, 8), (7, 3, 4), (2, 2, 8)])
def test_target_major_assignment_is_complete_and_balanced(T, C, W):
    a = sharding.assign_target_major(T, C, W)
    assert len(a) == W and sorted(sum(a, [])) == list(range(T * C))
    sizes = [len(x) for x in a]
    assert max(sizes) - min(sizes) <= 1
    assert all(x == sorted(x) for x in a)


def test_ten_by_ten_over_eight_ranks_meets_two_targets_each():
    """StdIcpVsChainICPrandomInitComparisonAll-style job: 100 items -> 13/13/13/13/12/12/12/12, every rank ONE whole target plus a
    piece of one of the two split ones (round-robin dealing made every rank meet all ten: ten contexts, ten cold first searches)."""
    a = sharding.assign_target_major(10, 10, 8)
    assert [len(x) for x in a] == [13, 13, 13, 13, 12, 12, 12, 12]
    assert [len(set(k // 10 for k in x)) for x in a] == [2] * 8
    rr = sharding.assign_work_items(100, 8)
    assert [len(set(k // 10 for k in x)) for x in rr] == [10] * 8
