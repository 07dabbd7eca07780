"""CPU placement of one process per GPU (cattus_amd/affinity.py): disjoint shares, NUMA-local where sysfs says so."""

import os

from cattus_amd import affinity as af


def test_parse_cpulist():
    assert af.parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    assert af.parse_cpulist("") == []


def test_plan_even_slices_without_topology():
    shares = af.plan(4, list(range(16)), [None] * 4)
    assert shares == [[0, 1, 2, 3], [4, 5, 6, 7], [8, 9, 10, 11], [12, 13, 14, 15]]


def test_plan_two_numa_nodes_eight_gpus():
    node0, node1 = list(range(0, 64)) + list(range(128, 192)), list(range(64, 128)) + list(range(192, 256))
    local = [node0] * 4 + [node1] * 4
    shares = af.plan(8, list(range(256)), local)
    flat = [c for s in shares for c in s]
    assert len(flat) == len(set(flat)) == 256  # disjoint, everything used
    for r in range(8):
        assert len(shares[r]) == 32 and set(shares[r]) <= set(local[r])


def test_plan_respects_the_allowed_set_and_falls_back():
    # the job may only use CPUs 0..7; GPU 1's node (CPUs 64..127) is outside of it: rank 1 falls back to a slice of the allowed ones
    shares = af.plan(2, list(range(8)), [list(range(0, 64)), list(range(64, 128))])
    assert set(shares[0]) <= set(range(8)) and shares[0]
    assert set(shares[1]) <= set(range(8)) and shares[1]


def test_gpu_local_cpus_reads_sysfs(tmp_path):
    d = tmp_path / "0000:c1:00.0"
    d.mkdir()
    (d / "local_cpulist").write_text("32-47,96-111\n")
    got = af.gpu_local_cpus(["0000:C1:00.0", "0000:ff:00.0"], sysfs=str(tmp_path))
    assert got[0] == list(range(32, 48)) + list(range(96, 112)) and got[1] is None


def test_pin_rank_single_rank_changes_nothing():
    before = os.sched_getaffinity(0)
    assert af.pin_rank(0, 1) == sorted(before)
    assert os.sched_getaffinity(0) == before
