"""Host side of the join tables (alaz_amd/csrc/join_host.hpp): the block table + cuckoo word image that replaces
ClusterInfo.PodIPToPodUid / ServiceIPToServiceUid (aggregator/cluster.go:13-17) under the churn processPod /
processSvc produce (aggregator/persist.go:55-71, 114-130).  Pure C++ on the CPU: tests/micro/join_host_test.cpp."""
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("seed", [1, 7])
def test_join_table_mirror_and_word_log_under_churn(tmp_path, seed):
    exe = tmp_path / "join_host_test"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", "-o", str(exe), os.path.join(HERE, "micro", "join_host_test.cpp")])
    out = subprocess.run([str(exe), str(seed)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = [l for l in out.stdout.splitlines() if l.startswith("ok ")]
    assert len(lines) == 5
    # dense clusters live entirely in the block table; incremental batches ship a few hundred words, not a table
    assert "ck_n 0" in lines[0] and "ck_n 0" in lines[4]


def test_level1_grown_incrementally_never_misresolves(tmp_path):
    """a failed level-1 insert must not drop a live /24 or alias two /24s onto one level-2 block.
    3000 seeds of a small first build followed by single upserts across 10..50 random /24s, every IP checked."""
    exe = tmp_path / "join_host_test"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", "-o", str(exe), os.path.join(HERE, "micro", "join_host_test.cpp")])
    out = subprocess.run([str(exe), "3000", "1"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "ok growth 3000 seeds" in out.stdout, out.stdout + out.stderr


def test_key_mix_is_a_bijection_and_balances_structured_keys(tmp_path):
    """sg_kmix (sg_hash.h): partition + remainder must name the edge uniquely (pass B recovers the endpoints from them)."""
    exe = tmp_path / "kmix_test"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", "-o", str(exe), os.path.join(HERE, "micro", "kmix_test.cpp")])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "ok kmix" in out.stdout, out.stdout + out.stderr
