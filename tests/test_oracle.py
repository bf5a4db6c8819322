"""Pins the CPU oracle against every golden vector the reference holds for the path
(SURVEY.md §8c).  CPU only."""
import hashlib
import os

import numpy as np
import pytest

import oracle


# ---- src/stack.rs:311-369 (from_overlap, -c 0) and :371-390 (coverage_upper_than_0, -c 2)
STACK_KATS = [
    ([(10, 990)], 1000, 0, [(0, 10), (990, 1000)]),                      # A
    ([(10, 90)], 1000, 0, [(0, 10), (90, 1000)]),                        # B
    ([(10, 490), (510, 990)], 1000, 0, [(0, 10), (490, 510), (990, 1000)]),  # C
    ([(0, 990)], 1000, 0, [(990, 1000)]),                                # D
    ([(10, 1000)], 1000, 0, [(0, 10)]),                                  # E
    ([(0, 490), (510, 1000)], 1000, 0, [(490, 510)]),                    # F
    ([(0, 425), (0, 450), (0, 475), (525, 1000), (550, 1000), (575, 1000)], 1000, 2,
     [(425, 575)]),
]


@pytest.mark.parametrize("ovl,length,cov,expect", STACK_KATS)
def test_stack_known_answers(ovl, length, cov, expect):
    assert oracle.compute_bad_part(ovl, length, cov) == expect
    # insertion order is irrelevant (the reference sorts): reversed input, same answer
    assert oracle.compute_bad_part(ovl[::-1], length, cov) == expect


# ---- src/editor/mod.rs:113-128 (read_type_assignation, -n 0.8)
TYPE_KATS = [
    ([(0, 10), (990, 1000)], 1000, oracle.NOT_BAD),
    ([(0, 10), (90, 1000)], 1000, oracle.NOT_COVERED),
    ([(0, 10), (490, 510), (990, 1000)], 1000, oracle.CHIMERIC),
    ([(990, 1000)], 1000, oracle.NOT_BAD),
    ([(0, 10)], 1000, oracle.NOT_BAD),
    ([(490, 510)], 1000, oracle.CHIMERIC),
]


@pytest.mark.parametrize("regions,length,expect", TYPE_KATS)
def test_type_known_answers(regions, length, expect):
    assert oracle.type_of_read(length, regions, 0.8) == expect


def test_type_unknown_read_is_notbad():
    # stack.rs:164-169 hands (vec![], 0) for an unknown id; 0/0 = NaN > n is false
    assert oracle.type_of_read(0, [], 0.8) == oracle.NOT_BAD


# ---- src/reads2ovl/mod.rs:173-237 (both sides of every record are ingested)
PAF_FILE = ("1\t12000\t20\t4500\t-\t2\t10000\t5500\t10000\t4500\t4500\t255\n"
            "1\t12000\t5500\t10000\t-\t3\t10000\t0\t4500\t4500\t4500\t255\n")
M4_FILE = ("1 2 0.1 2 0 20 4500 12000 0 5500 10000 10000\n"
           "1 3 0.1 2 0 5500 10000 12000 0 0 4500 10000\n")


@pytest.mark.parametrize("text,parser", [(PAF_FILE, oracle.parse_paf), (M4_FILE, oracle.parse_m4)])
def test_ingest_known_answers(text, parser):
    reads = parser(text.splitlines(True))
    assert set(reads) == {"1", "2", "3"}
    assert reads["1"][0] == [(20, 4500), (5500, 10000)]
    assert reads["2"][0] == [(5500, 10000)]
    assert reads["3"][0] == [(0, 4500)]
    assert reads["1"][1] == 12000 and reads["2"][1] == 10000 and reads["3"][1] == 10000


# ---- tests/reads.paf -> tests/truth.yacrd at defaults (-c 0 -n 0.8), tests/run.rs:95-117
def _fixture_reads(golden_dir):
    with open(os.path.join(golden_dir, "reads.paf")) as f:
        return oracle.parse_paf(f)


def test_fixture_matches_truth(golden_dir):
    reads = _fixture_reads(golden_dir)
    assert len(reads) == 230 and sum(len(v[0]) for v in reads.values()) == 2572
    got = set(oracle.report_lines(reads, 0, 0.8))
    with open(os.path.join(golden_dir, "truth.yacrd")) as f:
        truth = set(line.rstrip("\n") for line in f)
    assert got == truth  # unordered, exactly like diff_unorder (tests/run.rs:33-62)


def test_fixture_batch_driver_matches_truth(golden_dir):
    """Same fixture through the CSR batch driver (yo_run), 1 and 4 threads."""
    reads = _fixture_reads(golden_dir)
    names, offsets, intervals, lengths = oracle.to_csr(reads)
    with open(os.path.join(golden_dir, "truth.yacrd")) as f:
        truth = set(line.rstrip("\n") for line in f)
    for nt in (1, 4):
        bo, br, rt = oracle.run(offsets, intervals, lengths, 0, 0.8, n_threads=nt)
        assert set(oracle.report_from_csr(names, lengths, bo, br, rt)) == truth
        assert int(bo[-1]) == 462
        assert np.bincount(rt, minlength=3).tolist() == [226, 4, 0]


# ---- secondary cross-check vectors (independent Python restatement
# that reproduced truth.yacrd 230/230) for the -c/-n values the reference never tests.
SECONDARY = [
    (0, 0.8, (226, 4, 0), 462, "bfef1ecf6fb7bfaddad3605317e8d635f29fe8f24f10c0c2bfc0a73500cdc5b2"),
    (1, 0.8, (193, 6, 31), 435, "e7e9eb5b7e16ac1a56054b6566e431eddf69a95d23eed81fcf6c5cbbcfd29816"),
    (2, 0.4, (159, 4, 67), 408, "4b547044d130e382af3bb80a1621eb7dfdfd441824f25a5b3104c86405931e43"),
    (3, 0.4, (134, 3, 93), 390, "0b207320f179e75a93862fded411651225dfc94e982b1afa2f3e8ea3c113dce2"),
    (4, 0.4, (116, 4, 110), 385, "70e2f9e7873428e678c6d34e4233e3afaa6309a23fa4a028333d1bd172803df6"),
]


@pytest.mark.parametrize("cov,nc,counts,n_regions,sha", SECONDARY)
def test_fixture_secondary_vectors(golden_dir, cov, nc, counts, n_regions, sha):
    reads = _fixture_reads(golden_dir)
    names, offsets, intervals, lengths = oracle.to_csr(reads)
    bo, br, rt = oracle.run(offsets, intervals, lengths, cov, nc)
    assert tuple(np.bincount(rt, minlength=3).tolist()) == counts
    assert int(bo[-1]) == n_regions
    lines = sorted(l.encode() for l in oracle.report_from_csr(names, lengths, bo, br, rt))
    assert hashlib.sha256(b"".join(l + b"\n" for l in lines)).hexdigest() == sha


# ---- quirks enumerated in SURVEY.md §8a (semantics read off src/stack.rs:61-139)
def test_quirks():
    # abutting intervals give a zero-length gap, which makes the read Chimeric
    g = oracle.compute_bad_part([(0, 500), (500, 1000)], 1000, 0)
    assert g == [(500, 500)]
    assert oracle.type_of_read(1000, g, 0.8) == oracle.CHIMERIC
    # no interval at all (only reachable through add_length): whole read is bad
    assert oracle.compute_bad_part([], 1000, 0) == [(0, 1000)]
    # never covered above c: raw (0,first),(0,len) merged by equal begin
    assert oracle.compute_bad_part([(10, 900)], 1000, 1) == [(0, 1000)]
    # end > len: trailing region with begin > end, wrapping length in the report
    g = oracle.compute_bad_part([(0, 1200)], 1000, 0)
    assert g == [(1200, 1000)]
    assert oracle.report_line("r", 1000, g, 0).endswith("\t4294967096,1200,1000")
    # duplicate intervals, c = 1
    assert oracle.compute_bad_part([(0, 1000), (0, 1000)], 1000, 1) == []
