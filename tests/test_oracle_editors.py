"""Pins the Python editor restatement (oracle/editors.py) on the reference's golden FASTQ files
(tests/run.rs:162-300) so it can serve as the checker for synthetic end-to-end runs.  CPU only."""
import gzip
import os

import pytest

import oracle
from oracle import editors


@pytest.mark.parametrize("op", ["scrubb", "filter", "extract", "split"])
def test_python_editors_match_golden(golden_dir, op):
    with open(os.path.join(golden_dir, "reads.paf")) as f:
        reads = oracle.parse_paf(f)
    table = {k: (oracle.compute_bad_part(v[0], v[1], 0), v[1]) for k, v in reads.items()}
    with gzip.open(os.path.join(golden_dir, "reads.fastq.gz"), "rb") as f:
        data = f.read()
    with gzip.open(os.path.join(golden_dir, "truth.%s.fastq.gz" % op), "rb") as f:
        truth = f.read()
    assert editors.edit_fastq(op, data, table, 0.8) == truth
