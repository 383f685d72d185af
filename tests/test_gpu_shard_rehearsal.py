"""GPU suite at BASELINE configs[3]/[4]'s table size, on the one card of the test box: a 32.9e9-k-mer table of 8192
targets -- more than a card holds -- cut into 8 parts by minimizer exactly as 8 GPUs hold them, the parts played in turn
against the same reads, one k-way merge + top-2 (reference src/CuClarkDB.cu:516-559, :842-851, :909-928, :963-968).
tools/shard_rehearsal.py does the work and the checking (ground truth; a numpy model of the table on a sample, row for
row) in a process of its own, so that its 132 GB parts start from a clean heap; this test runs it and reads its verdict.
Skipped, with the reason, on a card that lacks the room."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("reads", ["150bp", "pairs_2x250bp"])
def test_a_table_that_needs_eight_cards_as_eight_parts_in_turn(reads):
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU")
    torch.cuda.empty_cache()
    free, total = torch.cuda.mem_get_info(0)
    if free < 200e9:
        pytest.skip("a 1/8 part of the 32.9e9-k-mer table takes 132 GB + the generator's chunks: %.0f GB free of %.0f" % (free / 1e9, total / 1e9))
    cmd = [sys.executable, os.path.join(ROOT, "tools", "shard_rehearsal.py"), "--reads", "100000", "--sample", "600",
           "--rate-reads", "500000", "--bg-reads", "200"] + (["--pairs"] if reads != "150bp" else [])
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    assert "== merged result of the 8 parts, row for row" in r.stdout and "ground truth:" in r.stdout, r.stdout[-3000:]
    assert r.stdout.count("streamed + built in") == 8
