"""GPU suite of the device-side FASTQ ingest (csrc/mc_ingest.hip, mc_text_*): the bytes of a batch of FASTQ records go to the
card, which cuts the records, packs the sequence lines as the reference's packer does (src/CuCLARK_hh.hh:1629-1707: parts at
every byte that is not one of acgtuACGTU, parts shorter than k dropped, 8 bases per container) and classifies them.  Checked
against the oracle's own indexer + packer + classifier on the same text, on every in-HBM index."""
import numpy as np
import pytest

from jn_cuclark_amd import synth
from helpers import small_db, mixed_fasta

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU: the product has no CPU fallback")
    from jn_cuclark_amd import CuClarkDB
    return CuClarkDB


@pytest.mark.parametrize("index,k", [("minimizer", 21), ("lines", 21), ("skm", 31), ("auto", 27)])
def test_fastq_text_in_final_rows_out(gpu, oracle, monkeypatch, index, k):
    monkeypatch.setenv("MC_INDEX", index)
    ht = 1000003
    genomes, sz, ky, lb = small_db(k=k, glen=6000)
    names, seqs = mixed_fasta(genomes, k, n=5000)          # N-split reads, short reads and parts, lower case, U, all-N, other symbols
    text = synth.fastq_text(names, seqs)
    ns, ne, sp, ep, ln = oracle.index_reads(text)
    rp, con = oracle.pack_reads(text, sp, ep, ln, k)
    want, _ = oracle.OracleDB.from_arrays(ht, sz, ky, lb).classify(k, rp, con, 15)
    with gpu(k=k, numBatches=1, numTargets=6, device=0, htsize=ht, maxhits=15) as db:
        db.read_arrays(sz, ky, lb)
        status, fin, hdr, seqlen = db.classify_text(text)
        # the same records without the newline behind the last one, and in a buffer that is exactly large enough
        status2, fin2, _, _ = db.classify_text(text[:-1] if text.endswith(b"\n") else text, max_reads=len(names), max_containers=int(con.size))
    assert status == 0 and status2 == 0
    assert np.array_equal(fin, want) and np.array_equal(fin2, want)
    assert np.array_equal(hdr, np.asarray(ns, dtype=np.int64) - 1)          # the '@' in front of every name
    assert np.array_equal(seqlen, np.asarray(ln, dtype=np.int64))
    assert (want[:, 2] > 0).sum() > 2000


def test_batches_this_code_does_not_vouch_for_are_handed_back(gpu, oracle, monkeypatch):
    ht, k = 1000003, 21
    genomes, sz, ky, lb = small_db(k=k)
    names, seqs = mixed_fasta(genomes, k, n=40)
    text = synth.fastq_text(names, seqs)
    with gpu(k=k, numBatches=1, numTargets=6, device=0, htsize=ht, maxhits=15) as db:
        db.read_arrays(sz, ky, lb)
        ok = db.classify_text(text)
        assert ok[0] == 0 and ok[1].shape[0] == 40
        lines = text.split(b"\n")
        assert db.classify_text(b"\n".join(lines[:-3]) + b"\n")[0] & 2                      # a record cut short: lines not a multiple of four
        bad = bytearray(text); bad[bad.index(b"\n@", 200) + 1] = ord(">")
        assert db.classify_text(bytes(bad))[0] & 1                                          # a header line that does not start with '@'
        blank = text.replace(b"@" + names[3], b"@ " + names[3], 1)
        assert db.classify_text(blank)[0] & 1                                               # a name that starts with a blank
        assert db.classify_text(text, max_reads=39)[0] & 4                                  # more reads than room
        assert db.classify_text(text, max_reads=40, max_containers=100)[0] & 4              # more containers than room
        again = db.classify_text(text)
        assert again[0] == 0 and np.array_equal(again[1], ok[1])
