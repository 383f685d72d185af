"""ctypes binding of oracle/liboracle.so (the CPU restatement, oracle/clark_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never by the product package (jn_cuclark_amd).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")


def build(force=False):
    """Compile liboracle.so (and, when /root/reference exists, oracle/_ref)."""
    src = os.path.join(_HERE, "clark_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        u64, u32, u16p, u8p = C.c_uint64, C.c_uint32, C.POINTER(C.c_uint16), C.POINTER(C.c_uint8)
        u64p, u32p, vp = C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.c_void_p
        L.orc_nt_code.restype = C.c_int
        L.orc_nt_code.argtypes = [C.c_uint8]
        L.orc_kmer_from_string.restype = C.c_int
        L.orc_kmer_from_string.argtypes = [C.c_char_p, C.c_int, u64p]
        L.orc_revcomp.restype = u64
        L.orc_revcomp.argtypes = [u64, C.c_int]
        L.orc_canonical.restype = u64
        L.orc_canonical.argtypes = [u64, C.c_int]
        L.orc_db_from_arrays.restype = vp
        L.orc_db_from_arrays.argtypes = [u64, vp, vp, C.c_int, vp, u64]
        L.orc_db_load.restype = vp
        L.orc_db_load.argtypes = [C.c_char_p, u64, C.c_int, u32]
        L.orc_db_free.restype = None
        L.orc_db_free.argtypes = [vp]
        L.orc_db_write.restype = C.c_int
        L.orc_db_write.argtypes = [C.c_char_p, u64, C.c_int, vp, vp, u64]
        L.orc_db_lookup.restype = C.c_int
        L.orc_db_lookup.argtypes = [vp, C.c_int, u64, u64, u64, u16p]
        L.orc_build_discriminative.restype = u64
        L.orc_build_discriminative.argtypes = [vp, vp, u64, C.c_int, u64, u32, vp, vp]
        L.orc_pack_reads.restype = C.c_size_t
        L.orc_pack_reads.argtypes = [vp, vp, vp, vp, C.c_size_t, C.c_int, vp, vp, C.c_size_t]
        L.orc_index_reads.restype = C.c_long
        L.orc_index_reads.argtypes = [vp, C.c_size_t, C.c_size_t, vp, vp, vp, vp, vp]
        L.orc_query_batch.restype = None
        L.orc_query_batch.argtypes = [vp, C.c_int, u32, vp, vp, C.c_size_t, u64, u64, vp, C.c_size_t, u64p]
        L.orc_merge_rows.restype = None
        L.orc_merge_rows.argtypes = [vp, vp, C.c_size_t, C.c_size_t, vp]
        L.orc_result_rows.restype = None
        L.orc_result_rows.argtypes = [vp, C.c_size_t, C.c_size_t, vp]
        L.orc_classify_batch.restype = None
        L.orc_classify_batch.argtypes = [vp, C.c_int, u32, u32, vp, vp, C.c_size_t, vp, u64p]
        L.orc_num_threads.restype = C.c_int
        L.orc_set_num_threads.restype = None
        L.orc_set_num_threads.argtypes = [C.c_int]
        L.orc_csv_line.restype = C.c_int
        L.orc_csv_line.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, vp, u64, C.c_int, C.c_char_p]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


_KEY_DT = {2: np.uint16, 4: np.uint32, 8: np.uint64}


def revcomp(x, k):
    return int(lib().orc_revcomp(int(x), k))


def canonical(x, k):
    return int(lib().orc_canonical(int(x), k))


def kmer_from_string(s, k):
    out = C.c_uint64()
    if lib().orc_kmer_from_string(s.encode(), k, C.byref(out)) != 0:
        raise ValueError("non-ACGT base in %r" % s)
    return out.value


class OracleDB:
    """The CSR arrays CuClarkDB::read builds (ref: src/CuClarkDB.cu:463-770)."""

    def __init__(self, handle, htsize, key_bytes):
        if not handle:
            raise RuntimeError("oracle DB creation failed")
        self.h = handle
        self.htsize = int(htsize)
        self.key_bytes = key_bytes

    @classmethod
    def from_arrays(cls, htsize, sz, keys, labels):
        sz = np.ascontiguousarray(sz, dtype=np.uint8)
        keys = np.ascontiguousarray(keys)
        labels = np.ascontiguousarray(labels, dtype=np.uint16)
        kb = keys.dtype.itemsize
        assert sz.size == htsize and keys.size == labels.size
        h = lib().orc_db_from_arrays(htsize, _p(sz), _p(keys), kb, _p(labels), keys.size)
        return cls(h, htsize, kb)

    @classmethod
    def load(cls, base, htsize, key_bytes=4, sampling=1):
        h = lib().orc_db_load(os.fsencode(base), htsize, key_bytes, sampling)
        return cls(h, htsize, key_bytes)

    def close(self):
        if self.h:
            lib().orc_db_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def lookup(self, k, kmer_fwd, part=(0, None)):
        lab = C.c_uint16()
        pe = self.htsize if part[1] is None else part[1]
        ok = lib().orc_db_lookup(self.h, k, int(kmer_fwd), part[0], pe, C.byref(lab))
        return (bool(ok), lab.value)

    def query_rows(self, k, reads_ptr, containers, maxhits, part=(0, None), num_targets=65535):
        reads_ptr = np.ascontiguousarray(reads_ptr, dtype=np.uint32)
        containers = np.ascontiguousarray(containers, dtype=np.uint16)
        n = reads_ptr.size - 1
        row_len = 2 * maxhits + 2
        rows = np.zeros((n, row_len), dtype=np.uint16)
        ovf = C.c_uint64()
        pe = self.htsize if part[1] is None else part[1]
        lib().orc_query_batch(self.h, k, num_targets, _p(reads_ptr), _p(containers), n,
                              part[0], pe, _p(rows), row_len, C.byref(ovf))
        return rows, ovf.value

    def classify(self, k, reads_ptr, containers, maxhits, num_targets=65535):
        reads_ptr = np.ascontiguousarray(reads_ptr, dtype=np.uint32)
        containers = np.ascontiguousarray(containers, dtype=np.uint16)
        n = reads_ptr.size - 1
        out = np.zeros((n, 5), dtype=np.uint16)
        ovf = C.c_uint64()
        lib().orc_classify_batch(self.h, k, num_targets, maxhits, _p(reads_ptr), _p(containers), n,
                                 _p(out), C.byref(ovf))
        return out, ovf.value


def merge_rows(a, b):
    a = np.ascontiguousarray(a, dtype=np.uint16)
    b = np.ascontiguousarray(b, dtype=np.uint16)
    out = np.zeros_like(a)
    lib().orc_merge_rows(_p(a), _p(b), a.shape[1], a.shape[0], _p(out))
    return out


def result_rows(rows):
    rows = np.ascontiguousarray(rows, dtype=np.uint16)
    out = np.zeros((rows.shape[0], 5), dtype=np.uint16)
    lib().orc_result_rows(_p(rows), rows.shape[1], rows.shape[0], _p(out))
    return out


def db_write(base, htsize, key_bytes, canon, labels):
    canon = np.ascontiguousarray(canon, dtype=np.uint64)
    labels = np.ascontiguousarray(labels, dtype=np.uint16)
    rc = lib().orc_db_write(os.fsencode(base), htsize, key_bytes, _p(canon), _p(labels), canon.size)
    if rc != 0:
        raise RuntimeError("orc_db_write failed: %d" % rc)


def build_discriminative(kmers_fwd, targets, k, htsize, min_count=0):
    kmers_fwd = np.ascontiguousarray(kmers_fwd, dtype=np.uint64)
    targets = np.ascontiguousarray(targets, dtype=np.uint16)
    oc = np.zeros(kmers_fwd.size, dtype=np.uint64)
    ol = np.zeros(kmers_fwd.size, dtype=np.uint16)
    n = lib().orc_build_discriminative(_p(kmers_fwd), _p(targets), kmers_fwd.size, k, htsize,
                                       min_count, _p(oc), _p(ol))
    return oc[:n].copy(), ol[:n].copy()


def index_reads(text):
    """(name_s, name_e, spos, epos, length) of every record of a FASTA/FASTQ image."""
    buf = np.frombuffer(text, dtype=np.uint8)
    buf = np.concatenate([buf, np.zeros(8, dtype=np.uint8)])  # the reference peeks one past
    cap = max(16, int((buf == ord(">")).sum() + (buf == ord("@")).sum()) + 2)
    arrs = [np.zeros(cap, dtype=np.uint64) for _ in range(5)]
    n = lib().orc_index_reads(_p(buf), len(text), cap, *[_p(a) for a in arrs])
    if n < 0:
        raise ValueError("orc_index_reads: %d" % n)
    return [a[:n].copy() for a in arrs]


def pack_reads(text, spos, epos, length, k):
    buf = np.frombuffer(text, dtype=np.uint8)
    spos = np.ascontiguousarray(spos, dtype=np.uint64)
    epos = np.ascontiguousarray(epos, dtype=np.uint64)
    length = np.ascontiguousarray(length, dtype=np.uint64)
    n = spos.size
    cap = int((epos - spos).sum()) + 4 * n + 16
    rp = np.zeros(n + 1, dtype=np.uint32)
    con = np.zeros(cap, dtype=np.uint16)
    cnt = lib().orc_pack_reads(_p(buf), _p(spos), _p(epos), _p(length), n, k, _p(rp), _p(con), cap)
    if cnt == C.c_size_t(-1).value:
        raise RuntimeError("orc_pack_reads: capacity")
    return rp, con[:cnt].copy()


def csv_line(name, res5, norm_len, k, assignment):
    res5 = np.ascontiguousarray(res5, dtype=np.uint16)
    buf = C.create_string_buffer(512)
    nb = name if isinstance(name, bytes) else name.encode()
    n = lib().orc_csv_line(buf, 512, nb, len(nb), _p(res5), int(norm_len), k, assignment.encode())
    return buf.raw[:n].decode()


def num_threads():
    return int(lib().orc_num_threads())


def host_cores():
    """CPU cores this process may really use: affinity mask capped by the cgroup quota."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except Exception:
        pass
    return n


def set_num_threads(n):
    lib().orc_set_num_threads(int(n))
