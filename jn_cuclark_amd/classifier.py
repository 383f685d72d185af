"""Python mirror of the reference's device-manager interface, over the C ABI.

Reference: ``class CuClarkDB<HKMERr>`` (src/CuClarkDB.cuh:39-153).  Method names and
argument meaning follow it (read / malloc / readyBatch / queryBatch / swapDbParts /
sync / waitForBatch / freeBatchMemory) so that tests read like calls of the original;
errors surface as ``McError`` where the reference printed and called ``exit(1)``.
All compute happens in libmcclark.so (HIP); nothing here touches ``oracle/``.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import MC_F_FINAL, MC_F_ROWS, MC_FINAL_ROW, McDbInfo, McStats, check

HTSIZE_FULL = 1610612741   # reference src/parameters.hh:37
HTSIZE_LIGHT = 57777779    # reference src/parameters_light_hh:39
MAXHITS_FULL = 15          # reference src/parameters.hh:44
MAXHITS_LIGHT = 23         # reference src/parameters_light_hh:45


def _np_view(addr, dtype, count):
    if count == 0 or not addr:
        return np.zeros(0, dtype=dtype)
    nbytes = int(count) * np.dtype(dtype).itemsize
    buf = (C.c_uint8 * nbytes).from_address(addr)
    return np.frombuffer(buf, dtype=dtype, count=int(count))


def _ptr(t):
    """device pointer of a torch tensor (or None)."""
    return None if t is None else C.c_void_p(t.data_ptr())


class CuClarkDB:
    """One GPU, one database shard.  ctor mirrors CuClarkDB.cu:94-241: ``numTargets`` is
    ``targetsName.size()-1``; ``device`` selects the HIP device (the reference's
    ``numDevices`` becomes one process per GPU, see DESIGN.md "Multi-GPU")."""

    def __init__(self, k, numBatches, numTargets, device=-1, htsize=HTSIZE_FULL,
                 maxhits=MAXHITS_FULL, verbose=False):
        self._lib = _lib.load_library()
        self.k, self.numBatches, self.numTargets = int(k), int(numBatches), int(numTargets)
        self.htsize, self.maxhits, self.verbose = int(htsize), int(maxhits), verbose
        self.row_len = 2 * self.maxhits + 2          # CuCLARK_hh.hh:1586-1589
        h = C.c_void_p()
        check(self._lib.mc_open(C.byref(h), device, self.k, self.htsize, self.numTargets, self.maxhits))
        self._h = h
        self._cycles_to_do = 1
        self._batch_meta = {}
        self._bufs = []

    # -- lifecycle -----------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self._lib.mc_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- database (CuClarkDB::read, :463-770) ---------------------------------------
    def read(self, filename, modCollision=1, key_bytes=4, shard=(0, 0)):
        """Returns False when the files cannot be opened -- the caller then rebuilds the
        database, as CuCLARK_hh.hh:622-684 does -- and raises on any other error."""
        rc = self._lib.mc_load_db(self._h, filename.encode(), key_bytes, int(modCollision),
                                  int(shard[0]), int(shard[1]))
        if rc == -2:      # MC_EIO
            return False
        check(rc)
        self._cycles_to_do = 1
        return True

    def read_arrays(self, sz, keys, labels, shard=(0, 0)):
        sz = np.ascontiguousarray(sz, dtype=np.uint8)
        keys = np.ascontiguousarray(keys)
        labels = np.ascontiguousarray(labels, dtype=np.uint16)
        if keys.dtype not in (np.uint16, np.uint32, np.uint64):
            raise ValueError("keys must be uint16, uint32 or uint64")
        check(self._lib.mc_load_db_host(self._h, sz.ctypes.data, keys.ctypes.data, keys.dtype.itemsize,
                                        labels.ctypes.data, keys.size, int(shard[0]), int(shard[1])))
        self._cycles_to_do = 1

    def read_device(self, d_sz, d_keys, d_labels, shard=(0, 0)):
        """Raw arrays already in HBM (torch uint8 / int16|int32|int64 / int16 tensors on this device)."""
        check(self._lib.mc_load_db_device(self._h, _ptr(d_sz), _ptr(d_keys), int(d_keys.element_size()),
                                          _ptr(d_labels), int(d_keys.numel()), int(shard[0]), int(shard[1])))
        self._cycles_to_do = 1

    def read_part(self, filename, part, n_parts, modCollision=1, key_bytes=4):
        """Line-range part `part` of `n_parts` of the table in the files (mc_load_db_part): what one GPU
        of a multi-GPU job holds; every part sees every read batch, rows add up (merge_result_device)."""
        rc = self._lib.mc_load_db_part(self._h, filename.encode(), key_bytes, int(modCollision), int(part), int(n_parts))
        if rc == -2:      # MC_EIO
            return False
        check(rc)
        self._cycles_to_do = 1
        return True

    def read_chunks(self, chunks, n_keys_total, part=0, n_parts=1, device=False):
        """Streamed index build (mc_index_*): `chunks` is a callable returning an iterator of
        (sz, keys, labels, bucket_begin, bucket_end) -- numpy arrays, or torch tensors on this GPU when
        device=True -- and is called twice (the table is fed once per pass)."""
        check(self._lib.mc_index_begin(self._h, int(n_keys_total), int(part), int(n_parts)))
        for p in range(2):
            for sz, keys, labels, b0, b1 in chunks():
                if device:
                    check(self._lib.mc_index_add_device(self._h, _ptr(sz), _ptr(keys), int(keys.element_size()),
                                                        _ptr(labels), int(keys.numel()), int(b0), int(b1)))
                else:
                    sz = np.ascontiguousarray(sz, dtype=np.uint8)
                    keys = np.ascontiguousarray(keys)
                    labels = np.ascontiguousarray(labels, dtype=np.uint16)
                    check(self._lib.mc_index_add_host(self._h, sz.ctypes.data, keys.ctypes.data, keys.dtype.itemsize,
                                                      labels.ctypes.data, keys.size, int(b0), int(b1)))
            check(self._lib.mc_index_next_pass(self._h) if p == 0 else self._lib.mc_index_end(self._h))
        self._cycles_to_do = 1

    def db_info(self):
        info = McDbInfo()
        check(self._lib.mc_get_db_info(self._h, C.byref(info)))
        return {f: getattr(info, f) for f, _ in McDbInfo._fields_}

    def stats(self):
        st = McStats()
        check(self._lib.mc_get_stats(self._h, C.byref(st)))
        return {f: getattr(st, f) for f, _ in McStats._fields_}

    def swapDbParts(self):
        """A single context keeps its table (or its part of one) resident in HBM: exactly one cycle here.  A table larger than all
        devices together is cycled by the group (include/mc_group.h: mc_group_set_cycle, MC_F_FOLLOWUP; reference :775-815)."""
        if self._cycles_to_do == 0:
            self._cycles_to_do = 1
            return False
        self._cycles_to_do -= 1
        return True

    # -- batches (malloc :321-421, readyBatch :820-828, queryBatch :835-987) --------
    def malloc(self, maxReads, maxReadsInContainers, isExtended=False):
        check(self._lib.mc_alloc_batches(self._h, self.numBatches, int(maxReads),
                                         int(maxReadsInContainers), 1 if isExtended else 0))
        self._maxReads, self._maxCon, self._ext = int(maxReads), int(maxReadsInContainers), bool(isExtended)
        readsPointer, readsInContainers, finals, fulls = [], [], [], []
        for b in range(self.numBatches):
            p, c, f, r = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
            check(self._lib.mc_batch_buffers(self._h, b, C.byref(p), C.byref(c), C.byref(f), C.byref(r)))
            readsPointer.append(_np_view(p.value, np.uint32, self._maxReads + 1))
            readsInContainers.append(_np_view(c.value, np.uint16, (self._maxCon + 7) // 8 * 8 if self._maxCon >= 8 else 8))
            finals.append(_np_view(f.value, np.uint16, self._maxReads * MC_FINAL_ROW))
            fulls.append(_np_view(r.value, np.uint16, self._maxReads * self.row_len) if isExtended else None)
        self._bufs = [readsPointer, readsInContainers, finals, fulls]
        return readsPointer, readsInContainers, finals, fulls

    def readyBatch(self, batchId, numReads, containerCount):
        self._batch_meta[batchId] = (int(numReads), int(containerCount))
        return True

    def queryBatch(self, batchId, isExtended=False, isFollowup=False):
        n, c = self._batch_meta[batchId]
        flags = MC_F_FINAL | (MC_F_ROWS if isExtended else 0)
        check(self._lib.mc_submit(self._h, batchId, n, c, flags))
        return True     # final results are always scheduled: one DB cycle

    def waitForBatch(self, batchId):
        check(self._lib.mc_wait(self._h, batchId))
        return True

    def sync(self):
        check(self._lib.mc_sync(self._h))
        return True

    def freeBatchMemory(self):
        self._bufs = []
        check(self._lib.mc_free_batches(self._h))

    # -- convenience: classify packed reads through the pinned-buffer path -----------
    def classify(self, reads_ptr, containers, extended=False):
        reads_ptr = np.ascontiguousarray(reads_ptr, dtype=np.uint32)
        containers = np.ascontiguousarray(containers, dtype=np.uint16)
        n = reads_ptr.size - 1
        saved = self.numBatches
        self.numBatches = 1
        try:
            rp, rc, fin, full = self.malloc(max(n, 1), max(containers.size, 8), extended)
            rp[0][: n + 1] = reads_ptr
            rc[0][: containers.size] = containers
            self.readyBatch(0, n, containers.size)
            self.queryBatch(0, extended)
            self.waitForBatch(0)
            final = fin[0][: n * MC_FINAL_ROW].reshape(n, MC_FINAL_ROW).copy()
            rows = full[0][: n * self.row_len].reshape(n, self.row_len).copy() if extended else None
            self.freeBatchMemory()
        finally:
            self.numBatches = saved
        return (final, rows) if extended else final

    # -- FASTQ text in, final rows out (mc_text_*: records cut and packed on the device) ----------
    def classify_text(self, text, max_reads=None, max_containers=None):
        """`text`: bytes of whole 4-line FASTQ records.  Returns (status, final rows [n, 5], header offsets [n], sequence
        lengths [n]); status != 0: the batch was handed back unclassified (see include/mc_api.h) and the arrays are empty."""
        n_bytes = len(text)
        max_reads = int(max_reads or n_bytes // 8 + 16)
        max_containers = int(max_containers or n_bytes // 4 + 64)
        check(self._lib.mc_text_alloc(self._h, 1, max(n_bytes + 1, 16), max_reads, max_containers))
        try:
            h, ln, f = C.c_void_p(), C.c_void_p(), C.c_void_p()
            check(self._lib.mc_text_buffers(self._h, 0, C.byref(h), C.byref(ln), C.byref(f)))
            arr = np.frombuffer(text, dtype=np.uint8)
            check(self._lib.mc_text_submit(self._h, 0, arr.ctypes.data, n_bytes))
            n, st = C.c_uint64(), C.c_uint32()
            check(self._lib.mc_text_wait(self._h, 0, C.byref(n), C.byref(st)))
            n = int(n.value)
            fin = _np_view(f.value, np.uint16, n * MC_FINAL_ROW).reshape(n, MC_FINAL_ROW).copy()
            return int(st.value), fin, _np_view(h.value, np.uint32, n).copy(), _np_view(ln.value, np.uint32, n).copy()
        finally:
            check(self._lib.mc_text_free(self._h))

    # -- device-resident entry points (torch tensors on this GPU) --------------------
    def query_device(self, reads_ptr_t, containers_t, final_t=None, rows_t=None, stream=None):
        flags = (MC_F_FINAL if final_t is not None else 0) | (MC_F_ROWS if rows_t is not None else 0)
        check(self._lib.mc_query_device(self._h, _ptr(reads_ptr_t), _ptr(containers_t),
                                        int(reads_ptr_t.numel()) - 1, int(containers_t.numel()), flags,
                                        _ptr(final_t), _ptr(rows_t),
                                        C.c_void_p(stream) if stream else None))

    def merge_rows_device(self, a_t, b_t, out_t, n_reads, stream=None):
        check(self._lib.mc_merge_rows_device(self._h, _ptr(a_t), _ptr(b_t), int(n_reads), _ptr(out_t),
                                             C.c_void_p(stream) if stream else None))

    def merge_result_device(self, srcs, n_reads, rows_t=None, final_t=None, stream=None):
        """k-way merge of the sparse rows of up to 16 shards (+ top-2) in one launch"""
        arr = (C.c_void_p * len(srcs))(*[t.data_ptr() for t in srcs])
        check(self._lib.mc_merge_result_device(self._h, arr, len(srcs), int(n_reads), _ptr(rows_t), _ptr(final_t),
                                               C.c_void_p(stream) if stream else None))

    def result_rows_device(self, rows_t, final_t, n_reads, stream=None):
        check(self._lib.mc_result_rows_device(self._h, _ptr(rows_t), int(n_reads), _ptr(final_t),
                                              C.c_void_p(stream) if stream else None))
