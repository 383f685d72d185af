// CuClarkDB_mc.cc -- drop-in replacement for the reference's src/CuClarkDB.cu.
//
// Implements the member functions of `template <typename HKMERr> class CuClarkDB`
// exactly as declared in the reference's src/CuClarkDB.cuh:39-153, on top of the C ABI
// of libmcclark.so (include/mc_api.h).  A maintainer builds cuCLARK / cuCLARK-l with
// this file in place of CuClarkDB.cu; nothing else in the
// reference's host code (src/main.cc, src/CuCLARK_hh.hh) changes.
//
// The class declaration stays the reference's, so this file cannot add members: the
// per-object state lives in a side table keyed by `this`.  No algorithm lives here --
// only the mapping between the two interfaces:
//   ctor            -> mc_group_open    (numDevices as the reference: 0 = all, CuClarkDB.cu:146-150;
//                                        HTSIZE / MAXHITS from the reference's parameters.hh)
//   read            -> mc_group_load_db (false when the files are missing, as the reference; replicas when
//                                        the table fits one GPU, shards + device-to-device row exchange
//                                        when it does not, include/mc_group.h)
//   malloc          -> mc_group_alloc_batches + mc_group_batch_buffers; the two whole-file result tables
//                      the host indexes by global read number are plain pinned-size host arrays
//   readyBatch      -> remembered sizes
//   queryBatch      -> mc_group_submit  (MC_F_FOLLOWUP for the re-queries of a cycled database; true in the last cycle)
//   swapDbParts     -> mc_group_set_cycle: the next cycle's parts, false when all are done (reference :775-815)
//   waitForBatch    -> mc_group_wait + copy of the batch's rows into the whole-file tables
//   sync / freeBatchMemory -> mc_group_sync / mc_group_free_batches
#include "CuClarkDB.cuh"          // the reference's header, unchanged
#include "mc_api.h"
#include "mc_group.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <map>
#include <mutex>

namespace {

struct State {
    mc_group *grp = nullptr;
    int cycles = 1, cycles_to_do = 1, cur = 0;      // database cycles per file (mc_group_info.n_cycles), still to do, loaded now
    bool extended = false;
    size_t row_len = 0, final_len = 0;
    RESULTS *full = nullptr, *final_ = nullptr;
    std::vector<ITYPE> index;             // first global read of every batch
    std::vector<size_t> n_reads, n_con;
};

std::mutex g_mu;
std::map<const void *, State> g_state;

State &st(const void *self)
{
    std::lock_guard<std::mutex> lk(g_mu);
    return g_state[self];
}

void check(int rc, const char *what)
{
    if (rc != MC_OK) {
        std::cerr << what << ": " << mc_last_error() << std::endl;   // reference CUERR: print + exit(1)
        exit(1);
    }
}

} // namespace

template <typename HKMERr>
CuClarkDB<HKMERr>::CuClarkDB(const size_t _numDevices, const uint8_t _k, const size_t _numBatches,
                             const size_t _numTargets, bool _verbose)
    : m_k(_k), m_numTargets(_numTargets), m_numBatches(_numBatches), m_verbose(_verbose), d_resultsFinal(nullptr)
{
    int n = 0;
    if (mc_device_count(&n) != MC_OK || n < 1) { std::cerr << "No HIP devices found. Abort.\n"; exit(1); }
    if ((size_t)n < _numDevices) { std::cerr << _numDevices << " devices requested. Insufficient devices found. Abort.\n"; exit(1); }
    m_numDevices = _numDevices ? _numDevices : (size_t)n;          // 0 = all (reference :146-150)
    m_dbParts = 1; m_dbPartsPerDevice = 1; m_cyclesPerDevice = 1; m_cyclesToDo = 1;
    State &s = st(this);
    check(mc_group_open(&s.grp, nullptr, (uint32_t)_numDevices, _k, (uint64_t)HTSIZE, (uint32_t)_numTargets, (uint32_t)MAXHITS),
          "mc_group_open");
}

template <typename HKMERr>
CuClarkDB<HKMERr>::~CuClarkDB()
{
    State &s = st(this);
    mc_group_close(s.grp);
    std::lock_guard<std::mutex> lk(g_mu);
    g_state.erase(this);
}

template <typename HKMERr>
bool CuClarkDB<HKMERr>::read(const char *_filename, size_t &_fileSize, size_t &_dbParts, const ITYPE &_modCollision)
{
    State &s = st(this);
    const int rc = mc_group_load_db(s.grp, _filename, (int)sizeof(HKMERr), _modCollision, MC_GROUP_AUTO);
    if (rc == MC_EIO) { std::cerr << mc_last_error() << std::endl; return false; }
    check(rc, "mc_group_load_db");
    mc_group_info info;
    check(mc_group_get_info(s.grp, &info), "mc_group_get_info");
    _fileSize = info.device_bytes_max;
    _dbParts = info.n_cycles ? info.n_cycles : 1;      // 1: everything is resident, however many devices
    s.cycles = s.cycles_to_do = (int)_dbParts;
    s.cur = 0;
    std::cerr << (m_verbose ? "DB loaded in HBM.\n" : "CuCLARK initialized.\n");
    return true;
}

template <typename HKMERr>
size_t CuClarkDB<HKMERr>::malloc(size_t _numReads, size_t _maxReads, size_t _maxReadsInContainers,
                                 std::vector<ITYPE> &_indexBatches, RESULTS *&_fullResults, size_t _resultRowSize,
                                 RESULTS *&_finalResults, size_t _finalResultsRowSize, bool _isExtended,
                                 std::vector<uint32_t *> &_readsPointer, std::vector<CONTAINER *> &_readsInCon)
{
    State &s = st(this);
    s.extended = _isExtended;
    s.row_len = _resultRowSize;
    s.final_len = _finalResultsRowSize;
    s.index = _indexBatches;
    s.n_reads.assign(m_numBatches, 0);
    s.n_con.assign(m_numBatches, 0);
    check(mc_group_alloc_batches(s.grp, (uint32_t)m_numBatches, _maxReads ? _maxReads : 1, _maxReadsInContainers,
                                 _isExtended || s.cycles > 1 ? 1 : 0), "mc_group_alloc_batches");
    _readsPointer.resize(m_numBatches);
    _readsInCon.resize(m_numBatches);
    for (size_t b = 0; b < m_numBatches; b++)
        check(mc_group_batch_buffers(s.grp, (uint32_t)b, &_readsPointer[b], &_readsInCon[b], nullptr, nullptr), "mc_group_batch_buffers");
    _fullResults = nullptr;
    if (_isExtended) _fullResults = (RESULTS *)std::calloc(_numReads * _resultRowSize + 1, sizeof(RESULTS));
    _finalResults = (RESULTS *)std::calloc(_numReads * _finalResultsRowSize + 1, sizeof(RESULTS));
    s.full = _fullResults;
    s.final_ = _finalResults;
    return (_maxReads + 1) * sizeof(uint32_t) + _maxReadsInContainers * sizeof(CONTAINER);
}

template <typename HKMERr>
void CuClarkDB<HKMERr>::freeBatchMemory()
{
    State &s = st(this);
    mc_group_free_batches(s.grp);
    std::free(s.full); std::free(s.final_);
    s.full = s.final_ = nullptr;
}

template <typename HKMERr>
bool CuClarkDB<HKMERr>::sync()
{
    check(mc_group_sync(st(this).grp), "mc_group_sync");
    return true;
}

template <typename HKMERr>
bool CuClarkDB<HKMERr>::readyBatch(const size_t _batchId, const size_t _numReads, const size_t _containerCount)
{
    State &s = st(this);
    s.n_reads[_batchId] = _numReads;
    s.n_con[_batchId] = _containerCount;
    return true;
}

template <typename HKMERr>
bool CuClarkDB<HKMERr>::queryBatch(const size_t _batchId, const bool _isExtended, const bool _isFollowup)
{
    // every batch keeps its own pinned buffers for the whole file (reference malloc): the sparse rows a batch got in the cycle
    // before are still in its row buffer when it is queried again (followup), and the library merges them in (:932-948)
    State &s = st(this);
    const bool last = s.cur + 1 == s.cycles;
    check(mc_group_submit(s.grp, (uint32_t)_batchId, s.n_reads[_batchId], s.n_con[_batchId],
                          (last ? MC_F_FINAL : 0u) | (_isExtended || s.cycles > 1 ? MC_F_ROWS : 0u) | (_isFollowup ? MC_F_FOLLOWUP : 0u)),
          "mc_group_submit");
    return last;                           // final results were scheduled (reference :963-979)
}

template <typename HKMERr>
bool CuClarkDB<HKMERr>::swapDbParts()
{
    State &s = st(this);
    if (s.cycles_to_do == 0) { s.cycles_to_do = s.cycles; return false; }      // reset for a possible next file (reference :778-783)
    s.cur = s.cycles - s.cycles_to_do;
    check(mc_group_set_cycle(s.grp, (uint32_t)s.cur), "mc_group_set_cycle");     // nothing to do when those parts are loaded
    s.cycles_to_do--;
    return true;
}

template <typename HKMERr>
bool CuClarkDB<HKMERr>::waitForBatch(size_t batchId)
{
    State &s = st(this);
    check(mc_group_wait(s.grp, (uint32_t)batchId), "mc_group_wait");
    uint16_t *fin = nullptr, *rows = nullptr;
    check(mc_group_batch_buffers(s.grp, (uint32_t)batchId, nullptr, nullptr, &fin, &rows), "mc_group_batch_buffers");
    const size_t n = s.n_reads[batchId], at = s.index[batchId];
    std::memcpy(s.final_ + at * s.final_len, fin, n * s.final_len * sizeof(RESULTS));
    if (s.extended) std::memcpy(s.full + at * s.row_len, rows, n * s.row_len * sizeof(RESULTS));
    return true;
}

template <typename HKMERr>
bool CuClarkDB<HKMERr>::checkBatch(size_t) { return true; }

template class CuClarkDB<uint16_t>;
template class CuClarkDB<uint32_t>;
template class CuClarkDB<uint64_t>;
