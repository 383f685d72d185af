// main.cc -- cuCLARK / cuCLARK-l compatible host driver over the C ABI.
//
// Same command line as the reference binary (src/main.cc:43-69, :103-212) so that an
// unmodified scripts/classify_metagenome.sh (:155-159) can exec it:
//   cuCLARK[-l] -k <k> -T <targets> -D <dbdir> (-O <file> | -P <f1> <f2>) -R <result>
//               [-t <minfreq>] [-n <threads>] [-b <batches>] [-d <devices>] [-g <gap>]
//               [-s <sampling>] [--tsk] [--extended] [--verbose]
// Output: <result>.csv with the reference's columns and number formatting
// (src/CuCLARK_hh.hh:1945-2122).  GPU work: include/mc_api.h only.
//
// -d N (0 or absent = all devices, as the reference, src/CuClarkDB.cu:146-150) goes to mc_group
// (include/mc_group.h): N replicas when the table fits one GPU, N shards with a device-to-device
// exchange of the per-read rows when it does not -- the reference always shards (:552-559).
// --tsk: the per-target .ht text files are written with the database (host/dbbuild.hpp); the reference's
// recovery path that rebuilds a vanished database from them (src/CuCLARK_hh.hh:633-684) is not.
// Ingest (classify_image): files of 8 MiB and more are cut into -b byte ranges at record starts and every range is
// indexed, packed and submitted by one task of a thread pool that also formats and writes the CSV slices; large
// regular mate pairs (-P) go the same way straight from their two files; small, irregular or --dump-batches
// inputs are indexed (and mates joined) as a whole first.  The CSV does not depend on the plan.
#include "../../include/mc_api.h"
#include "../../include/mc_group.h"
#include "common.hpp"
#include "dbbuild.hpp"
#include "reads.hpp"
#include "input.hpp"
#include "pairs.hpp"
#include "gzstream.hpp"
#include "format.hpp"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/time.h>
#include <unistd.h>

#include <condition_variable>
#include <cstring>
#include <deque>
#include <functional>
#include <array>
#include <iostream>
#include <memory>
#include <mutex>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#ifdef _OPENMP
#include <omp.h>
#endif

using namespace host;

namespace {

void print_usage()
{
    std::cout << "\ncuCLARK" << (LIGHT ? "-l" : "") << " (MI355X build) -- metagenomic classification with discriminative k-mers\n\n"
              << "./cuCLARK" << (LIGHT ? "-l" : "")
              << " -k <kmerSize> -t <minFreqTarget> -T <fileTargets> -D <directoryDB/> -O <fileObjects> -R <fileResults>"
                 " -n <numberofthreads> -b <numberofbatches> -d <numberofdevices> ...\n\n"
              << "-k <kmerSize>,       k-mer length: integer, >= 2 and <= 32 (fixed to 27 for cuCLARK-l)\n"
              << "-t <minFreqTarget>,  minimum of k-mer frequency in targets (default 0)\n"
              << "-T <fileTargets>,    targets definition: filename, label per line\n"
              << "-D <directoryDB/>,   directory of the database\n"
              << "-O <fileObjects>,    objects (reads) in fasta/fastq, or a file listing such files\n"
              << "-P <file1> <file2>,  paired-end fastq files\n"
              << "-R <fileResults>,    results file (\".csv\" is appended), or a file listing result names\n"
              << "-n <numberofthreads>, -b <numberofbatches>, -d <numberofdevices>\n"
              << "-g <gap> (light only), -s <samplingFactor>, --extended, --verbose, --version, --help\n\n";
}

[[noreturn]] void die(const std::string &m, int code = 1)
{
    std::cerr << m << std::endl;
    std::exit(code);
}

void mc_check(int rc, const char *what)
{
    if (rc != MC_OK) die(std::string(what) + ": " + mc_last_error());
}

// worker threads with one FIFO of tasks
class Pool {
public:
    explicit Pool(int n)
    {
        for (int i = 0; i < n; i++) th_.emplace_back([this]() { loop(); });
    }
    ~Pool() { finish(); }
    void run(std::function<void()> f)
    {
        { std::lock_guard<std::mutex> lk(mu_); q_.push_back(std::move(f)); }
        cv_.notify_one();
    }
    void finish()          // runs what is queued, then joins
    {
        { std::lock_guard<std::mutex> lk(mu_); stop_ = true; }
        cv_.notify_all();
        for (auto &t : th_) if (t.joinable()) t.join();
        th_.clear();
    }
private:
    void loop()
    {
        for (;;) {
            std::function<void()> f;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [this]() { return stop_ || !q_.empty(); });
                if (q_.empty()) return;
                f = std::move(q_.front());
                q_.pop_front();
            }
            f();
        }
    }
    std::vector<std::thread> th_;
    std::deque<std::function<void()>> q_;
    std::mutex mu_;
    std::condition_variable cv_;
    bool stop_ = false;
};

struct Options {
    size_t k = 31, cpu = 1, gap = 0, batches = 1, devices = 0;
    unsigned minT = 0, sfactor = 1;
    bool ext = false, verbose = false, tsk = false;
    const char *targets = nullptr, *folder = nullptr, *objects = nullptr, *objects2 = nullptr, *results = nullptr;
    const char *dump = nullptr;      // test hook: write the packed batches here
    bool gpu_build = false;          // --gpu-build / MC_GPU_BUILD=1: build a missing database on the GPU
    bool gpu_ingest = false;         // --gpu-ingest / MC_GPU_INGEST=1: large plain FASTQ files go to the card as TEXT, which cuts and
                                     // packs the records itself (mc_text_*).  Off by default: measured at the metric's size (40 M reads,
                                     // 12.7 GB, 16 threads) it is no faster than the host's indexer and packer -- 50-71 against 63-83
                                     // M reads/s: uploading 317 bytes of text per read from the page cache costs the host as many
                                     // thread-seconds as parsing them into 44
};

struct Classifier {
    Options opt;
    Targets T;
    std::string folder, dbbase;
    int key_bytes = 4;
    mc_group *grp = nullptr;
    bool paired = false;
    bool text_path = false;          // every member of the group holds the whole table: FASTQ batches may go to the card as text
    uint32_t db_cycles = 1, db_cycle = 0;      // > 1: the table is larger than all devices together; every file is classified once per cycle (db_cycle: the parts loaded now)
    // one pass of a file over the parts of one database cycle (reference: the swapDbParts loop, src/CuCLARK_hh.hh:1765-1772)
    struct Cycle { uint32_t i, n; std::vector<uint16_t> *rows; };      // rows: the sparse rows of every read of the file, kept between the passes
    // a gzip file that is classified segment by segment (gzstream.hpp): what one segment's pass leaves for the next
    struct Seg {
        bool first = true, last = false;
        uint64_t csv_bytes = 0;            // where the next segment's lines go
        size_t objects = 0;
        long nz_min = 0, nz_max = 0, nz_sum = 0;
    };
    size_t n_objects = 0;

    void open_devices()
    {
        int n = 0;
        if (mc_device_count(&n) != MC_OK || n < 1) die("No HIP devices found. Abort.");
        if ((size_t)n < opt.devices) die(std::to_string(opt.devices) + " devices requested. Insufficient devices found. Abort.");
        // (MC_GROUP_DEVICES=0,0,1 in the environment overrides the member list inside the library)
        mc_check(mc_group_open(&grp, nullptr, (uint32_t)opt.devices, (uint32_t)opt.k, HTSIZE, (uint32_t)(T.names.size() - 1), MAXHITS),
                 "mc_group_open");
    }

    // the database is built on the CPU when its files are missing (reference
    // CuCLARK ctor, src/CuCLARK_hh.hh:306-311) -- before any GPU is touched
    void build_if_missing()
    {
        dbbase = db_name(folder, (unsigned)opt.k, T.labels.size(), opt.minT, (unsigned)opt.gap);
        const bool present = file_readable((dbbase + ".sz").c_str()) && file_readable((dbbase + ".ky").c_str()) &&
                             file_readable((dbbase + ".lb").c_str());
        if (present) return;
        if (ht_files_present(T, folder, (unsigned)opt.k)) {
            // The per-target .ht files of an earlier --tsk run are all there: the reference does not rebuild from the target
            // files then (getTargetsData, src/CuCLARK_hh.hh:1826-1836); its load fails, and it either gives up or -- with
            // --tsk -- puts the database back together from those files and leaves (:633-684, exit(-1)).
            if (!opt.tsk) die("Failed to find the database.", -1);
            std::string err;
            if (!recover_database(T, folder, (unsigned)opt.k, opt.minT, key_bytes, dbbase, err)) die(err, -1);
            std::exit(-1);
        }
        if (opt.verbose && opt.tsk) std::cerr << "Creation of targets specific k-mers files requested " << std::endl;   // (:1914-1917)
        std::cerr << "Starting the creation of the database of targets specific " << opt.k
                  << "-mers from input files..." << std::endl;
        uint64_t stored = 0;
        std::string err;
        std::cerr << "Creating database in disk..." << std::endl;
        // --tsk: per-target text files of the target-specific k-mers next to the database (CPU builder only)
        const bool okb = opt.gpu_build && !opt.tsk
            ? build_database_gpu(T, (unsigned)opt.k, (unsigned)opt.gap, opt.minT, key_bytes, dbbase, stored, err)
            : build_database(T, (unsigned)opt.k, (unsigned)opt.gap, opt.minT, key_bytes, dbbase, stored, err, opt.tsk ? &folder : nullptr);
        if (!okb) die(err, -1);
    }

    void load()
    {
        if (opt.verbose) std::cerr << "Loading database [" << dbbase << ".*] (s=" << opt.sfactor << ")..." << std::endl;
        const int rc = mc_group_load_db(grp, dbbase.c_str(), key_bytes, opt.sfactor, MC_GROUP_AUTO);
        if (rc == MC_EIO) die("Failed to find the database.", -1);
        mc_check(rc, "mc_group_load_db");
        mc_group_info gi;
        mc_group_get_info(grp, &gi);
        text_path = gi.mode == MC_GROUP_REPLICAS;
        db_cycles = gi.n_cycles ? gi.n_cycles : 1;
        db_cycle = gi.cycle;
        if (opt.verbose) {
            mc_ctx *c0 = nullptr;
            mc_db_info info;
            mc_group_member(grp, 0, &c0);
            mc_get_db_info(c0, &info);
            std::cerr << "Devices: " << gi.n_members << " ("
                      << (gi.mode == MC_GROUP_SHARDS ? (gi.shard_kind == 1 ? "shards by minimizer line range" : "shards by bucket range")
                                                     : "replicas");
            if (gi.mode == MC_GROUP_SHARDS) std::cerr << ": " << gi.n_shards << " parts x " << gi.n_groups << " groups";
            if (gi.n_cycles > 1) std::cerr << ", " << gi.n_cycles << " database cycles per file";
            std::cerr << ", peer access " << (gi.peer_access ? "yes" : "no") << ")\n";
            std::cerr << "Total DB size in HBM:\t" << gi.device_bytes_max / 1000000 / 1000.0 << " GB per device (" << gi.n_keys
                      << " k-mers, " << (info.index_kind == MC_INDEX_MINIMIZER ? "minimizer index" : info.index_kind == MC_INDEX_SUPERKMER ? "super-k-mer index" : "bucket-line table")
                      << (info.index_fallback ? " [fallback: the minimizer index did not fit]" : "") << ", " << info.line_bytes
                      << "-byte lines)\n";
            std::cerr << "DB loaded.\n";
        } else {
            std::cerr << "CuCLARK initialized.\n";
        }
    }

    // one input file -> one CSV (reference runSimple + getObjectsDataComputeFullGPU + print*)
    void run_simple(const char *objects, const char *result)
    {
        std::cerr << "Classifying: " << objects << "\n";
        if (db_cycles == 1 && !opt.dump && !getenv("MC_GZ_WHOLE") && GzSegments::is_gzip(objects)) { run_gz_segments(objects, result); return; }
        InputImage img;                 // mmap, or inflated in memory when the file is gzip (database cycles, --dump-batches)
        std::string ierr;
        if (!img.load(objects, ierr, (int)opt.cpu)) { std::cerr << ierr << std::endl; return; }
        run_image(img.data(), img.size(), result);
    }

    // A gzip file goes through in segments of whole records (MC_GZ_SEGMENT_MB of text, 256 by default): a thread inflates the
    // next ones while this one is classified -- the streamed plan per segment, the whole-segment plan when that gives up -- and
    // every segment's lines are written behind the lines of the one before.  Three segments of text are in memory at a time
    // instead of the whole file; the first lines are out after one segment's worth of inflating; bgzip's blocks are inflated
    // by all -n threads at once, any other gzip file by one.  (The reference's wrapper
    // copies the file and gunzips the copy before the classifier starts: scripts/classify_metagenome.sh:118-137.)
    void run_gz_segments(const char *objects, const char *result)
    {
        size_t seg_bytes = (size_t)256 << 20;
        if (const char *e = getenv("MC_GZ_SEGMENT_MB")) { const long v = atol(e); if (v >= 1 && v <= 65536) seg_bytes = (size_t)v << 20; }
        if (const char *e = getenv("MC_GZ_SEGMENT_BYTES")) { const long long v = atoll(e); if (v >= 64) seg_bytes = (size_t)v; }       // (tests)
        GzSegments G;
        std::string err;
        if (!G.open(objects, seg_bytes, err, (int)opt.cpu)) { std::cerr << err << std::endl; return; }       // (-n threads inflate a BGZF file)
        struct timeval t0;
        gettimeofday(&t0, nullptr);
        size_t stream_min = 8u << 20;
        if (const char *e = getenv("MC_STREAM_MIN_BYTES")) stream_min = (size_t)std::strtoull(e, nullptr, 10);
        Seg st;
        st.nz_min = (long)T.names.size() - 1;
        GzSegments::Segment s;
        size_t n_seg = 0;
        while (G.next(s, err)) {
            if (s.size == 0 && st.first) { std::cerr << "Failed to open " << objects << std::endl; return; }       // (as an empty file)
            st.last = s.last;
            bool done = false;
            if (s.size >= stream_min && (s.data[0] == '>' || s.data[0] == '@')) done = classify_image(s.data, s.size, result, true, nullptr, nullptr, &st);
            if (!done) classify_image(s.data, s.size, result, false, nullptr, nullptr, &st);
            st.first = false;
            n_seg++;
        }
        if (!err.empty()) {
            // the text stops being gzip somewhere: nothing is classified when the first segment already fails (as the whole-file
            // path, which inflates first); later, the lines written so far stay and the run ends with an error
            std::cerr << err << std::endl;
            if (n_seg == 0) return;
            std::cerr << "ERROR: " << objects << ": the results file holds the first " << st.objects << " reads only." << std::endl;
            std::exit(1);
        }
        if (opt.verbose) {
            std::cerr << "gzip input classified in " << n_seg << " segment(s) of at most " << G.largest_segment() << " bytes of text";
            if (G.bgzf()) std::cerr << " (BGZF: " << G.bgzf_blocks() << " blocks inflated on " << opt.cpu << " threads)";
            std::cerr << "\n";
        }
        done_line(t0, result);
    }

    // paired FASTQ mates (the reference goes through a temporary FASTA file): classified straight from the two files
    // when they are large and regular, joined in memory first otherwise
    void run_paired(const char *f1, const char *f2, const char *result)
    {
        std::cerr << "Classifying: " << f1 << " + " << f2 << "\n";
        InputImage a, b, joined;
        std::string err;
        {
            // the two files side by side (gzip mates: two inflating threads, or -n / 2 threads per file for BGZF)
            std::string err2;
            bool ok2 = true;
            const int th = (int)std::max<size_t>(1, opt.cpu / 2);
            std::thread second([&]() { ok2 = b.load(f2, err2, th); });
            const bool ok1 = a.load(f1, err, th);
            second.join();
            if (!ok1 || !ok2) { std::cerr << (ok1 ? err2 : err) << std::endl; std::exit(1); }
        }
        struct timeval t0;
        gettimeofday(&t0, nullptr);
        size_t stream_min = 8u << 20;
        if (const char *e = getenv("MC_STREAM_MIN_BYTES")) stream_min = (size_t)std::strtoull(e, nullptr, 10);
        if (a.size() >= stream_min && b.size() >= stream_min && !opt.dump && a.data()[0] == '@' && b.data()[0] == '@' &&
            !getenv("MC_JOIN_MATES") && db_cycles == 1) {
            const Mates m{b.data(), b.size()};
            if (classify_image(a.data(), a.size(), result, true, &m)) { done_line(t0, result); return; }
        }
        uint8_t *buf = nullptr;
        size_t buf_len = 0;
        struct timeval tj0, tj1;
        gettimeofday(&tj0, nullptr);
        if (!merge_paired_parallel(a.data(), a.size(), b.data(), b.size(), (int)opt.cpu, &buf, &buf_len, err)) { perror(err.c_str()); std::exit(1); }   // as file.cc:220-259
        if (buf_len == 0) { std::free(buf); std::cerr << "Failed to open " << f1 << std::endl; return; }
        joined.adopt_raw(buf, buf_len);
        gettimeofday(&tj1, nullptr);
        if (opt.verbose)
            std::cerr << "timing: mates joined in memory (" << opt.cpu << " threads) "
                      << (tj1.tv_sec - tj0.tv_sec) + (tj1.tv_usec - tj0.tv_usec) / 1e6 << " s\n";
        run_image(joined.data(), joined.size(), result);
    }

    void done_line(const struct timeval &t0, const char *result)
    {
        struct timeval t1;
        gettimeofday(&t1, nullptr);
        const double diff = (t1.tv_sec - t0.tv_sec) + (t1.tv_usec - t0.tv_usec) / 1000000.0;
        char buf[256];
        std::snprintf(buf, sizeof buf, "Done in %.1fs (%zu reads/min, %zu reads)\n", diff,
                      (size_t)(((double)n_objects) / diff * 60.0), n_objects);
        std::cerr << buf << "Results: " << result << ".csv\n";
    }

    // One batch of the file on its way through the pipeline.  Its reads are either a slice of ONE index of the
    // whole file (small files, --dump-batches, or after the streamed attempt gave up) or its own index of a
    // byte range of the file (streamed: offsets relative to `text`).
    struct Batch {
        const ReadIndex *R = nullptr;
        ReadIndex own;
        const uint8_t *text = nullptr;
        size_t r0 = 0, n = 0;           // reads [r0, r0 + n) of *R
        size_t ncon = 0;
        bool indexed = false, submitted = false;
        ReadIndex own2;                 // mates classified straight from their two files: the records of file 2
        const uint8_t *text2 = nullptr;
        const uint32_t *dev_hdr = nullptr, *dev_len = nullptr;      // records cut on the card (mc_text_*): header offsets, sequence lengths
    };
    // file 2 of a pair, when the mates are not joined into one text first (streamed plan only)
    struct Mates { const uint8_t *b; size_t nb; };


    void run_image(const uint8_t *map, size_t nb, const char *result)
    {
        struct timeval t0;
        gettimeofday(&t0, nullptr);
        // Large files are STREAMED: cut into byte ranges at record starts, one per batch, and each range is
        // indexed, packed and submitted by one task -- no index of the whole file first, nothing waits for the
        // last indexer thread, the GPU starts after 1/nbatch of the scanning.  The pinned buffers are sized
        // from the head of the file; if a range turns out to hold more than they take, the attempt is given up
        // and the file goes through the plan that indexes it as a whole (as small files and --dump-batches do).
        size_t stream_min = 8u << 20;
        if (const char *e = getenv("MC_STREAM_MIN_BYTES")) stream_min = (size_t)std::strtoull(e, nullptr, 10);
        bool done = false;
        if (db_cycles > 1) {
            // The table does not fit the devices together: the file is classified once per database cycle -- the same batches every
            // time (the plan that indexes the file as a whole), their sparse rows kept here in between and merged on the card with
            // what the next cycle's parts find; the last pass writes the CSV.
            // The passes add up in any order: a file starts with the parts the file before it ended with (C - 1 loads per file, where
            // the reference goes back to its first parts for every file: C).
            std::vector<uint16_t> rows;
            for (uint32_t i = 0; i < db_cycles; i++) {
                if (i) db_cycle = (db_cycle + 1) % db_cycles;
                mc_check(mc_group_set_cycle(grp, db_cycle), "mc_group_set_cycle");
                Cycle cy{i, db_cycles, &rows};
                classify_image(map, nb, result, false, nullptr, &cy);
            }
            done_line(t0, result);
            return;
        }
        if (nb >= stream_min && !opt.dump && (map[0] == '>' || map[0] == '@')) done = classify_image(map, nb, result, true);
        if (!done) classify_image(map, nb, result, false);
        done_line(t0, result);
    }

    // false: the streamed attempt met a byte range its buffers do not take (nothing of the result is kept)
    // seg: this text is one segment of a file (run_gz_segments): lines go behind those of the segments before, the header
    // with the first, the closing messages with the last
    bool classify_image(const uint8_t *map, size_t nb, const char *result, const bool streamed, const Mates *mates = nullptr,
                        const Cycle *cyc = nullptr, Seg *seg = nullptr)
    {
        const std::string csv = std::string(result) + ".csv";
        const bool last_cycle = !cyc || cyc->i + 1 == cyc->n;
        const bool want_rows = opt.ext || cyc;
        const bool opens_csv = !seg || seg->first, closes_csv = !seg || seg->last;
        FILE *fout = std::fopen(csv.c_str(), opens_csv ? "w" : "r+");
        if (!fout) { std::cerr << "Failed to create/open file result: " << csv << std::endl; return true; }

        ReadIndex R;
        std::string err;
        auto now = []() { struct timeval tv; gettimeofday(&tv, nullptr); return tv.tv_sec + tv.tv_usec / 1e6; };
        const double ts0 = now();
        // Pinning the batch buffers takes 6-10 ms and needs no CPU: for a large file it runs NEXT TO the indexing, with
        // sizes guessed from the head of the file (10 % head room); the guess is checked against what the index
        // says (whole-file plan: the buffers are allocated again if it was too small; streamed: see above).
        size_t guess_reads = 0, guess_con = 0, guess_nbuf = 0;
        bool dev_ingest = false, hybrid = false;       // dev_ingest: ranges go up as text (hybrid: only card_share percent of them)
        std::thread early_alloc;
        int early_rc = MC_OK;
        std::string early_err;
        size_t nbatch_g = std::max<size_t>(1, opt.batches);
        // (card ingest: ranges of about MC_TEXT_RANGE_MB of text -- the first batch is up after a few milliseconds, and a ring of
        // a dozen buffers keeps uploads, kernels and the formatting of earlier batches going side by side)
        size_t text_range = 48u << 20;
        if (const char *e = getenv("MC_TEXT_RANGE_MB")) { const long v = atol(e); if (v >= 1 && v <= 2048) text_range = (size_t)v << 20; }
        // MC_CARD_SHARE (percent, with the card's ingest on): that share of the ranges goes up as text, the others are indexed and
        // packed here -- the two ways through a batch are bound by different things (the link: 317 bytes a read; the host's cores: two
        // passes over the text), so a file takes both at once
        size_t card_share = 100;
        if (const char *e = getenv("MC_CARD_SHARE")) { const long v = atol(e); if (v >= 0 && v <= 100) card_share = (size_t)v; }
        const bool card_wanted = streamed && opt.gpu_ingest && card_share > 0 && map[0] == '@' && !mates && !opt.ext && !opt.dump && text_path;
        if (card_wanted) nbatch_g = std::max(nbatch_g, (nb + text_range - 1) / text_range);
        if (nb >= (8u << 20) || streamed) {
            ReadIndex H;
            std::string herr;
            const size_t head = std::min<size_t>(nb, 1u << 20);
            if (!opt.dump && index_reads(map, head, H, herr) && H.size() > 8) {
                const size_t hn = H.size() - 1;                                  // the last record of the head is cut short
                const double per_read = (double)H.spos[hn] / (double)hn;         // bytes per record
                const double con_per_read = (double)container_bound(H, 0, hn, (unsigned)opt.k) / (double)hn;
                guess_reads = (size_t)((double)nb / per_read / (double)nbatch_g * 1.10) + 64;
                guess_con = (size_t)((double)guess_reads * con_per_read * (mates ? 2.2 : 1.05)) + 64;      // (a mate of its own per read)
                guess_nbuf = std::min(nbatch_g, std::max<size_t>(2, std::min<size_t>(opt.cpu, card_wanted ? 12 : 8)));
                // (plain FASTQ, streamed, final rows only: the card cuts and packs the records -- its buffers are allocated below,
                // once the byte ranges are known)
                dev_ingest = card_wanted && guess_con <= 0xFFFFFFFFull;
                hybrid = dev_ingest && card_share < 100 && nbatch_g >= 4;
                if (guess_con <= 0xFFFFFFFFull && !dev_ingest)
                    early_alloc = std::thread([&]() {
                        early_rc = mc_group_alloc_batches(grp, (uint32_t)guess_nbuf, guess_reads, guess_con, want_rows ? 1 : 0);
                        if (early_rc != MC_OK) early_err = mc_last_error();
                    });
            }
            if (streamed && !early_alloc.joinable() && !dev_ingest) { std::fclose(fout); return false; }     // no guess: whole-file plan
        }

        size_t nbatch, nbuf, cap_reads = 0, cap_con = 0;
        std::vector<Batch> B;
        std::vector<size_t> cut, cut2;                // streamed: byte range of batch b = [cut[b], cut[b + 1]) (cut2: of file 2)
        double ts1, ts2;
        if (streamed) {
            nbatch = nbatch_g;
            nbuf = guess_nbuf;
            cap_reads = guess_reads; cap_con = guess_con;
            const bool fastq = map[0] == '@';
            cut.resize(nbatch + 1);
            for (size_t b = 0; b <= nbatch; b++)
                cut[b] = b == nbatch ? nb : record_start_at_or_after(map, nb, (size_t)((unsigned __int128)nb * b / nbatch), fastq);
            if (mates) {
                // the same cuts in file 2, by id
                cut2.resize(nbatch + 1);
                bool found = true;
                for (size_t b = 0; b <= nbatch && found; b++) {
                    if (b == 0) cut2[b] = 0;
                    else if (cut[b] >= nb) cut2[b] = mates->nb;
                    else found = find_mate(map, nb, cut[b], mates->b, mates->nb, cut2[b]);
                }
                for (size_t b = 0; b < nbatch && found; b++) found = cut2[b] <= cut2[b + 1];
                if (!found) {
                    // mates that are not at the same relative place (reads trimmed to different lengths): count records
                    found = align_mates(map, nb, mates->b, mates->nb, cut, (int)opt.cpu, cut2);
                    if (found && opt.verbose) std::cerr << "mates located by counting records (not at the same relative place of the two files)\n";
                }
                if (!found) {          // not the regular pair of files this plan is for: join the mates first
                    if (opt.verbose) std::cerr << "streamed ingest of the two files given up (a record of file 1 has no mate at its place in file 2); joining the mates first\n";
                    early_alloc.join();
                    if (early_rc == MC_OK) mc_group_free_batches(grp);
                    std::fclose(fout);
                    return false;
                }
            }
            if (dev_ingest) {
                size_t max_text = 16;
                for (size_t b = 0; b < nbatch; b++) max_text = std::max(max_text, cut[b + 1] - cut[b] + 16);
                early_alloc = std::thread([&, max_text]() {
                    early_rc = mc_group_text_alloc(grp, (uint32_t)guess_nbuf, max_text, guess_reads, guess_con);
                    if (early_rc == MC_OK && hybrid)
                        early_rc = mc_group_alloc_batches(grp, (uint32_t)std::max<size_t>(2, std::min<size_t>(opt.cpu, 8)), guess_reads, guess_con, 0);
                    if (early_rc != MC_OK) early_err = mc_last_error();
                });
            }
            B.resize(nbatch);
            ts1 = ts2 = now();
        } else {
            if (!index_reads_parallel(map, nb, (int)opt.cpu, R, err)) { std::cerr << err << std::endl; std::exit(-1); }
            ts1 = now();
            const size_t n_all = R.size();
            nbatch = std::max<size_t>(1, std::min(opt.batches, n_all));
            B.resize(nbatch);
            size_t max_reads = 1, max_con = 8;
            {
                std::vector<size_t> bound(nbatch, 0);
                for (size_t b = 0; b < nbatch; b++) {
                    B[b].R = &R; B[b].text = map; B[b].r0 = n_all * b / nbatch; B[b].n = n_all * (b + 1) / nbatch - B[b].r0;
                    B[b].indexed = true;
                }
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
                for (long b = 0; b < (long)nbatch; b++) bound[b] = container_bound(R, B[b].r0, B[b].r0 + B[b].n, (unsigned)opt.k);
                for (size_t b = 0; b < nbatch; b++) {
                    max_reads = std::max(max_reads, B[b].n);
                    max_con = std::max(max_con, bound[b]);
                }
            }
            if (max_con > 0xFFFFFFFFull) die("ERROR: Batch overflow. Please increase the number of batches (-b <numberofbatches>).", -1);

            // The reference pins buffers for ALL batches of a file at once (CuClarkDB::malloc, CuClarkDB.cu:321-421).
            // Pinning costs ~0.3 ms per MB, so the batches go through a ring of a few buffer sets instead: batch b
            // uses set b % nbuf and is packed once batch b - nbuf has been formatted.
            nbuf = opt.dump ? nbatch : std::min(nbatch, std::max<size_t>(2, std::min<size_t>(opt.cpu, 8)));
            bool have_buffers = false;
            if (early_alloc.joinable()) {
                early_alloc.join();
                have_buffers = early_rc == MC_OK && guess_nbuf == nbuf && guess_reads >= max_reads && guess_con >= max_con;
                if (!have_buffers && early_rc == MC_OK) mc_group_free_batches(grp);          // the guess was too small
            }
            if (!have_buffers)
                mc_check(mc_group_alloc_batches(grp, (uint32_t)nbuf, max_reads, max_con, want_rows ? 1 : 0), "mc_group_alloc_batches");
            ts2 = now();
        }

        // Which ring of buffers a batch goes through (the card's text buffers or the pinned sets of packed reads), which slot of
        // it, and which batches use that slot before and after it: a batch is packed / uploaded once the one before it in its
        // slot has been formatted.
        const size_t NONE = (size_t)-1;
        std::vector<uint8_t> on_card(nbatch, dev_ingest ? 1 : 0);
        std::vector<uint32_t> slot_of(nbatch, 0);
        std::vector<size_t> next_in_slot(nbatch, NONE), prev_in_slot(nbatch, NONE);
        size_t ring_card = dev_ingest ? nbuf : 0, ring_host = dev_ingest ? 0 : nbuf;
        if (hybrid) {
            size_t n_card = 0;
            for (size_t b = 0; b < nbatch; b++) { on_card[b] = (b + 1) * card_share / 100 != b * card_share / 100; n_card += on_card[b]; }
            ring_card = std::min(n_card, nbuf);
            ring_host = std::min(nbatch - n_card, std::max<size_t>(2, std::min<size_t>(opt.cpu, 8)));
        }
        {
            std::vector<size_t> last_card(std::max<size_t>(ring_card, 1), NONE), last_host(std::max<size_t>(ring_host, 1), NONE);
            size_t jc = 0, jh = 0;
            for (size_t b = 0; b < nbatch; b++) {
                std::vector<size_t> &last = on_card[b] ? last_card : last_host;
                const size_t sl = on_card[b] ? jc++ % ring_card : jh++ % ring_host;
                slot_of[b] = (uint32_t)sl;
                prev_in_slot[b] = last[sl];
                if (last[sl] != NONE) next_in_slot[last[sl]] = b;
                last[sl] = b;
            }
        }

        const uint32_t flags = (last_cycle ? MC_F_FINAL : 0u) | (want_rows ? MC_F_ROWS : 0u) | (cyc && cyc->i ? MC_F_FOLLOWUP : 0u);
        const size_t rows_len = 2 * (size_t)MAXHITS + 2;
        if (cyc && cyc->i == 0) cyc->rows->assign(R.size() * rows_len, 0);

        // One pool of worker threads runs three kinds of tasks: INDEX (streamed plan: the records of a byte range),
        // PACK (2-bit pack a batch into its pinned buffers, then submit it: H2D, kernel, D2H are asynchronous) and
        // FORMAT (one slice of a finished batch's CSV lines).  The main thread waits for the batches in order,
        // hands their slices to the pool and writes them out -- so indexing, packing, the GPU, formatting and
        // writing overlap instead of running as phases (the reference serialises queryBatch behind an omp critical
        // and prints on one thread, src/CuCLARK_hh.hh:1737-1741, :2073-2121).
        Pool pool((int)std::max<size_t>(1, opt.cpu));
        std::mutex submit_mu, done_mu;
        std::condition_variable done_cv;
        bool buffers_ready = !streamed, gave_up = false, dev_gave_up = false;        // under done_mu
        double t_alloc = 0.0, t_submit = 0.0, t_wait = 0.0;                     // card ingest: seconds inside mc_group_text_submit (all tasks) / _wait (main thread)
        auto index_batch = [&](size_t b) {
            Batch &X = B[b];
            std::string ierr;
            const size_t len = cut[b + 1] - cut[b];
            X.R = &X.own; X.text = map + cut[b]; X.r0 = 0;
            if (on_card[b]) {          // the card will say where the records are
                { std::lock_guard<std::mutex> lk(done_mu); X.indexed = true; }
                done_cv.notify_all();
                return;
            }
            if (len && !index_reads(map + cut[b], len, X.own, ierr)) { std::cerr << ierr << std::endl; std::exit(-1); }
            X.n = X.own.size();
            bool mates_ok = true;
            if (mates) {
                // the same records of file 2; names become the ids, lengths the joined lengths (R1 'N' R2)
                const size_t len2 = cut2[b + 1] - cut2[b];
                X.text2 = mates->b + cut2[b];
                if (len2 && !index_reads(mates->b + cut2[b], len2, X.own2, ierr)) mates_ok = false;
                mates_ok = mates_ok && X.own2.size() == X.n;
                for (size_t i = 0; mates_ok && i < X.n; i++) {
                    size_t s1, e1, s2, e2;
                    mate_id(X.text, X.own.name_s[i] - 1, line_end(X.text, len, X.own.name_s[i] - 1), s1, e1);
                    mate_id(X.text2, X.own2.name_s[i] - 1, line_end(X.text2, len2, X.own2.name_s[i] - 1), s2, e2);
                    mates_ok = e1 - s1 == e2 - s2 && std::memcmp(X.text + s1, X.text2 + s2, e1 - s1) == 0;
                    X.own.name_s[i] = s1; X.own.name_e[i] = e1;
                    X.own.len[i] = X.own.len[i] + 1 + X.own2.len[i];
                }
            }
            const bool fits = mates_ok && X.n <= cap_reads && container_bound(X.own, 0, X.n, (unsigned)opt.k) <= cap_con;
            { std::lock_guard<std::mutex> lk(done_mu); X.indexed = true; if (!fits) gave_up = true; }
            done_cv.notify_all();
        };
        auto pack_batch = [&](size_t b) {
            Batch &X = B[b];
            bool stop;
            {
                std::unique_lock<std::mutex> lk(done_mu);
                done_cv.wait(lk, [&]() { return gave_up || (X.indexed && buffers_ready); });
                stop = gave_up;
            }
            if (!stop && on_card[b]) {
                // the bytes go up from where the file is mapped, on the buffer's own queue: several tasks upload at once
                const double t_a = now();
                mc_check(mc_group_text_submit(grp, slot_of[b], map + cut[b], cut[b + 1] - cut[b]), "mc_group_text_submit");
                { std::lock_guard<std::mutex> lk(done_mu); t_submit += now() - t_a; }
            } else if (!stop) {
                const uint32_t buf = slot_of[b];
                uint32_t *ptr; uint16_t *con;
                mc_check(mc_group_batch_buffers(grp, buf, &ptr, &con, nullptr, nullptr), "mc_group_batch_buffers");
                X.ncon = mates ? pack_mates(X.text, X.own, X.text2, X.own2, X.n, (unsigned)opt.k, ptr, con, (size_t)(map + nb - X.text),
                                            (size_t)(mates->b + mates->nb - X.text2))
                               : pack_reads(X.text, *X.R, X.r0, X.r0 + X.n, (unsigned)opt.k, ptr, con, (size_t)(map + nb - X.text));
                if (cyc && cyc->i) {
                    uint16_t *rows_in = nullptr;
                    mc_check(mc_group_batch_buffers(grp, buf, nullptr, nullptr, nullptr, &rows_in), "mc_group_batch_buffers");
                    std::memcpy(rows_in, cyc->rows->data() + X.r0 * rows_len, X.n * rows_len * 2);
                }
                std::lock_guard<std::mutex> lk(submit_mu);
                mc_check(mc_group_submit(grp, buf, X.n, X.ncon, flags), "mc_group_submit");
            }
            { std::lock_guard<std::mutex> lk(done_mu); X.submitted = true; }
            done_cv.notify_all();
        };
        auto enqueue_pack = [&](size_t b) { pool.run([&, b]() { pack_batch(b); }); };
        if (streamed) {
            // the first nbuf batches are indexed and packed by one task each; the ranges behind them are indexed
            // by whoever is free (their buffers are in use until an earlier batch has been formatted)
            for (size_t b = 0; b < nbatch; b++) if (prev_in_slot[b] == NONE) pool.run([&, b]() { index_batch(b); pack_batch(b); });
            for (size_t b = 0; b < nbatch; b++) if (prev_in_slot[b] != NONE) pool.run([&, b]() { index_batch(b); });
            early_alloc.join();
            t_alloc = now();
            if (early_rc != MC_OK) die(std::string(dev_ingest ? "mc_group_text_alloc: " : "mc_group_alloc_batches: ") + early_err);
            { std::lock_guard<std::mutex> lk(done_mu); buffers_ready = true; }
            done_cv.notify_all();
        } else {
            for (size_t b = 0; b < nbatch; b++) if (prev_in_slot[b] == NONE) enqueue_pack(b);
        }

        if (cyc && !last_cycle) {
            for (size_t b = 0; b < nbatch; b++) {
                { std::unique_lock<std::mutex> lk(done_mu); done_cv.wait(lk, [&]() { return B[b].submitted; }); }
                uint16_t *rows_out = nullptr;
                mc_check(mc_group_wait(grp, slot_of[b]), "mc_group_wait");
                mc_check(mc_group_batch_buffers(grp, slot_of[b], nullptr, nullptr, nullptr, &rows_out), "mc_group_batch_buffers");
                std::memcpy(cyc->rows->data() + B[b].r0 * rows_len, rows_out, B[b].n * rows_len * 2);
                if (next_in_slot[b] != NONE) enqueue_pack(next_in_slot[b]);
            }
            pool.finish();
            std::fclose(fout);
            if (opt.verbose) std::cerr << "timing: database cycle " << cyc->i + 1 << " of " << cyc->n << ": " << now() - ts0 << " s\n";
            mc_group_free_batches(grp);
            return true;
        }
        // header (reference :1951-1967)
        std::string head = "Object_ID";
        if (opt.ext) for (size_t t = 1; t < T.names.size(); t++) { head += ","; head += T.names[t]; }
        head += ",Gamma,Assignment,Score,Confidence\n";
        if (opens_csv) std::fwrite(head.data(), 1, head.size(), fout);
        std::fflush(fout);
        const int fd = fileno(fout);
        uint64_t file_off = opens_csv ? head.size() : seg->csv_bytes;             // slices are written with pwrite by the pool, in parallel
        if (opens_csv) std::cerr << (opt.ext ? "Writing extended results... " : "Writing results... ") << std::endl;

        const size_t row_len = 2 * (size_t)MAXHITS + 2;
        long nz_min = (long)T.names.size() - 1, nz_max = 0, nz_sum = 0;
        if (seg && !seg->first) { nz_min = seg->nz_min; nz_max = seg->nz_max; nz_sum = seg->nz_sum; }
        // a batch's lines are formatted in parallel slices (same printf conversions as the
        // reference, :2115-2118) and written in read order
        const int nfmt = (int)std::max<size_t>(1, opt.cpu);
        // a slice's text: a plain buffer sized for the worst case up front and filled with pointer bumps
        // (std::string appends and a 128-bit division per ratio were 3/4 of the formatting time)
        struct Text {
            char *p = nullptr;
            size_t n = 0;
            Text() = default;
            Text(const Text &) = delete;
            Text &operator=(const Text &) = delete;
            ~Text() { std::free(p); }
            void take(size_t cap) { std::free(p); p = static_cast<char *>(std::malloc(cap ? cap : 1)); n = 0; if (!p) die("out of memory", -1); }
            void drop() { std::free(p); p = nullptr; n = 0; }
        };
        struct Formatted {
            std::unique_ptr<Text[]> slice;
            std::vector<long> s_min, s_max, s_sum;
            int left = 0;
        };
        size_t assign_max = 2;          // "NA"
        for (const auto &nm : T.names) assign_max = std::max(assign_max, nm.size());
        std::vector<Formatted> fmt(nbatch);
        auto format_slice = [&](size_t b, int sl, const uint16_t *fin, const uint16_t *rows) {
            Formatted &F = fmt[b];
            const Batch &X = B[b];
            const ReadIndex &RI = *X.R;
            const uint8_t *text = X.text;
            const size_t r0 = X.r0, nr = X.n;
            const size_t i0 = r0 + nr * sl / nfmt, i1 = r0 + nr * (sl + 1) / nfmt;
            if (X.dev_hdr) {
                // records cut on the card: name = the bytes behind '@' up to the first blank / newline, length = the sequence line's
                ReadIndex &W = const_cast<Batch &>(X).own;
                const size_t avail = (size_t)(map + nb - text);
                for (size_t i = i0; i < i1; i++) {
                    size_t e = (size_t)X.dev_hdr[i] + 2;
                    while (e < avail && !is_sep(text[e])) e++;
                    W.name_s[i] = (uint64_t)X.dev_hdr[i] + 1; W.name_e[i] = e; W.len[i] = X.dev_len[i];
                }
            }
            // worst case: name + ",<gamma>," + assignment + ",<best>,<confidence>\n" (+ ",65535" per target when extended)
            size_t cap = 0;
            for (size_t i = i0; i < i1; i++) cap += std::min<size_t>(RI.name_e[i] - RI.name_s[i], OBJECTNAMEMAX - 1);
            cap += (i1 - i0) * (2 * 64 + 16 + assign_max + (opt.ext ? 6 * (T.names.size() - 1) : 0));
            Text &out = F.slice[sl];
            out.take(cap);
            char *o = out.p;
            F.s_min[sl] = (long)T.names.size() - 1; F.s_max[sl] = 0; F.s_sum[sl] = 0;
            // gamma = hits / (length - k + 1): nearly every read of a file has the same length, so the text of
            // hits / den is kept per hits value for the last den seen
            int64_t memo_den = -1;
            std::vector<std::array<char, 16>> memo;          // [0] = length, text from [1]
            // confidence = best / (best + second): the text per (best, second) pair while both stay below 128
            // (a 150 bp read has at most 120-124 hits)
            static thread_local std::vector<std::array<char, 16>> memo2(128 * 128, std::array<char, 16>{});      // (the text depends on the pair only)
            // The only bytes of a record this pass reads are its name: one cache line out of the five a 150 bp FASTQ record spans, gone
            // from the caches since the indexer ran over it.  MC_FMT_PREFETCH=<records> asks for it that many records ahead.  Off by
            // default: file -> CSV at the metric's size, interleaved runs, said 62-65 M reads/s without and 75-80 with 32 ahead on one
            // box, 64-83 without and 58-73 with 48 ahead on the next (profiles/r04_format_prefetch_ab.txt) -- the boxes' own spread
            // is larger than whatever this buys.
            static const size_t pf_ahead = []() { const char *e = getenv("MC_FMT_PREFETCH"); return e ? (size_t)atol(e) : (size_t)0; }();
            for (size_t i = i0; i < i1; i++) {
                if (pf_ahead && i + pf_ahead < i1) __builtin_prefetch(text + RI.name_s[i + pf_ahead], 0, 0);
                const uint16_t *r5 = fin + (i - r0) * MC_FINAL_ROW;
                const uint32_t total = r5[0], ibest = r5[1], best = r5[2], s_best = r5[4];
                size_t nl = RI.name_e[i] - RI.name_s[i];
                if (nl >= OBJECTNAMEMAX) nl = OBJECTNAMEMAX - 1;
                std::memcpy(o, text + RI.name_s[i], nl); o += nl;
                const uint32_t norm = (uint32_t)(paired ? RI.len[i] - NBN : RI.len[i]);     // ITYPE objectNorm
                if (opt.ext) {
                    // all scores, zeros for the targets not hit (reference :2006-2026)
                    const uint16_t *row = rows + (i - r0) * row_len;
                    size_t w = 0;
                    for (uint32_t h = 0; h < row[0]; h++) {
                        const size_t t = row[1 + 2 * h];
                        for (; w < t; w++) { *o++ = ','; *o++ = '0'; }
                        *o++ = ',';
                        o += fmt_u32(o, row[2 + 2 * h]);
                        w++;
                    }
                    for (; w < T.names.size() - 1; w++) { *o++ = ','; *o++ = '0'; }
                    F.s_max[sl] = std::max<long>(F.s_max[sl], row[0]); F.s_min[sl] = std::min<long>(F.s_min[sl], row[0]); F.s_sum[sl] += row[0];
                }
                // ",%g," gamma, assignment, ",%u,%g\n" best, confidence -- the two ratios without printf
                // where format.hpp covers them (it declines the odd cases: reads shorter than k, ties)
                *o++ = ',';
                const int64_t den = (int64_t)norm - (int64_t)opt.k + 1;
                if (den > 0 && den <= 4096 && total <= (uint64_t)den) {
                    if (den != memo_den) { memo.assign((size_t)den + 1, std::array<char, 16>{}); memo_den = den; }
                    std::array<char, 16> &mm = memo[total];
                    if (!mm[0]) {
                        char tmp[64];
                        int m = fmt_ratio_g(tmp, total, (uint64_t)den);
                        if (!m) m = std::snprintf(tmp, sizeof tmp, "%g", (double)total / (((double)norm - (double)opt.k) + 1.0));
                        if (m > 15) m = 0;          // (never: six digits and an exponent)
                        mm[0] = (char)m;
                        std::memcpy(&mm[1], tmp, (size_t)m);
                    }
                    if (mm[0]) { std::memcpy(o, &mm[1], 15); o += mm[0]; }
                    else o += std::snprintf(o, 64, "%g", (double)total / (((double)norm - (double)opt.k) + 1.0));
                } else {
                    const double gamma = (double)total / (((double)norm - (double)opt.k) + 1.0);
                    int m = den > 0 ? fmt_ratio_g(o, total, (uint64_t)den) : 0;
                    if (!m) m = std::snprintf(o, 64, "%g", gamma);
                    o += m;
                }
                *o++ = ',';
                if (ibest < T.names.size()) { const std::string &nm = T.names[ibest]; std::memcpy(o, nm.data(), nm.size()); o += nm.size(); }
                else { *o++ = 'N'; *o++ = 'A'; }
                *o++ = ',';
                o += fmt_u32(o, best);
                *o++ = ',';
                auto confidence = [&](char *dst) -> int {
                    int m = fmt_ratio_g(dst, best, (uint64_t)best + s_best ? (uint64_t)best + s_best : 1u);
                    if (!m) {
                        double delta = (double)(best + s_best);
                        delta = (delta < 0.001) ? 0 : ((double)best) / delta;
                        m = std::snprintf(dst, 64, "%g", delta);
                    }
                    return m;
                };
                if (best < 128u && s_best < 128u) {
                    std::array<char, 16> &mm = memo2[best * 128u + s_best];
                    if (!mm[0]) {
                        char tmp[64];
                        const int m = confidence(tmp);
                        if (m <= 15) { mm[0] = (char)m; std::memcpy(&mm[1], tmp, (size_t)m); }
                    }
                    if (mm[0]) { std::memcpy(o, &mm[1], 15); o += mm[0]; }
                    else o += confidence(o);
                } else {
                    o += confidence(o);
                }
                *o++ = '\n';
            }
            out.n = (size_t)(o - out.p);
        };
        // results of batch b are on the host -> its slices go to the pool; false: the streamed attempt was given up
        auto launch_format = [&](size_t b) -> bool {
            {
                std::unique_lock<std::mutex> lk(done_mu);
                done_cv.wait(lk, [&]() { return B[b].submitted; });
                if (gave_up) return false;
            }
            uint16_t *fin, *rows = nullptr;
            if (on_card[b]) {
                uint64_t n_dev = 0; uint32_t status = 0;
                uint32_t *hdr, *len;
                const double t_a = now();
                mc_check(mc_group_text_wait(grp, slot_of[b], &n_dev, &status), "mc_group_text_wait");
                t_wait += now() - t_a;
                mc_check(mc_group_text_buffers(grp, slot_of[b], &hdr, &len, &fin), "mc_group_text_buffers");
                if (status) {          // a batch the card does not vouch for: the whole file goes the host's way
                    std::lock_guard<std::mutex> lk(done_mu);
                    gave_up = true; dev_gave_up = true;
                    done_cv.notify_all();
                    return false;
                }
                Batch &X = B[b];
                X.n = (size_t)n_dev; X.dev_hdr = hdr; X.dev_len = len;
                X.own.name_s.resize(X.n); X.own.name_e.resize(X.n); X.own.len.resize(X.n);
            } else {
                mc_check(mc_group_wait(grp, slot_of[b]), "mc_group_wait");
                mc_group_batch_buffers(grp, slot_of[b], nullptr, nullptr, &fin, &rows);
            }
            Formatted &F = fmt[b];
            F.slice.reset(new Text[nfmt]); F.s_min.assign(nfmt, 0); F.s_max.assign(nfmt, 0); F.s_sum.assign(nfmt, 0);
            F.left = nfmt;
            for (int sl = 0; sl < nfmt; sl++)
                pool.run([&, b, sl, fin, rows]() {
                    format_slice(b, sl, fin, rows);
                    { std::lock_guard<std::mutex> lk(done_mu); fmt[b].left--; }
                    done_cv.notify_all();
                });
            return true;
        };
        size_t n_done = 0;
        // batch b+1 is formatted while batch b is written (card ingest, whose batches are small: up to four ahead)
        const size_t ahead = dev_ingest ? 4 : 1;
        size_t launched = 0;
        bool ok = launch_format(launched++);
        for (size_t b = 0; ok && b < nbatch; b++) {
            // (a batch further on has been packed or uploaded only if the batch before it in its slot is through: < b)
            while (ok && launched < nbatch && launched <= b + ahead && (prev_in_slot[launched] == NONE || prev_in_slot[launched] < b)) { if (!launch_format(launched)) ok = false; else launched++; }
            { std::unique_lock<std::mutex> lk(done_mu); done_cv.wait(lk, [&]() { return fmt[b].left == 0; }); }
            if (!ok) break;
            if (next_in_slot[b] != NONE) enqueue_pack(next_in_slot[b]);      // this batch's buffers are free again
            Formatted &F = fmt[b];
            const size_t nr = B[b].n;
            n_done += nr;
            for (int sl = 0; sl < nfmt; sl++) {
                const uint64_t at = file_off;
                file_off += F.slice[sl].n;
                if (opt.ext && nr) { nz_max = std::max(nz_max, F.s_max[sl]); nz_min = std::min(nz_min, F.s_min[sl]); nz_sum += F.s_sum[sl]; }
                if (F.slice[sl].n == 0) { F.slice[sl].drop(); continue; }
                pool.run([&, b, sl, at]() {
                    Text &str = fmt[b].slice[sl];
                    const char *p = str.p;
                    size_t left = str.n;
                    uint64_t off = at;
                    while (left) {
                        const ssize_t w = ::pwrite(fd, p, left, (off_t)off);
                        if (w <= 0) die(std::string("Failed to write ") + csv, -1);
                        p += w; left -= (size_t)w; off += (uint64_t)w;
                    }
                    str.drop();
                });
            }
            if (streamed) { B[b].own = ReadIndex(); B[b].own2 = ReadIndex(); }      // this range's index is not needed any more
        }
        pool.finish();
        if (!ok) {
            // a byte range holds more than the guessed buffers take: start over with the whole-file plan
            mc_group_sync(grp);
            mc_group_free_batches(grp);
            std::fclose(fout);
            if (opt.verbose) std::cerr << (dev_gave_up ? "ingest on the card given up (a record it does not vouch for, or a range that exceeds the guessed buffers); indexing the whole file on the host\n"
                                           : mates ? "streamed ingest of the two files given up (mates out of step, or a range exceeds the guessed buffers); joining the mates first\n"
                                                   : "streamed ingest given up (a range of the file exceeds the guessed buffers); indexing the whole file\n");
            return false;
        }
        n_objects = (seg ? seg->objects : 0) + n_done;
        if (seg) { seg->objects = n_objects; seg->csv_bytes = file_off; seg->nz_min = nz_min; seg->nz_max = nz_max; seg->nz_sum = nz_sum; }
        if (opt.dump) {
            FILE *dump = std::fopen(opt.dump, "wb");
            for (size_t b = 0; dump && b < nbatch; b++) {           // (nbuf == nbatch when dumping)
                uint32_t *ptr; uint16_t *con;
                mc_group_batch_buffers(grp, (uint32_t)b, &ptr, &con, nullptr, nullptr);
                const uint64_t n = B[b].n, c = B[b].ncon;
                std::fwrite(&n, 8, 1, dump); std::fwrite(&c, 8, 1, dump);
                std::fwrite(ptr, 4, n + 1, dump); std::fwrite(con, 2, c, dump);
            }
            if (dump) std::fclose(dump);
        }
        std::fclose(fout);
        if (opt.verbose) {
            if (streamed)
                std::cerr << "timing: streamed (" << nbatch << (mates ? " byte ranges of both files" : " byte ranges")
                          << (hybrid ? ", " + std::to_string(card_share) + " % of them as text to the card" : std::string())
                          << (dev_ingest ? ": copy+submit | records cut, packed and classified on the card | wait+format+write, all overlapped) "
                                         : ": index | pack+submit | wait+format+write, all overlapped) ")
                          << now() - ts0 << " s"
                          << (dev_ingest ? " (in mc_group_text_submit, all tasks: " + std::to_string(t_submit) + " s; main thread in mc_group_text_wait: " + std::to_string(t_wait) + " s; buffers ready after " + std::to_string(t_alloc - ts0) + " s)" : std::string()) << "\n";
            else
                std::cerr << "timing: index " << ts1 - ts0 << " s, alloc " << ts2 - ts1 << " s, pack+submit | wait+format+write (overlapped) "
                          << now() - ts2 << " s\n";
        }
        if (closes_csv) std::cerr << "Done." << std::endl;
        if (closes_csv && opt.ext && n_objects)
            std::cerr << "MIN targets: " << nz_min << ", MAX targets: " << nz_max << ", AVG targets: "
                      << (float)nz_sum / n_objects << "\n";
        mc_group_free_batches(grp);
        return true;
    }

    static bool looks_like_sequence_file(const char *path)
    {
        std::ifstream f(path);
        std::string line;
        std::getline(f, line);
        if (line.size() >= 2 && (unsigned char)line[0] == 0x1f && (unsigned char)line[1] == 0x8b) return true;   // gzip
        if (!line.empty() && (line[0] == '>' || line[0] == '@')) return true;
        return split_line(line, 4).size() == 2;     // reference run(): "ele.size() == 2"
    }

    // reference run() single/paired with their "file of files" mode (:383-506)
    void run()
    {
        if (!opt.objects2) {
            paired = false;
            if (!file_readable(opt.results) || looks_like_sequence_file(opt.objects)) { run_simple(opt.objects, opt.results); return; }
            std::ifstream o(opt.objects), r(opt.results);
            std::string ol, rl;
            while (std::getline(o, ol) && std::getline(r, rl)) run_simple(ol.c_str(), rl.c_str());
            return;
        }
        paired = true;
        auto one = [&](const char *f1, const char *f2, const char *res) { run_paired(f1, f2, res); };
        if (!file_readable(opt.results) || looks_like_sequence_file(opt.objects)) { one(opt.objects, opt.objects2, opt.results); return; }
        std::ifstream o1(opt.objects), o2(opt.objects2), r(opt.results);
        std::string a, b, rl;
        while (std::getline(o1, a) && std::getline(o2, b) && std::getline(r, rl)) one(a.c_str(), b.c_str(), rl.c_str());
    }
};

} // namespace

int main(int argc, char **argv)
{
    // more hardware queues than the runtime's default of 4, before it starts: the copy-in, compute and copy-out
    // queues of the batch interface must not share one (csrc/mc_api.hip, mc_open)
    setenv("GPU_MAX_HW_QUEUES", "8", 0);
    if (argc == 2) {
        const std::string v(argv[1]);
        if (v == "--help" || v == "--HELP") { print_usage(); return 0; }
        if (v == "--version" || v == "--VERSION") {
            std::cout << "Version: " << MC_HOST_VERSION << " (MI355X-native build of the cuCLARK classification path)" << std::endl;
            return 0;
        }
    }
    if (argc < 6) {
        std::cerr << "To run " << argv[0] << ", at least four  parameters are necessary:\n"
                  << "filename of the targets definition, directory of database, filename for objects, filename for results." << std::endl;
        print_usage();
        return -1;
    }
    Classifier C;
    Options &o = C.opt;
    for (int i = 1; i < argc; i++) {
        const std::string val(argv[i]);
        auto need = [&](const char *msg) { if (++i >= argc) die(msg); return argv[i]; };
        if (val == "-k") { o.k = (size_t)atoi(need("Please specify the k-mer length!")); if (o.k <= 1 || o.k > MAXK) die("The k-mer length should be in [2," + std::to_string(MAXK) + "]."); continue; }
        if (val == "-t") { o.minT = (unsigned)atoi(need("Please specify the minimum frequency (targets)!")); if (o.minT >= 65536) die("The min k-mer frequency should be in [0,65535]."); continue; }
        if (val == "-n") { o.cpu = (size_t)atoi(need("Please specify the number of threads!")); if (o.batches < o.cpu) o.batches = o.cpu; if (o.cpu < 1) die("The number of threads should be higher than 0."); continue; }
        if (val == "--tsk") { o.tsk = true; continue; }
        if (val == "--extended") { o.ext = true; continue; }
        if (val == "-T") { o.targets = need("Please specify the targets!"); if (!file_readable(o.targets)) die(std::string("Failed to find/read the file of the targets definition: ") + o.targets); continue; }
        if (val == "-O") { o.objects = need("Please specify the objects!"); if (!file_readable(o.objects)) die(std::string("Failed to find/read the filename of objects: ") + o.objects); continue; }
        if (val == "-P") {
            if (i + 2 >= argc) die("Please specify the paired-end reads!");
            o.objects = argv[++i]; o.objects2 = argv[++i];
            if (!file_readable(o.objects)) die(std::string("Failed to find/read ") + o.objects);
            if (!file_readable(o.objects2)) die(std::string("Failed to find/read ") + o.objects2);
            continue;
        }
        if (val == "-D") { o.folder = need("Please specify the database directory!"); if (!file_readable(o.folder)) die(std::string("Failed to find/read the directory:  ") + o.folder); continue; }
        if (val == "-R") { o.results = need("Please specify where to store results!"); continue; }
        if (val == "-g") { o.gap = (size_t)atoi(need("Please specify a gap value!")); if (o.gap < 4) die("The gap value should be >= 4."); continue; }
        if (val == "-s") { o.sfactor = (unsigned)atoi(need("Please specify a sampling factor value!")); if (o.sfactor < 2 || o.sfactor > SFACTORMAX) die("The sampling factor value should be in the interval [2," + std::to_string(SFACTORMAX) + "]."); continue; }
        if (val == "-b") { o.batches = (size_t)atoi(need("Please specify the number of batches!")); if (o.batches < o.cpu) die("The number of batches should be higher than the number of threads."); continue; }
        if (val == "-d") { o.devices = (size_t)atoi(need("Please specify the number of devices to use!")); if (o.devices < 1) die("The number of devices should be higher than 0."); continue; }
        if (val == "--verbose") { o.verbose = true; continue; }
        if (val == "--dump-batches") { o.dump = need("--dump-batches needs a file"); continue; }
        if (val == "--gpu-build") { o.gpu_build = true; continue; }
        if (val == "--host-ingest") { o.gpu_ingest = false; continue; }
        if (val == "--gpu-ingest") { o.gpu_ingest = true; continue; }
        die("Failed to recognize option: " + val);
    }
    // reference src/main.cc:214-228
    if (HTSIZE == LHTSIZE) { if (o.gap == 0) o.gap = 4; o.k = 27; o.sfactor = 1; }
    else o.gap = 0;
    if (!o.targets || !o.folder || !o.objects || !o.results) {
        std::cerr << "Failed to run " << argv[0] << ": at least four  parameters are necessary"
                  << ": file of targets, directory of database, file of objects, file for results." << std::endl;
        print_usage();
        return 1;
    }
    C.folder = o.folder;
    if (C.folder.back() != '/') C.folder.push_back('/');
    C.key_bytes = key_bytes_for((unsigned)o.k);
#ifdef _OPENMP
    omp_set_num_threads((int)o.cpu);
#endif
    if (const char *e = getenv("MC_GPU_BUILD")) o.gpu_build = o.gpu_build || atoi(e) != 0;
    if (const char *e = getenv("MC_GPU_INGEST")) o.gpu_ingest = atoi(e) != 0;
    std::string err;
    if (!read_targets(o.targets, C.T, err)) die(err, -1);
    C.build_if_missing();
    C.open_devices();
    C.load();
    C.run();
    mc_group_close(C.grp);
    return 0;
}
