// Input images for the host driver: a read file as one contiguous byte range, whatever it is on disk.
//
//  * plain FASTA/FASTQ: mmap, as the reference does (src/CuCLARK_hh.hh:1339-1352);
//  * gzip (magic 1f 8b, also concatenated members): inflated in memory with zlib -- the reference leaves
//    this to its wrapper, which copies the file and runs gunzip on the copy
//    (scripts/classify_metagenome.sh:118-137); BGZF blocks by several threads at once (gzstream.hpp, which also
//    holds the segment-by-segment path single files take);
//  * paired FASTQ: the mates are joined in memory into the FASTA records ">id\nR1NR2" that the reference
//    writes to a temporary "<file1>_ConcatenatedByCLARK.fa" first (mergePairedFiles, src/file.cc:205-268).
#pragma once
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include "gzstream.hpp"

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace host {

class InputImage {
public:
    InputImage() = default;
    InputImage(const InputImage &) = delete;
    InputImage &operator=(const InputImage &) = delete;
    ~InputImage() { release(); }

    const uint8_t *data() const { return data_; }
    size_t size() const { return size_; }
    bool gzipped() const { return gz_; }

    // false + err on failure; an empty file is a failure (as in the reference's size check).  threads: how many inflate a
    // BGZF file's blocks.
    bool load(const char *path, std::string &err, int threads = 1)
    {
        release();
        fd_ = open(path, O_RDONLY);
        struct stat st;
        if (fd_ == -1 || fstat(fd_, &st) != 0 || st.st_size == 0) {
            err = std::string("Failed to open ") + path;
            release();
            return false;
        }
        map_len_ = (size_t)st.st_size;
        map_ = mmap(nullptr, map_len_, PROT_READ, MAP_PRIVATE, fd_, 0);
        if (map_ == MAP_FAILED) { map_ = nullptr; err = "Failed to mmapping the file."; release(); return false; }
        const uint8_t *m = static_cast<const uint8_t *>(map_);
        if (map_len_ >= 2 && m[0] == 0x1f && m[1] == 0x8b) {
            gz_ = true;
            bool blocks = false;
            if (threads > 1 && !inflate_bgzf(m, map_len_, threads, raw_, size_, err, blocks)) { release(); return false; }
            if (!blocks && !inflate_all(m, map_len_, own_, err)) { release(); return false; }
            munmap(map_, map_len_); map_ = nullptr;
            close(fd_); fd_ = -1;
            if (blocks) data_ = raw_;
            else { data_ = own_.data(); size_ = own_.size(); }
            if (size_ == 0) { err = std::string("Failed to open ") + path; return false; }
        } else {
            data_ = m; size_ = map_len_;
        }
        return true;
    }

    // take over a buffer built in memory (paired mates)
    void adopt(std::vector<uint8_t> &&buf)
    {
        release();
        own_ = std::move(buf);
        data_ = own_.data(); size_ = own_.size();
    }
    // the same for a malloc'd block (never zero-filled: the parallel mate join writes every byte itself)
    void adopt_raw(uint8_t *p, size_t n)
    {
        release();
        raw_ = p;
        data_ = p; size_ = n;
    }

private:
    void release()
    {
        if (map_) munmap(map_, map_len_);
        if (fd_ != -1) close(fd_);
        map_ = nullptr; fd_ = -1; map_len_ = 0; data_ = nullptr; size_ = 0; gz_ = false;
        std::vector<uint8_t>().swap(own_);
        std::free(raw_); raw_ = nullptr;
    }

    // A file of BGZF blocks from end to end (trailing bytes that are no gzip member aside): the text's size is the sum of the
    // sizes the blocks carry, so it is inflated in place by `threads` threads.  blocks = false: not such a file, nothing done.
    static bool inflate_bgzf(const uint8_t *src, size_t n, int threads, uint8_t *&out, size_t &out_n, std::string &err, bool &blocks)
    {
        blocks = false;
        size_t pos = 0, total = 0, bl = 0, tl = 0;
        while (pos < n && BgzfInflater::parse(src + pos, n - pos, bl, tl)) { total += tl; pos += bl; }
        if (pos == 0 || (n - pos >= 2 && src[pos] == 0x1f && src[pos + 1] == 0x8b)) return true;      // other members: the plain inflater
        uint8_t *text = static_cast<uint8_t *>(std::malloc(total ? total : 1));
        if (!text) { err = "out of memory (gzip input)"; return false; }
        BgzfInflater B;
        B.init(src, n, threads);
        size_t filled = 0;
        bool eof = false, full = false;
        while (!eof) {
            size_t got = 0;
            if (!B.read(text + filled, total - filled, got, eof, full, err)) { std::free(text); return false; }
            filled += got;
            if (full) { err = "zlib: corrupt gzip input (bgzf block sizes)"; std::free(text); return false; }      // (never: the sizes were summed above)
        }
        out = text; out_n = filled;
        blocks = true;
        return true;
    }

    static bool inflate_all(const uint8_t *src, size_t n, std::vector<uint8_t> &out, std::string &err)
    {
        z_stream z;
        std::memset(&z, 0, sizeof z);
        if (inflateInit2(&z, 15 + 16) != Z_OK) { err = "zlib: inflateInit2 failed"; return false; }
        out.clear();
        out.reserve(n * 4);
        std::vector<uint8_t> chunk(8u << 20);
        size_t pos = 0;
        bool ended = false;                     // the current member is complete
        z.next_in = const_cast<Bytef *>(src);
        z.avail_in = 0;
        for (;;) {
            if (z.avail_in == 0) {              // avail_in is 32 bits wide: feed the input in pieces
                if (pos >= n) break;
                const size_t piece = std::min<size_t>(n - pos, (size_t)1 << 30);
                z.next_in = const_cast<Bytef *>(src + pos);
                z.avail_in = (uInt)piece;
                pos += piece;
            }
            z.next_out = chunk.data();
            z.avail_out = (uInt)chunk.size();
            const int rc = inflate(&z, Z_NO_FLUSH);
            if (rc != Z_OK && rc != Z_STREAM_END && rc != Z_BUF_ERROR) {
                err = std::string("zlib: corrupt gzip input (") + (z.msg ? z.msg : "?") + ")";
                inflateEnd(&z);
                return false;
            }
            out.insert(out.end(), chunk.data(), chunk.data() + (chunk.size() - z.avail_out));
            if (rc == Z_STREAM_END) {
                ended = true;
                if (z.avail_in == 0 && pos >= n) break;
                // another member may follow (bgzip, cat a.gz b.gz); anything else is trailing garbage
                if (z.avail_in >= 2 && !(z.next_in[0] == 0x1f && z.next_in[1] == 0x8b)) break;
                if (inflateReset(&z) != Z_OK) { err = "zlib: inflateReset failed"; inflateEnd(&z); return false; }
                ended = false;
            }
        }
        inflateEnd(&z);
        if (!ended) { err = "zlib: truncated gzip input"; return false; }
        return true;
    }

    const uint8_t *data_ = nullptr;
    size_t size_ = 0;
    void *map_ = nullptr;
    size_t map_len_ = 0;
    int fd_ = -1;
    bool gz_ = false;
    std::vector<uint8_t> own_;
    uint8_t *raw_ = nullptr;
};

// FASTQ mates -> FASTA records ">id\nR1NR2" in memory (reference mergePairedFiles, src/file.cc:205-268: ids
// must match after cutting at ' ', '/', '\t', '@'; same messages).
inline bool merge_paired(const uint8_t *a, size_t na, const uint8_t *b, size_t nb, std::vector<uint8_t> &out,
                         std::string &err, size_t reserve_hint = 0)
{
    struct Lines {
        const uint8_t *p, *end;
        bool next(const uint8_t *&s, size_t &n)              // std::getline semantics
        {
            if (p >= end) return false;
            const uint8_t *nl = static_cast<const uint8_t *>(std::memchr(p, '\n', (size_t)(end - p)));
            s = p;
            n = (size_t)((nl ? nl : end) - p);
            p = nl ? nl + 1 : end;
            return true;
        }
    } A{a, a + na}, B{b, b + nb};
    auto id_of = [](const uint8_t *l, size_t n, const uint8_t *&s, size_t &len) {
        auto sep = [](uint8_t c) { return c == ' ' || c == '/' || c == '\t' || c == '@'; };
        size_t i = 0;
        while (i < n && sep(l[i])) i++;
        size_t e = i;
        while (e < n && !sep(l[e])) e++;
        s = l + i; len = e - i;
    };
    out.clear();
    out.reserve(reserve_hint ? reserve_hint : na / 2 + nb / 2 + 1024);
    const uint8_t *l1, *l2;
    size_t n1, n2;
    bool first = true;
    while (A.next(l1, n1) && B.next(l2, n2)) {
        if (first) {
            first = false;
            if (n1 == 0 || n2 == 0 || l1[0] != l2[0]) { err = "Error: the files have different format!"; return false; }
            if (l1[0] != '@') { err = "Error: paired-end reads must be FASTQ files!"; return false; }
        }
        if (n1 == 0 || n2 == 0 || l1[0] != '@' || l2[0] != '@') continue;
        const uint8_t *i1, *i2;
        size_t k1, k2;
        id_of(l1, n1, i1, k1);
        id_of(l2, n2, i2, k2);
        if (k1 != k2 || std::memcmp(i1, i2, k1) != 0) { err = "Error: read id does not match between files!"; return false; }
        out.push_back('>');
        out.insert(out.end(), i1, i1 + k1);
        out.push_back('\n');
        if (!(A.next(l1, n1) && B.next(l2, n2))) { err = "Error: Found read without sequence"; return false; }
        out.insert(out.end(), l1, l1 + n1);
        out.push_back('N');
        out.insert(out.end(), l2, l2 + n2);
        out.push_back('\n');
        A.next(l1, n1); B.next(l2, n2);     // '+'
        A.next(l1, n1); B.next(l2, n2);     // quality
    }
    return true;
}

// The same join on several threads, for files of well-formed FASTQ (every record four lines, as many records in one
// file as in the other): file 1 is cut into byte ranges at record starts, the records of every range are counted (and
// checked: the line a record starts with begins with '@', the line two below with '+'), the matching record of file 2
// is found from the counts of ITS ranges, and every range pair is joined into a buffer of its own; the buffers are
// copied side by side into one block.  Whenever the files are not that regular -- a blank line, a different number of
// records, a count that does not divide -- the sequential join above decides, messages included.  `n_threads` plain
// threads.  The result is malloc'd: *out, *out_len (free() it).
inline bool merge_paired_parallel(const uint8_t *a, size_t na, const uint8_t *b, size_t nb, int n_threads,
                                  uint8_t **out, size_t *out_len, std::string &err);

} // namespace host
