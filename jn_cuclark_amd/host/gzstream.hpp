// A gzip read file as a sequence of SEGMENTS of whole records, inflated by a thread of its own while the
// segments before are classified -- instead of the whole text in memory before anything runs (InputImage,
// input.hpp; the reference's wrapper copies the file and runs gunzip on the copy first,
// scripts/classify_metagenome.sh:118-137).
//
//  * a segment is `seg_bytes` of inflated text cut back to the last record start in it (FASTA: a '>' that begins
//    a line; FASTQ: a line starting with '@' whose second-next line starts with '+', reads.hpp
//    record_start_at_or_after); what lies behind the cut opens the next segment;
//  * a record longer than a segment (a contig) makes its segment grow until the record ends;
//  * text that starts with neither '>' nor '@' is handed over whole, in one segment (the indexer refuses it with
//    the reference's message);
//  * three buffers go round: one with the classifier, one ready, one being filled;
//  * concatenated members (cat a.gz b.gz), truncated and corrupt input: as InputImage::inflate_all;
//  * BGZF (bgzip, htslib: members of at most 64 KB of text that carry their own size in a "BC" extra field) is inflated
//    by several threads at once: the block headers are walked without inflating anything, every block knows where its
//    text goes from the sizes before it, and the blocks of a segment are dealt to the threads.  A member that is not
//    such a block hands the rest of the file to the one-thread inflater.
#pragma once
#include "reads.hpp"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace host {

// zlib's inflate over a mapped gzip file, handed out in pieces of the caller's choosing
class GzInflater {
public:
    GzInflater() { std::memset(&z_, 0, sizeof z_); }
    GzInflater(const GzInflater &) = delete;
    GzInflater &operator=(const GzInflater &) = delete;
    ~GzInflater() { if (inited_) inflateEnd(&z_); }

    bool init(const uint8_t *src, size_t n, std::string &err)
    {
        src_ = src; n_ = n; pos_ = 0; ended_ = false; done_ = false;
        if (inflateInit2(&z_, 15 + 16) != Z_OK) { err = "zlib: inflateInit2 failed"; return false; }
        inited_ = true;
        z_.next_in = const_cast<Bytef *>(src);
        z_.avail_in = 0;
        return true;
    }

    // Up to `want` bytes of text into dst; `eof`: the stream is exhausted (got may still be > 0).  false + err: corrupt
    // input, or input that stops inside a member.
    bool read(uint8_t *dst, size_t want, size_t &got, bool &eof, std::string &err)
    {
        got = 0; eof = false;
        while (got < want && !done_) {
            if (z_.avail_in == 0) {             // avail_in is 32 bits wide: feed the input in pieces
                if (pos_ >= n_) { done_ = true; break; }
                const size_t piece = std::min<size_t>(n_ - pos_, (size_t)1 << 30);
                z_.next_in = const_cast<Bytef *>(src_ + pos_);
                z_.avail_in = (uInt)piece;
                pos_ += piece;
            }
            const size_t room = std::min<size_t>(want - got, (size_t)1 << 30);
            z_.next_out = dst + got;
            z_.avail_out = (uInt)room;
            const int rc = inflate(&z_, Z_NO_FLUSH);
            if (rc != Z_OK && rc != Z_STREAM_END && rc != Z_BUF_ERROR) {
                err = std::string("zlib: corrupt gzip input (") + (z_.msg ? z_.msg : "?") + ")";
                return false;
            }
            got += room - z_.avail_out;
            if (rc == Z_STREAM_END) {
                ended_ = true;
                if (z_.avail_in == 0 && pos_ >= n_) { done_ = true; break; }
                // another member may follow (bgzip, cat a.gz b.gz); anything else is trailing garbage
                if (z_.avail_in >= 2 && !(z_.next_in[0] == 0x1f && z_.next_in[1] == 0x8b)) { done_ = true; break; }
                if (inflateReset(&z_) != Z_OK) { err = "zlib: inflateReset failed"; return false; }
                ended_ = false;
            }
        }
        if (done_) {
            eof = true;
            if (!ended_) { err = "zlib: truncated gzip input"; return false; }
        }
        return true;
    }

private:
    z_stream z_;
    const uint8_t *src_ = nullptr;
    size_t n_ = 0, pos_ = 0;
    bool inited_ = false, ended_ = false, done_ = false;
};

// BGZF blocks inflated side by side (see the head of this file)
class BgzfInflater {
public:
    // the member at p is a BGZF block: its length in the file and the length of its text
    static bool parse(const uint8_t *p, size_t n, size_t &block_len, size_t &text_len)
    {
        if (n < 18 || p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || !(p[3] & 4)) return false;
        const size_t xend = 12 + ((size_t)p[10] | (size_t)p[11] << 8);
        if (xend > n) return false;
        bool found = false;
        for (size_t i = 12; i + 4 <= xend;) {
            const size_t slen = (size_t)p[i + 2] | (size_t)p[i + 3] << 8;
            if (p[i] == 'B' && p[i + 1] == 'C' && slen == 2 && i + 6 <= xend) {
                block_len = ((size_t)p[i + 4] | (size_t)p[i + 5] << 8) + 1;
                found = true;
                break;
            }
            i += 4 + slen;
        }
        if (!found || block_len < xend + 8 || block_len > n) return false;
        const uint8_t *t = p + block_len - 4;
        text_len = (size_t)t[0] | (size_t)t[1] << 8 | (size_t)t[2] << 16 | (size_t)t[3] << 24;
        return text_len <= 65536;
    }

    void init(const uint8_t *src, size_t n, int threads)
    {
        src_ = src; n_ = n; pos_ = 0; serial_ = false;
        threads_ = std::max(1, threads);
    }
    size_t blocks() const { return blocks_; }

    // As GzInflater::read, block-wise: `full` when the next block's text does not fit what is left of `room`.
    bool read(uint8_t *dst, size_t room, size_t &got, bool &eof, bool &full, std::string &err)
    {
        got = 0; eof = false; full = false;
        if (!serial_) {
            struct Blk { size_t in, len, out, text; };
            std::vector<Blk> blks;
            size_t sum = 0;
            while (pos_ < n_ && blks.size() < 32768) {
                size_t bl = 0, tl = 0;
                if (!parse(src_ + pos_, n_ - pos_, bl, tl)) {
                    if (!blks.empty()) break;               // first the blocks in hand; the next call comes back here
                    // not a block: nothing but trailing bytes (as the serial inflater: ignored), or members of another kind
                    if (n_ - pos_ < 2 || src_[pos_] != 0x1f || src_[pos_ + 1] != 0x8b) { pos_ = n_; break; }
                    if (!tail_.init(src_ + pos_, n_ - pos_, err)) return false;
                    serial_ = true;
                    break;
                }
                if (tl > room - sum) { full = true; break; }
                blks.push_back(Blk{pos_, bl, sum, tl});
                sum += tl;
                pos_ += bl;
            }
            if (!serial_) {
                std::atomic<size_t> nextb{0};
                std::atomic<bool> bad{false};
                std::mutex emu;
                auto work = [&]() {
                    z_stream z;
                    std::memset(&z, 0, sizeof z);
                    if (inflateInit2(&z, 15 + 16) != Z_OK) { bad = true; std::lock_guard<std::mutex> lk(emu); err = "zlib: inflateInit2 failed"; return; }
                    for (size_t i; !bad && (i = nextb++) < blks.size();) {
                        const Blk &b = blks[i];
                        if (b.text == 0) continue;
                        inflateReset(&z);
                        z.next_in = const_cast<Bytef *>(src_ + b.in); z.avail_in = (uInt)b.len;
                        z.next_out = dst + b.out; z.avail_out = (uInt)b.text;
                        const int rc = inflate(&z, Z_FINISH);
                        if (rc != Z_STREAM_END || z.avail_out != 0) {
                            bad = true;
                            std::lock_guard<std::mutex> lk(emu);
                            err = std::string("zlib: corrupt gzip input (bgzf block: ") + (z.msg ? z.msg : "size mismatch") + ")";
                        }
                    }
                    inflateEnd(&z);
                };
                const size_t nt = std::min<size_t>((size_t)threads_, (blks.size() + 7) / 8);
                if (nt <= 1) work();
                else {
                    std::vector<std::thread> th;
                    for (size_t t = 0; t + 1 < nt; t++) th.emplace_back(work);
                    work();
                    for (auto &t : th) t.join();
                }
                if (bad) return false;
                blocks_ += blks.size();
                got = sum;
                eof = pos_ >= n_;
                return true;
            }
        }
        if (!tail_.read(dst, room, got, eof, err)) return false;
        full = got == room;
        return true;
    }

private:
    const uint8_t *src_ = nullptr;
    size_t n_ = 0, pos_ = 0, blocks_ = 0;
    int threads_ = 1;
    bool serial_ = false;
    GzInflater tail_;
};

class GzSegments {
public:
    struct Segment {
        const uint8_t *data = nullptr;
        size_t size = 0;
        bool last = false;
    };

    GzSegments() = default;
    GzSegments(const GzSegments &) = delete;
    GzSegments &operator=(const GzSegments &) = delete;
    ~GzSegments() { close(); }

    // true when the file starts with the gzip magic (the caller keeps plain files on the mmap path)
    static bool is_gzip(const char *path)
    {
        const int fd = ::open(path, O_RDONLY);
        if (fd == -1) return false;
        uint8_t m[2] = {0, 0};
        const ssize_t r = ::read(fd, m, 2);
        ::close(fd);
        return r == 2 && m[0] == 0x1f && m[1] == 0x8b;
    }

    // threads: how many inflate a BGZF file's blocks (any other gzip file is one stream: one thread)
    bool open(const char *path, size_t seg_bytes, std::string &err, int threads = 1)
    {
        close();
        fd_ = ::open(path, O_RDONLY);
        struct stat st;
        if (fd_ == -1 || fstat(fd_, &st) != 0 || st.st_size == 0) { err = std::string("Failed to open ") + path; close(); return false; }
        map_len_ = (size_t)st.st_size;
        map_ = mmap(nullptr, map_len_, PROT_READ, MAP_PRIVATE, fd_, 0);
        if (map_ == MAP_FAILED) { map_ = nullptr; err = "Failed to mmapping the file."; close(); return false; }
        madvise(map_, map_len_, MADV_SEQUENTIAL);
        {
            size_t bl, tl;
            bgzf_ = threads > 1 && BgzfInflater::parse(static_cast<const uint8_t *>(map_), map_len_, bl, tl);
        }
        if (bgzf_) bgz_.init(static_cast<const uint8_t *>(map_), map_len_, threads);
        else if (!inf_.init(static_cast<const uint8_t *>(map_), map_len_, err)) { close(); return false; }
        seg_bytes_ = std::max<size_t>(seg_bytes, 64);
        made_ = 0; largest_ = 0;
        for (Buf &b : bufs_) free_.push_back(&b);
        quit_ = false; finished_ = false; failed_.clear();
        worker_ = std::thread([this]() { produce(); });
        return true;
    }

    // The next segment (blocks until it is inflated); what the call before handed out is released.  false: no more
    // segments -- err is empty at the end of the text and says why otherwise.
    bool next(Segment &s, std::string &err)
    {
        std::unique_lock<std::mutex> lk(mu_);
        if (held_) { free_.push_back(held_); held_ = nullptr; cv_.notify_all(); }
        cv_.wait(lk, [&]() { return !ready_.empty() || finished_; });
        if (ready_.empty()) { err = failed_; return false; }
        held_ = ready_.front();
        ready_.pop_front();
        s.data = held_->p; s.size = held_->n; s.last = held_->last;
        return true;
    }

    size_t segments_made() const { return made_; }
    size_t largest_segment() const { return largest_; }
    bool bgzf() const { return bgzf_; }
    size_t bgzf_blocks() const { return bgz_.blocks(); }

    void close()
    {
        {
            std::lock_guard<std::mutex> lk(mu_);
            quit_ = true;
        }
        cv_.notify_all();
        if (worker_.joinable()) worker_.join();
        for (Buf &b : bufs_) { std::free(b.p); b = Buf(); }
        free_.clear(); ready_.clear(); held_ = nullptr;
        if (map_) munmap(map_, map_len_);
        if (fd_ != -1) ::close(fd_);
        map_ = nullptr; fd_ = -1; map_len_ = 0;
    }

private:
    struct Buf {
        uint8_t *p = nullptr;
        size_t cap = 0, n = 0;
        bool last = false;
    };

    static bool reserve(Buf &b, size_t cap)
    {
        if (b.cap >= cap) return true;
        uint8_t *q = static_cast<uint8_t *>(std::realloc(b.p, cap));
        if (!q) return false;
        b.p = q; b.cap = cap;
        return true;
    }

    // nullptr: the consumer went away
    Buf *take_free()
    {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&]() { return !free_.empty() || quit_; });
        if (quit_) return nullptr;
        Buf *b = free_.front();
        free_.pop_front();
        return b;
    }

    void publish(Buf *b, size_t n, bool last)
    {
        b->n = n; b->last = last;
        made_++; largest_ = std::max(largest_, n);
        { std::lock_guard<std::mutex> lk(mu_); ready_.push_back(b); }
        cv_.notify_all();
    }

    void finish(const std::string &why)
    {
        { std::lock_guard<std::mutex> lk(mu_); failed_ = why; finished_ = true; }
        cv_.notify_all();
    }

    // the last record start of p[0, filled) that leaves whole records in front of it; `filled`: none
    static size_t cut_of(const uint8_t *p, size_t filled, bool fastq)
    {
        for (size_t w = (size_t)4 << 10;; w *= 8) {
            const size_t pos = filled > w ? filled - w : 1;
            size_t c = record_start_at_or_after(p, filled, pos, fastq);
            if (c > 0 && c < filled) {
                for (;;) {                          // the last one of the window
                    const size_t d = record_start_at_or_after(p, filled, c + 1, fastq);
                    if (d >= filled) return c;
                    c = d;
                }
            }
            if (pos == 1) return filled;
        }
    }

    void produce()
    {
        Buf *cur = take_free();
        if (!cur) return;
        size_t filled = 0;
        std::string err;
        bool known = false, cuttable = false, fastq = false;
        size_t target = seg_bytes_;            // bytes of text this segment is filled to before it is cut
        const size_t min_room = bgzf_ ? 65536 : 1;      // (a BGZF block's text goes in whole)
        for (;;) {
            target = std::max(target, filled + min_room);
            if (!reserve(*cur, target)) { finish("out of memory (gzip segment)"); return; }
            bool eof = false, full = false;
            while (!full && !eof) {
                size_t got = 0;
                if (bgzf_) { if (!bgz_.read(cur->p + filled, target - filled, got, eof, full, err)) { finish(err); return; } }
                else if (!inf_.read(cur->p + filled, target - filled, got, eof, err)) { finish(err); return; }
                filled += got;
                if (!bgzf_) full = filled >= target;
                { std::lock_guard<std::mutex> lk(mu_); if (quit_) return; }
            }
            if (!known && filled) { known = true; fastq = cur->p[0] == '@'; cuttable = fastq || cur->p[0] == '>'; }
            if (eof) {
                if (filled || made_ == 0) publish(cur, filled, true);
                finish(std::string());
                return;
            }
            const size_t cut = cuttable ? cut_of(cur->p, filled, fastq) : filled;
            if (cut >= filled) { target = filled + (filled >> 1) + 64; continue; }      // no record ends in here: the segment grows
            Buf *nxt = take_free();
            if (!nxt) return;
            const size_t carry = filled - cut;
            target = std::max(seg_bytes_, carry + (carry >> 1) + 64);
            if (!reserve(*nxt, target)) { finish("out of memory (gzip segment)"); return; }
            std::memcpy(nxt->p, cur->p + cut, carry);
            publish(cur, cut, false);
            cur = nxt;
            filled = carry;
        }
    }

    GzInflater inf_;
    BgzfInflater bgz_;
    bool bgzf_ = false;
    void *map_ = nullptr;
    size_t map_len_ = 0;
    int fd_ = -1;
    size_t seg_bytes_ = 0;
    Buf bufs_[3];
    std::mutex mu_;
    std::condition_variable cv_;
    std::deque<Buf *> free_, ready_;       // under mu_
    Buf *held_ = nullptr;                  // with the consumer
    bool quit_ = false, finished_ = false; // under mu_
    std::string failed_;                   // under mu_
    std::thread worker_;
    size_t made_ = 0, largest_ = 0;        // producer only (read after the end)
};

} // namespace host
