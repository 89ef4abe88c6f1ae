// mashmap_amd/host/seq_parse.hpp -- multi-threaded FASTA / FASTQ (optionally gzip / BGZF) ingest into flat batches.
//
// Same record semantics as seqiter::for_each_seq_in_file (src/common/seqiter.hpp:20-111, the non-htslib path):
//   * the first byte of the file decides the format ('>' FASTA, '@' FASTQ), anything else is fatal;
//   * name = header without its first character, cut at the first ' ' (:82);
//   * FASTA sequence = concatenation of the following lines up to the next line starting with '>';
//   * FASTQ: one sequence line, then two lines skipped (:104-107);
//   * records failing the keep_prefix / keep_seq filters are still reported, with an empty sequence (:84-97).
// What differs is the mechanics.  The reference pulls records one at a time through a 303-byte igzstream (src/common/gzstream.h:50)
// on one thread (~37 Mbp/s).  Here a file is taken in windows of raw bytes that end on a record boundary; a window
// is cut at record boundaries into one piece per thread; every thread finds its records (memchr over lines) and their sequence
// lengths, a prefix sum places them, and the threads copy the sequence bytes -- without the line breaks -- straight into the batch
// buffer (pinned host memory when the caller provides an allocator, so that the upload to the GPU is one DMA).  Plain files are
// mmap'ed; BGZF files (bgzip: independent <= 64 KiB deflate blocks, htslib's format -- the reference's data directory ships .gzi
// indexes of such files) are inflated block-parallel; any other gzip stream is inflated by one thread ahead of the parsers.
#pragma once
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sched.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <iostream>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "pack2bit.hpp"

namespace mmhost {

// CPUs this process may actually use at once: the smaller of the hardware threads, the affinity mask and the container's CPU quota
// (cgroup v2 cpu.max, v1 cpu.cfs_quota_us / cpu.cfs_period_us).  A container that shows 256 hardware threads and is given 16 CPUs'
// worth of time per 100 ms period *throttles* a process that runs more threads than that: every thread of it, the one that feeds the
// GPU included, stands still until the period ends (tens of milliseconds per burst -- profiles/r05b_*).  Stage widths are capped by
// this number.  MASHMAP_HIP_CPUS overrides.
inline unsigned availableCpus() {
  static const unsigned n = [] {
    if (const char* e = getenv("MASHMAP_HIP_CPUS")) { const int v = atoi(e); if (v > 0) return (unsigned)v; }
    unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) { const int c = CPU_COUNT(&set); if (c > 0) hw = std::min(hw, (unsigned)c); }
    double quota = -1, period = -1;
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
      char q[64] = {0};
      if (fscanf(f, "%63s %lf", q, &period) == 2 && strcmp(q, "max") != 0) quota = atof(q);
      fclose(f);
    } else {
      if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(g, "%lf", &quota) != 1) quota = -1; fclose(g); }
      if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(g, "%lf", &period) != 1) period = -1; fclose(g); }
    }
    if (quota > 0 && period > 0) hw = std::min(hw, (unsigned)std::max(1.0, quota / period + 0.5));
    return hw;
  }();
  return n;
}

struct ParsedBatch {
  std::vector<std::string> names;
  std::vector<int64_t> offs{0};      // record r owns bases[offs[r], offs[r+1])
  char* bases = nullptr;             // buffer from the reader's allocator
  size_t cap = 0;
  size_t size() const { return names.size(); }
  int64_t totalBases() const { return offs.back(); }
  // packed form (BatchReader with packOutput): the buffer holds the device layout of the batch instead of ASCII -- 2-bit codes, then
  // the N mask (pack2bit.hpp; every record starts on a 32-base boundary: record r at packed base packOffs[r]) -- 0.375 bytes per base
  bool packed = false;
  int64_t packedBases = 0;           // packOffs.back()
  std::vector<int64_t> packOffs;     // size() + 1 entries
  std::vector<int32_t> lens;         // bases of every record (offs differences)
  std::vector<uint8_t> hasN;         // record holds a base that is not A C G T a c g t
  int64_t maskBase = 0;              // the N mask words start behind maskBase / 4 bytes of code words (>= packedBases: the single-pass packer sizes the code area before it knows the records)
  // packed base behind the last base (rounded up to 32) of record r1 - 1: where a block of records [r0, r1) ends (the next record may start later: gaps)
  int64_t packEnd(size_t r0, size_t r1) const { return r1 > r0 ? packOffs[r1 - 1] + ((int64_t)lens[r1 - 1] + 31) / 32 * 32 : packOffs[r0]; }
  uint32_t* bases2() const { return (uint32_t*)bases; }
  uint32_t* nmask() const { return (uint32_t*)(bases + maskBase / 4); }
};

namespace detail {

}  // namespace detail

// Persistent workers for the host stages (a batch is parsed / post-processed in a handful of short parallel sections; starting
// a hundred std::threads for each of them costs milliseconds per batch).  run(n, fn) calls fn(0) .. fn(n - 1), each exactly once, on the
// workers and the calling thread, and returns when all have finished.  One run at a time per pool.
class WorkerPool {
 public:
  explicit WorkerPool(unsigned threads) { for (unsigned i = 1; i < threads; i++) th_.emplace_back([this] { loop(); }); }
  ~WorkerPool() {
    { std::lock_guard<std::mutex> lk(mu_); stop_ = true; gen_++; }
    cv_.notify_all();
    for (auto& t : th_) t.join();
  }
  WorkerPool(const WorkerPool&) = delete;
  WorkerPool& operator=(const WorkerPool&) = delete;
  unsigned size() const { return (unsigned)th_.size() + 1; }
  void run(unsigned n, const std::function<void(unsigned)>& fn) {
    if (n == 0) return;
    if (n == 1 || th_.empty()) { for (unsigned t = 0; t < n; t++) fn(t); return; }
    auto job = std::make_shared<Job>();
    job->fn = &fn; job->n = n; job->left.store(n);
    { std::lock_guard<std::mutex> lk(mu_); cur_ = job; gen_++; }
    cv_.notify_all();
    help(*job);
    std::unique_lock<std::mutex> lk(mu_);
    done_.wait(lk, [&] { return job->left.load() == 0; });
  }

 private:
  // every run has its own task counter: a worker that wakes up late holds the finished job and finds nothing left in it
  struct Job { const std::function<void(unsigned)>* fn = nullptr; unsigned n = 0; std::atomic<unsigned> next{0}, left{0}; };
  std::vector<std::thread> th_;
  std::mutex mu_; std::condition_variable cv_, done_;
  std::shared_ptr<Job> cur_; uint64_t gen_ = 0; bool stop_ = false;
  void help(Job& j) {
    for (;;) {
      const unsigned t = j.next.fetch_add(1);
      if (t >= j.n) return;
      (*j.fn)(t);
      if (j.left.fetch_sub(1) == 1) { std::lock_guard<std::mutex> lk(mu_); done_.notify_all(); }
    }
  }
  void loop() {
    uint64_t seen = 0;
    for (;;) {
      std::shared_ptr<Job> j;
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return gen_ != seen; });
        seen = gen_;
        if (stop_) return;
        j = cur_;
      }
      if (j) help(*j);
    }
  }
};

namespace detail {

inline void run_parallel(WorkerPool& pool, unsigned threads, const std::function<void(unsigned)>& fn) {
  if (threads <= 1) { fn(0); return; }
  pool.run(threads, fn);
}

// start of the next line at or after p (one past the next '\n'), or e
inline const char* next_line(const char* p, const char* e) {
  const char* nl = (const char*)memchr(p, '\n', (size_t)(e - p));
  return nl ? nl + 1 : e;
}
// is `p` (a line start) the first byte of a record?
inline bool record_starts_at(const char* p, const char* e, bool fasta) {
  if (p >= e) return false;
  if (fasta) return *p == '>';
  if (*p != '@') return false;
  const char* l2 = next_line(next_line(p, e), e);            // FASTQ: '@' header, and the line two below starts with '+' (a quality
  return l2 < e && *l2 == '+';                               // line that starts with '@' is followed, two lines on, by a sequence line)
}
// first record start at or after p (p need not be a line start), or e
inline const char* next_record(const char* base, const char* p, const char* e, bool fasta) {
  if (fasta) {                                               // a '>' right behind a line break (or at the very start): one memchr per '>' in the data, not per line
    for (const char* q = p; q < e; q++) {
      q = (const char*)memchr(q, '>', (size_t)(e - q));
      if (!q) return e;
      if (q == base || q[-1] == '\n') return q;
    }
    return e;
  }
  const char* q = (p == base || p[-1] == '\n') ? p : next_line(p, e);
  while (q < e && !record_starts_at(q, e, fasta)) q = next_line(q, e);
  return q;
}

// The record scanner both window parsers share (the one-pass packer and the two-pass form): seqiter's record semantics in one place.
// record_name: the name of the record whose header line starts at hdr and whose body starts at body -- the header without its marker,
// cut at the first ' ' only (a '\r' stays, like the reference's reader keeps it).
inline void record_name(const char* hdr, const char* body, const char*& nb, const char*& ne) {
  nb = hdr + 1;
  const char* he = body > hdr && body[-1] == '\n' ? body - 1 : body;   // header line without its '\n'
  if (he < nb) he = nb;
  const char* sp = (const char*)memchr(nb, ' ', (size_t)(he - nb));
  ne = sp ? sp : he;
}
// record_lines: hands every non-empty sequence line of the record whose body starts at body to onLine(line, bytes) and returns where the
// next record starts.  FASTA: lines up to the next '>' at a line start; FASTQ: one sequence line, then the '+' line and the quality line.
template <class F>
inline const char* record_lines(const char* body, const char* e, bool fasta, F&& onLine) {
  if (fasta) {
    const char* l = body;
    while (l < e && *l != '>') {
      const char* nl = (const char*)memchr(l, '\n', (size_t)(e - l));
      const char* le = nl ? nl : e;
      if (le > l) onLine(l, (size_t)(le - l));
      l = nl ? nl + 1 : e;
    }
    return l;
  }
  const char* nl = (const char*)memchr(body, '\n', (size_t)(e - body));
  const char* le = nl ? nl : e;
  if (body < e && le > body) onLine(body, (size_t)(le - body));
  return next_line(next_line(nl ? nl + 1 : e, e), e);                                   // '+' line and quality line skipped
}

struct Rec { const char* hdr; const char* body; const char* end; int64_t seqLen; bool keep; size_t seg0, seg1; };   // seg0..seg1: its sequence lines (Seg list of its thread)
struct Seg { const char* p; size_t n; };   // one line of sequence bytes (without the line break)

// source of raw (decompressed) bytes, window by window; every window ends on a record boundary
class RawSource {
 public:
  virtual ~RawSource() {}
  virtual bool next(const char*& p, size_t& n, bool fasta, bool first) = 0;   // false: exhausted
  virtual char first_byte() = 0;
  virtual bool fileMapped() const { return false; }                           // windows point into a file mapping (pages may not be mapped yet)
};

}  // namespace detail

// Mappings of plain (uncompressed, regular) input files made AHEAD of their use: skch::Sketch starts prefault() on the query files while
// the reference index is being built, so that when skch::Map's reader reaches a file its pages are already in the page tables -- the
// parser's workers then run at memory speed instead of taking a page fault per 64 KB (a third of their time on a 10 GB FASTA).  A file
// the cache does not hold is mapped by the reader itself, as before.
class MappedFileCache {
 public:
  struct Entry { int fd = -1; const char* data = nullptr; size_t size = 0; };
  static MappedFileCache& instance() { static MappedFileCache c; return c; }
  ~MappedFileCache() { wait(); for (auto& kv : map_) { if (kv.second.data) munmap((void*)kv.second.data, kv.second.size); if (kv.second.fd >= 0) close(kv.second.fd); } }
  // maps the files and starts `threads` background threads that populate the mappings front to back
  void prefault(const std::vector<std::string>& paths, unsigned threads) {
    std::lock_guard<std::mutex> lk(mu_);
    for (const auto& path : paths) {
      if (map_.count(path)) continue;
      struct stat st;
      if (stat(path.c_str(), &st) != 0 || !S_ISREG(st.st_mode) || st.st_size == 0) continue;
      Entry e; e.fd = open(path.c_str(), O_RDONLY);
      if (e.fd < 0) continue;
      unsigned char m[2] = {0, 0};
      if (pread(e.fd, m, 2, 0) == 2 && m[0] == 31 && m[1] == 139) { close(e.fd); continue; }      // gzip: read through zlib, not mapped
      e.size = (size_t)st.st_size;
      void* p = mmap(nullptr, e.size, PROT_READ, MAP_PRIVATE, e.fd, 0);
      if (p == MAP_FAILED) { close(e.fd); continue; }
      e.data = (const char*)p;
      map_[path] = e;
      const size_t piece = (size_t)64 << 20;
      for (size_t off = 0; off < e.size; off += piece) jobs_.push_back(Job{e.data + off, std::min(piece, e.size - off)});
    }
    if (workers_.empty() && !jobs_.empty())
      for (unsigned t = 0; t < std::max(1u, threads); t++) workers_.emplace_back([this] { run(); });
  }
  bool lookup(const std::string& path, Entry& out) { std::lock_guard<std::mutex> lk(mu_); auto it = map_.find(path); if (it == map_.end()) return false; out = it->second; return true; }
  bool allPopulated() { std::lock_guard<std::mutex> lk(mu_); return next_ >= jobs_.size() && running_ == 0; }   // every mapping made so far is in the page tables
  void wait() { std::vector<std::thread> w; { std::lock_guard<std::mutex> lk(mu_); w.swap(workers_); } for (auto& t : w) t.join(); }

 private:
  struct Job { const char* p; size_t n; };
  std::mutex mu_; std::unordered_map<std::string, Entry> map_; std::vector<Job> jobs_; size_t next_ = 0; std::vector<std::thread> workers_; int running_ = 0;
  void run() {
    bool busy = false;
    for (;;) {
      Job j;
      { std::lock_guard<std::mutex> lk(mu_); if (busy) { running_--; busy = false; } if (next_ >= jobs_.size()) return; j = jobs_[next_++]; running_++; busy = true; }
      const uintptr_t a = (uintptr_t)j.p & ~(uintptr_t)4095, z = ((uintptr_t)j.p + j.n + 4095) & ~(uintptr_t)4095;
      if (madvise((void*)a, (size_t)(z - a), 22 /* MADV_POPULATE_READ, Linux >= 5.14 */) != 0) {
        volatile char sink = 0;                                      // older kernel: touch every page
        for (size_t o = 0; o < j.n; o += 4096) sink = (char)(sink + j.p[o]);
        (void)sink;
      }
    }
  }
};

namespace detail {

class MmapSource : public RawSource {
  int fd_ = -1; const char* data_ = nullptr; size_t size_ = 0, pos_ = 0, window_; bool owned_ = true;
 public:
  MmapSource(const std::string& path, size_t window) : window_(window) {
    MappedFileCache::Entry ce;
    if (MappedFileCache::instance().lookup(path, ce)) { fd_ = ce.fd; data_ = ce.data; size_ = ce.size; owned_ = false; return; }   // mapped (and being populated) ahead
    fd_ = open(path.c_str(), O_RDONLY);
    if (fd_ < 0) return;
    struct stat st; if (fstat(fd_, &st) != 0) return;
    size_ = (size_t)st.st_size;
    if (size_) { void* m = mmap(nullptr, size_, PROT_READ, MAP_PRIVATE, fd_, 0); if (m == MAP_FAILED) { size_ = 0; return; } data_ = (const char*)m; madvise((void*)data_, size_, MADV_SEQUENTIAL); }
  }
  ~MmapSource() override { if (!owned_) return; if (data_) munmap((void*)data_, size_); if (fd_ >= 0) close(fd_); }
  bool prefaulted() const { return !owned_; }
  bool ok() const { return fd_ >= 0; }
  bool fileMapped() const override { return owned_ || !MappedFileCache::instance().allPopulated(); }   // a mapping from the cache whose pages are all in: the parser's workers need not ask again
  char first_byte() override { return size_ ? data_[0] : 0; }
  bool next(const char*& p, size_t& n, bool fasta, bool) override {
    if (pos_ >= size_) return false;
    const char* b = data_ + pos_; const char* e = data_ + size_;
    const char* cut = e;
    if ((size_t)(e - b) > window_) {
      // a record is never split between windows.  FASTA: the window ends at the last record start it holds (so that it fits the buffers
      // sized for one window: with 125 Mbp contigs the next boundary may be a quarter of a window away), and grows to the next one only
      // when a single record is longer than the window; FASTQ records are short: the next boundary
      cut = nullptr;
      if (fasta) {
        const char* q = b + window_;
        while (q > b + 1) {
          const char* g = (const char*)memrchr(b + 1, '>', (size_t)(q - (b + 1)));
          if (!g) break;
          if (g[-1] == '\n') { cut = g; break; }
          q = g;
        }
      }
      if (!cut) cut = next_record(data_, b + window_, e, fasta);
    }
    p = b; n = (size_t)(cut - b); pos_ += n;
    return true;
  }
};

// gzip / BGZF: inflated into an internal buffer; the tail behind the last record boundary is carried into the next window
class GzSource : public RawSource {
  std::string path_; size_t window_; unsigned threads_; WorkerPool* pool_;
  FILE* raw_ = nullptr; bool bgzf_ = false; gzFile gz_ = nullptr;
  std::vector<char> buf_; size_t carry_ = 0; bool eof_ = false;
  std::vector<unsigned char> comp_;

  bool fill_stream(size_t want) {
    while (buf_.size() < want && !eof_) {
      const size_t at = buf_.size(); buf_.resize(at + (8u << 20));
      const int got = gzread(gz_, buf_.data() + at, 8u << 20);
      buf_.resize(at + (got > 0 ? (size_t)got : 0));
      if (got <= 0) {
        // gzread() <= 0 is the end of the data only if zlib says so: a truncated or corrupt stream must not pass for a short file
        int zerr = Z_OK; const char* msg = gzerror(gz_, &zerr);
        if (got < 0 || (zerr != Z_OK && zerr != Z_STREAM_END)) {
          std::cerr << "[mashmap_hip] error while inflating " << path_ << ": " << (msg && *msg ? msg : "gzread failed") << std::endl; exit(1);
        }
        eof_ = true;
      }
    }
    return !buf_.empty();
  }
  // BGZF (SAM spec 4.1): gzip member with extra subfield 'B','C' holding BSIZE = block size - 1; payload = raw deflate; trailer CRC32 + ISIZE
  bool fill_bgzf(size_t want) {
    struct Blk { size_t cOff, cLen, uOff, uLen; uint32_t crc; };
    while (buf_.size() < want && !eof_) {
      std::vector<Blk> blks; comp_.clear();
      size_t uTot = 0;
      while (uTot < (64u << 20)) {
        unsigned char h[18];
        if (fread(h, 1, 18, raw_) != 18) { eof_ = true; break; }
        if (h[0] != 31 || h[1] != 139 || !(h[3] & 4) || h[12] != 'B' || h[13] != 'C') { std::cerr << "[mashmap_hip] malformed BGZF block in " << path_ << std::endl; exit(1); }
        const size_t bsize = (size_t)h[16] + ((size_t)h[17] << 8) + 1;
        const size_t at = comp_.size(); comp_.resize(at + bsize - 18);
        if (fread(comp_.data() + at, 1, bsize - 18, raw_) != bsize - 18) { std::cerr << "[mashmap_hip] truncated BGZF file " << path_ << std::endl; exit(1); }
        const unsigned char* t = comp_.data() + at + bsize - 18 - 4;
        const size_t isize = (size_t)t[0] | ((size_t)t[1] << 8) | ((size_t)t[2] << 16) | ((size_t)t[3] << 24);
        const uint32_t crc = (uint32_t)t[-4] | ((uint32_t)t[-3] << 8) | ((uint32_t)t[-2] << 16) | ((uint32_t)t[-1] << 24);
        blks.push_back(Blk{at, bsize - 18 - 8, uTot, isize, crc});
        uTot += isize;
      }
      const size_t base = buf_.size(); buf_.resize(base + uTot);
      std::atomic<size_t> nextB(0); std::atomic<int> bad(0);
      run_parallel(*pool_, std::max(1u, std::min<unsigned>(threads_, (unsigned)blks.size())), [&](unsigned) {
        z_stream zs;
        for (size_t i = nextB.fetch_add(1); i < blks.size(); i = nextB.fetch_add(1)) {
          if (!blks[i].uLen) continue;
          std::memset(&zs, 0, sizeof zs);
          if (inflateInit2(&zs, -15) != Z_OK) { bad = 1; return; }
          zs.next_in = comp_.data() + blks[i].cOff; zs.avail_in = (uInt)blks[i].cLen;
          zs.next_out = (Bytef*)(buf_.data() + base + blks[i].uOff); zs.avail_out = (uInt)blks[i].uLen;
          const int rc = inflate(&zs, Z_FINISH);
          inflateEnd(&zs);
          if (rc != Z_STREAM_END || zs.total_out != blks[i].uLen) bad = 1;
          else if ((uint32_t)crc32(crc32(0L, Z_NULL, 0), (const Bytef*)(buf_.data() + base + blks[i].uOff), (uInt)blks[i].uLen) != blks[i].crc) bad = 1;   // trailer: CRC32 of the payload
        }
      });
      if (bad) { std::cerr << "[mashmap_hip] corrupt BGZF block in " << path_ << std::endl; exit(1); }
    }
    return !buf_.empty();
  }

 public:
  // streamOnly: the path is not a regular file (FIFO, /dev/stdin, process substitution): it can be opened and read exactly once, so
  // there is no sniffing with a handle of its own -- gzopen reads plain data transparently and gzip (BGZF included: concatenated
  // members) by inflating it, which is what the reference's igzstream does with such inputs (src/common/gzstream.h:50)
  GzSource(const std::string& path, size_t window, unsigned threads, WorkerPool* pool, bool streamOnly = false) : path_(path), window_(window), threads_(threads), pool_(pool) {
    if (streamOnly) { gz_ = gzopen(path.c_str(), "rb"); if (gz_) gzbuffer(gz_, 1u << 20); return; }
    raw_ = fopen(path.c_str(), "rb");
    if (!raw_) return;
    unsigned char h[18] = {0};
    const size_t got = fread(h, 1, 18, raw_);
    bgzf_ = got == 18 && h[0] == 31 && h[1] == 139 && (h[3] & 4) && h[10] == 6 && h[11] == 0 && h[12] == 'B' && h[13] == 'C';
    if (bgzf_) fseek(raw_, 0, SEEK_SET);
    else { fclose(raw_); raw_ = nullptr; gz_ = gzopen(path.c_str(), "rb"); if (gz_) gzbuffer(gz_, 1u << 20); }
  }
  ~GzSource() override { if (raw_) fclose(raw_); if (gz_) gzclose(gz_); }
  bool ok() const { return raw_ || gz_; }
  char first_byte() override { if (buf_.empty()) { if (bgzf_) fill_bgzf(1); else fill_stream(1); } return buf_.empty() ? 0 : buf_[0]; }
  bool next(const char*& p, size_t& n, bool fasta, bool) override {
    if (carry_) { buf_.erase(buf_.begin(), buf_.begin() + (std::ptrdiff_t)carry_); carry_ = 0; }
    size_t want = window_;
    while (true) {
      if (bgzf_) fill_bgzf(want); else fill_stream(want);
      if (buf_.empty()) return false;
      const char* b = buf_.data(); const char* e = b + buf_.size();
      if (eof_) { p = b; n = buf_.size(); carry_ = n; return true; }
      // last record boundary inside the buffer (not at its start): everything before it is complete
      const char* cut = nullptr; const char* q = e;
      while (q > b + 1) {
        const char* nl = (const char*)memrchr(b, '\n', (size_t)(q - 1 - b));
        if (!nl) break;
        if (nl + 1 < e && record_starts_at(nl + 1, e, fasta) && (fasta || next_line(next_line(nl + 1, e), e) < e)) { cut = nl + 1; break; }
        q = nl + 1;
      }
      if (cut && cut > b) { p = b; n = (size_t)(cut - b); carry_ = n; return true; }
      want = buf_.size() + window_;                              // one record longer than the buffer: read on
    }
  }
};

}  // namespace detail

// Batches of records from a list of files.  Not thread-safe; next() itself runs `threads` workers.
class BatchReader {
 public:
  typedef std::function<char*(size_t)> Alloc;
  typedef std::function<void(char*)> Free;

  BatchReader(std::vector<std::string> files, size_t windowBytes, unsigned threads, std::unordered_set<std::string> keepSeq = {},
              std::string keepPrefix = "", Alloc a = nullptr, Free f = nullptr, bool packOutput = false)
      : files_(std::move(files)), window_(std::max<size_t>(windowBytes, 1u << 16)), threads_(std::max(1u, threads)), keepSeq_(std::move(keepSeq)),
        keepPrefix_(std::move(keepPrefix)), pool_(std::max(1u, threads)), alloc_(a ? a : [](size_t n) { return (char*)malloc(n); }), free_(f ? f : [](char* p) { free(p); }),
        packOutput_(packOutput) {}
  ~BatchReader() { delete src_; }

  // index of the file the batch returned last came from, and whether it was that file's last batch
  size_t fileIndex() const { return curFile_; }
  bool fileDone() const { return fileDone_; }
  size_t splitWindows() const { return splitWindows_; }   // windows that went through parseWindowPackedSplit (records cut across threads)
  void release(ParsedBatch& b) { if (b.bases) free_(b.bases); b.bases = nullptr; b.cap = 0; }

  bool next(ParsedBatch& out) {
    out.names.clear(); out.offs.assign(1, 0);
    while (true) {
      if (!src_) {
        if (nextFile_ >= files_.size()) return false;
        openFile(files_[nextFile_]); curFile_ = nextFile_++; firstWindow_ = true;
      }
      const char* p = nullptr; size_t n = 0;
      if (!src_->next(p, n, fasta_, firstWindow_)) { delete src_; src_ = nullptr; continue; }
      firstWindow_ = false;
      parseWindow(p, n, out);
      // peek: is this file exhausted?  (cheap for mmap; for gz the next call finds out)
      fileDone_ = false;
      return true;
    }
  }

 private:
  std::vector<std::string> files_; size_t window_; unsigned threads_;
  std::unordered_set<std::string> keepSeq_; std::string keepPrefix_;
  WorkerPool pool_;
  Alloc alloc_; Free free_;
  size_t splitWindows_ = 0;
  bool packOutput_ = false;          // batches carry 2-bit codes + N mask (what mm_reads_upload_packed takes) instead of ASCII
  detail::RawSource* src_ = nullptr; size_t nextFile_ = 0, curFile_ = 0; bool fasta_ = true, firstWindow_ = true, fileDone_ = false;

  void openFile(const std::string& path) {
    // mmap and the two-byte gzip sniff need a regular file; anything else (a FIFO, /dev/stdin, `-q <(zcat x.gz)`) is read once, as a stream
    struct stat st;
    const bool regular = stat(path.c_str(), &st) == 0 && S_ISREG(st.st_mode);
    bool gz = false;
    if (regular) { FILE* f = fopen(path.c_str(), "rb"); if (f) { unsigned char m[2] = {0, 0}; if (fread(m, 1, 2, f) == 2) gz = m[0] == 31 && m[1] == 139; fclose(f); } }
    bool ok = false;
    if (!regular) { auto* s = new detail::GzSource(path, window_, threads_, &pool_, true); ok = s->ok(); src_ = s; }
    else if (gz) { auto* s = new detail::GzSource(path, window_, threads_, &pool_); ok = s->ok(); src_ = s; }
    else { auto* s = new detail::MmapSource(path, window_); ok = s->ok(); src_ = s; }
    const char c = ok ? src_->first_byte() : 0;
    if (!ok || (c != '>' && c != '@')) {
      std::cerr << "[mashmap_hip::for_each_seq_in_file] unknown file format given to the sequence reader: " << path << std::endl;
      exit(1);
    }
    fasta_ = c == '>';
  }

  // the sequence bytes of the window's records as 2-bit codes + N mask: the workers normalise and pack while they drop the line breaks
  // (the bytes are touched once either way; the packed batch is 3/8 the size of the ASCII one on its way to the GPU)
  void packWindow(const std::vector<std::vector<detail::Rec>>& recs, const std::vector<std::vector<detail::Seg>>& segs, const std::vector<size_t>& first, unsigned T,
                  size_t nRec, ParsedBatch& out) {
    out.packOffs.assign(nRec + 1, 0); out.lens.assign(nRec, 0); out.hasN.assign(nRec, 0);
    int64_t pk = 0;
    for (size_t r = 0; r < nRec; r++) { const int64_t len = out.offs[r + 1] - out.offs[r]; out.lens[r] = (int32_t)len; out.packOffs[r] = pk; pk += (len + 31) / 32 * 32; }
    out.packOffs[nRec] = pk; out.packedBases = pk; out.maskBase = pk;
    const size_t need = (size_t)pk / 4 + (size_t)pk / 8 + 64;
    if (need > out.cap) {
      if (out.bases) free_(out.bases);
      out.bases = alloc_(need + (need >> 4) + 4096);
      if (!out.bases) { std::cerr << "[mashmap_hip] out of host memory for a batch of " << pk << " packed bases" << std::endl; exit(1); }
      out.cap = need + (need >> 4) + 4096;
    }
    uint32_t* b2 = out.bases2(); uint32_t* nm = out.nmask();
    detail::run_parallel(pool_, T, [&](unsigned t) {
      for (size_t i = 0; i < recs[t].size(); i++) {
        const detail::Rec& r = recs[t][i];
        const size_t ri = first[t] + i;
        if (!r.keep || !r.seqLen) continue;
        Pack2bitStream st(b2 + out.packOffs[ri] / 16, nm + out.packOffs[ri] / 32);
        for (size_t g = r.seg0; g < r.seg1; g++) st.feed(segs[t][g].p, segs[t][g].n);
        out.hasN[ri] = st.finish() ? 1 : 0;
      }
    });
  }

  // FASTA windows whose records are longer than a thread's piece (assembly contigs, chromosomes: a 512 MB window of 125 Mbp records
  // holds four of them): the pieces are cut at LINE boundaries instead, anywhere inside a record.  Pass A, every thread over its piece:
  // the records that start there and how many sequence bytes each (partial) record has in the piece -- a line scan, no packing.  A prefix
  // sum over the parts gives every part its place inside its record and every record its 32-aligned place in the batch.  Pass B, every
  // thread over its piece again: packs its parts where they belong.  The packed words hold 32 bases each, so a part that does not start
  // on a 32-base boundary of its record leaves its first few bases to the thread of the part before it, which reads on past its piece's
  // end until its last word is full (or the record ends).  Same ParsedBatch as the one-pass form, without gaps between records.
  struct SplitPart { const char* hdr; const char* body; const char* end; int64_t bases; size_t rec; int64_t off; uint8_t hasN; };
  bool parseWindowPackedSplit(const char* p, size_t n, unsigned T, ParsedBatch& out) {
    const char* b = p; const char* e = p + n;
    std::vector<const char*> cut(T + 1, e);
    cut[0] = b;
    for (unsigned t = 1; t < T; t++) { const char* q = b + n / T * t; cut[t] = q[-1] == '\n' ? q : detail::next_line(q, e); }
    for (unsigned t = 1; t <= T; t++) if (cut[t] < cut[t - 1]) cut[t] = cut[t - 1];
    const bool mapped = src_ && src_->fileMapped();
    const bool trace = getenv("MASHMAP_HIP_PARSE_TRACE") != nullptr;
    const auto tA = std::chrono::steady_clock::now();
    std::vector<std::vector<SplitPart>> parts(T);
    detail::run_parallel(pool_, T, [&](unsigned t) {
      const char* q = cut[t]; const char* pe = cut[t + 1];
      if (mapped && pe > q) {
        const uintptr_t a = (uintptr_t)q & ~(uintptr_t)4095, z = ((uintptr_t)pe + 4095) & ~(uintptr_t)4095;
        (void)madvise((void*)a, (size_t)(z - a), 22 /* MADV_POPULATE_READ */);
      }
      auto& P = parts[t];
      if (q < pe && *q != '>') P.push_back(SplitPart{nullptr, q, pe, 0, 0, 0, 0});          // the piece starts inside a record
      while (q < pe) {
        if (*q == '>') {                                       // q is a line start: a header line
          if (!P.empty()) P.back().end = q;
          const char* nl = (const char*)memchr(q, '\n', (size_t)(pe - q));
          const char* nx = nl ? nl + 1 : pe;
          P.push_back(SplitPart{q, nx, pe, 0, 0, 0, 0});
          q = nx;
          continue;
        }
        // sequence lines up to the next header line ('>' right behind a line break) or the piece's end: bytes minus line breaks
        size_t nls = 0;
        const char* h = scan_to_header(q, pe, &nls);           // one sweep: line breaks counted on the way to the next header line
        P.back().bases += (int64_t)(h - q) - (int64_t)nls;
        q = h;
      }
    });
    const auto tB = std::chrono::steady_clock::now();
    // parts -> records
    size_t nRec = 0;
    for (unsigned t = 0; t < T; t++) for (auto& pt : parts[t]) if (pt.hdr) nRec++;
    out.names.resize(nRec); out.offs.assign(nRec + 1, 0); out.packOffs.assign(nRec + 1, 0); out.lens.assign(nRec, 0); out.hasN.assign(nRec, 0);
    std::vector<int64_t> recLen(nRec, 0);
    std::vector<uint8_t> keep(nRec, 1);
    size_t r = 0; bool any = false;
    for (unsigned t = 0; t < T; t++) for (auto& pt : parts[t]) {
      if (pt.hdr) {
        if (any) r++;
        any = true;
        const char* nb; const char* ne;
        detail::record_name(pt.hdr, pt.body, nb, ne);
        out.names[r].assign(nb, ne);
        keep[r] = (keepPrefix_.empty() || out.names[r].compare(0, keepPrefix_.size(), keepPrefix_) == 0) && (keepSeq_.empty() || keepSeq_.count(out.names[r]));
      } else if (!any) return false;                         // a window always starts with a header line
      pt.rec = r; pt.off = recLen[r]; recLen[r] += pt.bases;
    }
    int64_t at = 0, pk = 0;
    for (size_t i = 0; i < nRec; i++) {
      if (!keep[i]) recLen[i] = 0;
      if (recLen[i] > 0x7fffffff) return false;              // refused further up with the reference's message (the two-pass form keeps 64-bit lengths)
      out.offs[i] = at; out.lens[i] = (int32_t)recLen[i]; out.packOffs[i] = pk;
      at += recLen[i]; pk += (recLen[i] + 31) / 32 * 32;
    }
    out.offs[nRec] = at; out.packOffs[nRec] = pk; out.packedBases = pk; out.maskBase = pk;
    const size_t need = (size_t)pk / 4 + (size_t)pk / 8 + 64;
    if (need > out.cap) {
      if (out.bases) free_(out.bases);
      out.bases = alloc_(need + (need >> 4) + 4096);
      if (!out.bases) { std::cerr << "[mashmap_hip] out of host memory for a batch of " << pk << " packed bases" << std::endl; exit(1); }
      out.cap = need + (need >> 4) + 4096;
    }
    uint32_t* b2 = out.bases2(); uint32_t* nm = out.nmask();
    const auto tC = std::chrono::steady_clock::now();
    detail::run_parallel(pool_, T, [&](unsigned t) {
      char tmp[16384 + 64];
      for (auto& pt : parts[t]) {
        if (!keep[pt.rec] || !pt.bases) continue;
        const int64_t D = out.packOffs[pt.rec] + pt.off;
        int64_t skip = (32 - D % 32) % 32;                    // these bases complete the previous part's last word: its thread packs them
        if (pt.bases <= skip) continue;
        Pack2bitStream st(b2 + (D + skip) / 16, nm + (D + skip) / 32);
        const char* q = pt.body;
        while (skip && q < pt.end) { if (*q != '\n') skip--; q++; }
        while (q < pt.end) {                                  // line breaks out, 16 KB at a time (stays in L1), then packed
          const size_t m = std::min<size_t>((size_t)(pt.end - q), sizeof(tmp) - 64);
          const size_t k = strip_newlines(q, m, tmp);
          if (k) st.feed(tmp, k);
          q += m;
        }
        int64_t want = (32 - (D + pt.bases) % 32) % 32;       // fill the last word from the lines behind the piece, if the record goes on
        q = pt.end;
        while (want && q < e && *q != '>') {
          const char* nl = (const char*)memchr(q, '\n', (size_t)(e - q));
          const char* le = nl ? nl : e;
          const size_t m = std::min<size_t>((size_t)(le - q), (size_t)want);
          if (m) { st.feed(q, m); want -= (int64_t)m; }
          q = nl ? nl + 1 : e;
        }
        pt.hasN = st.finish() ? 1 : 0;
      }
    });
    for (unsigned t = 0; t < T; t++) for (auto& pt : parts[t]) if (pt.hasN) out.hasN[pt.rec] = 1;
    out.packed = true;
    splitWindows_++;
    if (trace) {
      auto sec = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
      fprintf(stderr, "[mashmap_hip::parse] split window: %zu bytes, %zu records, %u threads: scan %.4f s, place + buffer %.4f s, pack %.4f s\n", n, nRec, T, sec(tA, tB), sec(tB, tC),
              sec(tC, std::chrono::steady_clock::now()));
    }
    return true;
  }

  // Packed batches in ONE pass over the window (the parser is bound by memory traffic: the two-pass form reads every byte twice).  Every
  // thread takes its piece of the window and, record by record, finds the lines and packs them straight into the batch buffer, into a
  // region of its own that starts where the piece's first byte would land plus some slack per thread -- a record's packed length is at
  // most 28 bases more than its bytes in the file, so a region overflows only on pathological input (thousands of near-empty records),
  // and then the window is redone by the two-pass path.  The regions leave gaps between them: mm_reads_upload_packed takes the
  // reads' start positions (packOffs) as they are.
  bool parseWindowPackedOnePass(const char* p, size_t n, ParsedBatch& out) {
    const char* b = p; const char* e = p + n;
    const unsigned T = (unsigned)std::min<size_t>(threads_, std::max<size_t>(1, n >> 16));
    std::vector<const char*> cut(T + 1, e);
    cut[0] = b;
    const bool fasta = fasta_;
    // a thread looks for the first record of its piece inside the piece only: when some piece holds no record start (records longer than
    // a piece: contigs, chromosomes) the window goes through the split form instead, which cuts inside records
    std::atomic<int> bare(0);
    detail::run_parallel(pool_, T, [&](unsigned t) {
      if (!t) return;
      const char* lim = t + 1 < T ? b + n / T * (t + 1) : e;
      cut[t] = detail::next_record(b, b + n / T * t, lim, fasta);
      if (cut[t] >= lim) { if (fasta && lim < e) bare = 1; if (lim < e) cut[t] = detail::next_record(b, lim, e, fasta); }
    });
    if (bare && fasta && !getenv("MASHMAP_HIP_NO_SPLIT_RECORDS")) {
      if (parseWindowPackedSplit(p, n, T, out)) return true;
      out.names.clear(); out.offs.assign(1, 0);               // (a record beyond int32: the two-pass form keeps 64-bit lengths for the caller's message)
      return false;
    }
    for (unsigned t = 1; t <= T; t++) if (cut[t] < cut[t - 1]) cut[t] = cut[t - 1];
    const int64_t slack = 1 << 16;
    std::vector<int64_t> G(T + 1);
    for (unsigned t = 0; t <= T; t++) G[t] = (((int64_t)(cut[t] - b) + (int64_t)t * slack) + 31) / 32 * 32;
    const size_t need = (size_t)G[T] / 4 + (size_t)G[T] / 8 + 64;
    if (need > out.cap) {
      if (out.bases) free_(out.bases);
      out.bases = alloc_(need + (need >> 4) + 4096);
      if (!out.bases) { std::cerr << "[mashmap_hip] out of host memory for a batch of " << G[T] << " packed bases" << std::endl; exit(1); }
      out.cap = need + (need >> 4) + 4096;
    }
    out.maskBase = G[T];
    uint32_t* b2 = out.bases2(); uint32_t* nm = out.nmask();
    struct Lite { std::string name; int32_t len; uint8_t hasN; int64_t start; };
    std::vector<std::vector<Lite>> recs(T);
    std::atomic<int> overflow(0);
    const bool mapped = src_ && src_->fileMapped();
    detail::run_parallel(pool_, T, [&](unsigned t) {
      const char* q = cut[t]; const char* pe = cut[t + 1];
      if (mapped && pe > q) {
        const uintptr_t a = (uintptr_t)q & ~(uintptr_t)4095, z = ((uintptr_t)pe + 4095) & ~(uintptr_t)4095;
        (void)madvise((void*)a, (size_t)(z - a), 22 /* MADV_POPULATE_READ */);
      }
      auto& R = recs[t];
      R.reserve((size_t)(pe - q) / 4096 + 16);
      int64_t cur = G[t];
      while (q < pe) {
        const char* body = detail::next_line(q, e);
        const char* nb; const char* ne;
        detail::record_name(q, body, nb, ne);
        Lite r; r.name.assign(nb, ne); r.start = cur; r.len = 0; r.hasN = 0;
        const bool keep = (keepPrefix_.empty() || r.name.compare(0, keepPrefix_.size(), keepPrefix_) == 0) && (keepSeq_.empty() || keepSeq_.count(r.name));
        // room for the longest record this piece could still hold is not known in advance: stop at the region's end, line by line
        Pack2bitStream st(b2 + cur / 16, nm + cur / 32);
        int64_t len = 0; bool fits = true;
        const char* end = detail::record_lines(body, e, fasta, [&](const char* l, size_t m) {
          if (!keep || !fits) return;
          if (cur + (len + (int64_t)m + 31) / 32 * 32 > G[t + 1]) { fits = false; return; }
          st.feed(l, m); len += (int64_t)m;
        });
        if (!fits || len > 0x7fffffff) { overflow = 1; return; }
        r.hasN = st.finish() ? 1 : 0; r.len = (int32_t)len;
        cur += (len + 31) / 32 * 32;
        R.push_back(std::move(r));
        q = end;
      }
    });
    if (overflow) return false;
    size_t nRec = 0;
    std::vector<size_t> first(T + 1, 0);
    for (unsigned t = 0; t < T; t++) { first[t] = nRec; nRec += recs[t].size(); }
    out.names.resize(nRec); out.offs.assign(nRec + 1, 0); out.packOffs.assign(nRec + 1, 0); out.lens.assign(nRec, 0); out.hasN.assign(nRec, 0);
    detail::run_parallel(pool_, T, [&](unsigned t) {
      for (size_t i = 0; i < recs[t].size(); i++) {
        auto& r = recs[t][i]; const size_t ri = first[t] + i;
        out.names[ri].swap(r.name); out.lens[ri] = r.len; out.hasN[ri] = r.hasN; out.packOffs[ri] = r.start;
      }
    });
    int64_t at = 0;
    for (size_t r = 0; r < nRec; r++) { out.offs[r] = at; at += out.lens[r]; }
    out.offs[nRec] = at;
    const int64_t endAt = nRec ? out.packOffs[nRec - 1] + ((int64_t)out.lens[nRec - 1] + 31) / 32 * 32 : 0;
    out.packOffs[nRec] = endAt; out.packedBases = endAt;
    out.packed = true;
    return true;
  }

  void parseWindow(const char* p, size_t n, ParsedBatch& out) {
    using detail::Rec;
    if (packOutput_ && out.names.empty() && !getenv("MASHMAP_HIP_TWO_PASS_PACK") && parseWindowPackedOnePass(p, n, out)) return;
    const char* b = p; const char* e = p + n;
    const unsigned T = (unsigned)std::min<size_t>(threads_, std::max<size_t>(1, n >> 16));
    // piece t = records that start in [cut[t], cut[t+1])
    std::vector<const char*> cut(T + 1, e);
    cut[0] = b;
    const bool fasta = fasta_;
    detail::run_parallel(pool_, T, [&](unsigned t) { if (t) cut[t] = detail::next_record(b, b + n / T * t, e, fasta); });   // may scan a long record: in parallel
    for (unsigned t = 1; t <= T; t++) if (cut[t] < cut[t - 1]) cut[t] = cut[t - 1];
    std::vector<std::vector<Rec>> recs(T);
    std::vector<std::vector<detail::Seg>> segs(T);        // the sequence lines found by the scan below: the copy / pack pass does not search again
    const bool mapped = src_ && src_->fileMapped();
    detail::run_parallel(pool_, T, [&](unsigned t) {
      const char* q = cut[t]; const char* pe = cut[t + 1];
      if (mapped && pe > q) {
        // page-cache pages of the file are mapped one fault (16 pages) at a time otherwise: one call maps this thread's whole piece
        // (MADV_POPULATE_READ, Linux >= 5.14; an older kernel refuses it and the faults happen as before)
        const uintptr_t a = (uintptr_t)q & ~(uintptr_t)4095, z = ((uintptr_t)pe + 4095) & ~(uintptr_t)4095;
        (void)madvise((void*)a, (size_t)(z - a), 22 /* MADV_POPULATE_READ */);
      }
      auto& R = recs[t]; auto& S = segs[t];
      while (q < pe) {
        Rec r; r.hdr = q; r.keep = true; r.seg0 = S.size();
        const char* body = detail::next_line(q, e);
        r.body = body;
        int64_t len = 0;
        r.end = detail::record_lines(body, e, fasta, [&](const char* l, size_t m) { S.push_back(detail::Seg{l, m}); len += (int64_t)m; });
        r.seqLen = len;
        r.seg1 = S.size();
        R.push_back(r);
        q = r.end;
      }
    });
    size_t nRec = 0;
    std::vector<size_t> first(T + 1, 0);
    for (unsigned t = 0; t < T; t++) { first[t] = nRec; nRec += recs[t].size(); }
    first[T] = nRec;
    const size_t base = out.names.size();
    out.names.resize(base + nRec); out.offs.resize(base + nRec + 1);
    // names + filters (per thread), then lengths -> offsets
    detail::run_parallel(pool_, T, [&](unsigned t) {
      for (size_t i = 0; i < recs[t].size(); i++) {
        Rec& r = recs[t][i];
        const char* nb; const char* ne;
        detail::record_name(r.hdr, r.body, nb, ne);
        std::string& name = out.names[base + first[t] + i];
        name.assign(nb, ne);
        r.keep = (keepPrefix_.empty() || name.compare(0, keepPrefix_.size(), keepPrefix_) == 0) && (keepSeq_.empty() || keepSeq_.count(name));
        if (!r.keep) r.seqLen = 0;
      }
    });
    int64_t at = out.offs[base];
    for (unsigned t = 0; t < T; t++) for (size_t i = 0; i < recs[t].size(); i++) { out.offs[base + first[t] + i] = at; at += recs[t][i].seqLen; }
    out.offs[base + nRec] = at;
    out.packed = packOutput_;
    if (packOutput_) { packWindow(recs, segs, first, T, nRec, out); return; }
    if ((size_t)at + 64 > out.cap) {
      char* nb = alloc_((size_t)at + ((size_t)at >> 4) + 4096);
      if (!nb) { std::cerr << "[mashmap_hip] out of host memory for a batch of " << at << " bases" << std::endl; exit(1); }
      if (out.bases) { if (out.offs[base]) std::memcpy(nb, out.bases, (size_t)out.offs[base]); free_(out.bases); }
      out.bases = nb; out.cap = (size_t)at + ((size_t)at >> 4) + 4096;
    }
    // sequence bytes, line breaks dropped
    char* dst0 = out.bases;
    detail::run_parallel(pool_, T, [&](unsigned t) {
      for (size_t i = 0; i < recs[t].size(); i++) {
        const Rec& r = recs[t][i];
        if (!r.keep || !r.seqLen) continue;
        char* d = dst0 + out.offs[base + first[t] + i];
        for (size_t g = r.seg0; g < r.seg1; g++) { std::memcpy(d, segs[t][g].p, segs[t][g].n); d += segs[t][g].n; }
      }
    });
  }
};

}  // namespace mmhost
