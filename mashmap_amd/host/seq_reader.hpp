// mashmap_amd/host/seq_reader.hpp -- FASTA / FASTQ (optionally gzip) record iterator.
//
// Same record semantics as seqiter::for_each_seq_in_file (src/common/seqiter.hpp:20-111, the non-htslib path):
//   * the first byte of the file decides the format ('>' FASTA, '@' FASTQ), anything else is fatal;
//   * name = header without its first character, cut at the first ' ' (:82);
//   * FASTA sequence = concatenation of the following lines up to the next line starting with '>';
//   * records failing the keep_prefix / keep_seq filters are still reported, with an empty sequence (:84-97);
//   * FASTQ: sequence line, then two lines skipped (:104-107).
// Differs only in mechanics: one gzread() stream with a 4 MiB buffer instead of a 303-byte igzstream
// (src/common/gzstream.h:50), which is what caps the reference's ingest at ~37 Mbp/s.
#pragma once
#include <zlib.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <iostream>
#include <string>
#include <unordered_set>
#include <vector>

namespace mmhost {

class LineReader {
  gzFile f_ = nullptr;
  std::vector<char> buf_;
  size_t pos_ = 0, end_ = 0;
  bool eof_ = false;

  bool fill() {
    if (eof_) return false;
    const int n = gzread(f_, buf_.data(), (unsigned)buf_.size());
    if (n <= 0) { eof_ = true; pos_ = end_ = 0; return false; }
    pos_ = 0; end_ = (size_t)n;
    return true;
  }

 public:
  explicit LineReader(const std::string& path) : buf_(4u << 20) {
    f_ = gzopen(path.c_str(), "rb");
    if (f_) gzbuffer(f_, 1u << 20);
  }
  ~LineReader() { if (f_) gzclose(f_); }
  bool ok() const { return f_ != nullptr; }
  // std::getline semantics: false only when nothing at all could be read
  bool getline(std::string& out) {
    out.clear();
    bool any = false;
    while (true) {
      if (pos_ == end_ && !fill()) return any;
      any = true;
      const char* b = buf_.data() + pos_;
      const char* nl = (const char*)memchr(b, '\n', end_ - pos_);
      if (nl) { out.append(b, (size_t)(nl - b)); pos_ += (size_t)(nl - b) + 1; return true; }
      out.append(b, end_ - pos_); pos_ = end_;
    }
  }
  // appends the line to `out` instead of replacing it; *first receives the line's first byte (0 if empty)
  bool appendline(std::string& out, char* first, bool commit_unless_header) {
    // peek the first byte to decide whether this line is the next header
    if (pos_ == end_ && !fill()) return false;
    *first = buf_[pos_];
    if (commit_unless_header && *first == '>') return true;      // caller re-reads it with getline()
    while (true) {
      const char* b = buf_.data() + pos_;
      const char* nl = (const char*)memchr(b, '\n', end_ - pos_);
      if (nl) { out.append(b, (size_t)(nl - b)); pos_ += (size_t)(nl - b) + 1; return true; }
      out.append(b, end_ - pos_); pos_ = end_;
      if (!fill()) return true;
    }
  }
  void skipline() {
    while (true) {
      if (pos_ == end_ && !fill()) return;
      const char* b = buf_.data() + pos_;
      const char* nl = (const char*)memchr(b, '\n', end_ - pos_);
      if (nl) { pos_ += (size_t)(nl - b) + 1; return; }
      pos_ = end_;
    }
  }
};

inline std::string record_name(const std::string& header) {      // seqiter.hpp:82
  return header.substr(1, header.find(" ") - 1);
}

inline void for_each_seq_in_file(const std::string& filename, const std::unordered_set<std::string>& keep_seq,
                                 const std::string& keep_prefix,
                                 const std::function<void(const std::string&, std::string&)>& func) {
  LineReader in(filename);
  std::string line;
  if (!in.ok() || !in.getline(line) || (line[0] != '>' && line[0] != '@')) {
    std::cerr << "[mashmap_hip::for_each_seq_in_file] unknown file format given to the sequence reader: " << filename << std::endl;
    exit(1);
  }
  const bool fasta = line[0] == '>';
  bool more = true;
  std::string seq;
  while (more) {
    const std::string name = record_name(line);
    const bool keep = (keep_prefix.empty() || name.substr(0, keep_prefix.length()) == keep_prefix) &&
                      (keep_seq.empty() || keep_seq.find(name) != keep_seq.end());
    seq.clear();
    if (fasta) {
      std::string drop;
      while (true) {
        char first = 0;
        if (!in.appendline(keep ? seq : drop, &first, true)) { more = false; break; }
        if (first == '>') { in.getline(line); break; }
        if (!keep) drop.clear();
      }
    } else {
      if (!in.getline(seq)) more = false;
      if (!keep) seq.clear();
      in.skipline(); in.skipline();
      if (!in.getline(line)) more = false;
    }
    func(name, seq);
  }
}

}  // namespace mmhost
