// mashmap_amd/host/pass_plan.hpp -- how skch::Map groups the reader's batches into device passes: the hand-over queue between two stages
// of its pipeline and the size a pass should have.  Kept apart from skch_map.hpp so that it can be exercised without a GPU
// (tests/hostlogic/pass_check.cpp).
#pragma once
#include <algorithm>
#include <condition_variable>
#include <cstddef>
#include <cstdint>
#include <deque>
#include <mutex>
#include <utility>
#include <vector>

namespace mmhost {

// Hand-over between two stages: at most `cap` items waiting.  T needs `size_t bases() const`.
template <class T>
class BatchChannel {
  std::deque<T> q; std::mutex mu; std::condition_variable cvFull, cvEmpty; bool done = false; size_t cap;

 public:
  explicit BatchChannel(size_t c) : cap(c) {}
  void put(T&& b) { std::unique_lock<std::mutex> lk(mu); cvFull.wait(lk, [&] { return q.size() < cap; }); q.emplace_back(std::move(b)); lk.unlock(); cvEmpty.notify_one(); }
  // blocks until there is room; with a single producer the put() that follows does not wait
  void waitSpace() { std::unique_lock<std::mutex> lk(mu); cvFull.wait(lk, [&] { return q.size() < cap; }); }
  bool get(T& b) {
    std::unique_lock<std::mutex> lk(mu);
    cvEmpty.wait(lk, [&] { return !q.empty() || done; });
    if (q.empty()) return false;
    b = std::move(q.front()); q.pop_front();
    lk.unlock(); cvFull.notify_one();
    return true;
  }
  void close() { { std::lock_guard<std::mutex> lk(mu); done = true; } cvEmpty.notify_all(); }
  // The items of one device pass: waits until `want` bases are queued -- or the producer is done, or the queue is full, or it holds
  // `maxItems` --, then takes items from the front until the pass holds `want` bases: at least one, at most `maxItems`.  false: the
  // producer is done and nothing is left.
  // takeAll: once the wait is over the pass takes everything that is queued (up to maxItems), not just `want` bases' worth -- with a small
  // `want` that is the greedy policy: a pass never waits for input that is not there yet and never leaves input behind that is.
  bool getGroup(std::vector<T>& g, size_t want, size_t maxItems, bool takeAll = false) {
    std::unique_lock<std::mutex> lk(mu);
    cvEmpty.wait(lk, [&] {
      if (done || q.size() >= cap || q.size() >= maxItems) return !q.empty() || done;
      size_t have = 0; for (const auto& b : q) have += b.bases();
      return !q.empty() && have >= want;
    });
    if (q.empty()) return false;
    size_t have = 0;
    while (!q.empty() && g.size() < maxItems && (g.empty() || takeAll || have < want)) { have += q.front().bases(); g.emplace_back(std::move(q.front())); q.pop_front(); }
    lk.unlock(); cvFull.notify_all();
    return true;
  }
  // every queued item in order (only the consumer removes items: what fn sees stays put until the consumer's next get)
  template <class F> void forEach(F fn) { std::lock_guard<std::mutex> lk(mu); for (auto& b : q) fn(b); }
};

// Bases the next device pass should hold, given what has been mapped so far: one batch at first (the post stage has work after one
// batch's worth of time), doubling up to `passBases` (1, 1, 2, 4, 4 ... batches), and, when the input's size is known, down again
// towards its end (a pass takes at most half of what is left: the last pass and its post stage are what the run waits for with nothing
// left to overlap them).  A sixteenth is
// taken off: a batch is a hair under `batchBases`, the parser cuts at a record boundary.
inline size_t passWant(size_t batchBases, size_t passBases, bool inputKnown, uint64_t inputBytes, uint64_t doneBases) {
  uint64_t want = std::min<uint64_t>(passBases, std::max<uint64_t>(batchBases, doneBases));
  if (inputKnown && inputBytes > doneBases) want = std::min<uint64_t>(want, std::max<uint64_t>(batchBases, (inputBytes - doneBases) / 2));
  return (size_t)(want - want / 16);
}

}  // namespace mmhost
