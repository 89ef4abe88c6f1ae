// tests/hostlogic/pass_check.cpp -- TEST HARNESS for mashmap_amd/host/pass_plan.hpp (no GPU): a producer thread and a consumer thread around
// a BatchChannel, the consumer taking device passes with passWant's ramp, under randomised timing.  Prints one line per scenario:
//   "ok <scenario> items <n> passes <p> sizes <s1,s2,...>"   or   "FAIL ..." (and exits 1)
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <thread>
#include "../../mashmap_amd/host/pass_plan.hpp"
#include "../../mashmap_amd/host/skch_types.hpp"

struct Item { size_t id; size_t n; size_t bases() const { return n; } };

static int scenario(const char* name, size_t items, size_t batchBases, size_t passBases, bool known, size_t cap, int producerUs, int consumerUs, unsigned seed) {
  const size_t maxGroup = std::max<size_t>(1, passBases / batchBases);
  mmhost::BatchChannel<Item> ch(cap);
  std::mt19937 rng(seed);
  std::vector<size_t> sizes(items);
  uint64_t total = 0;
  for (size_t i = 0; i < items; i++) { sizes[i] = batchBases - 1 - rng() % (batchBases / 50 + 1); if (i + 1 == items) sizes[i] = 1 + rng() % batchBases; total += sizes[i]; }
  std::thread prod([&] {
    std::mt19937 r2(seed * 7 + 1);
    for (size_t i = 0; i < items; i++) {
      if (producerUs) std::this_thread::sleep_for(std::chrono::microseconds(r2() % (2 * producerUs + 1)));
      ch.waitSpace();
      ch.put(Item{i, sizes[i]});
    }
    ch.close();
  });
  std::vector<Item> g; std::vector<size_t> passSizes;
  size_t next = 0; uint64_t done = 0; bool bad = false; std::string why;
  std::mt19937 r3(seed * 13 + 5);
  while (true) {
    const size_t want = maxGroup == 1 ? 0 : mmhost::passWant(batchBases, passBases, known, total + total / 1000, done);
    g.clear();
    if (!ch.getGroup(g, want, maxGroup)) break;
    if (g.empty() || g.size() > maxGroup) { bad = true; why = "group size"; break; }
    size_t have = 0;
    for (size_t i = 0; i < g.size(); i++) {
      if (g[i].id != next++) { bad = true; why = "order"; }
      if (i + 1 < g.size() && have + g[i].n >= want && want) { /* the group was full before its last item */ bad = true; why = "took more than it wanted"; }
      have += g[i].n;
    }
    if (bad) break;
    if (want > passBases) { bad = true; why = "want beyond the pass size"; break; }
    passSizes.push_back(g.size());
    for (const auto& it : g) done += it.n;
    if (consumerUs) std::this_thread::sleep_for(std::chrono::microseconds(r3() % (2 * consumerUs * g.size() + 1)));
  }
  prod.join();
  if (!bad && next != items) { bad = true; why = "items lost"; }
  if (!bad && done != total) { bad = true; why = "bases lost"; }
  printf("%s %s items %zu passes %zu sizes", bad ? "FAIL" : "ok", name, items, passSizes.size());
  for (size_t i = 0; i < passSizes.size(); i++) printf("%c%zu", i ? ',' : ' ', passSizes[i]);
  if (bad) printf(" (%s)", why.c_str());
  printf("\n");
  return bad ? 1 : 0;
}

// skch::queryBatchPlan (skch_types.hpp): batch / pass sizes and page-locked buffers for a query file of `bytes` bytes under the given
// environment; prints what it decided
static int plan_check(const char* path) {
  auto show = [&](const char* what, size_t ctxs) {
    const skch::QueryBatchPlan q = skch::queryBatchPlan({path}, ctxs);
    printf("plan %s ctxs %zu batch %zu pass %zu buffers %zu bufferBytes %zu known %d\n", what, ctxs, q.batchBases, q.passBases, q.buffers, q.bufferBytes, (int)q.inputKnown);
  };
  show("default", 1); show("default", 2);
  setenv("MASHMAP_HIP_COALESCE_MBP", "0", 1); show("coalesce0", 1);
  setenv("MASHMAP_HIP_COALESCE_MBP", "4096", 1); setenv("MASHMAP_HIP_BATCH_MBP", "256", 1); show("b256c4096", 1);
  setenv("MASHMAP_HIP_BATCH_MBP", "0.01", 1); setenv("MASHMAP_HIP_COALESCE_MBP", "2048", 1); show("tiny", 1);
  unsetenv("MASHMAP_HIP_BATCH_MBP"); unsetenv("MASHMAP_HIP_COALESCE_MBP");
  setenv("MASHMAP_HIP_ASCII_UPLOAD", "1", 1); show("ascii", 1); unsetenv("MASHMAP_HIP_ASCII_UPLOAD");
  return 0;
}

int main(int argc, char** argv) {
  if (argc == 3 && std::string(argv[1]) == "plan") return plan_check(argv[2]);
  int rc = 0;
  // a fast producer (the device is the bottleneck): the ramp 1, 1, 2, 4, 4 ... and down again when the size is known
  rc |= scenario("fast-producer-known", 20, 512, 2048, true, 4, 0, 300, 1);
  rc |= scenario("fast-producer-unknown", 20, 512, 2048, false, 4, 0, 300, 2);
  // a slow producer (the reader is the bottleneck): the consumer waits for its pass, nothing is lost, nothing deadlocks
  rc |= scenario("slow-producer-known", 23, 512, 2048, true, 4, 400, 50, 3);
  // one batch per pass (several contexts, ASCII uploads, MASHMAP_HIP_COALESCE_MBP=0)
  rc |= scenario("no-coalescing", 9, 512, 512, true, 2, 100, 100, 4);
  // tiny batches, many per pass (the small-batch PAF tests: MASHMAP_HIP_BATCH_MBP=0.05)
  rc |= scenario("tiny-batches", 300, 50, 3200, true, 64, 20, 20, 5);
  rc |= scenario("single-item", 1, 512, 2048, true, 4, 0, 0, 6);
  rc |= scenario("queue-smaller-than-pass", 40, 512, 4096, false, 3, 10, 10, 7);
  for (unsigned s = 10; s < 30; s++) rc |= scenario("random-timing", 37, 512, 2048, (s & 1) != 0, 4, (s * 37) % 200, (s * 53) % 200, s);
  return rc;
}
