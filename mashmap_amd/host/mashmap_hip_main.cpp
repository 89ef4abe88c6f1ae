// mashmap_amd/host/mashmap_hip_main.cpp -- `mashmap_hip`: the reference's command line (src/map/mash_map.cpp:22-57 +
// src/map/include/parseCmdArgs.hpp:30-657) in front of skch::Sketch / skch::Map built on libmashmap_hip.so.
//
// Inside the reference tree none of this file is needed: mash_map.cpp and parseCmdArgs.hpp compile unchanged against
// skch_sketch.hpp / skch_map.hpp (tests/test_dropin_compile.py).  This driver exists because the reference's
// sources (and its vendored argvparser) do not travel to the GPU box; it accepts the same option names, defaults and
// derived values, so the same command line produces the same PAF.
#include <chrono>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <map>
#include <sstream>
#include <string>

#include "skch_map.hpp"
#include "skch_sketch.hpp"
#include "skch_types.hpp"

namespace {

struct OptDef { const char* name; const char* alt; bool value; };
const OptDef kOpts[] = {
    {"help", "h", false}, {"version", "v", false}, {"ref", "r", true}, {"refList", "rl", true}, {"query", "q", true}, {"queryList", "ql", true},
    {"segLength", "s", true}, {"sketchSize", "J", true}, {"dense", "", false}, {"blockLength", "l", true}, {"chainGap", "c", true},
    {"numMappingsForSegment", "n", true}, {"numMappingsForShortSeq", "", true}, {"saveIndex", "", true}, {"loadIndex", "", true},
    {"noSplit", "", false}, {"perc_identity", "pi", true}, {"dropLowMapId", "K", false}, {"threads", "t", true}, {"output", "o", true},
    {"kmer", "k", true}, {"kmerThreshold", "", true}, {"kmerComplexity", "", true}, {"noHgFilter", "", false}, {"hgFilterAniDiff", "", true},
    {"hgFilterConf", "", true}, {"filterLengthMismatches", "", false}, {"lowerTriangular", "", false}, {"skipSelf", "X", false},
    {"skipPrefix", "Y", true}, {"targetPrefix", "", true}, {"targetList", "", true}, {"sparsifyMappings", "x", true}, {"noMerge", "M", false},
    {"filter_mode", "f", true}, {"legacy", "", false}, {"reportPercentage", "", false}};

struct Cmd {
  std::map<std::string, std::string> got;
  bool found(const std::string& n) const { return got.count(n) != 0; }
  template <class T> T value(const std::string& n) const { std::stringstream s; s << got.at(n); T v{}; s >> v; return v; }
};

[[noreturn]] void usage_error(const std::string& msg) {
  std::cerr << msg << std::endl;
  exit(1);
}

Cmd parse(int argc, char** argv) {
  Cmd c;
  for (int i = 1; i < argc; i++) {
    std::string a = argv[i];
    if (a.size() < 2 || a[0] != '-') usage_error("ERROR, unexpected argument '" + a + "'");
    std::string name = a.substr(a[1] == '-' ? 2 : 1), val;
    bool inlineVal = false;
    const size_t eq = name.find('=');
    if (eq != std::string::npos) { val = name.substr(eq + 1); name = name.substr(0, eq); inlineVal = true; }
    const OptDef* d = nullptr;
    for (const auto& o : kOpts) if (name == o.name || (*o.alt && name == o.alt)) d = &o;
    if (!d) usage_error("ERROR, unknown option '" + a + "'");
    if (d->value && !inlineVal) {
      if (i + 1 >= argc) usage_error(std::string("ERROR, option '") + d->name + "' requires a value");
      val = argv[++i];
    }
    c.got[d->name] = val;
  }
  return c;
}

void parseFileList(const std::string& listFile, std::vector<std::string>& out) {      // parseCmdArgs.hpp:141-161
  std::ifstream in(listFile);
  if (!in) usage_error("ERROR, skch::parseFileList, Could not open " + listFile);
  std::string line;
  while (std::getline(in, line)) out.push_back(line);
}

uint64_t referenceBytes(const std::vector<std::string>& files) {                        // commonFunc.hpp:591-603
  uint64_t n = 0;
  for (const auto& f : files) { std::ifstream in(f, std::ifstream::ate | std::ifstream::binary); n += (uint64_t)in.tellg(); }
  return n;
}

void fill(const Cmd& cmd, skch::Parameters& p) {                                        // parseCmdArgs.hpp:257-657, same order
  if (cmd.found("version")) { std::cerr << skch::fixed::VERSION << std::endl; exit(0); }
  if (!cmd.found("ref") && !cmd.found("refList")) usage_error("ERROR, skch::parseandSave, Provide reference file(s)");
  if (cmd.found("ref")) p.refSequences.push_back(cmd.value<std::string>("ref"));
  else parseFileList(cmd.value<std::string>("refList"), p.refSequences);
  p.referenceSize = (skch::offset_t)referenceBytes(p.refSequences);       // bytes on disk, through an int32 (map_parameters.hpp:41)
  if (cmd.found("query")) p.querySequences.push_back(cmd.value<std::string>("query"));
  else if (cmd.found("queryList")) parseFileList(cmd.value<std::string>("queryList"), p.querySequences);
  else p.querySequences = p.refSequences;
  p.lower_triangular = cmd.found("lowerTriangular");
  p.skip_self = cmd.found("skipSelf");                                    // overrides the no-query default, as :326-345 does
  if (cmd.found("skipPrefix")) { p.prefix_delim = cmd.value<char>("skipPrefix"); p.skip_prefix = true; }
  else { p.skip_prefix = false; p.prefix_delim = '\0'; }
  if (cmd.found("targetList")) p.target_list = cmd.value<std::string>("targetList");
  if (cmd.found("targetPrefix")) p.target_prefix = cmd.value<std::string>("targetPrefix");
  if (cmd.found("saveIndex")) p.saveIndexFilename = cmd.value<std::string>("saveIndex");
  if (cmd.found("loadIndex")) p.loadIndexFilename = cmd.value<std::string>("loadIndex");
  p.alphabetSize = 4;
  p.filterLengthMismatches = cmd.found("filterLengthMismatches");
  p.stage1_topANI_filter = !cmd.found("noHgFilter");
  p.filterMode = skch::filter::MAP;
  if (cmd.found("filter_mode")) {
    const std::string m = cmd.value<std::string>("filter_mode");
    if (m == "map") p.filterMode = skch::filter::MAP;
    else if (m == "one-to-one") p.filterMode = skch::filter::ONETOONE;
    else if (m == "none") { p.stage1_topANI_filter = false; p.filterMode = skch::filter::NONE; }
    else usage_error("ERROR, skch::parseandSave, Invalid option given for filter_mode");
  }
  p.split = !cmd.found("noSplit");
  p.mergeMappings = !cmd.found("noMerge");
  p.kmerSize = cmd.found("kmer") ? cmd.value<int>("kmer") : 19;
  p.segLength = 5000;
  if (cmd.found("segLength")) {
    p.segLength = cmd.value<skch::offset_t>("segLength");
    if (p.segLength < 100) usage_error("ERROR, skch::parseandSave, minimum segment length is required to be >= 100 bp.");
  }
  p.block_length = p.segLength;
  if (cmd.found("blockLength")) {
    p.block_length = cmd.value<skch::offset_t>("blockLength");
    if (p.block_length < 0) usage_error("[mashmap] ERROR, skch::parseandSave, min block length has to be a float value greater than or equal to 0.");
  }
  p.chain_gap = p.segLength;
  if (cmd.found("chainGap")) {
    const int64_t l = cmd.value<int64_t>("chainGap");
    if (l < 0) usage_error("[mashmap] ERROR, skch::parseandSave, chain gap has to be a float value greater than or equal to 0.");
    p.chain_gap = (skch::offset_t)l;
  }
  p.keep_low_pct_id = !cmd.found("dropLowMapId");
  p.kmer_pct_threshold = cmd.found("kmerThreshold") ? cmd.value<float>("kmerThreshold") : 0.001f;
  p.numMappingsForSegment = 1;
  if (cmd.found("numMappingsForSegment")) {
    p.numMappingsForSegment = cmd.value<uint32_t>("numMappingsForSegment");
    if (p.numMappingsForSegment == 0) usage_error("[mashmap] ERROR, skch::parseandSave, the number of mappings to retain for each segment has to be greater than 0.");
  }
  p.numMappingsForShortSequence = 1;
  if (cmd.found("numMappingsForShortSeq")) {
    p.numMappingsForShortSequence = cmd.value<uint32_t>("numMappingsForShortSeq");
    if (p.numMappingsForShortSequence == 0) usage_error("[mashmap] ERROR, skch::parseandSave, the number of mappings to retain for each short sequence has to be greater than 0.");
  }
  p.percentageIdentity = 0.85f;
  if (cmd.found("perc_identity")) {
    p.percentageIdentity = cmd.value<float>("perc_identity");
    if (p.percentageIdentity < 50) usage_error("ERROR, skch::parseandSave, minimum nucleotide identity requirement should be >= 50%");
    p.percentageIdentity /= 100.0;
  }
  p.kmerComplexityThreshold = cmd.found("kmerComplexity") ? cmd.value<float>("kmerComplexity") : 0.0f;
  p.ANIDiff = skch::fixed::ANIDiff;
  if (cmd.found("hgFilterAniDiff")) {
    p.ANIDiff = cmd.value<float>("hgFilterAniDiff");
    if (p.ANIDiff < 0 || p.ANIDiff > 100) usage_error("ERROR, skch::parseandSave, ANI difference must be between 0 and 100");
    p.ANIDiff /= 100;
  }
  p.ANIDiffConf = skch::fixed::ANIDiffConf;
  if (cmd.found("hgFilterConf")) {
    p.ANIDiffConf = cmd.value<float>("hgFilterConf");
    if (p.ANIDiffConf < 0 || p.ANIDiffConf > 100) usage_error("ERROR, skch::parseandSave, hypergeometric confidence must be between 0 and 100");
    p.ANIDiffConf /= 100;
  }
  p.stage2_full_scan = true;
  p.sparsity_hash_threshold = std::numeric_limits<uint64_t>::max();
  if (cmd.found("sparsifyMappings")) {
    const double frac = cmd.value<double>("sparsifyMappings");
    if (frac != 1) p.sparsity_hash_threshold = frac * std::numeric_limits<uint64_t>::max();
  }
  p.threads = cmd.found("threads") ? cmd.value<int>("threads") : 1;
  if (cmd.found("sketchSize")) p.sketchSize = cmd.value<int>("sketchSize");
  else if (cmd.found("dense")) {
    const double md = 1 - p.percentageIdentity;
    const double dens = 0.02 * (1 + (md / 0.05));
    p.sketchSize = dens * (p.segLength - p.kmerSize);
  } else {
    p.sketchSize = mmhost::Stat::recommendedSketchSize(skch::fixed::pval_cutoff, skch::fixed::confidence_interval, p.kmerSize, p.alphabetSize,
                                                       p.percentageIdentity, p.segLength, p.referenceSize);
  }
  p.outFileName = cmd.found("output") ? cmd.value<std::string>("output") : "mashmap.out";
  p.legacy_output = cmd.found("legacy");
  p.report_ANI_percentage = cmd.found("reportPercentage");
  std::cerr << "[mashmap_hip] reference files = " << p.refSequences.size() << ", query files = " << p.querySequences.size()
            << ", k = " << p.kmerSize << ", segLength = " << p.segLength << ", sketchSize = " << p.sketchSize
            << ", pi = " << 100 * p.percentageIdentity << "%, threads = " << p.threads << std::endl;
  for (const auto& f : p.refSequences) if (!std::ifstream(f)) usage_error("ERROR, skch::validateInputFiles, Could not open " + f);
  for (const auto& f : p.querySequences) if (!std::ifstream(f)) usage_error("ERROR, skch::validateInputFiles, Could not open " + f);
}

}  // namespace

int main(int argc, char** argv) {
  skch::Parameters parameters;
  fill(parse(argc, argv), parameters);
  auto t0 = skch::Time::now();
  skch::Sketch referSketch(parameters);
  std::chrono::duration<double> timeRefSketch = skch::Time::now() - t0;
  std::cerr << "[mashmap::map] time spent computing the reference index: " << timeRefSketch.count() << " sec" << std::endl;
  t0 = skch::Time::now();
  skch::Map mapper(parameters, referSketch);
  std::chrono::duration<double> timeMapQuery = skch::Time::now() - t0;
  std::cerr << "[mashmap::map] time spent mapping the query: " << timeMapQuery.count() << " sec" << std::endl;
  std::cerr << "[mashmap::map] mapping results saved in: " << parameters.outFileName << std::endl;
  return 0;
}
