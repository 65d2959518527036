// cli_flags.h -- minimal gflags-style parsing for the bench tools, so that they accept
// the reference's flag names (benchmark/bench_concat.cc:22-29, bench_conv.cc:22-37):
//   -name value | -name=value | --name value | --name=value | -boolflag | -noboolflag
#pragma once
#include <cstdlib>
#include <cstring>
#include <map>
#include <sstream>
#include <string>
#include <vector>

struct Flags {
  std::map<std::string, std::string> kv;
  Flags(int argc, char **argv) {
    for (int i = 1; i < argc; ++i) {
      std::string a = argv[i];
      if (a.size() < 2 || a[0] != '-') continue;
      a = a.substr(a[1] == '-' ? 2 : 1);
      size_t eq = a.find('=');
      if (eq != std::string::npos) { kv[a.substr(0, eq)] = a.substr(eq + 1); continue; }
      if (i + 1 < argc && argv[i + 1][0] != '-') { kv[a] = argv[++i]; continue; }
      if (a.rfind("no", 0) == 0) kv[a.substr(2)] = "false";
      else kv[a] = "true";
    }
  }
  int geti(const char *n, int d) const { auto it = kv.find(n); return it == kv.end() ? d : atoi(it->second.c_str()); }
  std::string gets(const char *n, const char *d) const { auto it = kv.find(n); return it == kv.end() ? d : it->second; }
  bool getb(const char *n, bool d) const {
    auto it = kv.find(n);
    if (it == kv.end()) return d;
    return !(it->second == "false" || it->second == "0");
  }
  static std::vector<int> split_ints(const std::string &s) {
    std::vector<int> out; std::stringstream ss(s); std::string item;
    while (std::getline(ss, item, ',')) if (!item.empty()) out.push_back(atoi(item.c_str()));
    return out;
  }
};

// seeded LCG data generators with the reference's value ranges (test/test_utils.h:49-63)
struct Lcg {
  uint32_t s;
  explicit Lcg(uint32_t seed) : s(seed) {}
  uint32_t next() { s = s * 1103515245u + 12345u; return (s >> 16) & 0x7fff; }
};
