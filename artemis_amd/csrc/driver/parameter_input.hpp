// Minimal reader for Parthenon-style input decks (the reference's inputs/*/*.in):
//   <block/name>            section header
//   key = value  # comment  (values may continue on the next line with a trailing `&`,
//                            inputs/blast/blast.in:21-24)
// plus command-line overrides `block/key=value` (tst/scripts/coords/blast.py:91-95).
// GetOrAdd* mirror parthenon::ParameterInput (upstream) so call sites read like the
// reference's Initialize functions (gas.cpp:55-208, dust.cpp:45-110).
#pragma once
#include <cstdlib>
#include <map>
#include <vector>
#include <set>
#include <sstream>
#include <stdexcept>
#include <string>

namespace artemis_host {

class ParameterInput {
 public:
  void LoadFromString(const std::string &text) {
    std::istringstream in(text);
    std::string line, block, pending_key, pending_val;
    bool cont = false;
    while (std::getline(in, line)) {
      const auto hash = line.find('#');
      if (hash != std::string::npos) line = line.substr(0, hash);
      line = trim(line);
      if (line.empty()) continue;
      if (cont) {
        cont = ends_with_amp(line);
        pending_val += " " + trim(strip_amp(line));
        if (!cont) Set(block, pending_key, trim(pending_val));
        continue;
      }
      if (line.front() == '<') {
        const auto close = line.find('>');
        if (close == std::string::npos) throw std::runtime_error("bad block header: " + line);
        block = trim(line.substr(1, close - 1));
        blocks_.insert(block);
        continue;
      }
      const auto eq = line.find('=');
      if (eq == std::string::npos) throw std::runtime_error("bad input line: " + line);
      pending_key = trim(line.substr(0, eq));
      std::string val = trim(line.substr(eq + 1));
      cont = ends_with_amp(val);
      pending_val = trim(strip_amp(val));
      if (!cont) Set(block, pending_key, pending_val);
    }
  }
  // "block/key=value" (block itself may contain '/', e.g. parthenon/mesh/nx1=64)
  void ApplyOverride(const std::string &arg) {
    const auto eq = arg.find('=');
    if (eq == std::string::npos) throw std::runtime_error("bad override: " + arg);
    const std::string path = arg.substr(0, eq);
    const auto slash = path.rfind('/');
    if (slash == std::string::npos) throw std::runtime_error("bad override: " + arg);
    Set(trim(path.substr(0, slash)), trim(path.substr(slash + 1)), trim(arg.substr(eq + 1)));
  }
  void Set(const std::string &block, const std::string &key, const std::string &val) {
    data_[block + "/" + key] = val;
    blocks_.insert(block);
  }
  bool DoesBlockExist(const std::string &block) const { return blocks_.count(block) > 0; }
  // names of the blocks that start with `prefix` (the reference walks its block list the same way,
  // nbody/nbody_setup.cpp:648-665)
  std::vector<std::string> BlocksWithPrefix(const std::string &prefix) const {
    std::vector<std::string> out;
    for (const std::string &b : blocks_)
      if (b.compare(0, prefix.size(), prefix) == 0) out.push_back(b);
    return out;
  }
  // comma-separated list (parthenon ParameterInput::GetVector, upstream)
  std::vector<double> GetVector(const std::string &b, const std::string &k) const {
    std::vector<double> out;
    std::istringstream in(GetString(b, k));
    std::string tok;
    while (std::getline(in, tok, ',')) {
      tok = trim(tok);
      if (!tok.empty()) out.push_back(std::strtod(tok.c_str(), nullptr));
    }
    return out;
  }
  bool DoesParameterExist(const std::string &block, const std::string &key) const {
    return data_.count(block + "/" + key) > 0;
  }
  std::string GetString(const std::string &block, const std::string &key) const {
    auto it = data_.find(block + "/" + key);
    if (it == data_.end())
      throw std::runtime_error("Parameter name '" + key + "' not found in block '" + block + "'");
    return it->second;
  }
  double GetReal(const std::string &b, const std::string &k) const {
    return std::strtod(GetString(b, k).c_str(), nullptr);
  }
  int GetInteger(const std::string &b, const std::string &k) const {
    // the reference's test scripts pass e.g. nx2=16.0 (repr(res / 2), linwave.py:48)
    return static_cast<int>(std::strtod(GetString(b, k).c_str(), nullptr));
  }
  std::string GetOrAddString(const std::string &b, const std::string &k, const std::string &d) {
    if (!DoesParameterExist(b, k)) Set(b, k, d);
    return GetString(b, k);
  }
  double GetOrAddReal(const std::string &b, const std::string &k, double d) {
    if (!DoesParameterExist(b, k)) {
      std::ostringstream o;
      o.precision(17);
      o << d;
      Set(b, k, o.str());
      return d;
    }
    return GetReal(b, k);
  }
  int GetOrAddInteger(const std::string &b, const std::string &k, int d) {
    if (!DoesParameterExist(b, k)) {
      Set(b, k, std::to_string(d));
      return d;
    }
    return GetInteger(b, k);
  }
  bool GetOrAddBoolean(const std::string &b, const std::string &k, bool d) {
    if (!DoesParameterExist(b, k)) {
      Set(b, k, d ? "true" : "false");
      return d;
    }
    const std::string v = GetString(b, k);
    return v == "true" || v == "True" || v == "1";
  }

 private:
  static std::string trim(const std::string &s) {
    const auto a = s.find_first_not_of(" \t\r\n");
    if (a == std::string::npos) return "";
    const auto b = s.find_last_not_of(" \t\r\n");
    return s.substr(a, b - a + 1);
  }
  static bool ends_with_amp(const std::string &s) {
    const std::string t = trim(s);
    return !t.empty() && t.back() == '&';
  }
  static std::string strip_amp(const std::string &s) {
    std::string t = trim(s);
    if (!t.empty() && t.back() == '&') t.pop_back();
    return t;
  }
  std::map<std::string, std::string> data_;
  std::set<std::string> blocks_;
};

} // namespace artemis_host
