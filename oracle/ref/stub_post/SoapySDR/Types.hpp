// TEST INFRASTRUCTURE ONLY (oracle/): the few SoapySDR names the reference's HEADERS mention, so that src/sdr/SDRPostThread.cpp can be
// compiled unmodified into oracle/_ref/libref_post.so (its routing code calls none of them).
#pragma once
#include <map>
#include <string>
#include <vector>
namespace SoapySDR {
typedef std::map<std::string, std::string> Kwargs;
typedef std::vector<Kwargs> KwargsList;
class Range { public: Range() : lo(0), hi(0), st(0) {} Range(double a, double b, double c = 0) : lo(a), hi(b), st(c) {} double minimum() const { return lo; } double maximum() const { return hi; } double step() const { return st; } private: double lo, hi, st; };
typedef std::vector<Range> RangeList;
class ArgInfo { public: std::string key, value, name, description, units; enum Type { BOOL, INT, FLOAT, STRING } type = STRING; Range range; std::vector<std::string> options, optionNames; };
typedef std::vector<ArgInfo> ArgInfoList;
}
