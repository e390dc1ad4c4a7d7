// Stand-in for <glog/logging.h>: LOG(severity) << ... ; FATAL aborts.  TEST INFRASTRUCTURE.
#pragma once
#include <cstdlib>
#include <iostream>
#include <sstream>
namespace dfx_test_glog {
struct Sink {
  bool fatal;
  std::ostringstream os;
  explicit Sink(bool f) : fatal(f) {}
  ~Sink() { if (fatal) { std::cerr << "LOG(FATAL): " << os.str() << std::endl; std::abort(); } }
  template <typename T> Sink& operator<<(const T& v) { os << v; return *this; }
};
struct SevINFO { static constexpr bool fatal = false; };
struct SevWARNING { static constexpr bool fatal = false; };
struct SevERROR { static constexpr bool fatal = false; };
struct SevFATAL { static constexpr bool fatal = true; };
}  // namespace dfx_test_glog
#define LOG(sev) ::dfx_test_glog::Sink(::dfx_test_glog::Sev##sev::fatal)
#define VLOG(n) ::dfx_test_glog::Sink(false)
#define CHECK(c) if (!(c)) ::dfx_test_glog::Sink(true) << "CHECK failed: " #c " "
