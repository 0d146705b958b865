// TEST INFRASTRUCTURE.  Minimal stand-in for the two glog macros the reference's runtime front-end uses
// (runtime/frontend/fbank.h:85,130: CHECK, CHECK_GE), so that header compiles here without the glog package
// (the reference fetches glog over the network at build time; there is no network in this image).
#ifndef WESEP_AMD_ORACLE_GLOG_SHIM_H_
#define WESEP_AMD_ORACLE_GLOG_SHIM_H_
#include <cstdio>
#include <cstdlib>
#define CHECK(cond)                                                            \
  do {                                                                         \
    if (!(cond)) {                                                             \
      std::fprintf(stderr, "CHECK failed: %s (%s:%d)\n", #cond, __FILE__, __LINE__); \
      std::abort();                                                            \
    }                                                                          \
  } while (0)
#define CHECK_GE(a, b) CHECK((a) >= (b))
#endif
