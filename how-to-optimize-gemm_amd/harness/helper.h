// helper.h -- abort-on-error wrappers for the harness (role of
// cuda/helper.h:7-17's checkCudaErrors: print file:line and exit).  The
// library itself never exits; only this driver does.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#include "../../include/mmult_hip.h"

#define HIP_CHECK(expr)                                                              \
  do {                                                                               \
    hipError_t err_ = (expr);                                                        \
    if (err_ != hipSuccess) {                                                        \
      std::fprintf(stderr, "HIP error at %s:%d code=%d (%s) \"%s\"\n", __FILE__,    \
                   __LINE__, (int)err_, hipGetErrorString(err_), #expr);            \
      std::exit(EXIT_FAILURE);                                                       \
    }                                                                                \
  } while (0)

#define MMH_CHECK(expr)                                                              \
  do {                                                                               \
    int st_ = (expr);                                                                \
    if (st_ != MMH_OK) {                                                             \
      std::fprintf(stderr, "mmult_hip error at %s:%d status=%d (%s; %s) \"%s\"\n",  \
                   __FILE__, __LINE__, st_, mmh_strerror(st_), mmh_last_error(),    \
                   #expr);                                                           \
      std::exit(EXIT_FAILURE);                                                       \
    }                                                                                \
  } while (0)
