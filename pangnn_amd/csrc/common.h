// Shared helpers for libpangnn_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/pangnn_hip.h"

namespace pangnn {

void set_error(const char* fmt, ...);

#define PG_CHECK_ARG(cond, code, ...)            \
  do {                                           \
    if (!(cond)) {                               \
      ::pangnn::set_error(__VA_ARGS__);          \
      return (code);                             \
    }                                            \
  } while (0)

#define PG_CHECK_LAUNCH(name)                                                          \
  do {                                                                                 \
    hipError_t e__ = hipGetLastError();                                                \
    if (e__ != hipSuccess) {                                                           \
      ::pangnn::set_error("%s: launch failed: %s", (name), hipGetErrorString(e__));    \
      return (int)e__;                                                                 \
    }                                                                                  \
  } while (0)

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

constexpr int kWave = 64;          // CDNA4 wavefront
constexpr int kBlock = 256;        // 4 waves per workgroup
constexpr int kWavesPerBlock = kBlock / kWave;

}  // namespace pangnn
