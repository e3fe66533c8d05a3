// parameters.h -- sweep defaults of the MI355X harness.
//
// The knobs are the reference's (cuda/parameters.h:5-24: first/last/increment of the square
// size p, optional fixed m/n/k, repeat count; armv7/parameters.h:38-46: optional fixed leading
// dimensions), but here they are DEFAULTS of run-time options, not compile-time macros: every
// one can be overridden by an environment variable (PFIRST, PLAST, PINC, M, N, K, NREPEATS,
// LDA, LDB, LDC) or a --NAME=value argument, so one binary covers all BASELINE.json configs.
#pragma once

namespace sweep_defaults {

constexpr int kFirstSize = 1024;   // p runs kFirstSize, kFirstSize + kStep, ... <= kLastSize
constexpr int kLastSize = 4096;
constexpr int kStep = 128;

constexpr int kBoundToP = -1;      // a dimension set to this follows p (the reference's "-1")
constexpr int kM = kBoundToP, kN = kBoundToP, kK = kBoundToP;

constexpr int kRepeats = 20;       // timed back-to-back calls per size

// leading dimensions; kBoundToP = dense (lda = k, ldb = n, ldc = n)
constexpr int kLda = kBoundToP, kLdb = kBoundToP, kLdc = kBoundToP;

}  // namespace sweep_defaults
