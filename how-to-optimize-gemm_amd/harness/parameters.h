// parameters.h -- sweep defaults of the MI355X harness.
//
// Same knobs as the reference's cuda/parameters.h:5-24 (PFIRST/PLAST/PINC,
// M/N/K = -1 binds the dimension to p, NREPEATS) and armv7/parameters.h:38-46
// (LDA/LDB/LDC = -1 binds the leading dimension to the row length), but they
// are DEFAULTS: every one can be overridden at run time by an environment
// variable of the same name or a --name=value argument, so one binary covers
// all five BASELINE.json configurations.
#pragma once

#define PFIRST 1024
#define PLAST 4096
#define PINC 128

#define M -1
#define N -1
#define K -1

#define NREPEATS 20

#define LDA -1
#define LDB -1
#define LDC -1
