// ab_build.hpp -- is this the tools build (libmmult_hip_ab.so)?  The kernel headers test kAbBuild with `if constexpr`
// where an A/B switch rides in a kernel argument's spare bits (raster group height, publish-on-the-spot): no
// preprocessor in the kernels, and nothing of the switches in the product's code objects.  Its own header, included
// by every header that tests it (it lived in internal.hpp and compiled by include order).
#pragma once

namespace mmh {
#ifdef MMH_AB_BUILD
constexpr bool kAbBuild = true;
#else
constexpr bool kAbBuild = false;
#endif
}  // namespace mmh
