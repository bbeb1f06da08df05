// Identity of the build: the hash of the kernel sources (nerflidar_hip/buildinfo.py) at compile time, passed by the Makefile.
#include "../../include/nerflidar_hip.h"

#ifndef NLR_SOURCE_SHA
#error "NLR_SOURCE_SHA must be defined by the build (see the Makefile rule for build/nlr_build.o)"
#endif

extern "C" const char *nlr_build_sha(void) { return NLR_SOURCE_SHA; }
