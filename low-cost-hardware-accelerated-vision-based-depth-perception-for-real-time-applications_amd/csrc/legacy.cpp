// Group (A) of include/stereo_vision_hip.h: the reference's exported symbols.  (Filled in by legacy.cpp proper.)
#include <stdio.h>

#include "../../include/stereo_vision_hip.h"

extern "C" {
Double3 *generatePointCloud(unsigned char *, unsigned char *, char *, int, int, bool, bool, bool, bool, int, int, const char *, const char *, const char *, bool, bool) {
    fprintf(stderr, "generatePointCloud: not wired yet\n");
    return nullptr;
}
void clean(void) {}
Uchar4 *getColor(void) { return nullptr; }
const unsigned char *sv_legacy_last_dmap(int *, int *) { return nullptr; }
}
