// Library-level entry points of libtsg_hip.so.
#include "tsg_common.h"

extern "C" int tsg_version(void) { return TSG_VERSION; }
