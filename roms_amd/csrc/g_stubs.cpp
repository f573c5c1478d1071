// g_stubs.cpp -- (no entry points are stubbed any more; kept so that the build list stays stable)
#include "roms_host.h"
