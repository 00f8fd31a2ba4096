// libsherf_hip_ops.so: error text shared by its translation units.
#include "ops_common.h"

char g_sherf_ops_err[256] = "";

extern "C" const char* sherf_ops_last_error(void) { return g_sherf_ops_err; }
