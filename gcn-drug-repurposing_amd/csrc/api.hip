// api.hip -- ABI version and error text.
#include "common.h"

namespace gss {
thread_local char g_err[512] = "";
}

extern "C" {
int gss_abi_version(void) { return GSS_ABI_VERSION; }
const char *gss_last_error(void) { return gss::g_err; }
}
