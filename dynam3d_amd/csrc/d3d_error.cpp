// d3d_error.cpp -- thread-local last-error string shared by the HIP launchers and the host state machine.
#include <string>

#include "../../include/dynam3d_hip.h"

static thread_local std::string g_d3d_err;

extern "C" {
const char* d3d_last_error(void) { return g_d3d_err.c_str(); }
int32_t d3d_version(void) { return 100; }
void d3d_set_error_(const char* msg) { g_d3d_err = msg ? msg : ""; }
}
