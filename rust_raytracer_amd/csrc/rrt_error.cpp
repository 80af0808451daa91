#include "rrt_error.hpp"

#include "../../include/rrt.h"

static thread_local std::string g_last_error;

int rrt_set_error(int code, const std::string& message) {
    g_last_error = message;
    return code;
}

extern "C" const char* rrt_last_error(void) { return g_last_error.c_str(); }
extern "C" const char* rrt_version(void) { return "rrt-mi355x 0.1 (gfx950)"; }
