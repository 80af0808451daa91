// Thread-local last-error storage behind rrt_last_error().
#pragma once
#include <string>

// Records `message` for the calling thread and returns `code` (so callers can `return rrt_set_error(...)`).
int rrt_set_error(int code, const std::string& message);
