// Thread-local last-error storage behind rrt_last_error(), and the exception barrier of the C ABI.
#pragma once
#include <exception>
#include <new>
#include <string>

// Records `message` for the calling thread and returns `code` (so callers can `return rrt_set_error(...)`).
int rrt_set_error(int code, const std::string& message);

// Runs the body of an extern "C" entry point; a C++ exception (std::bad_alloc from a vector sized by a hostile file, ...) becomes
// an error code instead of unwinding into a C or Rust caller, which would end the process.
template <class Body>
int rrt_guard(const char* entry, Body&& body) noexcept {
    try {
        return body();
    } catch (const std::bad_alloc&) {
        return rrt_set_error(4 /* RRT_ERR_LIMIT */, std::string(entry) + ": out of host memory");
    } catch (const std::exception& e) {
        return rrt_set_error(1 /* RRT_ERR_INVALID */, std::string(entry) + ": " + e.what());
    } catch (...) {
        return rrt_set_error(1 /* RRT_ERR_INVALID */, std::string(entry) + ": unknown exception");
    }
}
