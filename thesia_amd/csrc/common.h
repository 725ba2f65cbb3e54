// common.h — status/error plumbing shared by every translation unit of libthesia_amd.
#pragma once
#include <cstdarg>
#include <cstdio>
#include <exception>
#include <new>

#include "../../include/thesia_amd.h"
#include "../../include/thesia_amd_testing.h"  // (exported by the same library; a host binds none of it)

namespace th {

// thread-local last-error message (th_last_error)
void set_error(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
const char *get_error();

inline int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
inline int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    set_error("%s", buf);
    return code;
}

}  // namespace th

// No C++ exception may cross the C ABI: every entry point body is wrapped.
#define TH_TRY try {
#define TH_CATCH                                                          \
    }                                                                     \
    catch (const std::bad_alloc &) { return th::fail(TH_ERR_OOM, "host allocation failed"); } \
    catch (const std::exception &e) { return th::fail(TH_ERR_INTERNAL, "exception: %s", e.what()); } \
    catch (...) { return th::fail(TH_ERR_INTERNAL, "unknown exception"); }

#define TH_REQUIRE(cond, ...)                                             \
    do {                                                                  \
        if (!(cond)) return th::fail(TH_ERR_INVALID_ARG, __VA_ARGS__);    \
    } while (0)

#define TH_HIP(call)                                                      \
    do {                                                                  \
        hipError_t e_ = (call);                                           \
        if (e_ != hipSuccess)                                             \
            return th::fail(e_ == hipErrorOutOfMemory ? TH_ERR_OOM : TH_ERR_HIP, "%s failed: %s (%s:%d)", #call, \
                            hipGetErrorString(e_), __FILE__, __LINE__);   \
    } while (0)
