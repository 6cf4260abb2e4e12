// Shared host-side helpers for the lsfc HIP library (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdarg>
#include <cstdlib>
#include <cstdio>
#include <cstdint>
#include <string>
#include <stdexcept>

#include "../../include/lsfc.h"

namespace lsfc {

using cplx = double2;   // interleaved (re, im) == Julia Complex{Float64}

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& msg) : std::runtime_error(msg), code(c) {}
};

void set_last_error(const char* fmt, ...);

[[noreturn]] inline void fail(int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    throw Error(code, buf);
}

// (a failed call also leaves HIP's per-thread "last error" set: clear it so that a later hipGetLastError() check
// does not report this failure a second time)
#define LSFC_HIP(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { (void)hipGetLastError(); \
    ::lsfc::fail(e_ == hipErrorOutOfMemory ? LSFC_ENOMEM : LSFC_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); } } while (0)

#define LSFC_REQUIRE(cond, ...) do { if (!(cond)) ::lsfc::fail(LSFC_EINVAL, __VA_ARGS__); } while (0)

// Runs a C-ABI body; converts every C++ exception into an error code.
template <class F> int guarded(F&& body) {
    try { body(); return LSFC_OK; }
    catch (const Error& e) { set_last_error("%s", e.what()); return e.code; }
    catch (const std::bad_alloc&) { set_last_error("host allocation failed"); return LSFC_ENOMEM; }
    catch (const std::exception& e) { set_last_error("%s", e.what()); return LSFC_EHIP; }
    catch (...) { set_last_error("unknown error"); return LSFC_EHIP; }
}

// LSFC_PLAN_TIMING=1: wall-clock phases of plan creation on stderr (the device is synchronised at every mark)
struct PhaseTimer {
    bool on; std::chrono::steady_clock::time_point t0;
    PhaseTimer() : on(getenv("LSFC_PLAN_TIMING") && getenv("LSFC_PLAN_TIMING")[0] == '1'), t0(std::chrono::steady_clock::now()) {}
    void mark(const char* what) {
        if (!on) return;
        (void)hipDeviceSynchronize();
        const auto t1 = std::chrono::steady_clock::now();
        fprintf(stderr, "[lsfc plan] %-34s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
        t0 = t1;
    }
};

inline bool is_pow2(int64_t v) { return v > 0 && (v & (v - 1)) == 0; }

// RAII device buffer
template <class T> struct DevBuf {
    T* p = nullptr; size_t n = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete; DevBuf& operator=(const DevBuf&) = delete;
    DevBuf(DevBuf&& o) noexcept : p(o.p), n(o.n) { o.p = nullptr; o.n = 0; }
    DevBuf& operator=(DevBuf&& o) noexcept { if (this != &o) { release(); p = o.p; n = o.n; o.p = nullptr; o.n = 0; } return *this; }
    ~DevBuf() { release(); }
    void alloc(size_t count) { release(); if (count) { LSFC_HIP(hipMalloc((void**)&p, count * sizeof(T))); n = count; } }
    void release() { if (p) { (void)hipFree(p); p = nullptr; n = 0; } }
    size_t bytes() const { return n * sizeof(T); }
};

} // namespace lsfc
