// check.hpp -- minimal assertion helpers for the host-layer test programs.
// CLOSE uses a real relative tolerance (7e-5, the size of the reference's ASSERT_CLOSE band
// |log2 x - log2 y| <= 1e-4, test/test.hpp:39-46, which itself passes vacuously for
// negative numbers).
#pragma once
#include <cmath>
#include <cstdio>

static int g_failures = 0;

#define CHECK(cond)                                                                  \
    do {                                                                             \
        if (!(cond)) {                                                               \
            std::fprintf(stderr, "FAILURE: %s at %s:%d\n", #cond, __FILE__, __LINE__); \
            g_failures++;                                                            \
        }                                                                            \
    } while (0)

#define CHECK_EQ(x, y) CHECK((x) == (y))

#define CHECK_CLOSE(x, y)                                                                              \
    do {                                                                                               \
        const double x_ = (x), y_ = (y);                                                               \
        if (!(std::fabs(x_ - y_) <= 7e-5 * std::fabs(y_) + 6e-8)) {                                    \
            std::fprintf(stderr, "FAILURE: expected %.9g got %.9g for %s at %s:%d\n", y_, x_, #x, __FILE__, __LINE__); \
            g_failures++;                                                                              \
        }                                                                                              \
    } while (0)

#define RUN(fn, ...)                                                        \
    do {                                                                    \
        const int before = g_failures;                                      \
        fn(__VA_ARGS__);                                                    \
        std::printf("%s: %s\n", g_failures == before ? "TEST PASSED" : "TEST FAILED", #fn); \
    } while (0)
