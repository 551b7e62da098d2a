// mini_test.h -- a 40-line stand-in for gtest (not installed here; the reference fetches it from
// the network).  TEST(Suite, Name) { EXPECT_TRUE(..); EXPECT_EQ(..); }  + main().
#pragma once
#include <cstdio>
#include <exception>
#include <functional>
#include <string>
#include <vector>

namespace mini {
struct test_case { std::string name; std::function<void()> fn; };
inline std::vector<test_case>& registry() { static std::vector<test_case> r; return r; }
inline int& failures() { static int f = 0; return f; }
struct registrar { registrar(const char* s, const char* n, std::function<void()> f) { registry().push_back({std::string(s) + "." + n, std::move(f)}); } };
inline void fail(const char* file, int line, const char* expr) { std::printf("  FAILED %s:%d: %s\n", file, line, expr); ++failures(); }
inline int run_all() {
  int bad = 0;
  for (auto& t : registry()) {
    const int before = failures();
    std::printf("[ RUN  ] %s\n", t.name.c_str());
    try { t.fn(); } catch (std::exception const& e) { std::printf("  EXCEPTION: %s\n", e.what()); ++failures(); }
    const bool ok = failures() == before;
    std::printf("[ %s ] %s\n", ok ? " OK " : "FAIL", t.name.c_str());
    bad += !ok;
  }
  std::printf("%zu tests, %d failed\n", registry().size(), bad);
  return bad ? 1 : 0;
}
}  // namespace mini
#define TEST(S, N) static void test_##S##_##N(); static mini::registrar reg_##S##_##N(#S, #N, test_##S##_##N); static void test_##S##_##N()
#define EXPECT_TRUE(x) do { if (!(x)) mini::fail(__FILE__, __LINE__, #x); } while (0)
#define EXPECT_FALSE(x) EXPECT_TRUE(!(x))
#define EXPECT_EQ(a, b) do { if (!((a) == (b))) mini::fail(__FILE__, __LINE__, #a " == " #b); } while (0)
