// Minimal check harness + scene builders for the C++ tests of the host mirror.  The tests include the mirror through
// the reference's own header paths (audio/mesh2modes.h, audio/ModalAudio.h, audio/ContactModel.h) and call it the way
// the reference's ModalSolverTest / ModalRenderTest / ContactModelTest call the original, so they double as the
// "drops in unchanged" compile check; the assertions restate those tests' properties (tests/ModalSolverTest.cpp:
// 228-261, tests/ModalRenderTest.cpp:21-68, tests/ContactModelTest.cpp:55-125 of the reference).
#pragma once
#include <cmath>
#include <cstdio>
#include <functional>
#include <string>
#include <vector>

namespace check {
struct Case {
    std::string name;
    std::function<void()> body;
};
inline std::vector<Case> &cases() {
    static std::vector<Case> c;
    return c;
}
inline int &failures() {
    static int f = 0;
    return f;
}
struct Register {
    Register(const char *name, std::function<void()> body) { cases().push_back({name, std::move(body)}); }
};
inline void expect(bool ok, const char *expr, const char *file, int line, const std::string &note = {}) {
    if (ok) return;
    ++failures();
    std::fprintf(stderr, "  FAILED %s:%d: %s %s\n", file, line, expr, note.c_str());
}
inline bool near(double value, double target, double rel) { return std::abs(value - target) <= rel * std::abs(target); }
inline int run_all() {
    for (auto &c : cases()) {
        const int before = failures();
        std::printf("[ RUN  ] %s\n", c.name.c_str());
        try {
            c.body();
        } catch (const std::exception &e) {
            ++failures();
            std::fprintf(stderr, "  EXCEPTION %s\n", e.what());
        }
        std::printf("[ %s ] %s\n", failures() == before ? " OK " : "FAIL", c.name.c_str());
    }
    std::printf("%d case(s), %d failure(s)\n", int(cases().size()), failures());
    return failures() ? 1 : 0;
}
} // namespace check

#define CASE(name) static void name(); static check::Register reg_##name(#name, name); static void name()
#define EXPECT(cond) check::expect(bool(cond), #cond, __FILE__, __LINE__)
#define EXPECT_NOTE(cond, note) check::expect(bool(cond), #cond, __FILE__, __LINE__, note)
