// `expected<T, E>` for the front end's return types.  The reference declares tetra::Tetrahedralize and GenerateTets as
// std::expected<Result, std::string> (src/mesh/Tetrahedralize.h:61, src/mesh/Tets.h:16) and its callers use no more of it than
// `if (!tets)`, `tets.error()`, `tets->Mesh`, `*tets` (tests/ModalSolveTool.cpp:72-76, tests/ModalSolverBench.cpp:285-290).  With a
// C++23 standard library this IS std::expected; with an older one (libstdc++ 11 has no <expected>) the minimal class below stands in with
// the same member names, so a caller written against the reference compiles either way.
#pragma once
#include <version>

#if defined(__cpp_lib_expected) && __cpp_lib_expected >= 202202L
#include <expected>
namespace modal_compat {
template <class T, class E> using expected = std::expected<T, E>;
template <class E> using unexpected = std::unexpected<E>;
} // namespace modal_compat
#else
#include <exception>
#include <type_traits>
#include <utility>
#include <variant>

namespace modal_compat {
template <class E> class unexpected {
public:
    constexpr explicit unexpected(E e) : Error(std::move(e)) {}
    constexpr const E &error() const & noexcept { return Error; }
    constexpr E &error() & noexcept { return Error; }
    constexpr E &&error() && noexcept { return std::move(Error); }

private:
    E Error;
};
template <class E> unexpected(E) -> unexpected<E>;

template <class E> class bad_expected_access : public std::exception {
public:
    explicit bad_expected_access(E e) : Error(std::move(e)) {}
    const char *what() const noexcept override { return "bad access to expected without a value"; }
    const E &error() const & noexcept { return Error; }

private:
    E Error;
};

template <class T, class E> class expected {
public:
    using value_type = T;
    using error_type = E;
    using unexpected_type = unexpected<E>;

    constexpr expected() : State(std::in_place_index<0>) {}
    constexpr expected(const T &v) : State(std::in_place_index<0>, v) {}
    constexpr expected(T &&v) : State(std::in_place_index<0>, std::move(v)) {}
    template <class G, class = std::enable_if_t<std::is_constructible_v<E, const G &>>> constexpr expected(const unexpected<G> &u) : State(std::in_place_index<1>, u.error()) {}
    template <class G, class = std::enable_if_t<std::is_constructible_v<E, G>>> constexpr expected(unexpected<G> &&u) : State(std::in_place_index<1>, std::move(u).error()) {}

    constexpr bool has_value() const noexcept { return State.index() == 0; }
    constexpr explicit operator bool() const noexcept { return has_value(); }
    // as std::expected: dereferencing without a value, or error() with one, is a precondition violation
    constexpr T &operator*() & noexcept { return *std::get_if<0>(&State); }
    constexpr const T &operator*() const & noexcept { return *std::get_if<0>(&State); }
    constexpr T &&operator*() && noexcept { return std::move(*std::get_if<0>(&State)); }
    constexpr T *operator->() noexcept { return std::get_if<0>(&State); }
    constexpr const T *operator->() const noexcept { return std::get_if<0>(&State); }
    constexpr T &value() & {
        if (!has_value()) throw bad_expected_access<E>(error());
        return **this;
    }
    constexpr const T &value() const & {
        if (!has_value()) throw bad_expected_access<E>(error());
        return **this;
    }
    constexpr T &&value() && {
        if (!has_value()) throw bad_expected_access<E>(error());
        return std::move(**this);
    }
    constexpr E &error() & noexcept { return *std::get_if<1>(&State); }
    constexpr const E &error() const & noexcept { return *std::get_if<1>(&State); }
    constexpr E &&error() && noexcept { return std::move(*std::get_if<1>(&State)); }
    template <class U> constexpr T value_or(U &&fallback) const & { return has_value() ? **this : static_cast<T>(std::forward<U>(fallback)); }

private:
    std::variant<T, E> State;
};
} // namespace modal_compat
#endif
