// Stand-in for the one entt name the bank API uses (the reference forward-declares entt::entity the same way).
#pragma once
#include <cstdint>
namespace entt {
enum class entity : std::uint32_t {};
inline constexpr entity null{0xffffffffu};
}
