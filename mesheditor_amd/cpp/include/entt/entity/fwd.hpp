// Stand-in for the entt names the modal API mentions (the reference only forward-declares them as well): the entity
// handle, and the registry as an opaque type for the surface-contact hooks that take one by reference.
#pragma once
#include <cstdint>
namespace entt {
enum class entity : std::uint32_t {};
inline constexpr entity null{0xffffffffu};
class registry; // never defined here: callers of this build have no ECS, the hooks only pass the reference through
}
