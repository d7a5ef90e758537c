#pragma once
#include "../modal/surface.hpp"
