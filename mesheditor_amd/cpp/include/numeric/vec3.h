#pragma once
#include "../modal/math.hpp"
