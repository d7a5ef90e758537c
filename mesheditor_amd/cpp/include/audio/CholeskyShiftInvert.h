#pragma once
#include "../modal/shift_invert.hpp"
