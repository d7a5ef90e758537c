#pragma once
#include "../modal/solver.hpp"
