#pragma once
#include "../modal/tets.hpp"
