#pragma once
#include "../modal/types.hpp"
