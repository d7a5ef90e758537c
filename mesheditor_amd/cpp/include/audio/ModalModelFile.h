#pragma once
#include "../modal/model_io.hpp"
