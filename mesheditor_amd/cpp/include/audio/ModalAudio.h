#pragma once
#include "../modal/bank.hpp"
