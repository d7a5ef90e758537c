#pragma once
#include "../modal/contact.hpp"
