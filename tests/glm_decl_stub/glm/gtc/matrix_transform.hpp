// see glm.hpp in this directory: declarations only, for a syntax check
#pragma once
#include "../glm.hpp"
