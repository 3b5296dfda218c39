#pragma once
#include "Types.hpp"
