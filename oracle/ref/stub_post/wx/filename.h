#pragma once
#include "stubs.h"
