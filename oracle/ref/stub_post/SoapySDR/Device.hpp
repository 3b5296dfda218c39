#pragma once
#include "Types.hpp"
namespace SoapySDR { class Stream; class Device { public: virtual ~Device() {} }; }
