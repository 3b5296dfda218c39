// TEST INFRASTRUCTURE ONLY (oracle/): the wx value types AppConfig.h declares members / parameters with
#pragma once
#include <string>
struct wxPoint { int x = 0, y = 0; wxPoint() {} wxPoint(int a, int b) : x(a), y(b) {} };
struct wxSize { int x = 0, y = 0; wxSize() {} wxSize(int a, int b) : x(a), y(b) {} };
struct wxRect { int x = 0, y = 0, width = 0, height = 0; };
typedef std::string wxString;
