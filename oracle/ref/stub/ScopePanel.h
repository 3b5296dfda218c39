// TEST INFRASTRUCTURE ONLY (oracle/): stands in for src/panel/ScopePanel.h (a GL panel) when the reference's
// ScopeVisualProcessor.cpp is compiled unmodified into oracle/_ref: the processor only uses the mode enumeration (ScopePanel.h:11).
#pragma once
#include <cmath>     // (the real panel header brings the math declarations in through its GL / wx includes)
class ScopePanel {
public:
    typedef enum ScopeMode { SCOPE_MODE_Y, SCOPE_MODE_2Y, SCOPE_MODE_XY } ScopeMode;
};
