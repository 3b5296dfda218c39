// TEST INFRASTRUCTURE ONLY (oracle/): a stand-in for the application header, placed in front of the reference's include path when its
// hot-path sources are compiled unmodified into oracle/_ref (SpectrumVisualProcessor.cpp touches only wxGetApp().getSampleRate(), :306).
#pragma once
#include <cmath>
#include <cstdlib>
#include <cstring>
struct OracleApp {
    long long sampleRate = 2400000;
    long long getSampleRate() { return sampleRate; }
};
inline OracleApp &wxGetApp() { static OracleApp app; return app; }
