// TEST INFRASTRUCTURE ONLY (oracle/): stand-in for the application header when the reference's src/sdr/SDRPostThread.cpp is compiled
// unmodified into oracle/_ref/libref_post.so.  It reads the application centre frequency / sample rate and the demodulator manager
// (updateActiveDemodulators :44-98, runDemodChannels :303-398).
#pragma once
#include <cmath>
#include <cstdlib>
#include <cstring>
#include "DemodulatorMgr.h"
struct OracleApp {
    long long sampleRate = 2400000, frequency = 100000000;
    long long getSampleRate() { return sampleRate; }
    long long getFrequency() { return frequency; }
    void setFrequency(long long f) { frequency = f; }
    DemodulatorMgr &getDemodMgr();
};
inline OracleApp &wxGetApp() { static OracleApp app; return app; }
