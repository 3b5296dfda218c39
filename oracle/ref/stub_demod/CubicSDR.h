// TEST INFRASTRUCTURE ONLY (oracle/): stand-in for the application header when the reference's src/demod/DemodulatorThread.cpp is compiled
// unmodified into oracle/_ref/libref_demodthread.so.  It touches wxGetApp().getSoloMode(), getAppFrame()->isUserDemodBusy() and the
// demodulator manager's "active demodulator" calls (squelch break in solo mode, :202-213, :319-320).
#pragma once
#include <cmath>
#include <cstdlib>
#include <cstring>
#include "DemodulatorMgr.h"
struct OracleAppFrame { bool isUserDemodBusy() { return false; } };
struct OracleApp {
    long long sampleRate = 2400000;
    bool soloMode = false;
    long long getSampleRate() { return sampleRate; }
    bool getSoloMode() { return soloMode; }
    OracleAppFrame *getAppFrame() { static OracleAppFrame f; return &f; }
    DemodulatorMgr &getDemodMgr();
};
inline OracleApp &wxGetApp() { static OracleApp app; return app; }
