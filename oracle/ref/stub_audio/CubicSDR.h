// TEST INFRASTRUCTURE ONLY (oracle/): stand-in for the application header when the reference's AUDIO sources (src/audio/AudioThread.cpp,
// AudioFile.cpp, AudioFileWAV.cpp) are compiled unmodified into oracle/_ref/libref_audio.so.  They touch wxGetApp().getDemodMgr() (in
// AudioThread::setSampleRate, never reached by the harness) and wxGetApp().getConfig()->getRecordingPath() (the WAV writer's directory).
#pragma once
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include "DemodulatorMgr.h"
struct OracleConfig { std::string recordingPath; std::string getRecordingPath() { return recordingPath; } };
struct OracleApp {
    long long sampleRate = 2400000;
    long long getSampleRate() { return sampleRate; }
    DemodulatorMgr &getDemodMgr();
    OracleConfig *getConfig() { static OracleConfig c; return &c; }
};
inline OracleApp &wxGetApp() { static OracleApp app; return app; }
