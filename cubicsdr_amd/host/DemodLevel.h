// DemodLevel.h -- the host-side level / floor / ceil / squelch state machine of DemodulatorThread::run
// (reference src/demod/DemodulatorThread.cpp:142-220, linearToDb :59-67), one step per demodulated block.  The sums it starts
// from (level_accum / level_count) come from the device (csdr_block_result); everything here is a few scalar operations per
// block in the reference's own types (float trackers, double level), so it stays on the host and is pinned by
// tests/test_host_mirror.py::test_level_squelch_state_machine_matches_oracle against the checker's statement-by-statement restatement.
#pragma once
#include <cmath>

struct DemodLevelState {
    float signalLevel = -100.0f, signalFloor = -30.0f, signalCeil = 30.0f;     // DemodulatorThread ctor (:20-27)
    bool squelchBreak = false;
};

inline double demodLinearToDb(double linear) { if (linear <= 1e-20) linear = 1e-20; return 20.0 * std::log10(linear); }

// one block: `have_level` = the block produced audio (ati && !ati->data.empty(), :145), level_accum / level_count = the magnitude
// sum and its term count, sampleTime = input samples / input rate (:141).  Returns `squelched` (:198).
inline bool demodLevelStep(DemodLevelState &s, bool have_level, double level_accum, int level_count, double sampleTime,
                           bool squelchEnabled, float squelchLevel) {
    double currentSignalLevel = 0;
    if (have_level) {
        currentSignalLevel = demodLinearToDb(level_accum / double(level_count));
        float sf = s.signalFloor, sc = s.signalCeil, sl = squelchLevel;
        if (currentSignalLevel > sc) sc = (float)currentSignalLevel;
        if (currentSignalLevel < sf) sf = (float)currentSignalLevel;
        if (sl + 1.0f > sc) sc = sl + 1.0f;
        if ((sf + 2.0f) > sc) sc = sf + 2.0f;
        sc -= (sc - (currentSignalLevel + 2.0f)) * sampleTime * 0.05f;
        sf += ((currentSignalLevel - 5.0f) - sf) * sampleTime * 0.15f;
        s.signalFloor = sf; s.signalCeil = sc;
    }
    float lvl = s.signalLevel;
    if (currentSignalLevel > lvl) lvl = lvl + (currentSignalLevel - lvl) * 0.5;
    else lvl = lvl + (currentSignalLevel - lvl) * 0.05 * sampleTime * 30.0;
    s.signalLevel = lvl;
    const bool squelched = squelchEnabled && (lvl < squelchLevel);
    if (squelchEnabled) {
        if (!squelched && !s.squelchBreak) s.squelchBreak = true;              // (solo-mode squelch lock is GUI state: out of scope)
        else if (squelched && s.squelchBreak) s.squelchBreak = false;
    }
    return squelched;
}
