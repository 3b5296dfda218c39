// Modem.h -- the reference's modem plug-in surface (src/modules/modem/Modem.h:127-166, Modem.cpp:42-101) over the HIP library.
//
// In the reference a Modem object owns the demodulator arithmetic (demodulate()) and its kit.  Two kinds of modem exist here:
//   * the reference's nine analog modems (ModemHip<ID>): their arithmetic runs on the GPU inside csdr_bank_execute, the object is the
//     host-side DESCRIPTOR of the type -- name / type, default and admissible rates (checkSampleRate), the CSDR_MODEM_* id of its slot;
//   * any OTHER class registered through addModemFactory (an integrator's plug-in, e.g. a digital modem): it keeps the reference's
//     contract -- the pipeline runs DemodulatorPreThread's arithmetic (NCO shift + msresamp to the modem's rate) on the GPU in a
//     CSDR_MODEM_HOST slot, fetches the block's resampled IQ and calls the plug-in's own buildKit / demodulate(kit, iq, audioOut) on
//     the host thread, exactly where DemodulatorThread::run calls it (DemodulatorThread.cpp:119-135).
// The registry (addModemFactory / makeModem / getFactories / getModemDefaultSampleRate) and the settings calls keep the reference's
// signatures, so DemodulatorInstance and the GUI's modem menus bind unchanged.
#pragma once
#include <atomic>
#include <map>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/csdr_hip.h"
#include "DataTypes.h"

#define MIN_BANDWIDTH 500                      // Modem.h:13

class ModemKit {                                // Modem.h:15-24
public:
    ModemKit() : sampleRate(0), audioSampleRate(0) {}
    virtual ~ModemKit() = default;
    long long sampleRate;
    int audioSampleRate;
};
class ModemIQData {                             // Modem.h:26-37
public:
    std::vector<liquid_float_complex_t> data;
    long long sampleRate = 0;
    virtual ~ModemIQData() = default;
};
struct ModemArgInfo {                                                         // ModemArgInfo of Modem.h:22-60, the fields FM stereo's setting uses
    std::string key, value, name, description, units;
    std::vector<std::string> options, optionNames;
};   // the analog modems publish no settings but FM-stereo's de-emphasis
typedef std::vector<ModemArgInfo> ModemArgInfoList;
typedef std::map<std::string, std::string> ModemSettings;

class ModemBase {};
typedef ModemBase *(*ModemFactoryFn)();
typedef std::map<std::string, ModemFactoryFn> ModemFactoryList;
typedef std::map<std::string, int> DefaultRatesList;

class Modem : public ModemBase {
public:
    static void addModemFactory(ModemFactoryFn factoryFunc, std::string modemName, int defaultRate) {     // Modem.cpp:42-45
        std::lock_guard<std::mutex> g(registryMutex());
        factories()[modemName] = factoryFunc; defaultRates()[modemName] = defaultRate;
    }
    static ModemFactoryList getFactories() { registerBuiltins(); std::lock_guard<std::mutex> g(registryMutex()); return factories(); }
    static Modem *makeModem(std::string modemName) {                                                      // Modem.cpp:51-57
        registerBuiltins();
        ModemFactoryFn fn = nullptr;
        {
            std::lock_guard<std::mutex> g(registryMutex());
            auto it = factories().find(modemName);
            if (it != factories().end()) fn = it->second;
        }
        return fn ? static_cast<Modem *>(fn()) : nullptr;
    }
    static int getModemDefaultSampleRate(std::string modemName) {                                         // Modem.cpp:59-65
        registerBuiltins();
        std::lock_guard<std::mutex> g(registryMutex());
        auto it = defaultRates().find(modemName);
        return it == defaultRates().end() ? 0 : it->second;
    }

    virtual std::string getType() = 0;
    virtual std::string getName() = 0;
    Modem() { useSignalOutput(false); }
    virtual ~Modem() = default;

    virtual ModemArgInfoList getSettings() { return ModemArgInfoList(); }
    virtual int getDefaultSampleRate() { return 200000; }
    virtual void writeSetting(std::string, std::string) {}
    virtual void writeSettings(ModemSettings settings) { for (auto &kv : settings) writeSetting(kv.first, kv.second); }
    virtual std::string readSetting(std::string) { return ""; }
    virtual ModemSettings readSettings() {
        ModemSettings rs;
        for (auto &a : getSettings()) rs[a.key] = readSetting(a.key);
        return rs;
    }
    virtual int checkSampleRate(long long sampleRate, int audioSampleRate) = 0;
    virtual ModemKit *buildKit(long long sampleRate, int audioSampleRate) {          // the rates the bank slot is built with
        ModemKit *kit = new ModemKit; kit->sampleRate = sampleRate; kit->audioSampleRate = audioSampleRate; return kit;
    }
    virtual void disposeKit(ModemKit *kit) { delete kit; }
    // Modem.h:151: the plug-in's arithmetic.  Called by SDRPostThread::finishDemod for CSDR_MODEM_HOST slots, from the one thread
    // that owns the instance (the reference's rule).
    virtual void demodulate(ModemKit *kit, ModemIQData *input, AudioThreadInput *audioOut) = 0;
    bool shouldRebuildKit() { return refreshKit.load(); }
    void rebuildKit() { refreshKit.store(true); }
    void clearRebuildKit() { refreshKit.store(false); }
    bool useSignalOutput() { return _useSignalOutput.load(); }
    void useSignalOutput(bool useOutput) { _useSignalOutput.store(useOutput); }

    // the CSDR_MODEM_* id of this modem's bank slot (include/csdr_hip.h): a plug-in runs the front end only and demodulates on the host
    virtual int csdrModemId() { return CSDR_MODEM_HOST; }
    virtual int csdrModemArg() { return 0; }

    static void registerBuiltins();

private:
    static ModemFactoryList &factories() { static ModemFactoryList f; return f; }
    static DefaultRatesList &defaultRates() { static DefaultRatesList r; return r; }
    static std::mutex &registryMutex() { static std::mutex m; return m; }
    std::atomic_bool refreshKit{false}, _useSignalOutput{false};
};

// one descriptor class for the analog modems of the reference (src/modules/modem/analog/*.cpp)
template <int ID>
class ModemHip : public Modem {
public:
    static ModemBase *factory() { return new ModemHip<ID>(); }
    ModemHip() {
        // useSignalOutput(true): ModemAM.cpp:8, ModemUSB.cpp:12, ModemLSB.cpp:12, ModemDSB.cpp:7, ModemCW.cpp:24
        useSignalOutput(ID == CSDR_MODEM_AM || ID == CSDR_MODEM_USB || ID == CSDR_MODEM_LSB || ID == CSDR_MODEM_DSB || ID == CSDR_MODEM_CW);
    }
    std::string getType() override { return "analog"; }
    std::string getName() override {
        switch (ID) {
            case CSDR_MODEM_NBFM: return "NBFM"; case CSDR_MODEM_FM: return "FM"; case CSDR_MODEM_AM: return "AM";
            case CSDR_MODEM_USB: return "USB"; case CSDR_MODEM_LSB: return "LSB"; case CSDR_MODEM_IQ: return "I/Q";
            case CSDR_MODEM_CW: return "CW"; case CSDR_MODEM_DSB: return "DSB"; case CSDR_MODEM_FMS: return "FMS";
        }
        return "";
    }
    int getDefaultSampleRate() override {
        switch (ID) {
            case CSDR_MODEM_NBFM: return 12500; case CSDR_MODEM_FM: case CSDR_MODEM_FMS: return 200000; case CSDR_MODEM_AM: return 6000;
            case CSDR_MODEM_USB: case CSDR_MODEM_LSB: case CSDR_MODEM_DSB: return 5400; case CSDR_MODEM_IQ: return 48000;
            case CSDR_MODEM_CW: return MIN_BANDWIDTH;
        }
        return 200000;
    }
    int checkSampleRate(long long sampleRate, int audioSampleRate) override {
        if (ID == CSDR_MODEM_IQ) return audioSampleRate;                                   // ModemIQ.cpp:31-33
        if (ID == CSDR_MODEM_FMS) { if (sampleRate < 100000) return 100000; if (sampleRate < 1500) return 1500; return (int)sampleRate; }   // ModemFMStereo.cpp:41-49
        if (sampleRate < MIN_BANDWIDTH) return MIN_BANDWIDTH;                              // ModemAnalog.cpp:14-19
        if ((ID == CSDR_MODEM_USB || ID == CSDR_MODEM_LSB) && (sampleRate % 2)) return (int)sampleRate + 1;   // ModemUSB.cpp:29-37
        return (int)sampleRate;
    }
    int csdrModemId() override { return ID; }
    // the arithmetic of these nine modems runs on the device (csdr_bank_execute): nothing may route a block through the host entry
    void demodulate(ModemKit *, ModemIQData *, AudioThreadInput *) override {
        throw std::logic_error("ModemHip::demodulate: this modem's arithmetic runs on the device (csdr_bank_execute)");
    }
    // FM stereo's one setting, "demph" (ModemFMStereo.cpp:42-89): de-emphasis in microseconds, "0" = none; a write asks for a rebuild
    ModemArgInfoList getSettings() override {
        ModemArgInfoList args;
        if (ID != CSDR_MODEM_FMS) return args;
        ModemArgInfo a;
        a.key = "demph"; a.name = "De-emphasis"; a.value = std::to_string(demph_);
        a.description = "FM Stereo De-Emphasis, typically 75us in US/Canada, 50us elsewhere.";
        a.optionNames = {"None", "10us", "25us", "32us", "50us", "75us"};
        a.options = {"0", "10", "25", "32", "50", "75"};
        args.push_back(a);
        return args;
    }
    void writeSetting(std::string setting, std::string value) override {
        if (ID == CSDR_MODEM_FMS && setting == "demph") { demph_ = std::stoi(value); rebuildKit(); }
    }
    std::string readSetting(std::string setting) override { return (ID == CSDR_MODEM_FMS && setting == "demph") ? std::to_string(demph_) : ""; }
    // csdr_demod_params::modem_arg for this modem: FM stereo's de-emphasis (0 there means "the default", so "None" travels as -1)
    int csdrModemArg() override { return ID == CSDR_MODEM_FMS ? (demph_ ? demph_ : -1) : 0; }

private:
    int demph_ = 75;                                                          // ModemFMStereo::ModemFMStereo()
};

inline void Modem::registerBuiltins() {                                                   // CubicSDR.cpp:305-313
    // (instances are created from several threads: DemodulatorInstance's constructor ends up here)
    static std::once_flag once;
    std::call_once(once, [] {
    addModemFactory(ModemHip<CSDR_MODEM_FM>::factory, "FM", 200000);
    addModemFactory(ModemHip<CSDR_MODEM_NBFM>::factory, "NBFM", 12500);
    addModemFactory(ModemHip<CSDR_MODEM_FMS>::factory, "FMS", 200000);
    addModemFactory(ModemHip<CSDR_MODEM_AM>::factory, "AM", 6000);
    addModemFactory(ModemHip<CSDR_MODEM_CW>::factory, "CW", 500);
    addModemFactory(ModemHip<CSDR_MODEM_LSB>::factory, "LSB", 5400);
    addModemFactory(ModemHip<CSDR_MODEM_USB>::factory, "USB", 5400);
    addModemFactory(ModemHip<CSDR_MODEM_DSB>::factory, "DSB", 5400);
    addModemFactory(ModemHip<CSDR_MODEM_IQ>::factory, "I/Q", 48000);
    });
}
