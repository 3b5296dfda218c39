// modem_harness.cpp -- TEST INFRASTRUCTURE ONLY (oracle/).  Never linked into the product.
//
// C entry points over the reference's OWN modem classes: this file is compiled together with the unmodified
// /root/reference/src/modules/modem/{Modem,ModemAnalog}.cpp and analog/Modem{NBFM,FM,FMStereo,AM,USB,LSB,DSB,CW,IQ}.cpp (in place,
// by oracle/Makefile, output oracle/_ref/libref_modems.so) and linked against libliquid_ref.so, whose SysV wrappers carry the
// liquid names those sources call.  What runs is therefore CubicSDR's modem code on CubicSDR's liquid binary; the Python glue in
// oracle/cubicsdr_chain.py (RefDemod.demodulate) is pinned against it by tests/test_oracle_pin.py.
#include <cstring>
#include <string>
#include <vector>

#include "Modem.h"
#include "ModemAM.h"
#include "ModemCW.h"
#include "ModemDSB.h"
#include "ModemFM.h"
#include "ModemFMStereo.h"
#include "ModemIQ.h"
#include "ModemLSB.h"
#include "ModemNBFM.h"
#include "ModemUSB.h"

namespace {
struct RefModem {
    Modem *modem = nullptr;
    ModemKit *kit = nullptr;
    ModemIQData iq;
    AudioThreadInput out;
};
bool registered = false;
void register_once() {                      // CubicSDR.cpp:305-313
    if (registered) return;
    registered = true;
    Modem::addModemFactory(ModemFM::factory, "FM", 200000);
    Modem::addModemFactory(ModemNBFM::factory, "NBFM", 12500);
    Modem::addModemFactory(ModemFMStereo::factory, "FMS", 200000);
    Modem::addModemFactory(ModemAM::factory, "AM", 6000);
    Modem::addModemFactory(ModemCW::factory, "CW", 500);
    Modem::addModemFactory(ModemLSB::factory, "LSB", 5400);
    Modem::addModemFactory(ModemUSB::factory, "USB", 5400);
    Modem::addModemFactory(ModemDSB::factory, "DSB", 5400);
    Modem::addModemFactory(ModemIQ::factory, "I/Q", 48000);
}
}  // namespace

extern "C" {
void *refmodem_create(const char *name) {
    register_once();
    Modem *m = Modem::makeModem(name);
    if (!m) return nullptr;
    RefModem *r = new RefModem();
    r->modem = m;
    return r;
}
int refmodem_default_rate(const char *name) { register_once(); return Modem::getModemDefaultSampleRate(name); }
long long refmodem_check_rate(void *h, long long rate, int audio_rate) { return ((RefModem *)h)->modem->checkSampleRate(rate, audio_rate); }
int refmodem_use_signal_output(void *h) { return ((RefModem *)h)->modem->useSignalOutput() ? 1 : 0; }
void refmodem_write_setting(void *h, const char *key, const char *value) { ((RefModem *)h)->modem->writeSetting(key, value); }
// DemodulatorWorkerThread.cpp:72-77: buildKit at the checked rate
int refmodem_build(void *h, long long sample_rate, int audio_rate) {
    RefModem *r = (RefModem *)h;
    if (r->kit) r->modem->disposeKit(r->kit);
    r->kit = r->modem->buildKit(sample_rate, audio_rate);
    r->modem->clearRebuildKit();
    return r->kit ? 0 : -1;
}
// DemodulatorThread.cpp:127-130: demodulate one block of resampled IQ; returns the float count of audioOut->data (all channels)
int refmodem_demodulate(void *h, const float *iq, int n, long long sample_rate, float *audio, int cap, int *channels) {
    RefModem *r = (RefModem *)h;
    r->iq.sampleRate = sample_rate;
    r->iq.data.resize((size_t)n);
    std::memcpy(r->iq.data.data(), iq, (size_t)n * sizeof(liquid_float_complex));
    r->out.data.clear();
    r->out.channels = 0;
    r->modem->demodulate(r->kit, &r->iq, &r->out);
    const int m = (int)r->out.data.size();
    if (m > cap) return -m;
    if (m) std::memcpy(audio, r->out.data.data(), (size_t)m * sizeof(float));
    if (channels) *channels = r->out.channels;
    return m;
}
// ModemAnalog::getDemodOutputData (the scope tap's source), analog modems only
int refmodem_demod_output(void *h, float *dst, int cap) {
    ModemAnalog *a = dynamic_cast<ModemAnalog *>(((RefModem *)h)->modem);
    if (!a) return -1;
    std::vector<float> *d = a->getDemodOutputData();
    const int m = (int)d->size();
    if (m > cap) return -m;
    if (m) std::memcpy(dst, d->data(), (size_t)m * sizeof(float));
    return m;
}
void refmodem_destroy(void *h) {
    RefModem *r = (RefModem *)h;
    if (!r) return;
    if (r->kit) r->modem->disposeKit(r->kit);
    delete r->modem;
    delete r;
}
}
